#!/usr/bin/env python3
"""Headline benchmark: PS-VAE training step (encoder + decoder forward + backward,
global-norm clip + Adam) on synthetic ZINC250K-shape graphs.

    python bench.py --gpus N --steps K --warmup W

One "step" = one optimiser step over one batch of 128 graphs per GPU (vocab 300,
hidden 300: the configuration the metric is quoted on).  Batches are collated on
the host beforehand and resident in HBM when the timed region starts; parameters
are randomly initialised (no checkpoints exist offline).  With N > 1 every rank
processes its own 128 graphs (weak scaling; `--scaling strong --global-batch G`
splits G graphs over the ranks instead) and the flat gradient buffer is
all-reduced over RCCL each step, in three messages overlapped with backward.
Started without a launcher (`WORLD_SIZE` unset) the program starts the N rank
processes itself.  Rank 0 prints ONE JSON line; its "strong_scaling" object is a
second, shorter measurement of vocab 500 / global batch 1024 split over the GPUs.
"""
import argparse
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ps_vae_amd import PSVAEModel, Tokenizer, default_args, full_config, synth  # noqa: E402
from ps_vae_amd import _lib  # noqa: E402
from ps_vae_amd.data import BPEMolDataset, PackedMolDataset, to_device  # noqa: E402
from ps_vae_amd.parallel import GradReducer, OverlappedReducer, shard_range  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec
MFMA_F32_PEAK_TF = 157.3   # MI355X dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
# dense bf16 matrix peak / 6: the rate at which the cut-operand kernels (fp32 operands as three bf16 pieces, six
# v_mfma_*_bf16 per fp32-accurate product block) could deliver fp32-equivalent flops
MFMA_CUT_PEAK_TF = 2500.0 / 6
FLOP_CLASSES = ('gemm_forward', 'gemm_input_grad', 'gemm_weight_grad', 'gru_forward', 'gru_backward', 'gine_layer_fused',
                'scorer_fused', 'gine_layer_fused_bwd')


def make_batches(tok, vocab, shape, batch_size, count, seed, host_times=None):
    """`count` collated batches and the seconds stage 3 took per batch.  `host_times` (dict) also receives
    what one DataLoader worker pays per batch on the two host paths: stage 2 + 3 per item
    (`BPEMolDataset`) and the flattened data set with both stages native (`PackedMolDataset`)."""
    random.seed(seed)
    out, t_collate, t_item, t_packed, all_mols = [], 0.0, 0.0, 0.0, []
    for i in range(count):
        mols = synth.make_molecules(seed * 1000 + i, shape, vocab, batch_size)
        all_mols += mols
        t0 = time.perf_counter()
        step2 = [BPEMolDataset.process_step2(m, tok, lists=False) for m in mols]
        t1 = time.perf_counter()
        out.append(BPEMolDataset.process_step3(step2, tok))
        t2 = time.perf_counter()
        t_collate += t2 - t1
        t_item += t2 - t0
        if host_times is not None:
            state = random.getstate()
            ds = PackedMolDataset(mols, tok)
            t3 = time.perf_counter()
            ds.collate_fn(list(range(batch_size)))
            t_packed += time.perf_counter() - t3
            random.setstate(state)
    if host_times is not None:
        host_times.update(per_item_ms=t_item / count * 1e3, packed_ms=t_packed / count * 1e3, mols=all_mols)
    return out, t_collate / count


def _cpu_info():
    """(exact `model name` string, physical cores of the first socket, CPUs this process may use) from /proc/cpuinfo,
    the affinity mask and the cgroup's CPU quota."""
    model, cores, phys, core = 'unknown', set(), None, None
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                key, _, val = line.partition(':')
                key, val = key.strip(), val.strip()
                if key == 'model name' and model == 'unknown':
                    model = val
                elif key == 'physical id':
                    phys = val
                elif key == 'core id':
                    core = val
                elif key == '' and phys is not None:
                    cores.add((phys, core))
                    phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    first = min((p for p, _ in cores), default=None)
    socket_cores = len({c for p, c in cores if p == first}) or (os.cpu_count() or 1)
    usable = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
            if quota != 'max':
                usable = min(usable, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return model, socket_cores, usable


def gpus_without_touching_hip():
    """GPU count of this node from the kernel driver's topology files (agents with SIMDs): the process that
    starts the ranks must not initialise the HIP runtime itself.  None when the files cannot be read."""
    import glob
    n, seen = 0, False
    for path in glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties'):
        try:
            with open(path) as f:
                seen = True
                for line in f:
                    if line.startswith('simd_count') and int(line.split()[1]) > 0:
                        n += 1
        except (OSError, ValueError, IndexError):
            continue
    return n if seen else None


def _time_oracle(cpu_batch, vocab, budget_s, max_steps):
    from oracle import psvae_oracle as O
    d = O.default_dims(vocab)
    tr = O.Trainer(O.make_params(d), d)
    B = cpu_batch['x'].shape[0]
    eps = torch.randn(B, d['latent'])
    tr.step(cpu_batch, eps)  # warm-up
    t0, n = time.perf_counter(), 0
    while n < max_steps and (time.perf_counter() - t0 < budget_s or n == 0):
        tr.step(cpu_batch, eps)
        n += 1
    return B, n, time.perf_counter() - t0


def cpu_baseline(cpu_batch, qm9_batch, vocab, budget_s=7.0, max_steps=8):
    """The CPU oracle's training step (the reference's op sequence in plain PyTorch) on the host cores, at two thread
    counts: 8, and the physical cores of one socket - as far as this process may use them (its affinity mask and
    cgroup quota; more threads than that only queue behind each other).  The benchmark's own batch, and the
    reference's CPU-runnable case (QM9-shape, 32 graphs) beside it.  The headline `value` is the better of the two."""
    model_name, socket_cores, usable = _cpu_info()
    runs = []
    for threads in sorted({min(8, usable), min(socket_cores, usable)}):
        torch.set_num_threads(threads)
        B, n, dt = _time_oracle(cpu_batch, vocab, budget_s, max_steps)
        Bq, nq, dtq = _time_oracle(qm9_batch, vocab, 2.5, 30)
        runs.append({'cores': threads, 'value': B * n / dt, 'unit': 'molecules/s',
                     'sample': f'{n} training steps of the CPU oracle on one batch of {B} synthetic ZINC-shape graphs '
                               f'({dt / n * 1e3:.0f} ms/step, torch {torch.__version__} CPU)',
                     'qm9_b32': {'value': Bq * nq / dtq, 'unit': 'molecules/s',
                                 'sample': f'{nq} steps on one batch of {Bq} QM9-shape graphs ({dtq / nq * 1e3:.1f} ms/step)'}})
    best = max(runs, key=lambda r: r['value'])
    return {'value': best['value'], 'unit': 'molecules/s', 'cores': best['cores'], 'kind': 'port', 'cpu_model': model_name,
            'physical_cores_per_socket': socket_cores, 'cpus_usable_by_this_process': usable,
            'sample': best['sample'], 'qm9_b32': best['qm9_b32'], 'runs': runs,
            'note': 'torch_geometric cannot be installed here (no network): the oracle restates GINEConv with '
                    'index_select + index_add_, everything else is the op sequence of the reference modules'}


def step_flops(sz, dm, backward=True):
    """Algorithmic flops of one training step over a batch of these sizes (products only, 2 flops per multiply-add):
    both GINE stacks (node_trans + t applications of the two-layer MLP), pooling head, latent head, GRU + vocabulary
    logits, link scorer, property head; backward counts twice the forward (input and weight gradients)."""
    n, B, S, T = sz['n'], sz['B'], sz['S'], sz['L'] - 1
    H, G, Z, R = dm.hidden, dm.graph, dm.latent, dm.rnn
    Din, Dp, V2 = dm.d_atom + dm.d_piece + dm.d_pos, dm.d_piece, dm.num_piece
    P, P2 = 2 * H + Z, (2 * H + Z) // 2
    stacks = n * (2.0 * Din * H * 2 + (dm.t_enc + dm.t_dec) * 4.0 * H * H)
    heads = B * (2.0 * dm.t_enc * H * G + 4.0 * G * Z + 2.0 * Z * R)
    piece = B * T * (6.0 * R * (Dp + R) + 2.0 * R * V2)
    scorer = S * 2.0 * (P * P2 + P2 * P2 + P2 * P + 4.0 * P)
    pred = B * 2.0 * (Z * dm.pred_hidden + dm.pred_hidden ** 2 + dm.pred_hidden * dm.n_props)
    fwd = stacks + heads + piece + scorer + pred
    return fwd * (3.0 if backward else 1.0)


def encode_bulk(dev, graphs=4096, calls=10):
    """Bulk latent encoding (`PSVAEModel.encode` = the deterministic part of `get_z`) in batches of 4096 ZINC-shape
    graphs: the encoder-only configuration of the benchmark list (1 M graphs = 245 such calls)."""
    V = 300
    tok = Tokenizer.synthetic(V)
    torch.manual_seed(2021)
    model = PSVAEModel(full_config(default_args(), tok), tok).to(dev)
    random.seed(1)
    ds = PackedMolDataset(synth.make_molecules(11, 'zinc', V, 2 * graphs), tok)
    t = time.perf_counter()
    batches = [ds.collate_fn(list(range(i * graphs, (i + 1) * graphs))) for i in range(2)]
    host_ms = (time.perf_counter() - t) / 2 * 1e3
    batches = [to_device(b, dev) for b in batches]
    for b in batches:
        model.encode(b)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(calls):
        model.encode(batches[i % 2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / calls
    sz, dm = batches[0]['_native'].sizes, model._dims
    H, Din = dm.hidden, dm.d_atom + dm.d_piece + dm.d_pos
    flops = sz['n'] * (2.0 * Din * H + dm.t_enc * 4.0 * H * H) + graphs * (2.0 * dm.t_enc * H * dm.graph + 4.0 * dm.graph * dm.latent)
    return {'workload': f'encoder-only latent encoding, {graphs} ZINC-shape graphs per call ({sz["n"]} atoms, {sz["E"]} directed bonds), '
                        'vocab 300, hidden 300', 'calls': calls, 'ms_per_call': dt * 1e3, 'value': graphs / dt, 'unit': 'graphs/s',
            'seconds_per_million_graphs': 1e6 / (graphs / dt), 'TFLOPs': flops / dt / 1e12,
            'frac_fp32_mfma_peak': flops / dt / 1e12 / MFMA_F32_PEAK_TF, 'host_collate_ms_per_batch_packed_dataset': host_ms}


def stress_config(dev, steps=5):
    """The stress configuration of the benchmark list: 32 graphs of 60 atoms, 8 GINE iterations in both stacks, hidden
    512 (link scorer 1080 -> 540 -> 540 -> 1080 -> 4), EVERY cross-piece pair scored; one full training step."""
    V, B = 300, 32
    tok = Tokenizer.synthetic(V)
    torch.manual_seed(2021)
    cfg = full_config(default_args(node_hidden_dim=512), tok)
    cfg['encoder']['t'] = 8
    cfg['vae_piece_decoder']['t'] = 8
    model = PSVAEModel(cfg, tok).to(dev)
    opt = model.configure_optimizers()
    ds = PackedMolDataset(synth.make_molecules(12, 'stress', V, 2 * B), tok)
    batches = [to_device(ds.collate_fn(list(range(i * B, (i + 1) * B)), all_pairs=True), dev) for i in range(2)]

    def step(i):
        opt.zero_grad()
        model.train_step(batches[i % 2], opt)
        model.global_step += 1
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    sz = batches[0]['_native'].sizes
    flops = step_flops(sz, model._dims)
    return {'workload': f'training step, {B} graphs x 60 atoms, hidden 512, 8 GINE iterations per stack, all {sz["S"]} cross-piece '
                        'pairs scored, vocab 300', 'steps': steps, 'ms_per_step': dt * 1e3, 'value': B / dt, 'unit': 'graphs/s',
            'GFLOP_per_step': flops / 1e9, 'TFLOPs': flops / dt / 1e12, 'frac_fp32_mfma_peak': flops / dt / 1e12 / MFMA_F32_PEAK_TF}


def device_collate_ms(mols, tok, dev, batch_size, iters=20):
    """Milliseconds per batch of the GPU batch builder (data set resident in HBM), the read-back of the pair
    count included: what a training loop pays per batch on this path instead of the host collate."""
    from ps_vae_amd.device_data import DeviceMolDataset
    ds = DeviceMolDataset(mols, tok, dev)
    rng = random.Random(0)
    picks = [rng.sample(range(len(ds)), batch_size) for _ in range(iters + 2)]
    for p in picks[:2]:
        ds.collate(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k, p in enumerate(picks[2:]):
        ds.collate(p, epoch=k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def bulk_message_passing(dev, graphs=4096, H=300, iters=40):
    """The aggregation kernels at the bulk-encoding size (4096 ZINC-shape graphs, about 95 k atoms), where a
    launch moves a few hundred MB instead of the 7-11 MB of a 128-graph training step and is therefore
    bound by HBM rather than by launch latency.  Buffers rotate over more than 1 GB so that every launch
    streams from HBM as it does inside an encoder pass (a single buffer set would sit in the 256 MB
    last-level cache).  Timed with events on the stream the kernels are launched on; algorithmic bytes as
    in the per-step classes: forward 2 n 4H + 5E + 4(n+1), backward 3 n 4H + 5E + 4(n+1)."""
    from ps_vae_amd.ops import Csr
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    base = max(1, graphs // 4)
    mols = synth.make_molecules(77, 'zinc', 300, base)
    src, dst, typ, off = [], [], [], 0
    for m in mols:
        for (a, b), ty in zip(m['edge_index'], m['edge_attr']):
            src += [a + off, b + off]; dst += [b + off, a + off]; typ += [ty, ty]
        off += len(m['x'])
    reps = graphs // base
    ei = torch.tensor([src, dst], device=dev)
    ei = torch.cat([ei + r * off for r in range(reps)], 1)
    n = off * reps
    csr = Csr(ei, torch.tensor(typ, device=dev).repeat(reps), n)
    E = csr.E
    K = max(2, int(1.2e9 / (3.0 * n * H * 4)) + 1)
    xs = [torch.randn(n, H, device=dev) for _ in range(K)]
    das = [torch.randn(n, H, device=dev) for _ in range(K)]
    outs = [torch.empty(n, H, device=dev) for _ in range(K)]
    ew, eb = torch.randn(H, 4, device=dev), torch.randn(H, device=dev)
    part = torch.empty(max(1, L.psvae_gine_edge_grad_scratch(n, H)), device=dev)

    def fwd(k):
        _lib.check(L.psvae_gine_aggregate_forward(xs[k].data_ptr(), H, csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.etype.data_ptr(),
                                                  ew.data_ptr(), eb.data_ptr(), 1.0, outs[k].data_ptr(), H, n, H, st))

    def bwd(k):
        _lib.check(L.psvae_gine_aggregate_backward(das[k].data_ptr(), H, xs[k].data_ptr(), H, csr.rowptr_t.data_ptr(),
                                                   csr.col_t.data_ptr(), csr.etype_t.data_ptr(), ew.data_ptr(), eb.data_ptr(), 1.0,
                                                   None, 0, None, outs[k].data_ptr(), H, None, None, part.data_ptr(), n, H, st))

    def copy(k):
        outs[k].copy_(xs[k])

    def timed(fn):
        for i in range(K):
            fn(i)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(iters):
            fn(i % K)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e-3

    tf, tb, tc = timed(fwd), timed(bwd), timed(copy)
    bf, bb = 2.0 * n * 4 * H + 5.0 * E + 4 * (n + 1), 3.0 * n * 4 * H + 5.0 * E + 4 * (n + 1)
    return {'workload': f'{graphs} ZINC-shape graphs per launch ({n} atoms, {E} directed bonds, hidden {H}), HBM-resident rotating buffers',
            'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'gine_aggregate_fwd': {'us': tf * 1e6, 'achieved': bf / tf / 1e9, 'frac': bf / tf / 1e9 / HBM_PEAK_GBS},
            'gine_aggregate_bwd': {'us': tb * 1e6, 'achieved': bb / tb / 1e9, 'frac': bb / tb / 1e9 / HBM_PEAK_GBS},
            'plain_copy_same_rows': {'us': tc * 1e6, 'achieved': 2.0 * n * 4 * H / tc / 1e9, 'frac': 2.0 * n * 4 * H / tc / 1e9 / HBM_PEAK_GBS}}


PMC_FILE = 'profiles/r03_pmc_traffic.json'
FLAGS = None      # the parsed command line (set by main)


def pmc_traffic(kernel_class, args, per_gpu):
    """HBM-side bytes per launch of a kernel class, measured beforehand with rocprofv3 PMC passes of this
    same command (tools/pmc_traffic.py wrote the summary); None when no measurement of this workload exists."""
    if (args.shape, args.vocab, per_gpu) != ('zinc', 300, 128):
        return None
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            return json.load(f)['classes'][kernel_class]['hbm_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def make_sharded_batches(tok, vocab, shape, global_batch, count, seed, rank, world):
    """`count` global batches of `global_batch` synthetic graphs each; this rank's contiguous shard
    (`parallel.shard_range`) of every one, collated.  All ranks walk through all shards in order, so Python's
    `random` stream (the negative-pair draws of stage 3) is consumed exactly as one process collating the shards one
    after the other would consume it: the union of the ranks' batches does not depend on the number of ranks.
    Each batch remembers where its shard starts inside the global batch (`graph_base` keys the epsilon draw)."""
    out, mine = [], []
    for i in range(count):
        mols = synth.make_molecules(seed * 1000 + i, shape, vocab, global_batch)
        random.seed(seed * 7919 + i)
        for q in range(world):
            lo, hi = shard_range(global_batch, q, world)
            step2 = [BPEMolDataset.process_step2(m, tok, lists=False) for m in mols[lo:hi]]
            b = BPEMolDataset.process_step3(step2, tok)
            if q == rank:
                b['_native'].graph_base = lo
                out.append(b)
                mine.append(mols[lo:hi])
    return out, mine


def spawn_ranks(n, argv):
    """`bench.py --gpus N` started without a launcher: this process (which never touches a GPU) starts the N rank
    processes itself, one per GPU, and passes rank 0's result line through on its own standard output."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    code = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0 and code == 0:
                    code = rc
                    for other in pending:       # one rank failed: the others would wait for it in a collective
                        other.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return code


class Leg:
    """One measured configuration: a model, its optimiser, this rank's resident batches and the data-parallel
    exchange.  `step(i)` is one optimiser step (forward, backward, gradient all-reduce, clip + Adam)."""

    def __init__(self, args, dev, rank, world, dist_on, shared_gpu, vocab, per_gpu, global_batch, shape, n_distinct=8):
        self.args, self.dev, self.rank, self.world, self.dist_on, self.shared_gpu = args, dev, rank, world, dist_on, shared_gpu
        self.vocab, self.per_gpu, self.global_batch, self.n_distinct = vocab, per_gpu, global_batch, n_distinct
        tok = self.tok = Tokenizer.synthetic(vocab)
        torch.manual_seed(2021)
        self.model = model = PSVAEModel(full_config(default_args(), tok), tok).to(dev)
        self.opt = model.configure_optimizers()
        t0 = time.perf_counter()
        self.cpu_batches, self.mols = make_sharded_batches(tok, vocab, shape, global_batch, n_distinct, 2021, rank, world)
        self.collate_s = (time.perf_counter() - t0) / (n_distinct * world)      # every rank collated every shard
        self.batches = [to_device(b, dev) for b in self.cpu_batches]
        model.flat_buffers()
        # Global loss denominators (non-pad tokens, pairs, graphs, properties summed over ranks) depend on the batch
        # only.  --denoms selects how they are exchanged: "ride" (default) puts the NEXT batch's four local
        # counts behind the gradients, so that the last gradient message sums them too (no collective of their own);
        # "once" exchanges them once per collated batch before the timed region, as a loader that collates the
        # global batch would.
        self.denoms_mode = FLAGS.denoms if dist_on else 'none'
        tail = 4 if self.denoms_mode == 'ride' else 0
        self.message, self.tail = model.grad_message()
        if not dist_on:
            self.reducer = None
        elif FLAGS.dp_overlap:
            # three gradient messages in the order backward finishes them, the first two overlapped with it
            self.reducer = OverlappedReducer(model, always=True, tail=tail)
        else:
            self.reducer = GradReducer([0, model._n_trained + tail], always=True)     # one message after backward
        self.delivered = {}
        if self.denoms_mode == 'once':
            for b in self.batches:
                nb = b['_native']
                self.reducer.reduce_denoms(nb.local_denoms, out=nb.denoms)
            torch.cuda.synchronize()
        self.unit = model.unit_grad()       # loss.backward(unit): no fill / copy launches for the implicit gradient of 1
        # --autograd: the step through torch.autograd (training_step, backward, optimizer.step) instead of
        # PSVAEModel.train_step - same kernels, more host time
        self.one_call = not FLAGS.autograd

    def step(self, i):
        model, n = self.model, self.n_distinct
        batch = self.batches[i % n]
        nb = batch['_native']
        if self.denoms_mode == 'ride' and not self.delivered.pop(i, False):     # first step only: nothing has carried them yet
            self.reducer.reduce_denoms(nb.local_denoms, out=nb.denoms)
        self.opt.zero_grad()
        # epsilon: counter-based draw keyed by (seed, step, global graph index)
        if self.one_call:       # forward, backward (and, alone on the GPU, clip + Adam) queued by one library call
            loss = model.train_step(batch, None if self.dist_on else self.opt)[5]
        else:
            loss = model.training_step(batch)
            loss.backward(self.unit)
        if self.dist_on:
            if self.shared_gpu:
                torch.cuda.synchronize()
            if self.denoms_mode == 'ride':
                nxt = self.batches[(i + 1) % n]['_native']
                self.tail.copy_(nxt.local_denoms)
            if isinstance(self.reducer, OverlappedReducer):
                self.reducer.reduce()
            else:
                self.reducer.reduce_grads(self.message)
            if self.denoms_mode == 'ride':
                nxt.denoms.copy_(self.tail)
                self.delivered[i + 1] = True
            if self.shared_gpu:
                torch.cuda.synchronize()
        if self.dist_on or not self.one_call:
            self.opt.step()
        model.global_step += 1
        return loss

    def fence(self):
        if self.dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def run(self, steps, warmup):
        """`warmup` untimed steps, then exactly `steps` timed ones between two fences (barrier + device
        synchronise); the time is the maximum over ranks."""
        for i in range(warmup):
            self.step(i)
        self.fence()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            loss = self.step(warmup + i)
            if not FLAGS.no_marks or i == steps - 1:
                marks[i + 1].record()       # per-step spread (median, p10, p90); the headline stays total time / steps
        host_enqueue_s = time.perf_counter() - t0      # how long the host needed to queue the work
        self.fence()
        elapsed = time.perf_counter() - t0
        final = loss.detach().double().reshape(1).clone()
        if self.dist_on:
            # (host tensors: gloo's own path for device tensors stalls when ranks share a GPU; RCCL is given device tensors)
            on_dev = torch.distributed.get_backend() == 'nccl'
            t = torch.tensor([elapsed], device=self.dev if on_dev else 'cpu', dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t)
            # a rank's loss is its share of the global batch's loss (local sums over global counts)
            final = final if on_dev else final.cpu()
            torch.distributed.all_reduce(final, op=torch.distributed.ReduceOp.SUM)
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps)) if not FLAGS.no_marks else [elapsed / steps * 1e3] * steps
        spread = {'median': per_step[len(per_step) // 2], 'p10': per_step[int(0.1 * (len(per_step) - 1))],
                  'p90': per_step[int(0.9 * (len(per_step) - 1))]}
        if self.model.recurrence_gave_up():
            raise SystemExit('a step reported invalid results during the run')
        self.next_step = warmup + steps
        return {'elapsed': elapsed, 'ms_per_step': elapsed / steps * 1e3, 'value': self.global_batch * steps / elapsed,
                'step_ms': spread, 'host_enqueue_ms_per_step': host_enqueue_s / steps * 1e3, 'final_loss': float(final),
                # recurrences redone on one compute unit after the multi-CU kernels stopped waiting (0 on a GPU of our own)
                'gru_recurrences_redone': self.model.recurrence_redone()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch-size', type=int, default=128, help='graphs per GPU (weak scaling)')
    ap.add_argument('--vocab', type=int, default=300)
    ap.add_argument('--shape', default='zinc')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='weak: --batch-size graphs per GPU; strong: --global-batch graphs split over the GPUs')
    ap.add_argument('--global-batch', type=int, default=1024, help='graphs per step over all GPUs (strong scaling)')
    ap.add_argument('--config3-steps', type=int, default=40,
                    help='timed steps of the extra leg "vocab 500, global batch 1024 split over the GPUs" reported under '
                         '"strong_scaling" (0 = skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-bulk', action='store_true', help='skip the aggregation kernels at the bulk-encoding size')
    ap.add_argument('--no-configs', action='store_true', help='skip the encoder-only (4096 graphs per call) and stress '
                    '(60 atoms, hidden 512, 8 iterations, all pairs) configurations')
    ap.add_argument('--profile-steps', type=int, default=10)
    ap.add_argument('--denoms', choices=('ride', 'once'), default='ride', help='data-parallel runs: the loss denominators '
                    'ride behind the gradients of the previous step / are exchanged once per batch before the timed region')
    ap.add_argument('--dp-overlap', type=int, default=1, help='0: one gradient message after backward instead of three '
                    'overlapped with it')
    ap.add_argument('--autograd', action='store_true', help='the step through torch.autograd instead of the one-call train_step')
    ap.add_argument('--no-marks', action='store_true', help='no per-step event records inside the timed region')
    ap.add_argument('--watchdog', type=int, default=0, help='seconds after which all stacks are dumped and the run ends')
    ap.add_argument('--force-dist', action='store_true', help='run the collectives even with one rank')
    ap.add_argument('--late-streams', action='store_true', help='test switch: create the communicator before the side streams')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend; 'gloo' rehearses the multi-rank "
                    "path on a box with fewer GPUs than ranks (ranks then share devices)")
    args = ap.parse_args()
    global FLAGS
    FLAGS = args
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher around us: start the ranks here, before anything in this process touches a GPU
        # (counted from the driver's topology files: this process must not initialise HIP; the ranks check again)
        n_gpus = gpus_without_touching_hip()
        if args.backend == 'nccl' and n_gpus is not None and n_gpus < args.gpus:
            raise SystemExit(f'{args.gpus} ranks over RCCL need {args.gpus} GPUs, this node shows {n_gpus} '
                             "(--backend gloo rehearses the multi-rank path on fewer)")
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    # The result line is the ONLY thing this program writes to its standard output: libraries that chat on file
    # descriptor 1 (RCCL prints a five-line version banner there when its first communicator is created) are sent
    # to standard error for the whole run, and the JSON line goes to a private copy of the original descriptor.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    if args.watchdog > 0:     # seconds; on expiry every thread's stack goes to stderr and the run ends
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but the launcher started {world} ranks')
    n_dev = torch.cuda.device_count()
    if world > 1 and args.backend == 'nccl' and n_dev < world:
        raise SystemExit(f'{world} ranks over RCCL need {world} GPUs, this node shows {n_dev}')
    # Rehearsal only (gloo, fewer GPUs than ranks): processes that share a GPU must not block on a collective
    # while their own queue still holds work that waits for it, or the other process may never get the GPU; and the
    # multi-CU recurrence (workgroups that wait for each other) is replaced by the single-CU kernels there.
    shared_gpu = world > max(n_dev, 1)
    if shared_gpu:
        _lib.set_option('gru_multi_cu', 0)
    dev = torch.device('cuda', local_rank % max(n_dev, 1))
    torch.cuda.set_device(dev)
    # --force-dist: run the collectives even with one rank (rehearses the RCCL path on one GPU)
    dist_on = world > 1 or args.force_dist
    # the step's side streams are created before the communicator's, and checked to run beside the current stream
    def init_streams():
        torch.zeros(1, device=dev)
        torch.cuda.synchronize()
        _lib.check(_lib.lib().psvae_streams_init(torch.cuda.current_stream().cuda_stream), 'psvae_streams_init')
    late_streams = args.late_streams     # test switch: communicator first
    if not late_streams:
        init_streams()
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
    if late_streams:
        init_streams()

    strong = args.scaling == 'strong'
    if strong and args.global_batch % world:
        raise SystemExit('--global-batch must be a multiple of the number of GPUs')
    per_gpu = args.global_batch // world if strong else args.batch_size
    global_batch = per_gpu * world
    leg = Leg(args, dev, rank, world, dist_on, shared_gpu, args.vocab, per_gpu, global_batch, args.shape)
    res = leg.run(args.steps, args.warmup)
    model, tok = leg.model, leg.tok

    # roofline pass: same steps with every launch bracketed by HIP events on its stream
    L = _lib.lib()
    L.psvae_profile_enable(1)
    for i in range(args.profile_steps):
        leg.step(leg.next_step + i)
    torch.cuda.synchronize()
    raw = _lib.profile_collect()
    L.psvae_profile_enable(0)
    # an event pair around a launch also times the dispatch of the kernel and of the closing marker: calibrate
    # that on an empty kernel and take it off every launch, so durations are kernel time as a profiler reports it
    ev_ms = max(0.0, float(L.psvae_profile_event_overhead_ms(torch.cuda.current_stream().cuda_stream)))
    classes = {k: (max(ms - cnt * ev_ms, 0.05 * ms), work, cnt) for k, (ms, work, cnt) in raw.items()}

    # the other curve: vocab 500, global batch 1024 split over the GPUs (1024 / N graphs per GPU)
    strong_leg = None
    if args.config3_steps > 0 and not (strong and args.vocab == 500 and args.global_batch == 1024) and 1024 % world == 0:
        leg3 = Leg(args, dev, rank, world, dist_on, shared_gpu, 500, 1024 // world, 1024, 'zinc', n_distinct=4)
        r3 = leg3.run(args.config3_steps, max(2, min(args.warmup, 10)))
        strong_leg = {'workload': f'ZINC250K-shape training step, vocab 500, global batch 1024 = {1024 // world} graphs per GPU',
                      'scaling': 'strong', 'n_gpus': world, 'value': r3['value'], 'unit': 'molecules/s',
                      'ms_per_step': r3['ms_per_step'], 'steps': args.config3_steps, 'step_ms': r3['step_ms'],
                      'host_enqueue_ms_per_step': r3['host_enqueue_ms_per_step'], 'final_loss': r3['final_loss'],
                      'params': int(leg3.model._n_trained)}
        del leg3

    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    ms_per_step = res['ms_per_step']
    single = world == 1 and not dist_on
    dev_collate_ms = device_collate_ms([m for ms in leg.mols for m in ms], tok, dev, per_gpu) if single and not args.no_bulk else None
    comm = 'RCCL' if args.backend == 'nccl' else args.backend
    table = {}
    for name, (ms, work, cnt) in classes.items():
        per_launch_us = ms / cnt * 1e3
        rate = work / (ms * 1e-3) if ms > 0 else 0.0
        if name in FLOP_CLASSES:
            table[name] = {'us_per_step': ms / args.profile_steps * 1e3, 'launches_per_step': cnt / args.profile_steps,
                           'avg_us': per_launch_us, 'TFLOPs': rate / 1e12}
        else:
            table[name] = {'us_per_step': ms / args.profile_steps * 1e3, 'launches_per_step': cnt / args.profile_steps,
                           'avg_us': per_launch_us, 'GBs': rate / 1e9}
    if not classes:
        raise SystemExit('--profile-steps must be at least 1: the roofline object is measured there')
    dom = max(classes, key=lambda k: classes[k][0])
    dms, dwork, dcnt = classes[dom]
    if dom in FLOP_CLASSES:
        achieved, peak, unit, bound = dwork / (dms * 1e-3) / 1e12, MFMA_F32_PEAK_TF, 'TFLOP/s', 'mfma'
    else:
        achieved, peak, unit, bound = dwork / (dms * 1e-3) / 1e9, HBM_PEAK_GBS, 'GB/s', 'hbm'
    traffic = pmc_traffic(dom, args, per_gpu)
    roofline = {'kernel': dom, 'bound': bound, 'achieved': achieved, 'peak': peak, 'unit': unit, 'frac': achieved / peak,
                'traffic': traffic, 'traffic_unit': 'HBM-side bytes per launch (rocprofv3 PMC, see traffic_source)',
                'traffic_source': PMC_FILE if traffic is not None else None,
                'algorithmic_per_launch': dwork / dcnt, 'avg_launch_us': dms / dcnt * 1e3,
                'event_pair_overhead_us_subtracted': ev_ms * 1e3,
                'launches_per_step': dcnt / args.profile_steps}
    mp = [k for k in ('gine_aggregate_fwd', 'gine_aggregate_bwd') if k in classes]
    mp_ms = sum(classes[k][0] for k in mp)
    mp_bytes = sum(classes[k][1] for k in mp)
    out = {
        'metric': 'train molecules/sec (enc+dec fwd+bwd), ZINC-shape', 'value': res['value'], 'unit': 'molecules/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'step_ms': res['step_ms'], 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic', 'rccl_ranks': world if (dist_on and args.backend == 'nccl') else 0,
        'config': {'workload': f'ZINC250K-shape training step, vocab {args.vocab}, hidden 300, batch {per_gpu} per GPU '
                               f'(fwd + bwd + clip + Adam{f" + grad all-reduce over {comm}" if dist_on else ""})',
                   'global_batch': global_batch, 'parallelism': f'dp{world}', 'params': int(model._n_trained),
                   'grad_reduce': (type(leg.reducer).__name__ + f', denominators: {leg.denoms_mode}') if dist_on else None,
                   'final_loss': res['final_loss'], 'host_collate_ms_per_batch': leg.collate_s * 1e3,
                   'device_collate_ms_per_batch': dev_collate_ms,
                   'host_enqueue_ms_per_step': res['host_enqueue_ms_per_step'],
                   'arithmetic': 'fp32 values and fp32 accumulation everywhere; the fused GINE and link-scorer kernels and the '
                                 'weight-gradient products over fewer than 18 k rows form each fp32 product from exact three-way '
                                 'bf16 cuts of both operands (six v_mfma_*_bf16 per block), error against float64 at or below '
                                 'that of plain fp32 arithmetic (tests/test_gpu_ops.py); the other products run on v_mfma_f32_*'},
        'roofline': roofline,
        'message_passing': {'kernels': mp, 'bound': 'hbm', 'achieved': (mp_bytes / (mp_ms * 1e-3) / 1e9) if mp_ms else None,
                            'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': (mp_bytes / (mp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if mp_ms else None},
        'kernel_classes': table,
        'gru_recurrences_redone': res['gru_recurrences_redone'],
    }
    if 'gine_layer_fused' in classes:
        # aggregation fused into the MLP kernel: reported against max(HBM fraction, MFMA fraction), the HBM numerator
        # being the aggregation's compulsory bytes 2 n 4H + 5 E + 4 (n + 1) per stack (encoder: all bonds, decoder-side
        # stack: in-piece bonds).  Only when every launch of the class is a forward application (the fused backward of
        # the MLPs, which larger batches add to the class, is no aggregation).
        fms, fwork, fcnt = classes['gine_layer_fused']
        sizes = [b['_native'].sizes for b in leg.batches]
        t_layers = int(model._dims.t_enc)
        H = int(model._dims.hidden)
        # per stack: read x, write agg, hid and y (the training launch saves both intermediates) + CSR
        agg_bytes = sum(2 * (4.0 * z['n'] * 4 * H + 4.0 * (z['n'] + 1)) + 5.0 * (z['E'] + z['E_in']) for z in sizes) / len(sizes)
        if abs(fcnt / args.profile_steps - t_layers) < 0.5:
            hbm = agg_bytes / (fms / fcnt * 1e-3) / 1e9
            mf = fwork / (fms * 1e-3) / 1e12
            out['fused_layer'] = {'kernel': 'gine_layer_fused', 'avg_launch_us': fms / fcnt * 1e3,
                                  'mfma': {'achieved': mf, 'peak': MFMA_F32_PEAK_TF, 'unit': 'TFLOP/s', 'frac': mf / MFMA_F32_PEAK_TF,
                                           'instruction': 'v_mfma_f32_16x16x32_bf16, six per fp32 product block (operands cut into '
                                                          'three bf16 pieces; fp32-accurate, see tests/test_gpu_ops.py)',
                                           'peak_cut_operands': MFMA_CUT_PEAK_TF, 'frac_cut_operands': mf / MFMA_CUT_PEAK_TF},
                                  'hbm': {'achieved': hbm, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': hbm / HBM_PEAK_GBS,
                                          'algorithmic_bytes_per_launch': agg_bytes},
                                  'frac': max(mf / MFMA_F32_PEAK_TF, hbm / HBM_PEAK_GBS)}
    if strong_leg is not None:
        out['strong_scaling'] = strong_leg
    if single and not args.no_bulk:
        out['message_passing']['bulk'] = bulk_message_passing(dev)
    if single and not args.no_configs:
        out['encode_bulk'] = encode_bulk(dev)
        out['stress'] = stress_config(dev)
    if single and not args.no_cpu_baseline:
        host_times = {}
        make_batches(tok, args.vocab, args.shape, per_gpu, 2, seed=2021, host_times=host_times)
        out['config']['host_stage2_plus_3_ms_per_batch'] = host_times.get('per_item_ms')
        out['config']['host_packed_dataset_ms_per_batch'] = host_times.get('packed_ms')
        qm9, _ = make_batches(tok, args.vocab, 'qm9', 32, 1, seed=7)
        out['cpu_baseline'] = cpu_baseline(leg.cpu_batches[0], qm9[0], args.vocab)
    result_out.write(json.dumps(out) + '\n')
    result_out.flush()


if __name__ == '__main__':
    main()
