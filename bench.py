#!/usr/bin/env python3
"""Headline benchmark: PS-VAE training step (encoder + decoder forward + backward,
global-norm clip + Adam) on synthetic ZINC250K-shape graphs.

    python bench.py --gpus N --steps K --warmup W

One "step" = one optimiser step over one batch of 128 graphs per GPU (vocab 300,
hidden 300: the configuration the metric is quoted on).  Batches are collated on
the host beforehand and resident in HBM when the timed region starts; parameters
are randomly initialised (no checkpoints exist offline).  With N > 1 every rank
processes its own 128 graphs (weak scaling) and the flat gradient buffer is
all-reduced over RCCL each step.  Rank 0 prints ONE JSON line.
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
from ps_vae_amd.parallel import GradReducer, OverlappedReducer  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec
MFMA_F32_PEAK_TF = 157.3   # MI355X dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
FLOP_CLASSES = ('gemm_forward', 'gemm_input_grad', 'gemm_weight_grad', 'gru_forward', 'gru_backward')


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


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


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


def cpu_baseline(cpu_batch, qm9_batch, vocab, budget_s=12.0, max_steps=8):
    """The CPU oracle's training step (the reference's op sequence in plain PyTorch) on the host cores:
    the benchmark's own batch, and the reference's CPU-runnable case (QM9-shape, 32 graphs) beside it."""
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    B, n, dt = _time_oracle(cpu_batch, vocab, budget_s, max_steps)
    Bq, nq, dtq = _time_oracle(qm9_batch, vocab, 4.0, 30)
    return {'value': B * n / dt, 'unit': 'molecules/s', 'cores': threads, 'kind': 'port', 'cpu_model': _cpu_model(),
            'sample': f'{n} training steps of the CPU oracle on one batch of {B} synthetic ZINC-shape graphs '
                      f'({dt / n * 1e3:.0f} ms/step, torch {torch.__version__} CPU)',
            'qm9_b32': {'value': Bq * nq / dtq, 'unit': 'molecules/s',
                        'sample': f'{nq} steps on one batch of {Bq} QM9-shape graphs ({dtq / nq * 1e3:.1f} ms/step)'}}


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


PMC_FILE = 'profiles/r01_pmc_traffic.json'


def pmc_traffic(kernel_class, args):
    """HBM-side bytes per launch of a kernel class, measured beforehand with rocprofv3 PMC passes of this
    same command (tools/pmc_traffic.py wrote the summary); None when no measurement of this workload exists."""
    if (args.shape, args.vocab, args.batch_size) != ('zinc', 300, 128):
        return None
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            return json.load(f)['classes'][kernel_class]['hbm_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch-size', type=int, default=128, help='graphs per GPU')
    ap.add_argument('--vocab', type=int, default=300)
    ap.add_argument('--shape', default='zinc')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-bulk', action='store_true', help='skip the aggregation kernels at the bulk-encoding size')
    ap.add_argument('--profile-steps', type=int, default=10)
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend; 'gloo' rehearses the multi-rank "
                    "path on a box with fewer GPUs than ranks (ranks then share devices)")
    args = ap.parse_args()
    # The result line is the ONLY thing this program writes to its standard output: libraries that chat on file
    # descriptor 1 (RCCL prints a five-line version banner there when its first communicator is created) are sent
    # to standard error for the whole run, and the JSON line goes to a private copy of the original descriptor.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    if os.environ.get('PSVAE_BENCH_WATCHDOG'):     # seconds; on expiry every thread's stack goes to stderr and the run ends
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['PSVAE_BENCH_WATCHDOG']), exit=True)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    n_dev = torch.cuda.device_count()
    if world > 1 and args.backend == 'nccl' and n_dev < world:
        raise SystemExit(f'{world} ranks over RCCL need {world} GPUs, this node shows {n_dev}')
    dev = torch.device('cuda', local_rank % max(n_dev, 1))
    torch.cuda.set_device(dev)
    # Rehearsal only (gloo, fewer GPUs than ranks): processes that share a GPU must not block on a collective
    # while their own queue still holds work that waits for it, or the other process may never get the GPU.
    shared_gpu = world > max(n_dev, 1)
    # PSVAE_BENCH_FORCE_DIST=1: run the collectives even with one rank (rehearses the RCCL path on one GPU)
    dist_on = world > 1 or os.environ.get('PSVAE_BENCH_FORCE_DIST') == '1'
    # the step's side streams are created before the communicator's, and checked to run beside the current stream
    def init_streams():
        torch.zeros(1, device=dev)
        torch.cuda.synchronize()
        _lib.check(_lib.lib().psvae_streams_init(torch.cuda.current_stream().cuda_stream), 'psvae_streams_init')
    late_streams = os.environ.get('PSVAE_BENCH_LATE_STREAMS') == '1'     # test switch: communicator first
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
    tok = Tokenizer.synthetic(args.vocab)
    torch.manual_seed(2021)
    model = PSVAEModel(full_config(default_args(), tok), tok).to(dev)
    opt = model.configure_optimizers()
    n_distinct = 8
    host_times = {}
    cpu_batches, collate_s = make_batches(tok, args.vocab, args.shape, args.batch_size, n_distinct, seed=2021 + rank,
                                          host_times=host_times)
    batches = [to_device(b, dev) for b in cpu_batches]
    model.flat_buffers()
    reducer = GradReducer([0, model._n_trained], always=dist_on)     # one collective for the whole gradient (7.6 MB)
    # PSVAE_BENCH_OVERLAP=1: two gradient messages, the first one overlapped with the rest of backward
    overlap = OverlappedReducer(model, always=dist_on) if dist_on and os.environ.get('PSVAE_BENCH_OVERLAP') == '1' else None
    local_denoms = [b['_native'].denoms.clone() for b in batches]
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)

    # Global loss denominators (non-pad tokens, pairs, graphs, properties summed over ranks) are exchanged every
    # step, one step ahead (they depend on the batch only).  PSVAE_BENCH_DENOMS selects how: "ride" (default) puts
    # the next batch's four local counts behind the gradients, so that the gradient all-reduce sums them too;
    # "async" uses a collective of their own on the communicator's stream; "once" exchanges them once per collated
    # batch before the timed region, as a loader that collates the global batch would.
    denoms_mode = os.environ.get('PSVAE_BENCH_DENOMS', 'ride') if dist_on else 'none'
    if overlap is not None and denoms_mode == 'ride':
        denoms_mode = 'async'       # the overlapped reducer sends the gradients as two messages of its own
    denoms_per_step = denoms_mode == 'async'
    if denoms_mode == 'ride':
        message, tail = model.grad_message()
        message_reducer = GradReducer([0, message.numel()], always=dist_on)
        delivered = {}              # step index -> its batch holds the global denominators
    if denoms_mode == 'once':
        for j in range(n_distinct):
            reducer.reduce_denoms(local_denoms[j], out=batches[j]['_native'].denoms)
        torch.cuda.synchronize()

    pending = {}

    def exchange_denoms(i):
        # global loss denominators of batch i: needs only the batch, so it is issued one step ahead
        if shared_gpu:
            torch.cuda.synchronize()
        j = i % n_distinct
        pending[i] = reducer.reduce_denoms(local_denoms[j], out=batches[j]['_native'].denoms, async_op=True)[1]

    unit = model.unit_grad()       # loss.backward(unit): no fill / copy launches for the implicit gradient of 1

    def step(i):
        batch = batches[i % n_distinct]
        if denoms_per_step:
            if i not in pending:
                exchange_denoms(i)
            work = pending.pop(i)
            if work is not None:
                work.wait()
        if denoms_mode == 'ride' and not delivered.pop(i, False):     # first step only: nothing has carried them yet
            reducer.reduce_denoms(local_denoms[i % n_distinct], out=batch['_native'].denoms)
        eps = torch.randn(args.batch_size, 56, device=dev, generator=gen)
        opt.zero_grad()
        loss = model.training_step(batch, eps=eps)
        loss.backward(unit)
        if dist_on:
            if shared_gpu:
                torch.cuda.synchronize()
            if overlap is not None:
                overlap.reduce()
            elif denoms_mode == 'ride':
                nxt = (i + 1) % n_distinct
                tail.copy_(local_denoms[nxt])
                message_reducer.reduce_grads(message)
                batches[nxt]['_native'].denoms.copy_(tail)
                delivered[i + 1] = True
            else:
                reducer.reduce_grads(model.flat_buffers()[1])
            if shared_gpu:
                torch.cuda.synchronize()
            if denoms_per_step:
                exchange_denoms(i + 1)
        opt.step()
        model.global_step += 1
        return loss

    def fence():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step(args.warmup + i)
        marks[i + 1].record()       # per-step spread (median, p10, p90); the headline stays total time / steps
    host_enqueue_s = time.perf_counter() - t0      # how long the host needed to queue the work
    fence()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss.detach())
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    spread = {'median': per_step[len(per_step) // 2], 'p10': per_step[int(0.1 * (len(per_step) - 1))],
              'p90': per_step[int(0.9 * (len(per_step) - 1))]}
    if model.recurrence_gave_up():
        raise SystemExit('the multi-CU GRU recurrence timed out during the run: results are invalid')

    # roofline pass: same steps with every launch bracketed by HIP events on its stream
    L = _lib.lib()
    L.psvae_profile_enable(1)
    for i in range(args.profile_steps):
        step(args.warmup + args.steps + i)
    torch.cuda.synchronize()
    raw = _lib.profile_collect()
    L.psvae_profile_enable(0)
    # an event pair around a launch also times the dispatch of the kernel and of the closing marker: calibrate
    # that on an empty kernel and take it off every launch, so durations are kernel time as a profiler reports it
    ev_ms = max(0.0, float(L.psvae_profile_event_overhead_ms(torch.cuda.current_stream().cuda_stream)))
    classes = {k: (max(ms - cnt * ev_ms, 0.05 * ms), work, cnt) for k, (ms, work, cnt) in raw.items()}

    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    dev_collate_ms = device_collate_ms(host_times['mols'], tok, dev, args.batch_size) if world == 1 and not args.no_bulk else None
    comm = 'RCCL' if args.backend == 'nccl' else args.backend
    value = world * args.batch_size * args.steps / elapsed
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
    roofline = {'kernel': dom, 'bound': bound, 'achieved': achieved, 'peak': peak, 'unit': unit, 'frac': achieved / peak,
                'traffic': pmc_traffic(dom, args), 'traffic_unit': 'HBM-side bytes per launch (rocprofv3 PMC, see traffic_source)',
                'traffic_source': PMC_FILE if pmc_traffic(dom, args) is not None else None,
                'algorithmic_per_launch': dwork / dcnt, 'avg_launch_us': dms / dcnt * 1e3,
                'event_pair_overhead_us_subtracted': ev_ms * 1e3,
                'launches_per_step': dcnt / args.profile_steps}
    mp = [k for k in ('gine_aggregate_fwd', 'gine_aggregate_bwd') if k in classes]
    mp_ms = sum(classes[k][0] for k in mp)
    mp_bytes = sum(classes[k][1] for k in mp)
    out = {
        'metric': 'train molecules/sec (enc+dec fwd+bwd), ZINC-shape', 'value': value, 'unit': 'molecules/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'step_ms': spread, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'ZINC250K-shape training step, vocab {args.vocab}, hidden 300, batch {args.batch_size} per GPU '
                               f'(fwd + bwd + clip + Adam{f" + grad all-reduce over {comm}" if world > 1 else ""})',
                   'global_batch': world * args.batch_size, 'parallelism': f'dp{world}', 'params': int(model._n_trained),
                   'final_loss': final_loss, 'host_collate_ms_per_batch': collate_s * 1e3,
                   'host_stage2_plus_3_ms_per_batch': host_times.get('per_item_ms'),
                   'host_packed_dataset_ms_per_batch': host_times.get('packed_ms'),
                   'device_collate_ms_per_batch': dev_collate_ms,
                   'host_enqueue_ms_per_step': host_enqueue_s / args.steps * 1e3},
        'roofline': roofline,
        'message_passing': {'kernels': mp, 'bound': 'hbm', 'achieved': (mp_bytes / (mp_ms * 1e-3) / 1e9) if mp_ms else None,
                            'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': (mp_bytes / (mp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if mp_ms else None},
        'kernel_classes': table,
    }
    if world == 1 and not args.no_bulk:
        out['message_passing']['bulk'] = bulk_message_passing(dev)
    if world == 1 and not args.no_cpu_baseline:
        qm9, _ = make_batches(tok, args.vocab, 'qm9', 32, 1, seed=7)
        out['cpu_baseline'] = cpu_baseline(cpu_batches[0], qm9[0], args.vocab)
    result_out.write(json.dumps(out) + '\n')
    result_out.flush()


if __name__ == '__main__':
    main()
