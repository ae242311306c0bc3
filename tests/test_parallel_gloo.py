"""Data-parallel scheme on two CPU processes over gloo: per-rank loss sums divided by
the all-reduced denominators plus a SUM all-reduce of the flat gradient buffer must
equal the gradient of the undivided global batch.  The arithmetic of each rank is the
CPU oracle (float64, so the comparison is limited by summation order only); the
collective plumbing is `ps_vae_amd.parallel`, the code the GPU ranks run."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import psvae_oracle as O  # noqa: E402
from ps_vae_amd import Tokenizer, synth  # noqa: E402
from ps_vae_amd.data import BPEMolDataset  # noqa: E402
from ps_vae_amd.parallel import GradReducer, global_step_reference, shard_range  # noqa: E402

VOCAB, N_MOLS = 40, 7
DIMS = dict(hidden=32, graph=48, latent=8, rnn=24, pred_hidden=16, d_atom=8, d_piece=12, d_pos=4, t_enc=2, t_dec=2)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _collate(mols, tok):
    step2 = [BPEMolDataset.process_step2(m, tok) for m in mols]
    return BPEMolDataset.process_step3(step2, tok, all_pairs=True)   # no sampling: shards and whole batch agree


def _loss_sums(p, d, batch, eps):
    """Loss SUMS of one shard and the counts they are means over: [tokens, pairs, graphs, graphs*props]."""
    losses = O.forward(p, d, batch, eps)
    counts = batch['_native'].denoms.double()
    sums = {'piece': losses['piece'] * counts[0], 'edge': losses['edge'] * counts[1],
            'kl': losses['kl'] * counts[2], 'pred': losses['pred'] * counts[3]}
    return sums, counts


def _total(sums, denoms, step):
    scaled = {'pred': sums['pred'] / denoms[3], 'kl': sums['kl'] / denoms[2],
              'rec': sums['piece'] / denoms[0] + sums['edge'] / denoms[1]}
    return O.weighted_loss(scaled, step), scaled


def _setup(seed=91, n_mols=N_MOLS):
    d = O.default_dims(VOCAB, **DIMS)
    tok = Tokenizer.synthetic(VOCAB)
    mols = synth.make_molecules(seed, 'qm9', VOCAB, n_mols)
    p = {k: v.double() for k, v in O.make_params(d).items()}
    names = O.trainable_names(d)
    for n in names:
        p[n].requires_grad_(True)
    eps = torch.randn(n_mols, d['latent'], generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    return d, tok, mols, p, names, eps


def _model_layout(d, tok):
    """The product's gradient message for a model of these sizes: parameter names in flat-buffer order, their
    offsets (each tensor padded to a multiple of four floats), the four bucket bounds and the message length (four
    spare floats behind the gradients carry the next batch's loss denominators)."""
    from tests.util import config_from_dims
    from ps_vae_amd import PSVAEModel
    model = PSVAEModel(config_from_dims(d, tok), tok)
    message, tail = model.grad_message()
    return model.trained_names(), dict(model._offsets), list(model.bucket_bounds), message.numel()


def _message_grads(p, layout):
    names, offs, _, length = layout
    out = torch.zeros(length, dtype=torch.float64)
    for k in names:
        g = p[k].grad
        if g is not None:
            out[offs[k]:offs[k] + g.numel()] = g.flatten()
    return out


def _flat_grads(p, names):
    return torch.cat([(p[n].grad if p[n].grad is not None else torch.zeros_like(p[n])).flatten() for n in names])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        d, tok, mols, p, names, eps = _setup()
        lo, hi = shard_range(N_MOLS, rank, world)
        batch = _collate(mols[lo:hi], tok)
        sums, counts = _loss_sums(p, d, batch, eps[lo:hi])
        n_flat = sum(p[n].numel() for n in names)
        reducer = GradReducer([0, n_flat // 3, n_flat])            # two uneven buckets
        denoms = reducer.reduce_denoms(counts)
        total, _ = _total(sums, denoms, step=1500)
        total.backward()
        flat = _flat_grads(p, names)
        for w in reducer.reduce_grads(flat, async_op=True):
            w.wait()
        share = torch.tensor([float(total.detach())], dtype=torch.float64)
        dist.all_reduce(share)
        torch.save({'flat': flat, 'denoms': denoms, 'total': float(share)}, os.path.join(out_dir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


def _worker_message(rank, world, port, out_dir, n_mols):
    """One rank of a step as `bench.py` / `trainer.fit` drive it: its shard of the global batch, loss sums over the
    all-reduced denominators, the gradient message in the product's flat layout reduced in the three ranges
    `OverlappedReducer` uses, the next batch's four local counts riding behind the gradients."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        d, tok, mols, p, names, eps = _setup(n_mols=n_mols)
        layout = _model_layout(d, tok)
        lo, hi = shard_range(n_mols, rank, world)
        batch = _collate(mols[lo:hi], tok)
        sums, counts = _loss_sums(p, d, batch, eps[lo:hi])
        b = layout[2]
        reducer = GradReducer([b[0], b[1], b[3], b[4] + 4])
        denoms = reducer.reduce_denoms(counts)
        total, _ = _total(sums, denoms, step=1500)
        total.backward()
        message = _message_grads(p, layout)
        message[b[4]:] = counts            # (stands for the NEXT batch's counts)
        reducer.reduce_grads(message)
        torch.save({'message': message, 'denoms': denoms, 'shard': (lo, hi)}, os.path.join(out_dir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_eight_ranks_uneven_global_batch(tmp_path):
    """World size 8 (the node's GPU count) over a global batch of 1001 graphs: shards of 126 and 125 graphs, the four
    denominators, and the gradient message in the product's flat layout reduced in its three ranges with the
    denominators of the next batch riding in the last one.  Every rank must end with the undivided batch's gradient."""
    world, n_mols = 8, 1001
    mp.spawn(_worker_message, args=(world, _free_port(), str(tmp_path), n_mols), nprocs=world, join=True)
    got = [torch.load(tmp_path / f'rank{r}.pt') for r in range(world)]
    assert [g['shard'][1] - g['shard'][0] for g in got] == [126] + [125] * 7
    d, tok, mols, p, names, eps = _setup(n_mols=n_mols)
    layout = _model_layout(d, tok)
    batch = _collate(mols, tok)
    sums, counts = _loss_sums(p, d, batch, eps)
    total, _ = _total(sums, counts, step=1500)
    total.backward()
    want = _message_grads(p, layout)
    b = layout[2]
    assert b[0] == 0 and b[0] < b[1] < b[2] < b[3] < b[4] and b[4] + 4 == layout[3]
    scale = want.abs().max().item()
    for r in range(world):
        assert torch.equal(got[r]['denoms'], counts), 'all-reduced denominators differ from the global counts'
        assert torch.equal(got[r]['message'][b[4]:], counts), 'the counts riding behind the gradients'
        err = (got[r]['message'][:b[4]] - want[:b[4]]).abs().max().item()
        assert err <= 1e-10 * scale, f'rank {r}: gradient differs by {err}'
        assert torch.equal(got[r]['message'], got[0]['message']), 'ranks disagree after the all-reduce'


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 128, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_global_step_reference_is_the_global_mean():
    assert global_step_reference([6.0, 9.0], [2, 3]) == pytest.approx(15.0 / 5)


@pytest.mark.timeout(300)
def test_two_rank_gradients_equal_global_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(tmp_path / f'rank{r}.pt') for r in range(world)]

    d, tok, mols, p, names, eps = _setup()
    batch = _collate(mols, tok)
    sums, counts = _loss_sums(p, d, batch, eps)
    total, _ = _total(sums, counts, step=1500)
    # the same thing through the oracle's own means: dividing by the counts must change nothing
    direct = O.weighted_loss(O.forward(p, d, batch, eps), 1500)
    assert float(total.detach()) == pytest.approx(float(direct.detach()), rel=1e-12)
    total.backward()
    want = _flat_grads(p, names)

    for r in range(world):
        assert torch.equal(got[r]['denoms'], counts), 'all-reduced denominators differ from the global counts'
        assert got[r]['total'] == pytest.approx(float(total.detach()), rel=1e-10)
        err = (got[r]['flat'] - want).abs().max().item()
        assert err <= 1e-10 * want.abs().max().item(), f'rank {r}: gradient differs by {err}'
    assert torch.equal(got[0]['flat'], got[1]['flat']), 'ranks disagree after the all-reduce'
