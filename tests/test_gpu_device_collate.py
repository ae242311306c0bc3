"""Batches built on the GPU (`psvae_collate_device`) against the host builder: with every cross-piece pair
kept the two blobs must be identical field for field; with sampling, the kernels must select exactly the
pairs the restated counter-based sampler predicts, and the blob must equal the host builder's for those pairs."""
import numpy as np
import pytest
import torch

from ps_vae_amd import PSVAEModel, Tokenizer, default_args, full_config, synth
from ps_vae_amd.data import BPEMolDataset, PackedMolDataset, pack_native
from ps_vae_amd.device_data import DeviceLoader, DeviceMolDataset
from oracle import collate_oracle as CO

pytestmark = pytest.mark.gpu
V = 300
S_TOK = V + 1     # <s>

EDGE_CASES = [
    {'x': [2], 'edge_index': [], 'edge_attr': [], 'props': [0.1, 0.2, 0.3], 'pieces': [S_TOK, 3, S_TOK], 'groups': [[], [0], []]},
    {'x': [2, 7], 'edge_index': [[0, 1]], 'edge_attr': [2], 'props': [0.5, 0.5, 0.5], 'pieces': [S_TOK, 5, S_TOK],
     'groups': [[], [0, 1], []]},
    {'x': [1, 2, 3], 'edge_index': [], 'edge_attr': [], 'props': [0.3, 0.1, 0.7], 'pieces': [S_TOK, 7, 8, 9, S_TOK],
     'groups': [[], [0], [1], [2], []]},
    # atom 3 lies in no group (its row of the cross-piece mask stays 1, the diagonal included); bond 0-1 is listed
    # twice with two types (the later one wins)
    {'x': [2, 2, 6, 7, 2], 'edge_index': [[0, 1], [1, 2], [2, 3], [3, 4], [1, 0]], 'edge_attr': [1, 2, 1, 3, 2],
     'props': [0.9, 0.1, 0.4], 'pieces': [S_TOK, 0, 1, S_TOK], 'groups': [[], [4], [0, 1, 2], []]},
]


def records():
    return (EDGE_CASES + synth.make_molecules(3, 'zinc', V, 40) + synth.make_molecules(4, 'qm9', V, 12) +
            synth.make_molecules(5, 'stress', V, 3))


def field_counts(sz, tok):
    B, n, E, Ei, S, L = (sz[k] for k in ('B', 'n', 'E', 'E_in', 'S', 'L'))
    na, npc, mp, T = tok.num_atom_type(), tok.num_piece_type(), BPEMolDataset.max_pos, L - 1
    return [n, n, n, n, B + 1, n + 1, E, E, n + 1, E, E, n + 1, Ei, Ei, n + 1, Ei, Ei, S, S, S, S, n + 1, S, n + 1, S, B + 1,
            B * L, na + 1, n, npc + 1, n, mp + 1, n, npc + 1, B * T]


def host_batch_for_pairs(mols, tok, indices, select):
    """NativeBatch the host builder packs for the molecules `indices` when `select(b, mol_index, ad, cross)`
    names the kept pairs of every molecule."""
    hb = PackedMolDataset(mols, tok, full=True).collate_fn(indices, all_pairs=True)
    lens = (~hb['atom_mask']).sum(1).numpy().astype(np.int32)
    N, L = hb['x'].shape[1], hb['pieces'].shape[1]
    ei = hb['edge_index'].numpy()
    et = hb['edge_attr'].argmax(1).numpy() if hb['edge_attr'].shape[0] else np.zeros(0, np.int64)
    pg, pm, pn, pgold = [], [], [], []
    for b, mi in enumerate(indices):
        s2 = BPEMolDataset.process_step2(mols[mi], tok, lists=False)
        for (m, k, t) in select(b, mi, s2['_ad'], s2['_cross']):
            pg.append(b); pm.append(m); pn.append(k); pgold.append(t)
    props = hb['props'].reshape(len(indices), -1)[:, list(BPEMolDataset.selected_properties)].float().numpy()
    return pack_native(lens, N, L, hb['x'].numpy(), hb['x_pieces'].numpy(), hb['x_pos'].numpy(), ei[0], ei[1], et,
                       np.asarray(hb['in_piece_edge_idx'], dtype=np.int32), np.asarray(pg), np.asarray(pm), np.asarray(pn),
                       np.asarray(pgold), hb['pieces'].numpy(), props, tok, BPEMolDataset.max_pos)


def assert_same_native(got, want, tok):
    assert got.sizes == want.sizes, (got.sizes, want.sizes)
    from ps_vae_amd import _lib
    for name, cnt in zip(_lib.BLOB_FIELDS, field_counts(want.sizes, tok)):
        a = got.field(name)[:cnt].cpu()
        b = want.field(name)[:cnt].cpu()
        assert torch.equal(a, b), (name, a[:12].tolist(), b[:12].tolist())
    assert torch.equal(got.props.cpu(), want.props.cpu())
    assert torch.equal(got.denoms.cpu(), want.denoms.cpu()), (got.denoms, want.denoms)


@pytest.mark.parametrize('indices', [list(range(59)), [0], [3, 3, 1, 58, 20], list(range(58, -1, -3))])
def test_all_pairs_blob_equals_host_blob(indices):
    tok, mols = Tokenizer.synthetic(V), records()
    ds = DeviceMolDataset(mols, tok, 'cuda:0')
    got = ds.collate(indices, all_pairs=True)['_native']
    want = PackedMolDataset(mols, tok).collate_fn(indices, all_pairs=True)['_native']
    assert_same_native(got, want, tok)


def test_large_batch_spans_many_sort_tiles_and_scan_chunks():
    """1500 graphs (more than one graph per thread of the offset scans, about 140 tiles per counting sort)."""
    import random
    tok = Tokenizer.synthetic(V)
    mols = synth.make_molecules(21, 'zinc', V, 1500)
    order = list(range(1500))
    random.Random(5).shuffle(order)
    got = DeviceMolDataset(mols, tok, 'cuda:0').collate(order, all_pairs=True)['_native']
    want = PackedMolDataset(mols, tok).collate_fn(order, all_pairs=True)['_native']
    assert_same_native(got, want, tok)


@pytest.mark.parametrize('rate,epoch', [(0.022, 0), (0.022, 7), (0.3, 1), (0.0, 0)])
def test_sampled_pairs_are_the_restated_samplers(rate, epoch):
    tok, mols = Tokenizer.synthetic(V), records()
    seed = 2021
    ds = DeviceMolDataset(mols, tok, 'cuda:0', seed=seed, neg_rate=rate)
    indices = [5, 0, 3, 44, 17, 57, 2, 30, 1]
    got = ds.collate(indices, epoch=epoch)['_native']
    want = host_batch_for_pairs(mols, tok, indices, lambda b, mi, ad, cross: CO.device_select_pairs(ad, cross, seed, epoch, mi, rate))
    assert_same_native(got, want, tok)
    if rate == 0.0:      # only the bonded cross-piece pairs remain
        assert (want.field('pair_gold')[:want.sizes['S']] != 0).all()


def test_a_molecules_pairs_do_not_depend_on_its_batch():
    tok, mols = Tokenizer.synthetic(V), records()
    ds = DeviceMolDataset(mols, tok, 'cuda:0', neg_rate=0.2)

    def pairs_of(nb, b):
        lo, hi = (int(v) for v in nb.field('pair_graph_off')[b:b + 2].cpu())
        g0 = int(nb.field('graph_off')[b])
        return (nb.field('pair_src')[lo:hi] - g0).cpu().tolist(), (nb.field('pair_dst')[lo:hi] - g0).cpu().tolist()

    a = ds.collate([10, 20, 30], epoch=2)['_native']
    b = ds.collate([30, 7, 10], epoch=2)['_native']
    c = ds.collate([10], epoch=3)['_native']
    assert pairs_of(a, 0) == pairs_of(b, 2) and pairs_of(a, 2) == pairs_of(b, 0)
    assert pairs_of(a, 0) != pairs_of(c, 0)       # another epoch, another draw


def test_training_step_on_a_device_built_batch():
    tok, mols = Tokenizer.synthetic(V), records()
    seed, rate, indices = 11, 0.022, list(range(4, 36))
    ds = DeviceMolDataset(mols, tok, 'cuda:0', seed=seed, neg_rate=rate)
    dev_batch = ds.collate(indices, epoch=0)
    host_nb = host_batch_for_pairs(mols, tok, indices, lambda b, mi, ad, cross: CO.device_select_pairs(ad, cross, seed, 0, mi, rate))
    torch.manual_seed(0)
    model = PSVAEModel(full_config(default_args(), tok), tok).cuda()
    eps = torch.randn(len(indices), 56, device='cuda:0')
    outs = []
    for batch in (dev_batch, {'_native': host_nb.to('cuda:0')}):
        model.zero_grad()
        pred, (rec, piece, edge, kl) = model(batch, eps=eps)
        model.weighted_loss(pred, rec, kl).backward()
        outs.append((torch.stack([pred, rec, piece, edge, kl]).detach().clone(), model.flat_buffers()[1].clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[0][0]).all()


def test_loader_feeds_the_epoch_loop():
    """`fit` over a `DeviceLoader`: the batches it yields are the ones `collate` builds for the same indices and
    epoch, every molecule is seen once per epoch, and the loss goes down."""
    from ps_vae_amd.trainer import fit
    tok = Tokenizer.synthetic(V)
    mols = synth.make_molecules(8, 'qm9', V, 96)
    ds = DeviceMolDataset(mols, tok, 'cuda:0', seed=4)
    train = DeviceLoader(ds, 16, shuffle=True, seed=1)
    valid = DeviceLoader(DeviceMolDataset(mols[:32], tok, 'cuda:0', seed=4), 16)
    train.set_epoch(3)
    seen = []
    for idx, batch in zip(train.batches(), train):
        want = ds.collate(idx, epoch=3)['_native']
        got = batch['_native']
        assert got.sizes == want.sizes and torch.equal(got.blob, want.blob) and torch.equal(got.denoms, want.denoms)
        seen += list(idx)
    assert sorted(seen) == list(range(96)) and seen != list(range(96))
    torch.manual_seed(0)
    model = PSVAEModel(full_config(default_args(), tok), tok).cuda()

    class Epochs:       # what a training script does between epochs
        def __iter__(self):
            train.set_epoch(train.epoch + 1)
            return iter(train)
    history = fit(model, Epochs(), valid, max_epochs=4)
    assert len(history) == 4 and all(np.isfinite(h['val_loss']) for h in history)
    assert history[-1]['val_loss'] < history[0]['val_loss']
