"""Host-side batching (native C++ collate) must be bit-exact against the
reference's own process_step2 / process_step3 outputs and must consume Python's
random stream exactly like it."""
import random

import numpy as np
import pytest
import torch

from ps_vae_amd import synth
from ps_vae_amd.data import BPEMolDataset, ensure_native
from ps_vae_amd.tokenizer import Tokenizer
from oracle import collate_oracle as CO
from tests.util import load_golden

CASES = ['collate_qm9_b8.pt', 'collate_zinc_b6.pt', 'collate_edge_cases.pt']


def assert_same_batch(got, want):
    for k, v in want.items():
        if k.startswith('_'):
            continue
        if torch.is_tensor(v):
            assert got[k].dtype == v.dtype, (k, got[k].dtype, v.dtype)
            assert got[k].shape == v.shape, (k, got[k].shape, v.shape)
            assert torch.equal(got[k], v), k
        else:
            assert list(got[k]) == list(v), k


@pytest.mark.parametrize('case', CASES)
def test_native_collate_matches_reference_fixture(case):
    g = load_golden(case)
    tok = Tokenizer.synthetic(g['vocab_size'])
    step2 = [BPEMolDataset.process_step2(m, tok) for m in g['mols']]
    for mine, ref in zip(step2, g['step2']):
        for k in ('x', 'ad_mat', 'pieces', 'x_pieces', 'x_pos', 'edge_select'):
            assert mine[k] == ref[k], k
    random.seed(g['seed'])
    batch = BPEMolDataset.process_step3(step2, tok)
    assert_same_batch(batch, g['batch'])
    assert random.random() == g['next_random']


def test_native_collate_matches_oracle_on_larger_batches():
    tok = Tokenizer.synthetic(300)
    mols = synth.make_molecules(5, 'zinc', 300, 64)
    random.seed(99)
    want = CO.collate([CO.to_matrix_form(m) for m in mols], 300)
    nxt = random.random()
    random.seed(99)
    got = BPEMolDataset.process_step3([BPEMolDataset.process_step2(m, tok) for m in mols], tok)
    assert_same_batch(got, want)
    assert random.random() == nxt


def _csr_to_edges(ptr, col, typ):
    out = []
    for i in range(len(ptr) - 1):
        for e in range(ptr[i], ptr[i + 1]):
            out.append((int(col[e]), i, int(typ[e])))
    return out


def test_packed_index_blob_is_consistent():
    tok = Tokenizer.synthetic(300)
    mols = synth.make_molecules(6, 'zinc', 300, 16)
    random.seed(3)
    batch = BPEMolDataset.process_step3([BPEMolDataset.process_step2(m, tok) for m in mols], tok)
    nb = batch['_native']
    B, N = batch['x'].shape
    lens = (~batch['atom_mask']).sum(1).numpy()
    off = np.concatenate([[0], np.cumsum(lens)])
    n = int(off[-1])
    assert nb.sizes['n'] == n and nb.sizes['B'] == B
    f = lambda name, cnt: nb.field(name)[:cnt].numpy()
    assert list(f('graph_off', B + 1)) == list(off)
    packed = lambda q: int(off[q // N] + q % N)
    # packed ids
    flat_keep = (~batch['atom_mask']).flatten().numpy()
    assert list(f('atom_id', n)) == list(batch['x'].flatten().numpy()[flat_keep])
    assert list(f('piece_id', n)) == list(batch['x_pieces'].flatten().numpy()[flat_keep])
    # CSR by target keeps the reference's edge order within every row
    ei, et = batch['edge_index'].numpy(), batch['edge_attr'].argmax(1).numpy()
    E = ei.shape[1]
    want = sorted([(packed(int(ei[0, e])), packed(int(ei[1, e])), int(et[e]), e) for e in range(E)], key=lambda r: (r[1], r[3]))
    got = _csr_to_edges(f('rowptr', n + 1), f('col', E), f('etype', E))
    assert got == [(s, d, t) for s, d, t, _ in want]
    # transpose CSR: rows = sources
    got_t = _csr_to_edges(f('rowptr_t', n + 1), f('col_t', E), f('etype_t', E))
    want_t = sorted([(packed(int(ei[1, e])), packed(int(ei[0, e])), int(et[e]), e) for e in range(E)], key=lambda r: (r[1], r[3]))
    assert got_t == [(s, d, t) for s, d, t, _ in want_t]
    # in-piece subgraph
    idx = batch['in_piece_edge_idx']
    Ein = len(idx)
    assert nb.sizes['E_in'] == Ein
    want_in = sorted([(packed(int(ei[0, e])), packed(int(ei[1, e])), int(et[e]), k) for k, e in enumerate(idx)], key=lambda r: (r[1], r[3]))
    assert _csr_to_edges(f('in_rowptr', n + 1), f('in_col', Ein), f('in_etype', Ein)) == [(s, d, t) for s, d, t, _ in want_in]
    # pairs in nonzero order
    pg, pm, pn = np.nonzero(batch['edge_select'].numpy())
    S = len(pg)
    assert nb.sizes['S'] == S
    assert list(f('pair_src', S)) == [int(off[g] + m) for g, m in zip(pg, pm)]
    assert list(f('pair_dst', S)) == [int(off[g] + m) for g, m in zip(pg, pn)]
    assert list(f('pair_gold', S)) == list(batch['golden_edge'].numpy())
    bysrc_ptr, bysrc_idx = f('pair_by_src_ptr', n + 1), f('pair_by_src_idx', S)
    src = f('pair_src', S)
    for j in range(n):
        seg = bysrc_idx[bysrc_ptr[j]:bysrc_ptr[j + 1]]
        assert all(src[s] == j for s in seg) and list(seg) == sorted(seg)
    assert bysrc_ptr[n] == S
    # denominators: target tokens that are not padding, S, B, B*2
    ntok = int((batch['pieces'][:, 1:] != tok.pad_idx()).sum())
    assert list(nb.denoms.numpy()) == [ntok, S, B, 2 * B]
    # a batch dict without the native entry packs to the same blob
    foreign = {k: v for k, v in batch.items() if k != '_native'}
    nb2 = ensure_native(foreign, tok)
    assert torch.equal(nb2.blob, nb.blob) and nb2.offsets == nb.offsets
    assert torch.equal(nb2.props, nb.props) and torch.equal(nb2.denoms, nb.denoms)


@pytest.mark.parametrize('case', CASES)
def test_packed_dataset_matches_reference_fixture(case):
    """The flattened data set + native stage 2/3 against the reference's own batches."""
    from ps_vae_amd.data import PackedMolDataset
    g = load_golden(case)
    tok = Tokenizer.synthetic(g['vocab_size'])
    ds = PackedMolDataset(g['mols'], tok, full=True)
    random.seed(g['seed'])
    batch = ds.collate_fn(list(range(len(g['mols']))))
    assert_same_batch(batch, g['batch'])
    assert random.random() == g['next_random']


@pytest.mark.parametrize('shape,count,picks', [('zinc', 40, [3, 17, 5, 39, 0, 22, 8]), ('qm9', 64, list(range(63, -1, -3)))])
def test_packed_dataset_equals_per_item_dataset(shape, count, picks):
    """Same blob, same dict, same state of `random` as the per-item path, for arbitrary index selections; and
    the array form of stage 2 equals its list form."""
    from ps_vae_amd.data import PackedMolDataset
    tok = Tokenizer.synthetic(50)
    mols = synth.make_molecules(9, shape, 50, count)
    mols[1] = dict(mols[1], edge_index=mols[1]['edge_index'] + [mols[1]['edge_index'][0]],
                   edge_attr=mols[1]['edge_attr'] + [3])           # a duplicate bond: the later type wins
    ds, per_item = PackedMolDataset(mols, tok, full=True), BPEMolDataset(mols, tok)
    for a, b in zip([per_item[i] for i in (0, 1)], [BPEMolDataset.process_step2(mols[i], tok) for i in (0, 1)]):
        for k in ('ad_mat', 'edge_select', 'x_pieces', 'x_pos'):
            assert np.asarray(a[k]).tolist() == b[k], k
    for sel in (picks, [1, 0, 2]):
        random.seed(77)
        want = per_item.collate_fn([per_item[i] for i in sel])
        state = random.getstate()
        random.seed(77)
        got = ds.collate_fn([ds[i] for i in sel])
        assert random.getstate() == state
        assert_same_batch(got, want)
        assert torch.equal(got['_native'].blob, want['_native'].blob) and got['_native'].offsets == want['_native'].offsets
        assert torch.equal(got['_native'].props, want['_native'].props) and torch.equal(got['_native'].denoms, want['_native'].denoms)
    slim = PackedMolDataset(mols, tok).collate_fn(picks)
    assert 'edge_select' not in slim and slim['x'].shape[0] == len(picks)
