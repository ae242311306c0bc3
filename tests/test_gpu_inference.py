"""Batched generation front half (piece sampling, link scoring) against the reference's
one-molecule loop restated in the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import psvae_oracle as O
from tests.test_gpu_step import build
from tests.util import check_close

pytestmark = pytest.mark.gpu


def _fragments(seed, count, V):
    """Fragment sets shaped like what the sampling loop hands to the scorer: atoms numbered piece by
    piece, positions from 1, in-piece bonds only (`src/modules/vae_piece_decoder.py:168-200`)."""
    from ps_vae_amd.synth import make_molecules
    out = []
    for mol in make_molecules(seed, 'qm9' if seed % 2 else 'zinc', V, count):
        groups = [g for g in mol['groups'] if len(g)]
        pieces = mol['pieces'][1:-1]
        new_id, x, xp, xq, new_groups = {}, [], [], [], []
        for pos, (g, pid) in enumerate(zip(groups, pieces)):
            ng = []
            for a in g:
                new_id[a] = len(x)
                ng.append(len(x))
                x.append(mol['x'][a]); xp.append(pid); xq.append(pos + 1)
            new_groups.append(ng)
        gid = {a: i for i, g in enumerate(new_groups) for a in g}
        bonds, types = [], []
        for (a, b), t in zip(mol['edge_index'], mol['edge_attr']):
            a, b = new_id[a], new_id[b]
            if gid[a] == gid[b]:
                bonds.append([a, b]); types.append(t)
        out.append({'x': x, 'x_pieces': xp, 'x_pos': xq, 'bonds': bonds, 'bond_types': types, 'groups': new_groups})
    return out


@pytest.mark.parametrize('seed,count', [(3, 9), (4, 33)])
def test_score_links_matches_the_reference_loop(seed, count):
    from ps_vae_amd.inference import candidate_order, score_links
    V = 60
    d = O.default_dims(V)
    model, tok, sd = build(d, V)
    frags = _fragments(seed, count, V)
    frags[0] = {'x': [2, 6], 'x_pieces': [5, 5], 'x_pos': [1, 1], 'bonds': [[0, 1]], 'bond_types': [1],
                'groups': [[0, 1]]}                       # one piece only: no candidate pair
    frags[1] = {'x': [2, 7, 2], 'x_pieces': [1, 2, 3], 'x_pos': [1, 2, 3], 'bonds': [], 'bond_types': [],
                'groups': [[0], [1], [2]]}                # single-atom pieces: no in-piece bond at all
    z = torch.randn(len(frags), d['latent'], generator=torch.Generator().manual_seed(seed))
    got = score_links(model, z.cuda(), frags)
    assert len(got) == len(frags) and len(got[0]['pairs']) == 0 and len(got[1]['pairs']) == 3
    for i in list(range(0, len(frags), 4)) + [1]:
        f = frags[i]
        ei, ea = [], []
        for (a, b), t in zip(f['bonds'], f['bond_types']):
            ei += [[a, b], [b, a]]; ea += [t, t]
        with torch.no_grad():
            idx, probs = O.score_links(sd, d, z[i], f['x'], f['x_pieces'], f['x_pos'], ei, ea, f['groups'])
        assert np.array_equal(got[i]['pairs'], idx.numpy()), i
        if len(idx) == 0:
            continue
        check_close(torch.from_numpy(got[i]['probs']), probs, f'link probabilities of fragment set {i}', rtol=2e-4, atol_scale=2e-5)
        conf, typ = probs.max(dim=-1)
        check_close(torch.from_numpy(got[i]['confidence']), conf, f'confidence of fragment set {i}', rtol=2e-4, atol_scale=2e-5)
        clear = (probs.topk(2, dim=-1).values[:, 0] - probs.topk(2, dim=-1).values[:, 1]) > 1e-4
        assert np.array_equal(got[i]['bond_type'][clear.numpy()], typ[clear].numpy().astype(np.int32)), i
        order = candidate_order(got[i], 0.3)
        assert all(got[i]['confidence'][a] >= got[i]['confidence'][b] for a, b in zip(order, order[1:]))
        assert all(got[i]['bond_type'][a] != 0 and got[i]['confidence'][a] >= 0.3 for a in order)


@pytest.mark.parametrize('temperature,max_atoms,check_every', [(1.0, 25, 8), (0.7, 1000, 0), (1.5, 6, 3)])
def test_sample_pieces_matches_the_reference_loop(temperature, max_atoms, check_every):
    from ps_vae_amd.inference import sample_pieces
    V, Bz = 60, 67
    d = O.default_dims(V)
    model, tok, sd = build(d, V)
    g = torch.Generator().manual_seed(11)
    z = torch.randn(Bz, d['latent'], generator=g)
    u = torch.rand(Bz, d['max_pos'], generator=g)
    atoms = [int(v) for v in np.random.default_rng(5).integers(1, 6, size=V)] + [0, 0]
    # a bias that makes the end token likely enough for sequences to end by themselves
    with torch.no_grad():
        model.decoder.to_vocab.bias[V + 1] += 2.0
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    got = sample_pieces(model, z.cuda(), max_atoms, temperature, uniforms=u.cuda(), piece_atoms=atoms, check_every=check_every)
    assert len(got) == Bz
    same = 0
    for b in range(Bz):
        with torch.no_grad():
            want = O.sample_pieces(sd, d, z[b], atoms, max_atoms, temperature, u[b])
        assert all(0 <= t < V + 2 and t != V for t in got[b])          # never <pad>
        assert len(got[b]) <= d['max_pos'] - 1
        same += int(got[b] == want)
    # A draw that falls within rounding distance of an interval boundary may pick the neighbouring piece
    # (fp32 logits differ in the last bits), after which that sequence goes its own way; such draws are rare.
    assert same >= Bz - 2, f'{Bz - same} of {Bz} sequences differ from the one-at-a-time loop'
    lens = [len(s) for s in got]
    if max_atoms == 6:
        assert max(lens) <= 6                                       # every piece has at least one atom
    if max_atoms == 1000:
        assert any(l > 3 for l in lens)


def test_sample_pieces_draws_from_the_softmax():
    """Frequencies of the first piece over many sequences follow softmax(logits / T) of the first step."""
    from ps_vae_amd.inference import sample_pieces
    V, Bz = 60, 4096
    d = O.default_dims(V)
    model, tok, sd = build(d, V)
    z = torch.randn(1, d['latent'], generator=torch.Generator().manual_seed(2)).repeat(Bz, 1)
    got = sample_pieces(model, z.cuda(), 1, 1.0, piece_atoms=[5] * V + [0, 0], generator=None)
    assert all(len(s) == 0 for s in got)            # the first piece already exceeds one atom and is dropped
    emb = sd['decoder.piece_embedding.weight'][V + 1].view(1, 1, -1)
    h0 = torch.nn.functional.linear(z[:1], sd['decoder.latent_to_rnn_hidden.weight'], sd['decoder.latent_to_rnn_hidden.bias'])
    h = O.gru_sequence(emb, h0, sd)[:, 0]
    logits = torch.nn.functional.linear(h, sd['decoder.to_vocab.weight'], sd['decoder.to_vocab.bias']).squeeze()
    logits[V] = logits[V + 1] = float('-inf')
    p = torch.softmax(logits, -1).numpy()
    # without an atom limit every sequence keeps its first draw (the end token cannot come first)
    got = sample_pieces(model, z.cuda(), 1000, 1.0, piece_atoms=[5] * V + [0, 0])
    assert all(len(s) >= 1 for s in got)
    first = np.array([s[0] for s in got])
    freq = np.bincount(first, minlength=V + 2)[:V] / len(first)
    assert np.abs(freq - p[:V]).max() < 4.0 * np.sqrt(p[:V].max() / len(first)) + 5e-3


def _fixture_model(g):
    """The model the generation fixtures were recorded with: seeded parameters plus the recorded bias shifts."""
    d, V = g['dims'], g['vocab_size']
    model, tok, sd = build(d, V, g['param_seed'])
    with torch.no_grad():
        model.decoder.to_vocab.bias[V + 1] += g.get('end_bias', 0.0)
        model.decoder.edge_predictor[1].bias[0] += g.get('none_bias', 0.0)
    return model, d


def test_front_half_matches_the_reference_fixture():
    """The batched GPU stages against what the reference's own `VAEPieceDecoder.inference` computed
    (`tests/golden/inference.pt`, recorded by `tools/make_fixtures.py` from the reference's file): the pieces drawn
    for every latent (same inversion rule at the recorded uniforms; cases whose draws fall near an interval boundary
    are not in the file) and the link probabilities of the fragment sets."""
    from tests.util import load_golden
    from ps_vae_amd.inference import candidate_order, sample_pieces, score_links
    g = load_golden('inference.pt')
    model, d = _fixture_model(g)
    cases = g['cases']
    for c in cases:       # sampling: one call per (temperature, atom limit) setting
        got = sample_pieces(model, c['z'].cuda().unsqueeze(0), c['max_atom_num'], c['temperature'],
                            uniforms=c['uniforms'].cuda().unsqueeze(0), piece_atoms=g['piece_atoms'], check_every=0)
        assert got[0] == c['drawn_ids'][:-1], (got[0], c['drawn_ids'])
    with_pairs = [c for c in cases if c['pair_logits'] is not None]
    frags = []
    for c in with_pairs:
        groups, start = [], 0
        for pid in c['drawn_ids'][:-1]:
            groups.append(list(range(start, start + g['piece_atoms'][pid])))
            start += g['piece_atoms'][pid]
        bonds = [e for e in c['edge_index'][::2]]          # each in-piece bond once (the loop appends both directions)
        frags.append({'x': c['x'], 'x_pieces': c['x_pieces'], 'x_pos': c['x_pos'], 'bonds': bonds,
                      'bond_types': c['edge_attr'][::2], 'groups': groups})
    got = score_links(model, torch.stack([c['z'] for c in with_pairs]).cuda(), frags)
    for c, r in zip(with_pairs, got):
        want = torch.softmax(c['pair_logits'], dim=-1)
        check_close(torch.from_numpy(r['probs']), want, 'link probabilities', rtol=2e-4, atol_scale=2e-5)
        conf, typ = want.max(dim=-1)
        top2 = want.topk(2, dim=-1).values
        if bool(((top2[:, 0] - top2[:, 1]) > 1e-4).all()):       # no near-tie between two bond classes
            order = candidate_order(r, c['add_edge_th'])
            assert [int(r['bond_type'][j]) for j in order] == [t for t, _ in c['tried']]
