"""The CPU oracle against outputs of the reference's own modules (golden files)."""
import random

import pytest
import torch

from oracle import collate_oracle as CO
from oracle import psvae_oracle as O
from tests.util import check_close, load_golden

MODEL_CASES = ['model_small_edge.pt', 'model_small_t3.pt', 'model_zinc_v300.pt', 'model_qm9_v500.pt']
COLLATE_CASES = ['collate_qm9_b8.pt', 'collate_zinc_b6.pt', 'collate_edge_cases.pt']


def _same_batch(got, want):
    for k, v in want.items():
        if torch.is_tensor(v):
            assert got[k].dtype == v.dtype, k
            assert got[k].shape == v.shape, (k, got[k].shape, v.shape)
            assert torch.equal(got[k], v), k
        else:
            assert list(got[k]) == list(v), k


@pytest.mark.parametrize('case', COLLATE_CASES)
def test_collate_oracle_bit_exact(case):
    g = load_golden(case)
    step2 = [CO.to_matrix_form(m) for m in g['mols']]
    for mine, ref in zip(step2, g['step2']):
        for k in ('x', 'ad_mat', 'pieces', 'x_pieces', 'x_pos', 'edge_select'):
            assert mine[k] == ref[k], k
    random.seed(g['seed'])
    batch = CO.collate(step2, g['vocab_size'])
    _same_batch(batch, g['batch'])
    assert random.random() == g['next_random']  # same number of draws consumed


@pytest.mark.parametrize('case', MODEL_CASES)
def test_model_oracle_matches_reference(case):
    g = load_golden(case)
    d = g['dims']
    random.seed(g['collate_seed'])
    batch = CO.collate([CO.to_matrix_form(m) for m in g['mols']], g['vocab_size'])
    tr = O.Trainer(O.make_params(d, g['param_seed']), d)
    for s, ref in enumerate(g['steps']):
        if s == 0:
            keep = {}
            for n in tr.names:
                tr.p[n].grad = None
            losses = O.forward(tr.p, d, batch, g['eps'][0], keep)
            a = ref['acts']
            check_close(keep['enc_layers'][0], a['enc_h0'], 'enc_h0')
            for l, want in enumerate(a['enc_layers']):
                check_close(keep['enc_layers'][l + 1], want, f'enc layer {l}')
            check_close(keep['dec_layers'][0], a['dec_h0'], 'dec_h0')
            for l, want in enumerate(a['dec_layers']):
                check_close(keep['dec_layers'][l + 1], want, f'dec layer {l}')
            for k in ('cond', 'z_mean', 'rnn_out', 'logits', 'pair_in', 'pair_logits', 'pred'):
                check_close(keep[k], a[k], k)
            total = O.weighted_loss(losses, 0)
            total.backward()
            for n, want in ref['grads'].items():
                if want is None:
                    assert tr.p[n].grad is None, n
                else:
                    check_close(tr.p[n].grad, want, f'grad {n}')
        total, losses = tr.step(batch, g['eps'][s])
        for k, v in ref['losses'].items():
            got = total if k == 'total' else losses[k]
            assert abs(got - v) <= 1e-5 * max(1.0, abs(v)), (s, k, got, v)
        for n, want in ref['params_after'].items():
            check_close(tr.p[n], want, f'step {s} param {n}', rtol=1e-5)
    assert list(O.accuracies(tr.p, d, batch, g['eps'][0])) == pytest.approx(g['accus_after'], abs=1e-6)
