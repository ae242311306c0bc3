"""Batched latent-space property optimisation against the reference's one-vector loop (CPU oracle)."""
import pytest
import torch

from oracle import psvae_oracle as O
from tests.test_gpu_step import build
from tests.util import check_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('target', [[0.9, None, 0.3], [None, None, 0.5]])
def test_direct_optimize_rows_follow_the_reference_loop(target):
    from ps_vae_amd.latent_opt import direct_optimize
    V, n = 60, 37
    d = O.default_dims(V)
    model, tok, sd = build(d, V)
    z0 = torch.randn(n, d['latent'], generator=torch.Generator().manual_seed(4))
    lr, max_iter, patience = 0.1, 40, 3
    best_z, best_loss, steps = direct_optimize(model, z0.cuda(), target, lr=lr, max_iter=max_iter, patience=patience)
    assert best_z.shape == z0.shape and int(steps.max()) <= max_iter
    for row in (0, 5, 11, 36):
        z_ref, loss_ref, steps_ref = O.direct_optimize(sd, d, z0[row], target, lr, max_iter, patience)
        assert int(steps[row]) == steps_ref, (row, int(steps[row]), steps_ref)
        assert abs(float(best_loss[row]) - loss_ref) <= 1e-4 * max(1e-3, abs(loss_ref)) + 1e-7
        check_close(best_z[row], z_ref, f'best z of row {row}', rtol=1e-4, atol_scale=1e-4)
    # the optimisation must actually lower the loss
    with torch.no_grad():
        first = direct_optimize(model, z0.cuda(), target, lr=lr, max_iter=1, patience=patience)[1]
    assert float(best_loss.mean()) < float(first.mean())


def test_direct_optimize_matches_the_reference_fixture():
    """All nine recorded runs of the reference's own `generate.direct_optimize` (`tests/golden/latent_opt.pt`) as rows of
    batched calls: step counts of the patience rule, best loss and the returned z."""
    from tests.util import load_golden
    from ps_vae_amd.latent_opt import direct_optimize
    g = load_golden('latent_opt.pt')
    model, tok, sd = build(g['dims'], g['vocab_size'], g['param_seed'])
    cases = g['cases']
    settings = sorted({(tuple(c['target']), c['lr'], c['max_iter'], c['patience']) for c in cases}, key=str)
    assert len(settings) == 3
    for target, lr, max_iter, patience in settings:
        rows = [c for c in cases if (tuple(c['target']), c['lr'], c['max_iter'], c['patience']) == (target, lr, max_iter, patience)]
        z0 = torch.stack([c['z0'] for c in rows]).cuda()
        best_z, best_loss, steps = direct_optimize(model, z0, list(target), lr=lr, max_iter=max_iter, patience=patience)
        for i, c in enumerate(rows):
            assert int(steps[i]) == len(c['losses']), (i, int(steps[i]), len(c['losses']))
            want = min(c['losses'])
            assert abs(float(best_loss[i]) - want) <= 1e-4 * max(1e-3, abs(want)) + 1e-7
            check_close(best_z[i], c['best_z'], f'best z of run {i}', rtol=1e-4, atol_scale=1e-4)
