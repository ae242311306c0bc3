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
