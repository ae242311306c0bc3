"""Data-parallel `fit`: two ranks (sharing the one GPU of the test box, exchanging over gloo) must end an
epoch with the parameters a single process reaches when it takes every step on both ranks' batches with the
global loss denominators and the summed gradient."""
import os
import random
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_fit_two_ranks_equals_the_undivided_batches(tmp_path):
    from ps_vae_amd import Tokenizer
    from tests import dp_fit_worker as W
    out = str(tmp_path / 'dp')
    os.makedirs(out + '.ckpt', exist_ok=True)
    env = dict(os.environ, GLOO_SOCKET_IFNAME='lo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29551', os.path.join(ROOT, 'tests', 'dp_fit_worker.py'), out]
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert run.returncode == 0, run.stderr[-3000:]
    got = [torch.load(f'{out}.{r}.pt', weights_only=True) for r in range(2)]
    assert torch.equal(got[0]['params'], got[1]['params']), 'ranks must hold the same parameters'
    assert got[0]['history'] == got[1]['history'], 'ranks must see the same validation record'
    assert os.path.exists(os.path.join(out + '.ckpt', 'best.ckpt'))

    # resident batches come round again in the second epoch: their global counts must still be the sum of the two
    # ranks' own counts, not that sum multiplied up once more
    assert torch.equal(got[0]['denoms'], got[0]['local_denoms'] + got[1]['local_denoms'])
    assert torch.equal(got[0]['denoms'], got[1]['denoms'])

    # the same steps in one process: every step takes both ranks' batches (the second backward accumulates) with the
    # global counts, and each graph gets the epsilon of its place in the global batch
    tok = Tokenizer.synthetic(W.V)
    batches, valid = zip(*[W.resident_batches(tok, r, 'cuda') for r in range(2)])
    model = W.build_model(tok)
    opt = model.configure_optimizers()
    draw = 0
    for epoch in range(W.EPOCHS):
        opt.set_epoch(epoch)
        for k in range(W.STEPS):
            total = batches[0][k]['_native'].local_denoms + batches[1][k]['_native'].local_denoms
            opt.zero_grad()
            for r in range(2):
                batches[r][k]['_native'].denoms.copy_(total)
                model.eps_draws = draw
                eps = model.draw_eps(W.PER_RANK, r * W.PER_RANK)
                model.training_step(batches[r][k], eps=eps).backward()      # the second backward accumulates
            opt.step()
            model.global_step += 1
            draw += 1
        draw += len(valid[0])          # every validation batch of a rank takes one draw as well
    want = model.flat_buffers()[0].detach().cpu()
    err = (got[0]['params'] - want).abs().max().item()
    assert err <= 2e-6, f'two-rank fit differs from the undivided steps by {err:.3e}'
    # the sampled latent of the two ranks' first batches = the sample one process draws for the 32 graphs together
    # (under the ranks' own parameters: the 2e-6 allowed above would otherwise show in z)
    with torch.no_grad():
        model.flat_buffers()[0].copy_(got[0]['params'])
    for r in range(2):
        assert got[r]['draws'] == draw and got[r]['graph_base'] == r * W.PER_RANK
        mean, lvar = model.encode(batches[r][0])
        assert torch.allclose(mean.cpu(), got[r]['mean'], rtol=1e-5, atol=1e-6), f'latent mean of rank {r}'
        assert torch.allclose(lvar.cpu(), got[r]['log_var'], rtol=1e-5, atol=1e-6), f'latent log-variance of rank {r}'
        model.eps_draws = draw          # every draw advances the counter
        z = model.get_z(batches[r][0], eps=model.draw_eps(W.PER_RANK, r * W.PER_RANK))
        err = (z.cpu() - got[r]['z']).abs().max().item()
        assert torch.allclose(z.cpu(), got[r]['z'], rtol=1e-5, atol=1e-6), f'sampled z of rank {r}: {err:.3e}'
    model.eps_draws = draw
    both = model.draw_eps(2 * W.PER_RANK, 0)
    model.eps_draws = draw
    assert torch.equal(both[W.PER_RANK:], model.draw_eps(W.PER_RANK, W.PER_RANK)), 'epsilon rows depend on the global index only'
