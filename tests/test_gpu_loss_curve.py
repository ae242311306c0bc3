"""Loss curve of 1000 optimiser steps (forward, backward, clip + Adam, beta schedule) on the GPU against the
same 1000 steps of the CPU oracle: same parameters, batches and eps draws.

Training trajectories are sensitive to rounding: the fp32 CPU oracle (the reference's own arithmetic) itself
parts from its float64 twin by ~5e-2 relative within 50 steps and by tens of percent within 1000 (Adam turns
rounding-level gradient differences into steps of size lr; `tools/loss_curve_study.py` prints the numbers).
A per-step bound of 1e-3 over 1000 steps is therefore not a property any fp32 implementation can have.  What
is asserted:
  * the first 10 steps agree with the float64 oracle within 1e-3 relative;
  * in every later window of steps, the GPU's largest deviation from the float64 oracle is at most 4 times the
    fp32 oracle's own largest deviation from it (the yardstick the gradient tests use as well);
  * the run trains, and ends at the same loss level as the float64 run (mean of the last 100 steps within
    the fp32 oracle's own margin).
The criterion is asserted on a small shape over 1000 steps and at the benchmark shape over 200 steps."""
import pytest
import torch

from oracle import psvae_oracle as O
from tests.test_gpu_step import build, collate_both

pytestmark = pytest.mark.gpu
WINDOWS = ((10, 50), (50, 100), (100, 200), (200, 500), (500, 1000))


def run_three_trajectories(steps, V, B, shape, n_batches):
    """`steps` optimiser steps of the GPU step, the fp32 CPU oracle and the float64 CPU oracle on the same
    parameters, batches and epsilon draws; returns the three loss curves (float64 tensors)."""
    from ps_vae_amd import synth
    d = O.default_dims(V)
    model, tok, sd = build(d, V)
    pairs = [collate_both(synth.make_molecules(500 + i, shape, V, B), tok, V, 40 + i) for i in range(n_batches)]
    eps_all = torch.randn(steps, B, d['latent'], generator=torch.Generator().manual_seed(77))
    ref32 = O.Trainer(sd, d)
    ref64 = O.Trainer({k: v.double() for k, v in sd.items()}, d)
    opt = model.configure_optimizers()
    gpu, c32, c64 = [], [], []
    for s in range(steps):
        ref_batch, batch = pairs[s % n_batches]
        opt.zero_grad()
        loss = model.training_step(batch, eps=eps_all[s])
        loss.backward()
        opt.step()
        model.global_step += 1
        gpu.append(loss.detach())
        c32.append(ref32.step(ref_batch, eps_all[s])[0])
        c64.append(ref64.step(dict(ref_batch, props=ref_batch['props'].double()), eps_all[s].double())[0])
    return (torch.stack(gpu).cpu().double(), torch.tensor(c32, dtype=torch.double), torch.tensor(c64, dtype=torch.double))


def check_trajectories(gpu, c32, c64, steps, tail):
    def rel(x):
        return (x - c64).abs() / c64.abs().clamp_min(1e-12)
    g, c = rel(gpu), rel(c32)
    assert float(g[:10].max()) <= 1e-3, f'first steps: {float(g[:10].max()):.2e}'
    report = [f'steps 0-10: gpu {float(g[:10].max()):.2e}, fp32 oracle {float(c[:10].max()):.2e}']
    for lo, hi in WINDOWS:
        if lo >= steps:
            break
        gw, cw = float(g[lo:min(hi, steps)].max()), float(c[lo:min(hi, steps)].max())
        report.append(f'steps {lo}-{min(hi, steps)}: gpu {gw:.2e}, fp32 oracle {cw:.2e}')
        assert gw <= 4.0 * max(cw, 1e-3), report[-1]
    end_g, end_c, end_64 = float(gpu[-tail:].mean()), float(c32[-tail:].mean()), float(c64[-tail:].mean())
    print('; '.join(report), f'; final level gpu {end_g:.4f} fp32 {end_c:.4f} fp64 {end_64:.4f}')
    return end_g, end_c, end_64


@pytest.mark.timeout(900)
def test_loss_curve_follows_the_oracle_for_1000_steps():
    steps = 1000
    gpu, c32, c64 = run_three_trajectories(steps, V=60, B=8, shape='qm9', n_batches=16)
    assert float(c64[-50:].mean()) < 0.7 * float(c64[:50].mean()), 'the run did not train'
    end_g, end_c, end_64 = check_trajectories(gpu, c32, c64, steps, tail=100)
    # same loss level at the end of training as the float64 run (mean of the last 100 steps)
    assert abs(end_g - end_64) <= 4.0 * max(abs(end_c - end_64), 1e-3 * end_64), (end_g, end_c, end_64)


@pytest.mark.timeout(900)
def test_loss_curve_at_the_benchmark_shape():
    """The same criterion at the shape the throughput is quoted on (ZINC-shape, 128 graphs, vocabulary 300), over
    200 optimiser steps: first 10 steps within 1e-3 of the float64 trajectory, later windows within 4 times the fp32
    oracle's own deviation from it.  (200 steps end in the middle of training, where a mean over a few steps still
    carries the step-to-step differences the window bound already limits: the level printed at the end is information,
    the 1000-step run above asserts it.)"""
    steps = 200
    gpu, c32, c64 = run_three_trajectories(steps, V=300, B=128, shape='zinc', n_batches=8)
    assert float(c64[-20:].mean()) < 0.9 * float(c64[:20].mean()), 'the run did not train'
    check_trajectories(gpu, c32, c64, steps, tail=20)
