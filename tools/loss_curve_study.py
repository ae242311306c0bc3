#!/usr/bin/env python3
"""How fast do training trajectories of the same run part?  GPU fp32 vs CPU-oracle fp32 vs CPU-oracle fp64,
same parameters, batches and eps.  Usage: loss_curve_study.py [steps] [batch] [n_batches] [shape] [vocab] [out.json]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import psvae_oracle as O  # noqa: E402
from tests.test_gpu_step import build, collate_both  # noqa: E402
from ps_vae_amd import synth  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    shape = sys.argv[4] if len(sys.argv) > 4 else 'qm9'
    V = int(sys.argv[5]) if len(sys.argv) > 5 else 60
    out_path = sys.argv[6] if len(sys.argv) > 6 else None
    d = O.default_dims(V)
    model, tok, sd = build(d, V)
    pairs = [collate_both(synth.make_molecules(500 + i, shape, V, B), tok, V, 40 + i) for i in range(nb)]
    eps_all = torch.randn(steps, B, d['latent'], generator=torch.Generator().manual_seed(77))
    r32 = O.Trainer(sd, d)
    sd64 = {k: v.double() for k, v in sd.items()}
    r64 = O.Trainer(sd64, d)
    opt = model.configure_optimizers()
    g, a, b = [], [], []
    t0 = time.time()
    for s in range(steps):
        ref_batch, batch = pairs[s % nb]
        opt.zero_grad()
        loss = model.training_step(batch, eps=eps_all[s])
        loss.backward()
        opt.step()
        model.global_step += 1
        g.append(loss.detach())
        a.append(r32.step(ref_batch, eps_all[s])[0])
        rb = dict(ref_batch, props=ref_batch['props'].double())
        b.append(r64.step(rb, eps_all[s].double())[0])
        if (s + 1) % 25 == 0:
            print(f'step {s + 1}: gpu {float(g[-1]):.6f} cpu32 {float(a[-1]):.6f} cpu64 {float(b[-1]):.6f} ({time.time() - t0:.0f} s)', flush=True)
    g = torch.stack(g).cpu().double(); a = torch.tensor(a, dtype=torch.double); b = torch.tensor(b, dtype=torch.double)
    rel = lambda x, y: (x - y).abs() / y.abs().clamp_min(1e-12)
    print(f'steps {steps} B {B} batches {nb} {shape}: loss {float(b[0]):.4f} -> {float(b[-1]):.4f}')
    rows = []
    for lo, hi in ((0, 10), (10, 50), (50, 100), (100, 200), (200, 500), (500, 1000)):
        if lo >= steps:
            break
        sl = slice(lo, min(hi, steps))
        rows.append({'steps': [lo, min(hi, steps)], 'gpu32_vs_cpu64': float(rel(g, b)[sl].max()),
                     'cpu32_vs_cpu64': float(rel(a, b)[sl].max()), 'gpu32_vs_cpu32': float(rel(g, a)[sl].max())})
        print(f'  steps {lo:4d}-{hi:4d}: max rel  gpu32-vs-cpu64 {float(rel(g, b)[sl].max()):.2e}   cpu32-vs-cpu64 {float(rel(a, b)[sl].max()):.2e}   '
              f'gpu32-vs-cpu32 {float(rel(g, a)[sl].max()):.2e}', flush=True)
    if out_path:
        every = max(1, steps // 100)
        json.dump({'what': 'max relative difference of the training loss between trajectories of the same run (same initial '
                           'parameters, batches and epsilon): GPU fp32 step, CPU oracle fp32, CPU oracle fp64',
                   'steps': steps, 'batch': B, 'distinct_batches': nb, 'shape': shape, 'vocab': V, 'windows': rows,
                   'loss_every_%d_steps' % every: {'gpu32': g[::every].tolist(), 'cpu32': a[::every].tolist(),
                                                   'cpu64': b[::every].tolist()}}, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
