#!/usr/bin/env python3
"""Which kernels account for the GPU's distance from the float64 trajectory in the first optimiser steps?

Ten steps at the benchmark shape (ZINC-shape, 128 graphs, vocabulary 300) on the float64 and the fp32 CPU oracle once,
then on the GPU once per option setting (same parameters, batches, epsilon): per setting the largest relative
difference of the step loss from the float64 run over steps 0-10, and of step 0 alone (the forward pass before any
update).  Usage: first_steps_study.py [steps] > profiles/r03_first_steps.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import psvae_oracle as O  # noqa: E402
from tests.test_gpu_step import build, collate_both  # noqa: E402
from ps_vae_amd import _lib, synth  # noqa: E402

SETTINGS = [
    ('default', {}),
    ('fused GINE / scorer kernels on the fp32 matrix instruction', {'split_bf16': 0}),
    ('weight gradients on the fp32 64-tile kernel', {'split_gemm': 0, 'wide_gemm': 0}),
    ('both of the above (no product on cut operands)', {'split_bf16': 0, 'split_gemm': 0, 'wide_gemm': 0}),
    ('no fused kernels (aggregate + GEMM chain everywhere)', {'fused_layer': 0}),
    ('no fused kernels, weight gradients on the fp32 kernel', {'fused_layer': 0, 'split_gemm': 0, 'wide_gemm': 0}),
    ('single-CU recurrence', {'gru_multi_cu': 0}),
    ('no side streams', {'streams': 0}),
]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    V, B, nb = 300, 128, 8
    d = O.default_dims(V)
    model, tok, sd = build(d, V)
    pairs = [collate_both(synth.make_molecules(500 + i, 'zinc', V, B), tok, V, 40 + i) for i in range(nb)]
    eps_all = torch.randn(steps, B, d['latent'], generator=torch.Generator().manual_seed(77))
    r32, r64 = O.Trainer(sd, d), O.Trainer({k: v.double() for k, v in sd.items()}, d)
    c32, c64, first32, first64 = [], [], None, None
    for s in range(steps):
        c32.append(r32.step(pairs[s % nb][0], eps_all[s])[0])
        c64.append(r64.step(dict(pairs[s % nb][0], props=pairs[s % nb][0]['props'].double()), eps_all[s].double())[0])
        if s == 0:      # the first update of every trained parameter
            first32 = {n: (r32.p[n].detach() - sd[n]).double() for n in r32.names}
            first64 = {n: r64.p[n].detach() - sd[n].double() for n in r64.names}
    c32, c64 = torch.tensor(c32, dtype=torch.double), torch.tensor(c64, dtype=torch.double)
    lr = r64.cfg['lr']

    def moved_otherwise(first):
        """entries whose first update differs from the float64 run's by more than half a learning rate (Adam's first
        update is lr * g / (|g| + eps): lr * sign(g) wherever |g| >> eps, so such an entry took the other sign or
        sits at |g| ~ eps)"""
        bad = sum(int(((first[n] - first64[n]).abs() > 0.5 * lr).sum()) for n in first64 if n in first)
        return bad, sum(first64[n].numel() for n in first64)

    def rel(x):
        return (x - c64).abs() / c64.abs()
    print(f'{steps} optimiser steps, ZINC-shape, {B} graphs, vocabulary {V}; largest relative difference of the step loss from the float64 CPU run')
    print(f'  {"fp32 CPU oracle":64s} steps 0-{steps}: {float(rel(c32).max()):.2e}   step 0: {float(rel(c32)[0]):.2e}')
    print('    per step: ' + ' '.join(f'{float(v):.1e}' for v in rel(c32)))
    bad, tot = moved_otherwise(first32)
    print(f'    parameters whose FIRST update differs from the float64 run by more than lr/2: {bad} of {tot}')
    for name, setting in SETTINGS:
        old = {k: _lib.set_option(k, v) for k, v in setting.items()}
        try:
            m, _, _ = build(d, V)          # same seed, same initial parameters
            opt = m.configure_optimizers()
            g = []
            for s in range(steps):
                opt.zero_grad()
                loss = m.training_step(pairs[s % nb][1], eps=eps_all[s])
                loss.backward()
                opt.step()
                m.global_step += 1
                g.append(loss.detach())
                if s == 0 and not setting:
                    now = m.state_dict()
                    firstg = {n: (now[n].detach().cpu() - sd[n]).double() for n in first64 if n in now}
            g = torch.stack(g).cpu().double()
        finally:
            for k, v in old.items():
                _lib.set_option(k, v)
        print(f'  GPU, {name:59s} steps 0-{steps}: {float(rel(g).max()):.2e}   step 0: {float(rel(g)[0]):.2e}', flush=True)
        print('    per step: ' + ' '.join(f'{float(v):.1e}' for v in rel(g)))
        if not setting:
            bad, tot = moved_otherwise(firstg)
            print(f'    parameters whose FIRST update differs from the float64 run by more than lr/2: {bad} of {tot} '
                  f'({len(firstg)} of {len(first64)} tensors compared)')


if __name__ == '__main__':
    main()
