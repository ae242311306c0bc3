#!/usr/bin/env python3
"""Host time of one training step split into the two library calls and everything around them
(autograd, Python glue): wraps psvae_step_forward / psvae_step_backward_signal with a timer."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ps_vae_amd import PSVAEModel, Tokenizer, _lib, default_args, full_config  # noqa: E402
from ps_vae_amd.data import to_device  # noqa: E402


class Timed:
    def __init__(self, fn):
        self.fn, self.t, self.n = fn, 0.0, 0

    def __call__(self, *a):
        t = time.perf_counter()
        r = self.fn(*a)
        self.t += time.perf_counter() - t
        self.n += 1
        return r


def main():
    dev = torch.device('cuda:0')
    tok = Tokenizer.synthetic(300)
    torch.manual_seed(2021)
    model = PSVAEModel(full_config(default_args(), tok), tok).to(dev)
    opt = model.configure_optimizers()
    cpu_batches, _ = bench.make_batches(tok, 300, 'zinc', 128, 4, seed=2021)
    batches = [to_device(b, dev) for b in cpu_batches]
    gen = torch.Generator(device=dev).manual_seed(1)
    L = _lib.lib()
    fwd, bwd, adam = Timed(L.psvae_step_forward), Timed(L.psvae_step_backward_signal), Timed(L.psvae_clip_adam)
    L.psvae_step_forward, L.psvae_step_backward_signal, L.psvae_clip_adam = fwd, bwd, adam
    unit = model.unit_grad()
    n, tot = 300, 0.0
    for i in range(n + 20):
        if i == 20:
            torch.cuda.synchronize()
            fwd.t = bwd.t = adam.t = 0.0
            tot = 0.0
        if i % 25 == 24:
            torch.cuda.synchronize()
        t = time.perf_counter()
        eps = torch.randn(128, 56, device=dev, generator=gen)
        opt.zero_grad()
        loss = model.training_step(batches[i % 4], eps=eps)
        loss.backward(unit)
        opt.step()
        model.global_step += 1
        tot += time.perf_counter() - t
    torch.cuda.synchronize()
    print(f'host per step {tot / n * 1e6:.0f} us: psvae_step_forward {fwd.t / n * 1e6:.0f}, psvae_step_backward {bwd.t / n * 1e6:.0f}, '
          f'psvae_clip_adam {adam.t / n * 1e6:.0f}, Python/autograd around them {(tot - fwd.t - bwd.t - adam.t) / n * 1e6:.0f}')


if __name__ == '__main__':
    main()
