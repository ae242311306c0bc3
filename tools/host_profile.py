#!/usr/bin/env python3
"""Where the HOST time of one training step goes (no device synchronisation inside the loop)."""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ps_vae_amd import PSVAEModel, Tokenizer, default_args, full_config  # noqa: E402
from ps_vae_amd.data import to_device  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    tok = Tokenizer.synthetic(300)
    torch.manual_seed(2021)
    model = PSVAEModel(full_config(default_args(), tok), tok).to(dev)
    opt = model.configure_optimizers()
    shape, B = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ('zinc', 128)
    cpu_batches, _ = bench.make_batches(tok, 300, shape, B, 4, seed=2021)
    batches = [to_device(b, dev) for b in cpu_batches]
    gen = torch.Generator(device=dev).manual_seed(1)
    acc = defaultdict(float)

    def lap(name, t):
        now = time.perf_counter()
        acc[name] += now - t
        return now

    n = 300
    for i in range(n + 20):
        if i == 20:
            torch.cuda.synchronize()
            acc.clear()
            t_all = time.perf_counter()
        if i % 50 == 49:
            torch.cuda.synchronize()      # keep the launch queue from filling up and blocking the host
        t = time.perf_counter()
        batch = batches[i % 4]
        eps = torch.randn(B, 56, device=dev, generator=gen); t = lap('randn', t)
        opt.zero_grad(); t = lap('zero_grad', t)
        loss = model.training_step(batch, eps=eps); t = lap('forward', t)
        loss.backward(); t = lap('backward', t)
        opt.step(); t = lap('opt.step', t)
        model.global_step += 1
    total = time.perf_counter() - t_all
    torch.cuda.synchronize()
    for k, v in acc.items():
        print(f'{k:10s} {v / n * 1e6:8.1f} us/step')
    print(f'host total {total / n * 1e6:.1f} us/step')


if __name__ == '__main__':
    main()
