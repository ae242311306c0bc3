#!/usr/bin/env python3
"""Is the training step limited by the host?  One optimiser step (forward, backward, clip + Adam) of a fixed batch is
captured into a HIP graph and replayed: the replay needs one launch call per step, so its time per step is what
the GPU alone needs for the same kernels.  (Measurement aid: a replayed graph repeats ONE batch.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ps_vae_amd import PSVAEModel, Tokenizer, default_args, full_config  # noqa: E402
from ps_vae_amd.data import to_device  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device('cuda:0')
    tok = Tokenizer.synthetic(300)
    torch.manual_seed(2021)
    model = PSVAEModel(full_config(default_args(), tok), tok).to(dev)
    opt = model.configure_optimizers()
    cpu, _ = bench.make_sharded_batches(tok, 300, 'zinc', B, 1, 2021, 0, 1)
    batch = to_device(cpu[0], dev)
    eps = torch.randn(B, 56, device=dev)
    unit = model.unit_grad()

    def step():
        opt.zero_grad()
        loss = model.training_step(batch, eps=eps)
        loss.backward(unit)
        opt.step()
        return loss

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 200 * 1e3
    print(f'B={B} eager: {eager:.3f} ms/step (host enqueue {t_host / 200 * 1e3:.3f} ms/step)', flush=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    replay = (time.perf_counter() - t0) / 200 * 1e3
    print(f'B={B} graph replay: {replay:.3f} ms/step (host {t_host / 200 * 1e3:.3f} ms/step)', flush=True)


if __name__ == '__main__':
    main()
