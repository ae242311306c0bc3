#!/usr/bin/env python3
"""Micro-benchmark of the GINE aggregation kernels at the training step's shape, beside a plain copy of
the same number of bytes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_vae_amd import _lib  # noqa: E402
from ps_vae_amd.ops import Csr  # noqa: E402
from ps_vae_amd.synth import make_molecules  # noqa: E402


ITERS = 200


def timeit(fn, iters=None):
    iters = iters or ITERS
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    """Buffers rotate over enough copies to exceed the 256 MB last-level cache several times, so every launch
    streams from HBM as it does inside an encoder pass (one fixed buffer set of 2 x 114 MB would be served from
    that cache and flatter the kernels by 1.5x)."""
    L = _lib.lib()
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    H = 300
    global ITERS
    sizes = [int(v) for v in sys.argv[1:] if v.isdigit()] or [128, 1024, 4096]
    for v in sys.argv[1:]:
        if v.startswith('--iters='):
            ITERS = int(v.split('=')[1])
    for B in sizes:
        mols = make_molecules(2021, 'zinc', 300, B)
        src, dst, typ, off = [], [], [], 0
        for m in mols:
            for (a, b), ty in zip(m['edge_index'], m['edge_attr']):
                src += [a + off, b + off]; dst += [b + off, a + off]; typ += [ty, ty]
            off += len(m['x'])
        n, E = off, len(src)
        csr = Csr(torch.tensor([src, dst], device=dev), torch.tensor(typ, device=dev), n)
        K = max(2, int(1.2e9 / (3.0 * n * H * 4)) + 1)      # buffer sets in rotation: > 1.2 GB in total
        xs = [torch.randn(n, H, device=dev) for _ in range(K)]
        outs = [torch.empty(n, H, device=dev) for _ in range(K)]
        das = [torch.randn(n, H, device=dev) for _ in range(K)]
        ew, eb = torch.randn(H, 4, device=dev), torch.randn(H, device=dev)
        part = torch.empty(max(1, L.psvae_gine_edge_grad_scratch(n, H)), device=dev)
        it = [0]

        def fwd():
            k = it[0] = (it[0] + 1) % K
            L.psvae_gine_aggregate_forward(xs[k].data_ptr(), H, csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.etype.data_ptr(),
                                           ew.data_ptr(), eb.data_ptr(), 1.0, outs[k].data_ptr(), H, n, H, st)

        def bwd():
            k = it[0] = (it[0] + 1) % K
            L.psvae_gine_aggregate_backward(das[k].data_ptr(), H, xs[k].data_ptr(), H, csr.rowptr_t.data_ptr(),
                                            csr.col_t.data_ptr(), csr.etype_t.data_ptr(), ew.data_ptr(), eb.data_ptr(), 1.0,
                                            None, 0, None, outs[k].data_ptr(), H, None, None, part.data_ptr(), n, H, st)

        def copy():
            k = it[0] = (it[0] + 1) % K
            outs[k].copy_(xs[k])

        def copy2():
            k = it[0] = (it[0] + 1) % K
            torch.add(xs[k], das[k], out=outs[k])

        f, b, c, c2 = timeit(fwd), timeit(bwd), timeit(copy), timeit(copy2)
        fb, bb = 2.0 * n * 4 * H + 5.0 * E + 4 * (n + 1), 3.0 * n * 4 * H + 5.0 * E + 4 * (n + 1)
        print(f'B={B} n={n} E={E} ({K} buffer sets): fwd {f:6.1f}us {fb / f / 1e3:6.0f} GB/s | bwd {b:6.1f}us {bb / b / 1e3:6.0f} GB/s | '
              f'copy [n,H] {c:6.1f}us {2.0 * n * 4 * H / c / 1e3:6.0f} GB/s | a+b [n,H] {c2:6.1f}us {3.0 * n * 4 * H / c2 / 1e3:6.0f} GB/s',
              flush=True)


if __name__ == '__main__':
    main()
