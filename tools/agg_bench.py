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


def timeit(fn, iters=200):
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
    L = _lib.lib()
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    H, t = 300, 4
    for B in (128, 1024, 4096):
        mols = make_molecules(2021, 'zinc', 300, B)
        src, dst, typ, off = [], [], [], 0
        for m in mols:
            for (a, b), ty in zip(m['edge_index'], m['edge_attr']):
                src += [a + off, b + off]; dst += [b + off, a + off]; typ += [ty, ty]
            off += len(m['x'])
        n, E = off, len(src)
        csr = Csr(torch.tensor([src, dst], device=dev), torch.tensor(typ, device=dev), n)
        stack = torch.randn(n, (t + 1) * H, device=dev)
        x = stack[:, :H]
        out = torch.empty(n, H, device=dev)
        ew, eb = torch.randn(H, 4, device=dev), torch.randn(H, device=dev)
        da, dx = torch.randn(n, H, device=dev), torch.empty(n, H, device=dev)
        dw, db = torch.zeros(H, 4, device=dev), torch.zeros(H, device=dev)
        part = torch.empty(max(1, L.psvae_gine_edge_grad_scratch(n, H)), device=dev)
        f = timeit(lambda: L.psvae_gine_aggregate_forward(x.data_ptr(), (t + 1) * H, csr.rowptr.data_ptr(), csr.col.data_ptr(),
                                                          csr.etype.data_ptr(), ew.data_ptr(), eb.data_ptr(), 1.0, out.data_ptr(), H, n, H, st))
        b = timeit(lambda: L.psvae_gine_aggregate_backward(da.data_ptr(), H, x.data_ptr(), (t + 1) * H, csr.rowptr_t.data_ptr(),
                                                           csr.col_t.data_ptr(), csr.etype_t.data_ptr(), ew.data_ptr(), eb.data_ptr(), 1.0,
                                                           None, 0, None, dx.data_ptr(), H, None, None, part.data_ptr(), n, H, st))
        c = timeit(lambda: out.copy_(da))
        xd = x.contiguous()
        f_dense = timeit(lambda: L.psvae_gine_aggregate_forward(xd.data_ptr(), H, csr.rowptr.data_ptr(), csr.col.data_ptr(),
                                                                csr.etype.data_ptr(), ew.data_ptr(), eb.data_ptr(), 1.0, out.data_ptr(), H, n, H, st))
        empty = Csr(torch.zeros(2, 0, dtype=torch.long, device=dev), torch.zeros(0, dtype=torch.long, device=dev), n)
        f_empty = timeit(lambda: L.psvae_gine_aggregate_forward(xd.data_ptr(), H, empty.rowptr.data_ptr(), empty.col.data_ptr(),
                                                                empty.etype.data_ptr(), ew.data_ptr(), eb.data_ptr(), 1.0, out.data_ptr(), H, n, H, st))
        print(f'   fwd dense x {f_dense:6.1f}us | fwd no edges {f_empty:6.1f}us')
        fb, bb = 2.0 * n * 4 * H + 5.0 * E + 4 * (n + 1), 3.0 * n * 4 * H + 5.0 * E + 4 * (n + 1)
        print(f'B={B} n={n} E={E}: fwd {f:6.1f}us {fb / f / 1e3:6.0f} GB/s | bwd {b:6.1f}us {bb / b / 1e3:6.0f} GB/s | '
              f'copy [n,H] {c:6.1f}us {2.0 * n * 4 * H / c / 1e3:6.0f} GB/s', flush=True)


if __name__ == '__main__':
    main()
