#!/usr/bin/env python3
"""One GINE application: the fused launch against aggregation + two GEMM launches, at the training step's size
(two stacks of 2964 rows are one grouped launch there; here one stack of n rows) and at the bulk-encoding size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_vae_amd import _lib, ops, synth  # noqa: E402


def graph(graphs, dev):
    mols = synth.make_molecules(77, 'zinc', 300, graphs)
    src, dst, typ, off = [], [], [], 0
    for m in mols:
        for (a, b), ty in zip(m['edge_index'], m['edge_attr']):
            src += [a + off, b + off]; dst += [b + off, a + off]; typ += [ty, ty]
        off += len(m['x'])
    return ops.Csr(torch.tensor([src, dst], device=dev), torch.tensor(typ, device=dev), off), off


def main():
    dev = torch.device('cuda:0')
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    H = 300
    for graphs in (128, 256, 1024, 4096):
        csr, n = graph(graphs, dev)
        x = torch.randn(n, H, device=dev)
        ew, eb = torch.randn(H, 4, device=dev), torch.randn(H, device=dev)
        w1, w2 = torch.randn(H, H, device=dev) * 0.05, torch.randn(H, H, device=dev) * 0.05
        b1, b2 = torch.randn(H, device=dev), torch.randn(H, device=dev)
        agg, hid, y = (torch.empty(n, H, device=dev) for _ in range(3))
        scratch = torch.empty(L.psvae_linear_scratch_bytes(n, H, H, 1), dtype=torch.uint8, device=dev)
        pk = torch.empty(L.psvae_gine_layer_scratch_bytes(H), dtype=torch.uint8, device=dev)

        def fused(save=True):
            _lib.check(L.psvae_gine_layer_forward(x.data_ptr(), H, csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.etype.data_ptr(),
                                                  ew.data_ptr(), eb.data_ptr(), 1.0, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                                  b2.data_ptr(), agg.data_ptr() if save else None, hid.data_ptr() if save else None,
                                                  y.data_ptr(), H, n, H, pk.data_ptr(), pk.numel(), st))

        def unfused():
            _lib.check(L.psvae_gine_aggregate_forward(x.data_ptr(), H, csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.etype.data_ptr(),
                                                      ew.data_ptr(), eb.data_ptr(), 1.0, agg.data_ptr(), H, n, H, st))
            _lib.check(L.psvae_linear_forward(agg.data_ptr(), H, w1.data_ptr(), b1.data_ptr(), hid.data_ptr(), H, n, H, H, 1,
                                              scratch.data_ptr(), scratch.numel(), st))
            _lib.check(L.psvae_linear_forward(hid.data_ptr(), H, w2.data_ptr(), b2.data_ptr(), y.data_ptr(), H, n, H, H, 0,
                                              scratch.data_ptr(), scratch.numel(), st))

        def timed(fn, iters=50):
            for _ in range(5):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / iters * 1e3

        tf, ti, tu = timed(fused), timed(lambda: fused(False)), timed(unfused)
        fl = 4.0 * n * H * H
        print(f'{graphs} graphs, n={n}: fused {tf:.1f} us ({fl / tf / 1e6:.1f} TF), fused inference {ti:.1f} us '
              f'({fl / ti / 1e6:.1f} TF), three launches {tu:.1f} us ({fl / tu / 1e6:.1f} TF)', flush=True)


if __name__ == '__main__':
    main()
