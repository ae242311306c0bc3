#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM entry points at the shapes of the training step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_vae_amd import _lib  # noqa: E402


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
    return a.elapsed_time(b) / iters * 1e3  # us


def main():
    L = _lib.lib()
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    shapes = [(2964, 300, 300), (2964, 300, 200), (3000, 328, 656), (3000, 656, 328), (1408, 600, 100), (1408, 302, 200),
              (128, 400, 1200), (128, 200, 200), (23000, 300, 300), (95000, 300, 300)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
    for M, N, K in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * K ** -0.5
        b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        dy = torch.randn(M, N, device=dev)
        dx = torch.empty(M, K, device=dev)
        dw = torch.empty(N, K, device=dev)
        db = torch.empty(N, device=dev)
        nbytes = L.psvae_linear_scratch_bytes(M, N, K, 1)
        sc = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        fl = 2.0 * M * N * K
        t_f = timeit(lambda: L.psvae_linear_forward(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, N, K, 1,
                                                    sc.data_ptr(), sc.numel(), st))
        t_x = timeit(lambda: L.psvae_linear_backward_input(dy.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, M, N, K, None, 0, 0, st))
        t_w = timeit(lambda: L.psvae_linear_backward_weight(dy.data_ptr(), N, 0, x.data_ptr(), K, 0, 1, dw.data_ptr(), db.data_ptr(),
                                                            M, N, K, sc.data_ptr(), sc.numel(), st))
        t_ref = timeit(lambda: torch.nn.functional.linear(x, w, b))
        print(f'M{M} N{N} K{K}: fwd {t_f:7.1f}us {fl / t_f / 1e6:6.1f}TF | dx {t_x:7.1f}us {fl / t_x / 1e6:6.1f}TF | '
              f'dw {t_w:7.1f}us {fl / t_w / 1e6:6.1f}TF | torch/hipBLASLt fwd {t_ref:7.1f}us {fl / t_ref / 1e6:6.1f}TF', flush=True)


if __name__ == '__main__':
    main()
