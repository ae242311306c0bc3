#!/usr/bin/env python3
"""Weight-gradient GEMM with and without the bias column, beside torch's dy^T x."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_vae_amd import _lib  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

L = _lib.lib(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
shapes = [(95000, 300, 300), (95000, 320, 320), (95000, 256, 256), (11856, 300, 300)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev); dy = torch.randn(M, N, device=dev)
    dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    sc = torch.empty(max(L.psvae_linear_scratch_bytes(M, N, K, 1), 256), dtype=torch.uint8, device=dev)
    fl = 2.0 * M * N * K
    a = timeit(lambda: L.psvae_linear_backward_weight(dy.data_ptr(), N, 0, x.data_ptr(), K, 0, 1, dw.data_ptr(), db.data_ptr(), M, N, K, sc.data_ptr(), sc.numel(), st))
    b = timeit(lambda: L.psvae_linear_backward_weight(dy.data_ptr(), N, 0, x.data_ptr(), K, 0, 1, dw.data_ptr(), None, M, N, K, sc.data_ptr(), sc.numel(), st))
    ref = timeit(lambda: torch.mm(dy.t(), x))
    print(f'M{M} N{N} K{K}: dW with bias {a:7.1f}us {fl/a/1e6:5.1f}TF | without {b:7.1f}us {fl/b/1e6:5.1f}TF | torch dy^T x {ref:7.1f}us {fl/ref/1e6:5.1f}TF', flush=True)
