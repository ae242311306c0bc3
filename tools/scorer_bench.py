#!/usr/bin/env python3
"""The link scorer's forward pass: the fused launch against gather + three GEMMs + head, alone on the GPU."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_vae_amd import _lib  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    H, Z = 300, 56
    P, P2 = 656, 328
    for S, n, B in ((3000, 2964, 128), (6000, 5900, 256), (24000, 23700, 1024)):
        emb, z = torch.randn(n, H, device=dev), torch.randn(B, Z, device=dev)
        idx = [torch.randint(0, hi, (S,), device=dev, dtype=torch.int32) for hi in (n, n, B, 4)]
        ws = [torch.randn(*sh, device=dev) * 0.05 for sh in ((P2, P), (P2,), (P2, P2), (P2,), (P, P2), (P,), (4, P), (4,))]
        out = [torch.empty(*sh, device=dev) for sh in ((S, P), (S, P2), (S, P2), (S, P), (S, 4), (S,), (S,))]
        nbytes = L.psvae_scorer_scratch_bytes(H, Z)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        gs = torch.empty(L.psvae_linear_scratch_bytes(S, P, P, 1), dtype=torch.uint8, device=dev)

        def fused():
            _lib.check(L.psvae_scorer_forward(emb.data_ptr(), H, z.data_ptr(), *[t.data_ptr() for t in idx], *[t.data_ptr() for t in ws],
                                              *[t.data_ptr() for t in out], S, H, Z, scratch.data_ptr(), nbytes, st))

        def gemms():
            _lib.check(L.psvae_linear_forward(out[0].data_ptr(), P, ws[0].data_ptr(), ws[1].data_ptr(), out[1].data_ptr(), P2, S, P2, P, 1, gs.data_ptr(), gs.numel(), st))
            _lib.check(L.psvae_linear_forward(out[1].data_ptr(), P2, ws[2].data_ptr(), ws[3].data_ptr(), out[2].data_ptr(), P2, S, P2, P2, 1, gs.data_ptr(), gs.numel(), st))
            _lib.check(L.psvae_linear_forward(out[2].data_ptr(), P2, ws[4].data_ptr(), ws[5].data_ptr(), out[3].data_ptr(), P, S, P, P2, 0, gs.data_ptr(), gs.numel(), st))

        def timed(fn, iters=30):
            for _ in range(3):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / iters * 1e3

        fl = 2.0 * S * (P * P2 * 2 + P2 * P2)
        tf, tg = timed(fused), timed(gemms)
        print(f'S={S}: fused (with pack, gather, head) {tf:.1f} us ({fl / tf / 1e6:.1f} TF); three GEMM launches alone {tg:.1f} us '
              f'({fl / tg / 1e6:.1f} TF)', flush=True)


if __name__ == '__main__':
    main()
