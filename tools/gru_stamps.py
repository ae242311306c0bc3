#!/usr/bin/env python3
"""Where a time step of the multi-CU GRU forward recurrence goes: psvae_gru_forward_stamped at the benchmark's shape
(128 sequences, hidden 200), the split of a step between the wait for the other workgroups' h_t (the cross-CU hop), the
products with their K-split sum through LDS, and the gate arithmetic with the publication of h_{t+1}.

    python tools/gru_stamps.py [B] [T] [R] > profiles/r03_gru_stamps.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_vae_amd import _lib  # noqa: E402


def main():
    B, T, R = (int(v) for v in (sys.argv[1:4] + ['128', '12', '200'][len(sys.argv) - 1:]))
    L = _lib.lib()
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device='cpu').manual_seed(1)
    gi = torch.randn(B, T, 3 * R, generator=g).to(dev)
    h0 = torch.randn(B, R, generator=g).to(dev)
    whh = (torch.randn(3 * R, R, generator=g) * R ** -0.5).to(dev)
    bhh = torch.randn(3 * R, generator=g).to(dev)
    hseq, hprev = torch.empty(B, T, R, device=dev), torch.empty(B, T, R, device=dev)
    gates = torch.empty(B, T, 4 * R, device=dev)
    nx = L.psvae_gru_exchange_bytes(B, T, R)
    exch = torch.zeros(nx // 8 + 1, dtype=torch.int64, device=dev)
    stamps = torch.zeros(T, 4, dtype=torch.int64, device=dev)
    want = torch.empty_like(hseq)
    _lib.check(L.psvae_gru_forward(gi.data_ptr(), h0.data_ptr(), whh.data_ptr(), bhh.data_ptr(), want.data_ptr(), hprev.data_ptr(),
                                   gates.data_ptr(), B, T, R, exch.data_ptr(), nx, st), 'psvae_gru_forward')
    rows = []
    for rep in range(60):
        _lib.check(L.psvae_gru_forward_stamped(gi.data_ptr(), h0.data_ptr(), whh.data_ptr(), bhh.data_ptr(), hseq.data_ptr(),
                                               hprev.data_ptr(), gates.data_ptr(), B, T, R, exch.data_ptr(), nx,
                                               stamps.data_ptr(), st), 'psvae_gru_forward_stamped')
        torch.cuda.synchronize()
        if rep >= 10:
            rows.append(stamps.cpu().double() * 0.01)      # 100 MHz ticks -> microseconds
    assert torch.equal(hseq, want), 'the stamped kernel must produce the same bits'
    s = torch.stack(rows)                                   # [reps, T, 4]
    wait = s[:, 1:, 1] - s[:, 1:, 0]                         # steps 1..T-1 (step 0 reads h0 from memory, no hop)
    prod = s[:, 1:, 2] - s[:, 1:, 1]
    gate = s[:, 1:, 3] - s[:, 1:, 2]
    step = s[:, 1:, 0] - s[:, :-1, 0]
    hop = s[:, 1:, 1] - s[:, :-1, 3]                         # own publication of h_t -> everybody's h_t here
    print(f'multi-CU GRU forward, {B} sequences x {T} steps, hidden {R}: {R // 16 + (R % 16 > 0)} workgroups per sequence tile, '
          f'{(B + 15) // 16} tiles; stamps of workgroup (0,0), {len(rows)} runs, microseconds (10 ns clock)')
    for name, v in (('whole time step', step), ('wait for h_t of the other workgroups (top of step -> all words carry the tag)', wait),
                    ('  of which after this workgroup published its own h_t (cross-CU hop + slowest producer)', hop),
                    ('products on the matrix cores + K-split partial sums through LDS + barrier', prod),
                    ('gate arithmetic, publication of h_{t+1}, stores of h / gates', gate)):
        print(f'  {name}: mean {float(v.mean()):.2f}, median {float(v.median()):.2f}, p10 {float(v.quantile(0.1)):.2f}, p90 {float(v.quantile(0.9)):.2f}')
    print(f'  whole recurrence (first stamp -> last stamp): mean {float((s[:, -1, 3] - s[:, 0, 0]).mean()):.1f}')


if __name__ == '__main__':
    main()
