"""The counter-based epsilon draw (`psvae_normal_rows`) that stands where the reference calls
`torch.randn_like` (`src/modules/vae_piece_decoder.py:61`).

The reference holds no vector for its draw (it comes from torch's global generator), so parity of the VALUES is
unpinned by the reference; what is pinned: the generator core against the published known-answer vectors of
Philox-4x32-10 (Random123 `kat_vectors`), the device kernel against the numpy restatement, and the property the
draw exists for - a row depends on its global index only, however the rows are dealt to calls."""
import numpy as np
import pytest
import torch

from oracle import psvae_oracle as O


def test_philox_known_answers():
    kat = [
        ([0, 0, 0, 0], (0, 0), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, (0xffffffff, 0xffffffff), [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0), [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        got = [int(np.asarray(w).reshape(-1)[0]) for w in O.philox4x32_10([np.array([c], dtype=np.uint64) for c in ctr], *key)]
        assert got == want


def test_restated_draw_is_standard_normal_and_keyed_by_row():
    a = O.normal_rows(2021, 3, 0, 8192, 56)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1.0) < 5e-3
    assert abs(float((a ** 4).mean()) - 3.0) < 0.1                       # kurtosis of N(0,1)
    assert torch.equal(O.normal_rows(2021, 3, 5000, 100, 56), a[5000:5100])
    assert not torch.equal(O.normal_rows(2021, 4, 0, 16, 56), a[:16])    # another draw index, other numbers
    assert not torch.equal(O.normal_rows(2022, 3, 0, 16, 56), a[:16])    # another seed
    wide = O.normal_rows(7, 1, 0, 4, 10)                                 # a width that is no multiple of 4
    assert tuple(wide.shape) == (4, 10) and torch.equal(wide[:, :8], O.normal_rows(7, 1, 0, 4, 8))


@pytest.mark.gpu
def test_device_draw_matches_the_restatement():
    from ps_vae_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for seed, draw, first, rows, width in [(2021, 0, 0, 128, 56), (2021, 17, 384, 512, 56), ((5 << 32) | 9, (3 << 32) | 1, (1 << 33) + 5, 33, 10)]:
        out = torch.full((rows, width), float('nan'), device='cuda')
        _lib.check(L.psvae_normal_rows(seed, draw, first, rows, width, out.data_ptr(), st))
        want = O.normal_rows(seed, draw, first, rows, width)
        assert torch.allclose(out.cpu(), want, rtol=1e-4, atol=2e-5), (seed, draw, first, (out.cpu() - want).abs().max())
    # shards of one global batch: bitwise the rows of the undivided draw
    whole = torch.empty(1024, 56, device='cuda')
    _lib.check(L.psvae_normal_rows(1, 2, 0, 1024, 56, whole.data_ptr(), st))
    for lo, hi in [(0, 128), (128, 256), (512, 1024)]:
        part = torch.empty(hi - lo, 56, device='cuda')
        _lib.check(L.psvae_normal_rows(1, 2, lo, hi - lo, 56, part.data_ptr(), st))
        assert torch.equal(part, whole[lo:hi])
    assert L.psvae_normal_rows(1, 2, -1, 4, 56, whole.data_ptr(), st) == -2      # PSVAE_E_SHAPE
