"""Helpers shared by the parity tests."""
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# Floating-point tolerance of the parity contract: fp32 activations and
# gradients within rtol 1e-4 of the reference (atol scaled to the tensor).
RTOL = 1e-4


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


def check_close(got, want, what, rtol=RTOL, atol_scale=1e-5):
    """`want` is either a tensor or a compressed record from the fixture
    script ({'full': t} or {'sample','stride','l2','sum'})."""
    got = got.detach().float().cpu()
    if isinstance(want, dict):
        if 'full' in want:
            ref = want['full']
            got = got.reshape(ref.shape)
        else:
            flat = got.flatten()
            assert tuple(got.shape) == tuple(want['shape']) or flat.numel() == int(torch.tensor(want['shape']).prod()), what
            l2 = float(flat.double().norm())
            assert abs(l2 - want['l2']) <= rtol * max(want['l2'], 1e-12) + 1e-7, f'{what}: l2 {l2} vs {want["l2"]}'
            got = flat[::want['stride']]
            ref = want['sample']
    else:
        ref = want.detach().float().cpu()
        got = got.reshape(ref.shape)
    scale = float(ref.abs().max()) if ref.numel() else 0.0
    atol = atol_scale * max(scale, 1e-6)
    ok = torch.allclose(got, ref, rtol=rtol, atol=atol)
    if not ok:
        err = (got - ref).abs()
        i = int(err.argmax())
        raise AssertionError(f'{what}: max abs err {float(err.max()):.3e} at {i} '
                             f'(got {float(got.flatten()[i]):.6e}, want {float(ref.flatten()[i]):.6e}, scale {scale:.3e})')
