"""Helpers shared by the parity tests."""
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# Floating-point tolerance of the parity contract: fp32 activations and
# gradients within rtol 1e-4 of the reference (atol scaled to the tensor).
RTOL = 1e-4


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


def check_close(got, want, what, rtol=RTOL, atol_scale=1e-5, bad_frac=0.0, outlier_scale=2e-3):
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
    bad = ~torch.isclose(got, ref, rtol=rtol, atol=atol)
    ok = not bool(bad.any())
    if not ok and bad_frac > 0.0:
        # tolerated outliers (see the caller's comment): few, and bounded relative to the tensor's scale
        ok = int(bad.sum()) <= max(bad_frac * bad.numel(), 4) and float((got - ref).abs().max()) <= outlier_scale * max(scale, 1e-6)
    if not ok:
        err = (got - ref).abs()
        i = int(err.argmax())
        raise AssertionError(f'{what}: max abs err {float(err.max()):.3e} at {i} '
                             f'(got {float(got.flatten()[i]):.6e}, want {float(ref.flatten()[i]):.6e}, scale {scale:.3e})')


def config_from_dims(d, tok):
    """Model config dict (the structure `src/train.py:130-132` assembles) from an oracle dims dict."""
    return {
        'lr': 1e-3, 'alpha': 0.1, 'beta': 0.0, 'max_beta': 0.01, 'step_beta': 0.002,
        'kl_warmup': 0, 'kl_anneal_iter': 1000, 'selected_properties': [0, 2],
        'predictor': {'dim_feature': d['latent'], 'dim_hidden': d['pred_hidden'], 'num_property': d['n_props']},
        'encoder': {'dim_in': d['d_atom'] + d['d_piece'] + d['d_pos'], 'num_edge_type': d['num_bond'],
                    'dim_hidden': d['hidden'], 'dim_out': d['graph'], 't': d['t_enc']},
        'vae_piece_decoder': {'atom_embedding_dim': d['d_atom'], 'piece_embedding_dim': d['d_piece'],
                              'max_pos': d['max_pos'], 'pos_embedding_dim': d['d_pos'],
                              'piece_hidden_dim': d['rnn'], 'node_hidden_dim': d['hidden'],
                              'num_edge_type': d['num_bond'], 'cond_dim': d['graph'],
                              'latent_dim': d['latent'], 'tokenizer': tok, 't': d['t_dec']},
    }


def check_mostly_close(got, want, what, rtol=1e-4, atol=2e-5, max_bad_frac=2e-4, max_step=6e-3):
    """For post-optimiser parameters: Adam's first steps move a weight by ~lr * sign(g),
    so an element whose gradient sits at rounding-noise level may differ by up to 2*lr.
    Everything else must agree; the tolerated fraction of such elements is tiny (at least one
    element, since some tensors are compared on a sample of ~150 values) and none may be further away than
    `max_step` (3 steps of lr 1e-3, both directions)."""
    got = got.detach().float().cpu().flatten()
    if isinstance(want, dict):
        if 'full' in want:
            ref = want['full'].flatten()
        else:
            got, ref = got[::want['stride']], want['sample']
    else:
        ref = want.detach().float().cpu().flatten()
    bad = ~torch.isclose(got, ref, rtol=rtol, atol=atol)
    allowed = max(1, int(max_bad_frac * bad.numel()))
    worst = float((got - ref).abs().max()) if bad.numel() else 0.0
    assert int(bad.sum()) <= allowed, f'{what}: {int(bad.sum())}/{bad.numel()} elements differ (max {worst:.3e})'
    assert worst <= max_step, f'{what}: an element moved {worst:.3e} away, more than Adam can move it'
