"""The training step under every kernel-selection option (`psvae_option_set`): each setting replaces one or more
kernels of the default step by their alternative form - the fused kernels on the fp32 matrix instruction, a GINE
application as three launches, everything on one stream, the link scorer as one launch per direction or as a GEMM
chain, the recurrence on one compute unit, weight gradients on the fp32 / cut-operand / 160-tile kernels - and the
step must match the CPU oracle at the benchmark shape (ZINC, 128 graphs, vocabulary 300) under each of them, to the
same bounds as under the defaults.  One process, options flipped between calls."""
import pytest
import torch

from oracle import psvae_oracle as O
from tests.test_gpu_step import build, collate_both, compare_forward_backward

pytestmark = pytest.mark.gpu

SETTINGS = [
    {'split_bf16': 0},                       # fused GINE + scorer kernels on v_mfma_f32_16x16x4_f32
    {'split_bf16': 0, 'fused_scorer': 1},    # ... and the fp32 one-launch scorer backward
    {'fused_layer': 0},                      # aggregate + GEMM + GEMM, scorer as a GEMM chain, GEMM pair backward
    {'streams': 0},                          # no side streams
    {'fused_scorer': 0},                     # scorer as pair gather + GEMM chain + head
    {'fused_scorer': 1},                     # one-launch scorer backward at a size where the chain is the default
    {'fused_bwd': 0},                        # GINE MLP backward as two GEMMs with the ReLU mask in the epilogue
    {'gru_multi_cu': 0},                     # single-CU recurrence
    {'gru_tiles': 2},                        # two sequence tiles per workgroup of the multi-CU recurrence
    {'split_gemm': 0, 'wide_gemm': 0},       # every weight gradient on the fp32 64-tile kernel
    {'split_gemm': 1, 'wide_gemm': 0},       # every weight gradient on the cut-operand 64-tile kernel
    {'wide_gemm': 1},                        # 160-tile kernel wherever the reduction is long enough
    {'split_rows': 64},                      # 64-node workgroups in the cut-operand GINE kernels
]


@pytest.fixture(scope='module')
def case():
    from ps_vae_amd import synth
    V, B = 300, 128
    d = O.default_dims(V)
    model, tok, sd = build(d, V)
    mols = synth.make_molecules(100 + B, 'zinc', V, B)
    ref_batch, batch = collate_both(mols, tok, V, 7)
    eps = torch.randn(B, d['latent'], generator=torch.Generator().manual_seed(5))
    return model, d, sd, ref_batch, batch, eps


@pytest.mark.parametrize('setting', SETTINGS, ids=lambda s: ','.join(f'{k}={v}' for k, v in s.items()))
def test_step_matches_oracle_under_option(case, setting):
    from ps_vae_amd import _lib
    model, d, sd, ref_batch, batch, eps = case
    old = {k: _lib.set_option(k, v) for k, v in setting.items()}
    try:
        compare_forward_backward(model, d, sd, ref_batch, batch, eps, with_fp64=True)
        torch.cuda.synchronize()
    finally:
        for k, v in old.items():
            _lib.set_option(k, v)
    assert not model.recurrence_gave_up()


def test_recurrence_redone_on_one_cu_when_the_multi_cu_kernels_give_up(case):
    """Option gru_spin_ticks = 0 makes every wait of the multi-CU recurrence give up at once.  The single-CU
    kernels queued behind it must then redo the recurrence inside the same step: results match the oracle, the
    status word counts one redo per direction, and without the option nothing is redone."""
    from ps_vae_amd import _lib
    model, d, sd, ref_batch, batch, eps = case
    before = model.recurrence_redone()
    compare_forward_backward(model, d, sd, ref_batch, batch, eps, with_fp64=True)
    assert model.recurrence_redone() == before, 'a default step must not redo its recurrence'
    old = _lib.set_option('gru_spin_ticks', 0)
    try:
        compare_forward_backward(model, d, sd, ref_batch, batch, eps, with_fp64=True)
        torch.cuda.synchronize()
    finally:
        _lib.set_option('gru_spin_ticks', old)
    assert model.recurrence_redone() == before + 2      # forward and backward
    assert not model.recurrence_gave_up()
    g_redo = model.flat_buffers()[1].clone()
    # the redo is the single-CU recurrence: same bits as forcing it
    old = _lib.set_option('gru_multi_cu', 0)
    try:
        model.zero_grad()
        pred, (rec, _, _, kl) = model(batch, eps=eps)
        model.weighted_loss(pred, rec, kl).backward()
    finally:
        _lib.set_option('gru_multi_cu', old)
    assert torch.equal(model.flat_buffers()[1], g_redo)


@pytest.mark.parametrize('split', [-1, 0])
def test_link_scorer_width_outside_the_one_launch_kernels(split):
    """hidden = 340, latent = 56: P/2 = 368 lies inside the one-launch scorer kernels' column range but, with cut
    operands, its images do not fit the LDS - the step must take the GEMM chain by itself (and the one-launch
    kernels on the fp32 instruction, where they fit) and match the oracle."""
    from ps_vae_amd import _lib, synth
    V, B = 300, 24
    d = O.default_dims(V, hidden=340)
    L = _lib.lib()
    old = _lib.set_option('split_bf16', split)
    try:
        assert (L.psvae_scorer_scratch_bytes(340, 56) > 0) == (split == 0)
        assert L.psvae_scorer_scratch_bytes(300, 56) > 0
        model, tok, sd = build(d, V)
        mols = synth.make_molecules(641, 'zinc', V, B)
        ref_batch, batch = collate_both(mols, tok, V, 9)
        eps = torch.randn(B, d['latent'], generator=torch.Generator().manual_seed(6))
        compare_forward_backward(model, d, sd, ref_batch, batch, eps, with_fp64=True)
    finally:
        _lib.set_option('split_bf16', old)
