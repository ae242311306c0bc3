"""`VAEPieceDecoder.forward` called on its own (the fused step in decoder-only mode) against the
CPU oracle run on the same graph conditions: z, the four losses, the gradient with respect to the
conditions, the gradient that comes back through the returned z, and the decoder's parameter
gradients."""
import pytest
import torch

from oracle import psvae_oracle as O
from tests.test_gpu_step import build, collate_both
from tests.util import check_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape,B,V', [('zinc', 24, 300), ('qm9', 9, 120)])
def test_decoder_alone_matches_oracle(shape, B, V):
    from ps_vae_amd import synth
    d = O.default_dims(V)
    model, tok, sd = build(d, V)
    mols = synth.make_molecules(300 + B, shape, V, B)
    ref_batch, batch = collate_both(mols, tok, V, 5)
    gen = torch.Generator().manual_seed(8)
    eps = torch.randn(B, d['latent'], generator=gen)
    conds = 0.5 * torch.randn(B, d['graph'], generator=gen)
    wz = torch.randn(B, d['latent'], generator=gen)           # the caller's use of z: loss term sum(z * wz)

    # oracle on the same conditions
    p = {k: v.clone() for k, v in sd.items()}
    names = [n for n in O.trainable_names(d) if n.startswith('decoder.')]
    for n in names:
        p[n].requires_grad_(True)
    c_ref = conds.clone().requires_grad_(True)
    keep = {}
    ref = O.forward(p, d, ref_batch, eps, keep, conds=c_ref)
    (ref['rec'] + 0.5 * ref['kl'] + (keep['z'] * wz).sum()).backward()

    # HIP decoder on its own, with the arguments PSVAEModel.forward would pass (ps_vae_model.py:45-52)
    dec = model.decoder
    idx = torch.tensor(ref_batch['in_piece_edge_idx'], dtype=torch.long)
    c = conds.cuda().requires_grad_(True)
    for q in dec.parameters():
        q.grad = None
    z, (rec, piece, edge, kl) = dec(batch['x'], batch['x_pieces'], batch['x_pos'], batch['edge_index'][:, idx.cuda()],
                                    batch['edge_attr'][idx.cuda()], batch['pieces'], c, batch['edge_select'],
                                    batch['golden_edge'], eps=eps.cuda())
    check_close(z, keep['z'], 'z')
    for name, got in (('rec', rec), ('piece', piece), ('edge', edge), ('kl', kl)):
        want = float(ref[name].detach())
        assert abs(float(got.detach()) - want) <= 1e-4 * max(1.0, abs(want)), (name, float(got.detach()), want)
    (rec + 0.5 * kl + (z * wz.cuda()).sum()).backward()
    check_close(c.grad, c_ref.grad, 'd conds', atol_scale=5e-5)
    named = dict(dec.named_parameters())
    for n in names:
        g_ref = p[n].grad
        q = named[n[len('decoder.'):]]
        if g_ref is None:
            assert q.grad is None or float(q.grad.abs().max()) == 0.0, n
        else:
            check_close(q.grad, g_ref, f'grad {n}', atol_scale=5e-5, bad_frac=1e-3)
    # accuracies of the same call
    _, (_, accus) = dec(batch['x'], batch['x_pieces'], batch['x_pos'], batch['edge_index'][:, idx.cuda()],
                        batch['edge_attr'][idx.cuda()], batch['pieces'], c.detach(), batch['edge_select'],
                        batch['golden_edge'], return_accu=True, eps=eps.cuda())
    logits = keep['logits'].view(-1, keep['logits'].shape[-1])
    gold = ref_batch['pieces'][:, 1:].flatten()
    not_pad = gold != d['num_piece'] - 2
    want_piece = ((logits.argmax(-1) == gold) & not_pad).sum().item() / not_pad.sum().item()
    want_edge = (keep['pair_logits'].argmax(-1) == ref_batch['golden_edge']).sum().item() / len(ref_batch['golden_edge'])
    assert accus == pytest.approx((want_piece, want_edge), abs=1e-6)
