"""The remaining benchmark configurations as parity cases: encoder-only latent encoding over
batches of 4096 graphs, and the stress shape (60-atom graphs, 8 GINE iterations, hidden 512,
every cross-piece pair scored).  Small instances are compared with the CPU oracle directly; the
full sizes are checked through properties that do not depend on size: graphs of a batch are
independent, so results of a big batch must equal those of its parts, and loss sums add up."""
import pytest
import torch

from oracle import collate_oracle as CO
from oracle import psvae_oracle as O
from tests.test_gpu_step import build, compare_forward_backward
from tests.util import check_close

pytestmark = pytest.mark.gpu


class _AlwaysSelect:
    """Stands in for `random` in the oracle collate: every cross-piece pair is kept."""
    @staticmethod
    def random():
        return 0.0


def collate_all_pairs(mols, tok, vocab_size):
    from ps_vae_amd.data import BPEMolDataset, to_device
    ref_batch = CO.collate([CO.to_matrix_form(m) for m in mols], vocab_size, rng=_AlwaysSelect)
    batch = BPEMolDataset.process_step3([BPEMolDataset.process_step2(m, tok) for m in mols], tok, all_pairs=True)
    return ref_batch, to_device(batch, 'cuda:0')


def collate_gpu(mols, tok, all_pairs=False, seed=11):
    import random
    from ps_vae_amd.data import BPEMolDataset, to_device
    random.seed(seed)
    return to_device(BPEMolDataset.process_step3([BPEMolDataset.process_step2(m, tok) for m in mols], tok, all_pairs=all_pairs),
                     'cuda:0')


STRESS = dict(hidden=512, t_enc=8, t_dec=8)


def test_stress_shape_matches_oracle():
    from ps_vae_amd import synth
    V, B = 300, 3
    d = O.default_dims(V, **STRESS)
    model, tok, sd = build(d, V)
    mols = synth.make_molecules(500, 'stress', V, B)
    ref_batch, batch = collate_all_pairs(mols, tok, V)
    assert batch['_native'].sizes['S'] == int(ref_batch['edge_select'].sum()) > 2000 * B
    eps = torch.randn(B, d['latent'], generator=torch.Generator().manual_seed(9))
    compare_forward_backward(model, d, sd, ref_batch, batch, eps, with_fp64=True)


def _loss_sums(model, batch, eps):
    """The four loss sums (piece, edge, kl, pred) of a batch = means times their denominators."""
    pred, (rec, piece, edge, kl) = model(batch, eps=eps)
    den = batch['_native'].denoms.double().cpu()
    return torch.stack([piece.detach().double().cpu() * den[0], edge.detach().double().cpu() * den[1],
                        kl.detach().double().cpu() * den[2], pred.detach().double().cpu() * den[3]])


def test_stress_full_size_is_sum_of_its_parts():
    """32 graphs of 60 atoms, all pairs (about 100 k scorer rows): loss sums of the whole batch equal the
    sums over its two halves, gradients of the whole batch equal the combination of the halves' gradients,
    and two runs are bitwise identical."""
    from ps_vae_amd import synth
    V, B = 300, 32
    d = O.default_dims(V, **STRESS)
    model, tok, sd = build(d, V)
    mols = synth.make_molecules(501, 'stress', V, B)
    eps = torch.randn(B, d['latent'], generator=torch.Generator().manual_seed(3)).cuda()
    whole = collate_gpu(mols, tok, all_pairs=True)
    halves = [collate_gpu(mols[:B // 2], tok, all_pairs=True), collate_gpu(mols[B // 2:], tok, all_pairs=True)]
    assert whole['_native'].sizes['S'] > 90000

    def grads_of(batch, e, denoms=None):
        if denoms is not None:
            batch['_native'].denoms.copy_(denoms)       # divide by the whole batch's counts
        model.zero_grad()
        model.global_step = 1000
        model.training_step(batch, eps=e).backward()
        return model.flat_buffers()[1].clone()

    sums_whole = _loss_sums(model, whole, eps)
    sums_parts = _loss_sums(model, halves[0], eps[:B // 2]) + _loss_sums(model, halves[1], eps[B // 2:])
    assert torch.allclose(sums_whole, sums_parts, rtol=2e-5), (sums_whole, sums_parts)
    g_whole = grads_of(whole, eps)
    assert torch.equal(g_whole, grads_of(whole, eps)), 'two runs of the same step differ'
    den = whole['_native'].denoms.clone()
    g_parts = grads_of(halves[0], eps[:B // 2], den) + grads_of(halves[1], eps[B // 2:], den)
    check_close(g_parts, g_whole, 'gradient of the whole batch vs sum over halves', rtol=1e-4, atol_scale=2e-5, bad_frac=1e-3)


@pytest.mark.parametrize('B', [256, 512, 1024])
def test_vocab500_per_gpu_workloads_of_the_data_parallel_configuration(B):
    """Vocabulary 500 with 256, 512 and 1024 graphs per GPU: what one GPU holds when the global batch of 1024 is split
    over 4, 2 and 1 of them (at 1024 the single-CU recurrence, 64-node GINE workgroups, the one-launch scorer backward
    and the 192-workgroup split of the 160-tile weight gradients are the defaults).  A 48-graph slice of the batch is checked against the oracle (activations, losses,
    all gradients); the whole batch through size-independent properties: its loss sums equal the sums over its
    four quarters, its gradient equals the quarters' gradients (each divided by the whole batch's counts, as the
    data-parallel step divides) added up, and two runs are bitwise identical."""
    import random
    from ps_vae_amd import synth
    V = 500
    d = O.default_dims(V)
    model, tok, sd = build(d, V)
    mols = synth.make_molecules(900 + B, 'zinc', V, B)
    eps = torch.randn(B, d['latent'], generator=torch.Generator().manual_seed(21))
    # oracle on a slice
    from ps_vae_amd.data import BPEMolDataset, to_device
    random.seed(5)
    ref_slice = CO.collate([CO.to_matrix_form(m) for m in mols[:48]], V)
    random.seed(5)
    slice_ = to_device(BPEMolDataset.process_step3([BPEMolDataset.process_step2(m, tok) for m in mols[:48]], tok), 'cuda:0')
    compare_forward_backward(model, d, sd, ref_slice, slice_, eps[:48], with_fp64=True)
    # whole batch = its quarters
    eps = eps.cuda()
    q = B // 4
    random.seed(6)          # one stream of negative-pair draws, consumed quarter after quarter = whole batch in order
    parts = [to_device(BPEMolDataset.process_step3([BPEMolDataset.process_step2(m, tok) for m in mols[k * q:(k + 1) * q]], tok), 'cuda:0')
             for k in range(4)]
    random.seed(6)
    whole = to_device(BPEMolDataset.process_step3([BPEMolDataset.process_step2(m, tok) for m in mols], tok), 'cuda:0')
    assert whole['_native'].sizes['S'] == sum(p['_native'].sizes['S'] for p in parts)
    assert whole['_native'].sizes['B'] == B and whole['_native'].sizes['n'] > 20 * B

    def grads_of(batch, e, denoms=None):
        if denoms is not None:
            batch['_native'].denoms.copy_(denoms)
        model.zero_grad()
        model.global_step = 1000
        model.training_step(batch, eps=e).backward()
        return model.flat_buffers()[1].clone()

    sums_whole = _loss_sums(model, whole, eps)
    sums_parts = sum(_loss_sums(model, parts[k], eps[k * q:(k + 1) * q]) for k in range(4))
    assert torch.allclose(sums_whole, sums_parts, rtol=2e-5), (sums_whole, sums_parts)
    g_whole = grads_of(whole, eps)
    assert torch.equal(g_whole, grads_of(whole, eps)), 'two runs of the same step differ'
    den = whole['_native'].denoms.clone()
    g_parts = sum(grads_of(parts[k], eps[k * q:(k + 1) * q], den) for k in range(4))
    check_close(g_parts, g_whole, 'gradient of the whole batch vs sum over quarters', rtol=1e-4, atol_scale=2e-5, bad_frac=1e-3)
    assert not model.recurrence_gave_up()


def test_encode_4096_graphs_equals_its_parts_and_the_oracle():
    """Encoder-only path at the size of the bulk-encoding configuration (4096 ZINC-shape graphs, about
    95 k atoms): every graph's latent mean must equal what a 64-graph batch holding it gives, and that
    small batch is checked against the oracle."""
    from ps_vae_amd import synth
    V, B, part = 300, 4096, 64
    d = O.default_dims(V)
    model, tok, sd = build(d, V)
    mols = synth.make_molecules(777, 'zinc', V, B)
    whole = collate_gpu(mols, tok)
    assert whole['_native'].sizes['n'] > 80000
    mean, lvar = model.encode(whole)
    assert mean.shape == (B, d['latent']) and bool(torch.isfinite(mean).all()) and bool((lvar <= 0).all())
    for start in (0, 1984, B - part):
        sub = mols[start:start + part]
        m_sub, lv_sub = model.encode(collate_gpu(sub, tok))
        check_close(mean[start:start + part], m_sub, f'z_mean of graphs {start}..', rtol=1e-4, atol_scale=1e-5)
        check_close(lvar[start:start + part], lv_sub, f'z_log_var of graphs {start}..', rtol=1e-4, atol_scale=1e-5)
    import random
    random.seed(11)
    ref_batch = CO.collate([CO.to_matrix_form(m) for m in mols[:part]], V)
    mean_ref, lvar_ref = O.get_z_mean(sd, d, ref_batch)
    check_close(mean[:part], mean_ref, 'z_mean vs oracle')
    check_close(lvar[:part], lvar_ref, 'z_log_var vs oracle')
