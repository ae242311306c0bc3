"""Epoch-loop plumbing without a GPU: early stopping follows `VAEEarlyStopping`'s rules, and a
checkpoint written here restores the model (state dict, step counters) through a weights-only load."""
import math

import torch

from ps_vae_amd import PSVAEModel, Tokenizer, default_args, full_config
from ps_vae_amd.trainer import VAEEarlyStopping, finish_anneal_step, load_checkpoint, save_checkpoint


def test_early_stopping_waits_for_the_kl_schedule():
    es = VAEEarlyStopping(finish_anneal_step=1000, patience=2)
    # while beta is still being raised nothing counts: the record is reset after every validation
    assert [es.on_validation_end(v, step) for v, step in [(5.0, 100), (6.0, 400), (7.0, 900)]] == [False, False, False]
    assert es.best_score == math.inf and es.wait_count == 0
    # afterwards: improvement resets the counter, `patience` misses in a row stop the run
    assert es.on_validation_end(4.0, 1200) is False and es.best_score == 4.0
    assert es.on_validation_end(4.5, 1500) is False and es.wait_count == 1
    assert es.on_validation_end(3.9, 1800) is False and es.wait_count == 0
    assert es.on_validation_end(3.95, 2100) is False
    assert es.on_validation_end(3.91, 2400) is True


def test_finish_anneal_step_matches_the_training_script():
    cfg = dict(kl_warmup=0, max_beta=0.01, step_beta=0.002, kl_anneal_iter=1000)
    assert finish_anneal_step(cfg) == 0 + (0.01 // 0.002 - 1) * 1000


def _model(seed, V=25):
    tok = Tokenizer.synthetic(V)
    torch.manual_seed(seed)
    return PSVAEModel(full_config(default_args(), tok), tok)


def _with_moments(model):
    """An optimiser whose two moment buffers hold, for every parameter, values that name it."""
    from ps_vae_amd.optim import FusedClipAdam
    opt = FusedClipAdam(model, lr=1e-3)
    params, _ = model.flat_buffers()
    opt.exp_avg, opt.exp_avg_sq = torch.zeros_like(params), torch.zeros_like(params)
    opt.step_count, opt.epoch = 7, 2
    named = dict(model.named_parameters())
    for i, k in enumerate(model.trained_names()):
        o, n = model._offsets[k], named[k].numel()
        opt.exp_avg[o:o + n] = float(i + 1) + torch.arange(n) * 1e-6
        opt.exp_avg_sq[o:o + n] = float(100 + i)
    return opt


def test_checkpoint_round_trip(tmp_path):
    a = _model(1)
    a.global_step, a.eps_draws, a.eps_seed = 321, 77, 5
    opt_a = _with_moments(a)
    path = tmp_path / 'run' / 'best.ckpt'
    save_checkpoint(str(path), a, opt_a, epoch=4)
    ckpt = torch.load(path, weights_only=True)          # nothing in the file needs unpickling of code
    assert set(ckpt) == {'epoch', 'global_step', 'state_dict', 'optimizer_states', 'hyper_parameters', 'eps_state'}
    assert len(ckpt['state_dict']) == 51 and ckpt['hyper_parameters']['config']['alpha'] == a.config['alpha']
    # Adam's moments are stored per parameter name
    st = ckpt['optimizer_states'][0]
    assert set(st['exp_avg']) == set(a.trained_names()) and st['exp_avg']['decoder.to_vocab.weight'].shape == (27, 200)
    b = _model(2)
    opt_b = b.configure_optimizers()
    assert load_checkpoint(str(path), b, opt_b) == 4 and b.global_step == 321
    assert (b.eps_seed, b.eps_draws) == (5, 77), 'the epsilon draw goes on where the saved run stood'
    for (k, v), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k == k2 and torch.equal(v, v2), k
    assert opt_b.step_count == 7 and opt_b.epoch == 2
    assert torch.equal(opt_b.exp_avg, opt_a.exp_avg) and torch.equal(opt_b.exp_avg_sq, opt_a.exp_avg_sq)


def test_optimizer_state_survives_a_change_of_the_flat_order():
    """Moments written while the flat buffer had ANOTHER parameter order land on the right parameters: per name
    directly; as a flat buffer only together with the order it was laid out in, and refused without it."""
    from ps_vae_amd import model as M
    a = _model(3)
    opt = _with_moments(a)
    named = dict(a.named_parameters())
    per_name = opt.state_dict()
    # a flat buffer in the order of an earlier layout (predictor first, piece branch behind the scorer)
    names = a.trained_names()
    old_order = [k for k in names if k.startswith('predictor.')] + [k for k in names if not k.startswith('predictor.')][::-1]
    assert old_order != names and sorted(old_order) == sorted(names)
    total = sum((named[k].numel() + 3) // 4 * 4 for k in old_order)
    old_flat = {key: torch.zeros(total) for key in ('exp_avg', 'exp_avg_sq')}
    off = 0
    for k in old_order:
        n = named[k].numel()
        for key in old_flat:
            old_flat[key][off:off + n] = per_name[key][k].reshape(-1)
        off += (n + 3) // 4 * 4
    assert not torch.equal(old_flat['exp_avg'], opt.exp_avg)
    b = _model(4)
    opt_b = b.configure_optimizers()
    opt_b.load_state_dict({'step': 7, 'epoch': 2, 'flat_order': old_order, **old_flat})
    assert torch.equal(opt_b.exp_avg, opt.exp_avg) and torch.equal(opt_b.exp_avg_sq, opt.exp_avg_sq)
    import pytest
    with pytest.raises(ValueError, match='flat_order'):
        b.configure_optimizers().load_state_dict({'step': 7, 'epoch': 2, **old_flat})
    # and the per-name form does not care how FLAT_ORDER is arranged when it is read back
    saved = list(M.FLAT_ORDER)
    try:
        M.FLAT_ORDER[:] = old_order
        c = _model(5)
        opt_c = c.configure_optimizers()
        opt_c.load_state_dict(per_name)
        named_c = dict(c.named_parameters())
        for i, k in enumerate(names):
            o, n = c._offsets[k], named_c[k].numel()
            assert float(opt_c.exp_avg[o]) == float(names.index(k) + 1) and float(opt_c.exp_avg_sq[o + n - 1]) == float(100 + names.index(k)), k
    finally:
        M.FLAT_ORDER[:] = saved
