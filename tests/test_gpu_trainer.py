"""The epoch loop on the GPU: a few epochs over a small synthetic set lower the validation loss,
the best checkpoint restores the model exactly, and a resumed run continues bit for bit."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(tok, seed=3):
    from ps_vae_amd import PSVAEModel, default_args, full_config
    torch.manual_seed(seed)
    return PSVAEModel(full_config(default_args(), tok), tok).cuda()


def test_fit_validates_checkpoints_and_resumes(tmp_path):
    from ps_vae_amd import Tokenizer, synth
    from ps_vae_amd.data import get_dataloader, to_device
    from ps_vae_amd.trainer import fit, load_checkpoint, save_checkpoint, validate
    V = 40
    tok = Tokenizer.synthetic(V)
    train = get_dataloader(synth.make_molecules(1, 'qm9', V, 96), tok, batch_size=16, shuffle=False, num_workers=0)
    valid = get_dataloader(synth.make_molecules(2, 'qm9', V, 32), tok, batch_size=16, shuffle=False, num_workers=0)
    random.seed(0)
    torch.manual_seed(0)
    model = _model(tok)
    before = validate(model, valid, 'cuda')
    history = fit(model, train, valid, max_epochs=3, save_dir=str(tmp_path), patience=5)
    assert len(history) == 3 and history[-1]['global_step'] == 18
    assert history[-1]['val_loss'] < before['val_loss'] and 0.0 <= history[-1]['edge accu'] <= 1.0
    # the best checkpoint holds the parameters of its epoch
    fresh = _model(tok, seed=9)
    opt = fresh.configure_optimizers()
    epoch = load_checkpoint(str(tmp_path / 'best.ckpt'), fresh, opt)
    assert epoch == min(range(3), key=lambda e: history[e]['val_loss'])
    # resume: save now, take two more steps, restore, take the same two steps -> identical parameters
    opt_a = model.configure_optimizers()
    batches = [to_device(b, 'cuda') for b in list(train)[:2]]
    eps = [torch.randn(16, 56, device='cuda', generator=torch.Generator(device='cuda').manual_seed(i)) for i in range(2)]

    def two_steps(m, o):
        for b, e in zip(batches, eps):
            o.zero_grad()
            m.training_step(b, eps=e).backward()
            o.step()
            m.global_step += 1
        return m.flat_buffers()[0].clone()

    two_steps(model, opt_a)                      # builds the optimiser state
    save_checkpoint(str(tmp_path / 'resume.ckpt'), model, opt_a, epoch=3)
    want = two_steps(model, opt_a)
    other = _model(tok, seed=11)
    opt_b = other.configure_optimizers()
    load_checkpoint(str(tmp_path / 'resume.ckpt'), other, opt_b)
    got = two_steps(other, opt_b)
    assert torch.equal(got, want)
