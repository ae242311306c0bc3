"""The fused HIP training step (PSVAEModel.forward / backward / FusedClipAdam)
against the CPU oracle on the same seeded inputs, and against the golden
fixtures that hold the reference's own outputs.  Floating point: rtol 1e-4."""
import random

import pytest
import torch

from oracle import collate_oracle as CO
from oracle import psvae_oracle as O
from tests.util import check_close, check_mostly_close, config_from_dims, load_golden

pytestmark = pytest.mark.gpu
MODEL_CASES = ['model_small_edge.pt', 'model_small_t3.pt', 'model_zinc_v300.pt', 'model_qm9_v500.pt']


def build(d, vocab_size, seed=2021):
    from ps_vae_amd import PSVAEModel, Tokenizer
    tok = Tokenizer.synthetic(vocab_size)
    model = PSVAEModel(config_from_dims(d, tok), tok)
    sd = O.make_params(d, seed)
    model.load_state_dict(sd)
    return model.cuda(), tok, sd


def collate_both(mols, tok, vocab_size, seed):
    from ps_vae_amd.data import BPEMolDataset, to_device
    random.seed(seed)
    ref_batch = CO.collate([CO.to_matrix_form(m) for m in mols], vocab_size)
    random.seed(seed)
    batch = BPEMolDataset.process_step3([BPEMolDataset.process_step2(m, tok) for m in mols], tok)
    return ref_batch, to_device(batch, 'cuda:0')


def unpad(t, mask):
    """[B*N, F] padded rows -> packed rows of real atoms."""
    return t[~mask.flatten()]


def fp64_grads(d, sd, ref_batch, eps):
    """Gradients of the same step evaluated in float64 (the yardstick for fp32 noise)."""
    tr = O.Trainer({k: v.double() for k, v in sd.items()}, d)
    tr.loss_and_grads(ref_batch, eps.double())
    return {n: tr.p[n].grad for n in tr.names}


# Gradient tensors (of 47) that may need the float64 yardstick instead of the flat band, per case.  Printed by
# every run (`pytest -s`); measured: 0 for the reference fixtures, 0 / 1 / 2 for the three benchmark shapes.
MAX_FP64_FALLBACKS = 3


def compare_forward_backward(model, d, sd, ref_batch, batch, eps, with_fp64=False, fallbacks=None):
    """`fallbacks` (list) receives the names of the gradients that were only accepted against float64."""
    tr = O.Trainer(sd, d)
    fallbacks = [] if fallbacks is None else fallbacks
    g64 = fp64_grads(d, sd, ref_batch, eps) if with_fp64 else None
    keep = {}
    losses_ref = O.forward(tr.p, d, ref_batch, eps, keep)
    total_ref = O.weighted_loss(losses_ref, 0)
    total_ref.backward()
    model.zero_grad()
    pred, (rec, piece, edge, kl) = model(batch, eps=eps)
    total = model.weighted_loss(pred, rec, kl)
    got = dict(pred=pred, rec=rec, piece=piece, edge=edge, kl=kl)
    for k, v in losses_ref.items():
        assert abs(float(got[k]) - float(v)) <= 1e-4 * max(1.0, abs(float(v))), (k, float(got[k]), float(v))
    nb = batch['_native']
    n, B, S, L = (nb.sizes[k] for k in ('n', 'B', 'S', 'L'))
    H, te, td, T = d['hidden'], d['t_enc'], d['t_dec'], L - 1
    mask = ref_batch['atom_mask']
    enc = model.activation('enc_stack', (te + 1) * n, H)      # slice-major: [t+1][n][H]
    dec = model.activation('dec_stack', (td + 1) * n, H)
    for l in range(te + 1):
        check_close(enc[l * n:(l + 1) * n], unpad(keep['enc_layers'][l], mask), f'encoder layer stack slice {l}')
    for l in range(td + 1):
        check_close(dec[l * n:(l + 1) * n], unpad(keep['dec_layers'][l], mask), f'decoder layer stack slice {l}')
    check_close(model.activation('x0', n, d['d_atom'] + d['d_piece'] + d['d_pos']), unpad(keep['x0'].flatten(0, 1), mask), 'x0')
    check_close(model.activation('pooled', B, te * H), keep['pooled'], 'pooled')
    check_close(model.activation('cond', B, d['graph']), keep['cond'], 'cond')
    check_close(model.activation('mulv', B, 2 * d['latent'])[:, :d['latent']], keep['z_mean'], 'z_mean')
    check_close(model.activation('z', B, d['latent']), keep['z'], 'z')
    check_close(model.activation('rnn_out', B * T, d['rnn']), keep['rnn_out'].flatten(0, 1), 'rnn_out')
    V2 = d['num_piece']
    check_close(model.activation('logits', B * T, V2, (V2 + 3) // 4 * 4), keep['logits'].flatten(0, 1), 'logits')
    P = 2 * H + d['latent']
    check_close(model.activation('pair_in', S, P), keep['pair_in'], 'pair_in')
    check_close(model.activation('pair_logits', S, 4), keep['pair_logits'], 'pair_logits')
    check_close(model.activation('pred', B, d['n_props'], (d['n_props'] + 3) // 4 * 4), keep['pred'], 'pred')
    total.backward()
    named = dict(model.named_parameters())
    for name in tr.names:
        g_ref = tr.p[name].grad
        if g_ref is None:
            assert named[name].grad is None, name
        else:
            # Weight gradients are sums over thousands of rows: fp32 summation-order noise scales with
            # the tensor's magnitude, so the absolute floor is 5e-5 of the tensor's max (rtol stays 1e-4).
            # Two fp32 implementations also disagree on the ReLU gate of the few hidden units whose
            # pre-activation is within rounding of zero; each such flip moves a handful of gradient
            # entries by a discrete amount, so up to 0.1 % of the entries may sit outside the band,
            # bounded by 2e-3 of the tensor's max.
            try:
                check_close(named[name].grad, g_ref, f'grad {name}', atol_scale=5e-5, bad_frac=1e-3)
            except AssertionError:
                # Deep in the network (embedding tables, node_trans) one flipped gate perturbs a whole
                # row of dL/dx0 and through it every element of the weight gradient by about 1/rows of
                # the tensor's scale.  There the yardstick is the float64 evaluation of the same step:
                # the HIP gradient must be as close to it as the fp32 CPU oracle is (within 4x in max
                # norm), or within 2e-3 of the tensor's max (the same bound the outliers above get).
                if g64 is None:
                    raise
                exact = g64[name]
                e_hip = float((named[name].grad.detach().double().cpu() - exact).abs().max())
                e_ref = float((g_ref.double() - exact).abs().max())
                assert e_hip <= max(4 * e_ref, 2e-3 * float(exact.abs().max())), (name, e_hip, e_ref)
                fallbacks.append((name, e_hip, e_ref))
    print(f'gradients accepted against the float64 yardstick: {len(fallbacks)} of {len(tr.names)}: '
          + ', '.join(f'{n} (hip {a:.2e}, fp32 oracle {b:.2e})' for n, a, b in fallbacks))
    assert len(fallbacks) <= MAX_FP64_FALLBACKS, fallbacks
    return tr


@pytest.mark.parametrize('case', MODEL_CASES)
def test_step_matches_oracle_and_reference_fixture(case):
    g = load_golden(case)
    d, V = g['dims'], g['vocab_size']
    model, tok, sd = build(d, V, g['param_seed'])
    ref_batch, batch = collate_both(g['mols'], tok, V, g['collate_seed'])
    compare_forward_backward(model, d, sd, ref_batch, batch, g['eps'][0])
    # three optimiser steps against the reference's recorded trajectory
    model, tok, sd = build(d, V, g['param_seed'])
    opt = model.configure_optimizers()
    for s, ref in enumerate(g['steps']):
        model.global_step = s
        opt.zero_grad()
        pred, (rec, piece, edge, kl) = model(batch, eps=g['eps'][s])
        total = model.weighted_loss(pred, rec, kl)
        total.backward()
        opt.step()
        got = dict(pred=pred, rec=rec, piece=piece, edge=edge, kl=kl, total=total)
        for k, v in ref['losses'].items():
            assert abs(float(got[k]) - v) <= 2e-4 * max(1.0, abs(v)), (s, k, float(got[k]), v)
        assert abs(float(opt.grad_norm) - ref['grad_norm']) <= 1e-4 * ref['grad_norm']
        named = dict(model.named_parameters())
        for n_, want in ref['params_after'].items():
            check_mostly_close(named[n_], want, f'step {s} param {n_}')
    pred, (losses, accus) = model(batch, return_accu=True, eps=g['eps'][0])
    assert list(accus) == pytest.approx(g['accus_after'], abs=1e-6)


@pytest.mark.parametrize('shape,B,V', [('zinc', 128, 300), ('qm9', 32, 300), ('zinc', 48, 500)])
def test_step_matches_oracle_at_benchmark_shapes(shape, B, V):
    from ps_vae_amd import synth
    d = O.default_dims(V)
    model, tok, sd = build(d, V)
    mols = synth.make_molecules(100 + B, shape, V, B)
    ref_batch, batch = collate_both(mols, tok, V, 7)
    eps = torch.randn(B, d['latent'], generator=torch.Generator().manual_seed(5))
    compare_forward_backward(model, d, sd, ref_batch, batch, eps, with_fp64=True)


def test_encode_and_foreign_batch_dict():
    from ps_vae_amd import synth
    d, V, B = O.default_dims(300), 300, 16
    model, tok, sd = build(d, V)
    mols = synth.make_molecules(31, 'zinc', V, B)
    ref_batch, batch = collate_both(mols, tok, V, 3)
    mean_ref, lvar_ref = O.get_z_mean(sd, d, ref_batch)
    mean, lvar = model.encode(batch)
    check_close(mean, mean_ref, 'z_mean')
    check_close(lvar, lvar_ref, 'z_log_var')
    # a batch dict without the packed entry (as foreign code would hand it over) gives the same result
    foreign = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ref_batch.items()}
    mean2, _ = model.encode(foreign)
    assert torch.equal(mean2, mean)
    z = model.get_z(batch, eps=torch.zeros(B, d['latent']))
    assert torch.equal(z, mean)


def test_gradient_accumulation_and_determinism():
    from ps_vae_amd import synth
    d, V, B = O.default_dims(300), 300, 32
    model, tok, sd = build(d, V)
    mols = synth.make_molecules(77, 'zinc', V, B)
    _, batch = collate_both(mols, tok, V, 11)
    eps = torch.randn(B, d['latent'], generator=torch.Generator().manual_seed(1))

    def grads():
        model.zero_grad()
        pred, (rec, _, _, kl) = model(batch, eps=eps)
        model.weighted_loss(pred, rec, kl).backward()
        return model.flat_buffers()[1].clone()
    g1, g2 = grads(), grads()
    assert torch.equal(g1, g2), 'two runs of the same step must be bitwise identical'
    from ps_vae_amd import _lib
    st = torch.cuda.current_stream().cuda_stream
    assert _lib.lib().psvae_streams_init(st) == 0      # idempotent once the side streams exist; results unchanged
    assert _lib.lib().psvae_streams_init(None) == 0
    assert torch.equal(grads(), g1)
    pred, (rec, _, _, kl) = model(batch, eps=eps)      # second backward without zero_grad accumulates
    model.weighted_loss(pred, rec, kl).backward()
    check_close(model.flat_buffers()[1], 2 * g1, 'accumulated gradient', rtol=1e-6)


def test_train_step_entry_equals_the_three_calls():
    """`PSVAEModel.train_step` (one `psvae_train_step` call: forward, backward, clip + Adam) against `training_step`,
    `backward`, `optimizer.step()` through autograd: losses and parameters bit for bit over three steps; without an
    optimiser it leaves the gradients of that step and refuses to run on top of gradients that are still there."""
    from ps_vae_amd import synth
    d, V, B = O.default_dims(300), 300, 24
    mols = [synth.make_molecules(300 + i, 'zinc', V, B) for i in range(3)]
    eps = [torch.randn(B, d['latent'], generator=torch.Generator().manual_seed(20 + i)) for i in range(3)]
    runs = []
    for one_call in (False, True):
        model, tok, sd = build(d, V)
        opt = model.configure_optimizers()
        losses = []
        for i in range(3):
            _, batch = collate_both(mols[i], tok, V, 50 + i)
            if one_call:
                six = model.train_step(batch, opt, eps=eps[i])
                losses.append(six.clone())
            else:
                opt.zero_grad()
                pred, (rec, piece, edge, kl) = model(batch, eps=eps[i])
                total = model._last_total
                total.backward()
                opt.step()
                losses.append(torch.stack([pred, rec, piece, edge, kl, total]).detach().clone())
            model.global_step += 1
        runs.append((torch.stack(losses), model.flat_buffers()[0].clone(), opt.grad_norm.clone()))
    assert torch.equal(runs[0][0], runs[1][0]), 'losses'
    assert torch.equal(runs[0][1], runs[1][1]), 'parameters after three steps'
    assert torch.equal(runs[0][2], runs[1][2]), 'gradient norm of the last step'
    # forward + backward only: the gradients equal those of the autograd path, and a second call must not overwrite them
    model, tok, sd = build(d, V)
    _, batch = collate_both(mols[0], tok, V, 50)
    model.zero_grad()
    model.training_step(batch, eps=eps[0]).backward()
    want = model.flat_buffers()[1].clone()
    model.zero_grad()
    model.train_step(batch, None, eps=eps[0])
    assert torch.equal(model.flat_buffers()[1], want)
    with pytest.raises(RuntimeError):
        model.train_step(batch, None, eps=eps[0])


def test_train_step_draws_epsilon_itself():
    """`train_step` without `eps`: the library draws rsample's epsilon inside the call, on a side stream.  Same numbers
    as `draw_eps` (same seed, same draw counter) handed in by the caller: losses and parameters bit for bit."""
    from ps_vae_amd import synth
    d, V, B = O.default_dims(300), 300, 24
    runs = []
    for own_draw in (False, True):
        model, tok, sd = build(d, V)
        model.eps_seed = 1234
        opt = model.configure_optimizers()
        losses = []
        for i in range(3):
            _, batch = collate_both(synth.make_molecules(300 + i, 'zinc', V, B), tok, V, 50 + i)
            eps = None if own_draw else model.draw_eps(B, 0)
            losses.append(model.train_step(batch, opt, eps=eps).clone())
            model.global_step += 1
        assert model.eps_draws == 3
        runs.append((torch.stack(losses), model.flat_buffers()[0].clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


def test_two_host_threads_step_on_one_device():
    """Two host threads, each with a model, a stream and batches of its own, run optimiser steps on the same GPU at the
    same time.  The step calls share the device's side streams and events; the library serialises the threads while a
    call enqueues, so each model must end exactly where it ends when the two runs happen one after the other.  (A
    smoke test of the threaded use: two Python threads overlap inside the library only now and then, so a pass does not
    prove the absence of a race; the guarantee is the per-device mutex in the step entry points.)"""
    import threading
    from ps_vae_amd import synth
    d, V, B, steps = O.default_dims(300), 300, 24, 40
    tok0 = build(d, V)[1]
    jobs = []
    for t in range(2):
        batches = [collate_both(synth.make_molecules(700 + 10 * t + i, 'zinc', V, B), tok0, V, 60 + i)[1] for i in range(2)]
        eps = [torch.randn(B, d['latent'], generator=torch.Generator().manual_seed(90 + 10 * t + i)) for i in range(steps)]
        jobs.append((batches, eps))

    def run(job, out, slot, stream):
        batches, eps = job
        with torch.cuda.stream(stream):
            model = build(d, V, seed=2021 + slot)[0]
            opt = model.configure_optimizers()
            for i in range(steps):
                model.train_step(batches[i % 2], opt, eps=eps[i])
                model.global_step += 1
            stream.synchronize()
            out[slot] = model.flat_buffers()[0].clone()

    serial, threaded = [None, None], [None, None]
    for t in range(2):
        run(jobs[t], serial, t, torch.cuda.Stream())
    torch.cuda.synchronize()
    threads = [threading.Thread(target=run, args=(jobs[t], threaded, t, torch.cuda.Stream())) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    for t in range(2):
        assert threaded[t] is not None and torch.equal(serial[t], threaded[t]), f'model of thread {t}'


def test_module_level_encoder_matches_oracle():
    """Encoder.embed_node / embed_graph and Predictor through the single-operator API."""
    from ps_vae_amd import synth
    d, V, B = O.default_dims(300), 300, 8
    model, tok, sd = build(d, V)
    mols = synth.make_molecules(41, 'qm9', V, B)
    ref_batch, batch = collate_both(mols, tok, V, 3)
    p = {k: v.clone().requires_grad_(not k.endswith('eps')) for k, v in sd.items()}
    x0 = torch.randn(B * ref_batch['x'].shape[1], 200, generator=torch.Generator().manual_seed(3))
    layers = O.embed_node(x0, ref_batch['edge_index'], ref_batch['edge_attr'], p, 'encoder.', 4)
    all_x_ref = torch.cat(layers[1:], -1)
    N = ref_batch['x'].shape[1]
    gid = torch.repeat_interleave(torch.arange(B), N)
    _, cond_ref = O.embed_graph(all_x_ref, gid, ref_batch['atom_mask'].flatten(), p, 'encoder.')
    cond_ref.sum().backward()
    model.zero_grad()
    for q in model.parameters():
        q.grad = None
    x_last, all_x = model.encoder.embed_node(x0.cuda(), batch['edge_index'], batch['edge_attr'])
    cond = model.encoder.embed_graph(all_x, gid.cuda(), batch['atom_mask'].flatten())
    check_close(all_x, all_x_ref, 'all_x')
    check_close(cond, cond_ref, 'graph embedding')
    cond.sum().backward()
    for name in ('node_trans.weight', 'edge_trans.weight', 'edge_trans.bias', 'conv.nn.seq.0.weight', 'conv.nn.seq.2.bias', 'linear.weight'):
        check_close(dict(model.encoder.named_parameters())[name].grad, p['encoder.' + name].grad, f'grad {name}')
    z = torch.randn(B, 56, generator=torch.Generator().manual_seed(4))
    q = torch.relu(torch.nn.functional.linear(z, sd['predictor.mlp.0.weight'], sd['predictor.mlp.0.bias']))
    q = torch.relu(torch.nn.functional.linear(q, sd['predictor.mlp.2.weight'], sd['predictor.mlp.2.bias']))
    want = torch.nn.functional.linear(q, sd['predictor.output.weight'], sd['predictor.output.bias'])
    check_close(model.predictor(z.cuda()), want, 'predictor')


def test_encoder_forward_on_a_pyg_style_batch():
    """`Encoder.forward(batch, return_x)` (`src/modules/encoder.py:71-80`): the PyG-`Batch` flavour takes packed
    nodes (`x`, `edge_index`, one-hot `edge_attr`), the graph of each node in `batch.batch` and `num_graphs`;
    no padding mask.  Checked against the oracle's embed_node + embed_graph on the same arrays, values and
    gradients, including a graph without bonds and `return_x=True`."""
    from types import SimpleNamespace
    from ps_vae_amd import synth
    d, V = O.default_dims(300), 300
    model, tok, sd = build(d, V)
    mols = synth.make_molecules(63, 'qm9', V, 6)
    mols[2] = dict(mols[2], edge_index=[], edge_attr=[])             # an edgeless graph in the middle
    src, dst, typ, gid, off = [], [], [], [], 0
    for i, m in enumerate(mols):
        for (a, b), ty in zip(m['edge_index'], m['edge_attr']):
            src += [a + off, b + off]; dst += [b + off, a + off]; typ += [ty, ty]
        gid += [i] * len(m['x'])
        off += len(m['x'])
    n = off
    x0 = torch.randn(n, 200, generator=torch.Generator().manual_seed(8))
    edge_index = torch.tensor([src, dst], dtype=torch.long)
    edge_attr = torch.nn.functional.one_hot(torch.tensor(typ), 4)
    gid = torch.tensor(gid)
    p = {k: v.clone().requires_grad_(not k.endswith('eps')) for k, v in sd.items()}
    layers = O.embed_node(x0, edge_index, edge_attr, p, 'encoder.', 4)
    _, res_ref = O.embed_graph(torch.cat(layers[1:], -1), gid, torch.zeros(n, dtype=torch.bool), p, 'encoder.')
    (res_ref.sum() + layers[-1].pow(2).sum()).backward()
    for q in model.parameters():
        q.grad = None
    batch = SimpleNamespace(x=x0.cuda(), edge_index=edge_index.cuda(), edge_attr=edge_attr.cuda(), batch=gid.cuda(),
                            num_graphs=len(mols))
    res, x_last = model.encoder(batch, return_x=True)
    assert tuple(res.shape) == (len(mols), d['graph']) and tuple(x_last.shape) == (n, d['hidden'])
    check_close(res, res_ref, 'Encoder.forward graph embedding')
    check_close(x_last, layers[-1], 'Encoder.forward last-layer x')
    assert torch.equal(model.encoder(batch), res)                    # return_x=False: the embedding alone
    (res.sum() + x_last.pow(2).sum()).backward()
    for name in ('node_trans.weight', 'node_trans.bias', 'edge_trans.weight', 'conv.nn.seq.0.weight', 'conv.nn.seq.2.weight',
                 'linear.weight', 'linear.bias'):
        check_close(dict(model.encoder.named_parameters())[name].grad, p['encoder.' + name].grad, f'grad {name}',
                    atol_scale=5e-5)


def test_training_step_total_equals_weighted_loss():
    """`training_step` forms alpha*rec + (1-alpha)*pred + beta*kl inside the fused step; value and
    gradients must equal the torch-side `weighted_loss` route, with and without the side streams."""
    from ps_vae_amd import synth
    d, V, B = O.default_dims(300), 300, 24
    model, tok, sd = build(d, V)
    mols = synth.make_molecules(55, 'zinc', V, B)
    _, batch = collate_both(mols, tok, V, 4)
    eps = torch.randn(B, d['latent'], generator=torch.Generator().manual_seed(2))
    model.global_step = 2500   # beta = 0.006
    model.zero_grad()
    pred, (rec, _, _, kl) = model(batch, eps=eps)
    total_a = model.weighted_loss(pred, rec, kl)
    total_a.backward()
    g_a = model.flat_buffers()[1].clone()
    model.zero_grad()
    total_b = model.training_step(batch, eps=eps)
    total_b.backward()
    g_b = model.flat_buffers()[1].clone()
    assert abs(float(total_a.detach()) - float(total_b.detach())) <= 1e-6 * abs(float(total_a.detach()))
    check_close(g_b, g_a, 'gradients via fused total', rtol=1e-5, atol_scale=1e-6)
    # `backward(model.unit_grad())` takes the constant upstream-gradient vector: same gradients bit for bit
    model.zero_grad()
    total_c = model.training_step(batch, eps=eps)
    total_c.backward(model.unit_grad())
    assert torch.equal(model.flat_buffers()[1], g_b)
    model.zero_grad()
    total_d = model.training_step(batch, eps=eps)
    total_d.backward(torch.full((), 0.5, device='cuda:0'))       # any other gradient still goes through the copy
    check_close(model.flat_buffers()[1], 0.5 * g_b, 'gradients for an upstream gradient of 0.5', rtol=1e-5, atol_scale=1e-6)
