"""CPU oracle for the PS-VAE training step.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch on the CPU, the arithmetic of the
reference training step so that the HIP path can be checked against it.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import it; the product package never does.

Parity status: PINNED.  `tools/make_fixtures.py` runs the reference's own
modules (imported from the read-only reference tree, with inert stand-ins for
the absent rdkit / pytorch_lightning / torch_geometric packages) on seeded
inputs and stores their outputs under `tests/golden/`; `tests/test_oracle_golden.py`
checks this restatement against those files.  The one piece of arithmetic that
lives in an absent dependency is `torch_geometric.nn.GINEConv` (un-vendored,
version unpinned by the reference); its published formula
    out_i = nn( (1 + eps) * x_i + sum_{j -> i} relu(x_j + e_ji) )
is restated in `gine_conv` below and in the stand-in the fixture script gives
the reference, so that single formula is "parity unpinned" against PyG itself.

Every function follows the reference lines cited in its docstring.  The batch
is the reference's padded batch dict (`src/data/bpe_dataset.py:142-155`) and
the parameters are a dict keyed by the reference's state-dict names, so the
cost profile (padded rows, dense `[B,N,N,*]` pair tensors) is the reference's.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# sizes and parameter recipe
# --------------------------------------------------------------------------
def default_dims(vocab_size=300, **over):
    """Layer sizes from the training CLI defaults (`src/train.py:92-110`,
    `src/utils/nn_utils.py:180-206`, `src/utils/chem_utils.py:14-15`)."""
    d = dict(num_piece=vocab_size + 2, num_atom=11, num_bond=4, max_pos=50,
             d_atom=50, d_piece=100, d_pos=50, hidden=300, graph=400, latent=56,
             rnn=200, pred_hidden=200, n_props=2, t_enc=4, t_dec=4)
    d.update(over)
    return d


def param_shapes(d):
    """State-dict names and shapes in the reference's registration order
    (`src/pl_models/ps_vae_model.py:25-27` and the module constructors)."""
    H, node = d['hidden'], d['d_atom'] + d['d_piece'] + d['d_pos']
    pin = 2 * H + d['latent']
    out = []

    def lin(name, o, i):
        out.append((name + '.weight', (o, i)))
        out.append((name + '.bias', (o,)))

    def enc(prefix, dim_out, t):
        lin(prefix + 'node_trans', H, node)
        lin(prefix + 'edge_trans', H, d['num_bond'])
        out.append((prefix + 'conv.eps', (1,)))
        lin(prefix + 'conv.nn.seq.0', H, H)
        lin(prefix + 'conv.nn.seq.2', H, H)
        lin(prefix + 'linear', dim_out, H * t)

    lin('predictor.mlp.0', d['pred_hidden'], d['latent'])
    lin('predictor.mlp.2', d['pred_hidden'], d['pred_hidden'])
    lin('predictor.output', d['n_props'], d['pred_hidden'])
    enc('encoder.', d['graph'], d['t_enc'])
    out.append(('decoder.atom_embedding.weight', (d['num_atom'], d['d_atom'])))
    out.append(('decoder.piece_embedding.weight', (d['num_piece'], d['d_piece'])))
    out.append(('decoder.pos_embedding.weight', (d['max_pos'], d['d_pos'])))
    lin('decoder.latent_to_rnn_hidden', d['rnn'], d['latent'])
    out.append(('decoder.rnn.weight_ih_l0', (3 * d['rnn'], d['d_piece'])))
    out.append(('decoder.rnn.weight_hh_l0', (3 * d['rnn'], d['rnn'])))
    out.append(('decoder.rnn.bias_ih_l0', (3 * d['rnn'],)))
    out.append(('decoder.rnn.bias_hh_l0', (3 * d['rnn'],)))
    lin('decoder.to_vocab', d['num_piece'], d['rnn'])
    enc('decoder.graph_embedding.', 1, d['t_dec'])
    lin('decoder.edge_predictor.0.seq.0', pin // 2, pin)
    lin('decoder.edge_predictor.0.seq.2', pin // 2, pin // 2)
    lin('decoder.edge_predictor.0.seq.4', pin, pin // 2)
    lin('decoder.edge_predictor.1', d['num_bond'], pin)
    lin('decoder.W_mean', d['latent'], d['graph'])
    lin('decoder.W_log_var', d['latent'], d['graph'])
    return out


BUFFER_SUFFIX = 'conv.eps'


def make_params(d, seed=2021):
    """Seeded random parameters (a recipe of this build, not the reference's
    initialiser): weights U(-1/sqrt(fan_in), 1/sqrt(fan_in)), embedding tables
    0.5*N(0,1), biases U(-0.1, 0.1), drawn from one CPU generator in state-dict
    order, so every process regenerates the same values."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(d):
        if name.endswith(BUFFER_SUFFIX):
            sd[name] = torch.zeros(shape)
        elif 'embedding.weight' in name:
            sd[name] = 0.5 * torch.randn(shape, generator=g)
        elif len(shape) == 2:
            a = 1.0 / math.sqrt(shape[1])
            sd[name] = torch.empty(shape).uniform_(-a, a, generator=g)
        else:
            sd[name] = torch.empty(shape).uniform_(-0.1, 0.1, generator=g)
    return sd


def trainable_names(d):
    return [n for n, _ in param_shapes(d) if not n.endswith(BUFFER_SUFFIX)]


# --------------------------------------------------------------------------
# model arithmetic
# --------------------------------------------------------------------------
def gine_conv(x, src, dst, e, p, pre):
    """One GINEConv application: message relu(x_src + e), sum at dst, then the
    2-layer MLP on agg + (1+eps)*x  (`src/modules/encoder.py:48,58`;
    `src/modules/common_nn.py:15-23`)."""
    msg = torch.relu(x.index_select(0, src) + e)
    agg = torch.zeros_like(x).index_add_(0, dst, msg)
    h = agg + (1.0 + p[pre + 'conv.eps']) * x
    h = torch.relu(F.linear(h, p[pre + 'conv.nn.seq.0.weight'], p[pre + 'conv.nn.seq.0.bias']))
    return F.linear(h, p[pre + 'conv.nn.seq.2.weight'], p[pre + 'conv.nn.seq.2.bias'])


def embed_node(x, edge_index, edge_attr_onehot, p, pre, t):
    """`Encoder.embed_node` (`src/modules/encoder.py:53-61`)."""
    w = p[pre + 'node_trans.weight']
    h = F.linear(x.to(w.dtype), w, p[pre + 'node_trans.bias'])
    e = F.linear(edge_attr_onehot.to(w.dtype), p[pre + 'edge_trans.weight'], p[pre + 'edge_trans.bias'])
    src, dst = edge_index[0], edge_index[1]
    layers = [h]
    for _ in range(t):
        h = gine_conv(h, src, dst, e, p, pre)
        layers.append(h)
    return layers  # [h0, x1 .. xt]


def embed_graph(all_x, graph_ids, node_mask, p, pre):
    """`Encoder.embed_graph` (`src/modules/encoder.py:63-69`)."""
    n_graph = int(graph_ids[-1]) + 1
    pooled = torch.zeros(n_graph, all_x.shape[-1], dtype=all_x.dtype)
    keep = ~node_mask
    pooled.index_add_(0, graph_ids[keep], all_x[keep])
    return pooled, F.linear(pooled, p[pre + 'linear.weight'], p[pre + 'linear.bias'])


def gru_sequence(x_seq, h0, p):
    """Single-layer batch-first GRU, gate order (r, z, n) as `torch.nn.GRU`
    stores it (`src/modules/vae_piece_decoder.py:29,89`)."""
    w_ih, w_hh = p['decoder.rnn.weight_ih_l0'], p['decoder.rnn.weight_hh_l0']
    b_ih, b_hh = p['decoder.rnn.bias_ih_l0'], p['decoder.rnn.bias_hh_l0']
    R = h0.shape[-1]
    h, outs = h0, []
    for t in range(x_seq.shape[1]):
        gi = F.linear(x_seq[:, t], w_ih, b_ih)
        gh = F.linear(h, w_hh, b_hh)
        r = torch.sigmoid(gi[:, :R] + gh[:, :R])
        u = torch.sigmoid(gi[:, R:2 * R] + gh[:, R:2 * R])
        c = torch.tanh(gi[:, 2 * R:] + r * gh[:, 2 * R:])
        h = (1.0 - u) * c + u * h
        outs.append(h)
    return torch.stack(outs, dim=1)


def forward(p, d, batch, eps, keep=None, conds=None):
    """`PSVAEModel.forward` (`src/pl_models/ps_vae_model.py:33-58`) with the
    decoder's `forward` (`src/modules/vae_piece_decoder.py:71-117`) inlined.
    `eps` replaces the `randn_like` draw of `rsample` (:61).  Returns a dict of
    the five losses; intermediate tensors are added to `keep` when given.
    With `conds` the encoder is skipped and the decoder runs on the given graph
    conditions, as when `VAEPieceDecoder.forward` is called on its own (:71)."""
    A = keep if keep is not None else {}
    x_ids, xp, xq = batch['x'], batch['x_pieces'], batch['x_pos']
    B, N = x_ids.shape
    # embed_atom (vae_piece_decoder.py:65-69)
    x0 = torch.cat([p['decoder.atom_embedding.weight'][x_ids],
                    p['decoder.piece_embedding.weight'][xp],
                    p['decoder.pos_embedding.weight'][xq]], dim=-1)
    A['x0'] = x0
    node_dim = x0.shape[-1]
    graph_ids = torch.repeat_interleave(torch.arange(B), N)
    ei, ea = batch['edge_index'], batch['edge_attr']
    if conds is None:
        enc_layers = embed_node(x0.view(-1, node_dim), ei, ea, p, 'encoder.', d['t_enc'])
        A['enc_layers'] = enc_layers
        all_x = torch.cat(enc_layers[1:], dim=-1)
        pooled, cond = embed_graph(all_x, graph_ids, batch['atom_mask'].flatten(), p, 'encoder.')
        A['pooled'] = pooled
    else:
        cond = conds
    A['cond'] = cond
    # rsample (vae_piece_decoder.py:56-63)
    z_mean = F.linear(cond, p['decoder.W_mean.weight'], p['decoder.W_mean.bias'])
    z_log_var = -torch.abs(F.linear(cond, p['decoder.W_log_var.weight'], p['decoder.W_log_var.bias']))
    kl = -0.5 * torch.sum(1.0 + z_log_var - z_mean * z_mean - torch.exp(z_log_var)) / B
    z = z_mean + torch.exp(z_log_var / 2) * eps
    A['z_mean'], A['z_log_var'], A['z'] = z_mean, z_log_var, z
    # piece branch (vae_piece_decoder.py:81-90)
    pieces = batch['pieces']
    gold_piece = pieces[:, 1:].flatten()
    emb = p['decoder.piece_embedding.weight'][pieces]
    h0 = F.linear(z, p['decoder.latent_to_rnn_hidden.weight'], p['decoder.latent_to_rnn_hidden.bias'])
    rnn_out = gru_sequence(emb[:, :-1], h0, p)
    logits = F.linear(rnn_out, p['decoder.to_vocab.weight'], p['decoder.to_vocab.bias'])
    A['rnn_out'], A['logits'] = rnn_out, logits
    # decoder graph branch over in-piece bonds (ps_vae_model.py:45-48; decoder :93-96)
    idx = batch['in_piece_edge_idx']
    idx_t = torch.tensor(idx, dtype=torch.long)
    dec_layers = embed_node(x0.view(-1, node_dim), ei[:, idx_t], ea[idx_t], p,
                            'decoder.graph_embedding.', d['t_dec'])
    A['dec_layers'] = dec_layers
    node_emb = dec_layers[-1].view(B, N, -1)
    # pair inputs exactly as the reference builds them (:97-109)
    sel = batch['edge_select']
    src = torch.repeat_interleave(node_emb, N, dim=1).view(B, N, N, -1)[sel]
    dst = torch.repeat_interleave(node_emb, N, dim=0).view(B, N, N, -1)[sel]
    lat = torch.repeat_interleave(z, N * N, dim=0).view(B, N, N, -1)[sel]
    pair_in = torch.cat([src, dst, lat], dim=-1)
    A['pair_in'] = pair_in
    h = pair_in
    for k in (0, 2):
        h = torch.relu(F.linear(h, p[f'decoder.edge_predictor.0.seq.{k}.weight'],
                                p[f'decoder.edge_predictor.0.seq.{k}.bias']))
    h = F.linear(h, p['decoder.edge_predictor.0.seq.4.weight'], p['decoder.edge_predictor.0.seq.4.bias'])
    pair_logits = F.linear(h, p['decoder.edge_predictor.1.weight'], p['decoder.edge_predictor.1.bias'])
    A['pair_logits'] = pair_logits
    # losses (:53-54,113-117)
    piece_loss = F.cross_entropy(logits.view(-1, logits.shape[-1]), gold_piece,
                                 ignore_index=d['num_piece'] - 2)
    edge_loss = F.cross_entropy(pair_logits, batch['golden_edge'])
    # predictor + MSE (predictor.py:19-21; ps_vae_model.py:54-57)
    q = torch.relu(F.linear(z, p['predictor.mlp.0.weight'], p['predictor.mlp.0.bias']))
    q = torch.relu(F.linear(q, p['predictor.mlp.2.weight'], p['predictor.mlp.2.bias']))
    pred = F.linear(q, p['predictor.output.weight'], p['predictor.output.bias'])
    A['pred'] = pred
    golden = batch['props'].reshape(B, -1)[:, batch.get('selected_properties', [0, 2])].to(pred.dtype)
    pred_loss = F.mse_loss(pred, golden)
    return {'pred': pred_loss, 'rec': piece_loss + edge_loss, 'piece': piece_loss,
            'edge': edge_loss, 'kl': kl}


def accuracies(p, d, batch, eps):
    """The `return_accu` branch (`src/modules/vae_piece_decoder.py:120-124`)."""
    keep = {}
    with torch.no_grad():
        forward(p, d, batch, eps, keep)
    logits = keep['logits'].view(-1, keep['logits'].shape[-1])
    gold = batch['pieces'][:, 1:].flatten()
    not_pad = gold != d['num_piece'] - 2
    piece_accu = ((logits.argmax(-1) == gold) & not_pad).sum().item() / not_pad.sum().item()
    edge_accu = (keep['pair_logits'].argmax(-1) == batch['golden_edge']).sum().item() / len(batch['golden_edge'])
    return piece_accu, edge_accu


def get_z_mean(p, d, batch):
    """Encoder-only path of `PSVAEModel.get_z` (`ps_vae_model.py:160-170`),
    returning the deterministic mean and log-variance instead of a sample."""
    keep = {}
    x_ids = batch['x']
    B, N = x_ids.shape
    x0 = torch.cat([p['decoder.atom_embedding.weight'][x_ids],
                    p['decoder.piece_embedding.weight'][batch['x_pieces']],
                    p['decoder.pos_embedding.weight'][batch['x_pos']]], dim=-1)
    layers = embed_node(x0.view(B * N, -1), batch['edge_index'], batch['edge_attr'], p, 'encoder.', d['t_enc'])
    graph_ids = torch.repeat_interleave(torch.arange(B), N)
    _, cond = embed_graph(torch.cat(layers[1:], -1), graph_ids, batch['atom_mask'].flatten(), p, 'encoder.')
    z_mean = F.linear(cond, p['decoder.W_mean.weight'], p['decoder.W_mean.bias'])
    z_log_var = -torch.abs(F.linear(cond, p['decoder.W_log_var.weight'], p['decoder.W_log_var.bias']))
    return z_mean, z_log_var


def beta_at(step, cfg):
    """`PSVAEModel.cal_beta` (`ps_vae_model.py:60-69`)."""
    if step < cfg['kl_warmup']:
        beta = 0
    else:
        beta = min(cfg['max_beta'], ((step - cfg['kl_warmup']) // cfg['kl_anneal_iter'] + 1) * cfg['step_beta'])
    return beta + cfg['beta']


TRAIN_CFG = dict(lr=1e-3, alpha=0.1, beta=0.0, max_beta=0.01, step_beta=0.002,
                 kl_warmup=0, kl_anneal_iter=1000, grad_clip=10.0)


def weighted_loss(losses, step, cfg=TRAIN_CFG):
    """`PSVAEModel.weighted_loss` (`ps_vae_model.py:71-74`)."""
    return cfg['alpha'] * losses['rec'] + (1 - cfg['alpha']) * losses['pred'] + beta_at(step, cfg) * losses['kl']


class Trainer:
    """Plain-loop equivalent of what Lightning drives for the reference:
    forward, weighted loss, backward, `clip_grad_norm_(10)` (`src/train.py:47`)
    and Adam(lr) (`ps_vae_model.py:123-127`).  Parameters that receive no
    gradient (the decoder-side encoder's unused `linear`) are skipped by both,
    as torch does for `grad is None`."""

    def __init__(self, sd, d, cfg=TRAIN_CFG):
        self.d, self.cfg = d, dict(cfg)
        self.p = {k: v.clone() for k, v in sd.items()}
        self.names = trainable_names(d)
        for n in self.names:
            self.p[n].requires_grad_(True)
        self.opt = torch.optim.Adam([self.p[n] for n in self.names], lr=cfg['lr'])
        self.step_idx = 0

    def loss_and_grads(self, batch, eps):
        for n in self.names:
            self.p[n].grad = None
        losses = forward(self.p, self.d, batch, eps)
        total = weighted_loss(losses, self.step_idx, self.cfg)
        total.backward()
        return total, losses

    def step(self, batch, eps, epoch=0):
        total, losses = self.loss_and_grads(batch, eps)
        params = [self.p[n] for n in self.names if self.p[n].grad is not None]
        torch.nn.utils.clip_grad_norm_(params, self.cfg['grad_clip'])
        for g in self.opt.param_groups:
            g['lr'] = self.cfg['lr'] / (epoch + 1)
        self.opt.step()
        self.step_idx += 1
        return float(total.detach()), {k: float(v.detach()) for k, v in losses.items()}


def direct_optimize(p, d, z, target, lr, max_iter, patience, selected=(0, 2), keep=None):
    """`direct_optimize` (`src/generate.py:67-89`) with `get_loss` (`:30-44`) and the padding of
    `predict_props` (`ps_vae_model.py:146-153`) for ONE latent vector `z` [latent].  `keep` (a list) receives
    the loss of every step.  Parity status: PINNED by `tests/golden/latent_opt.pt` (the reference's own
    `generate.direct_optimize`: per-step losses and the returned z for nine runs)."""
    def predict_props(zz):
        q = torch.relu(F.linear(zz, p['predictor.mlp.0.weight'], p['predictor.mlp.0.bias']))
        q = torch.relu(F.linear(q, p['predictor.mlp.2.weight'], p['predictor.mlp.2.bias']))
        props = F.linear(q, p['predictor.output.weight'], p['predictor.output.bias'])
        padded = [torch.tensor(0.0) for _ in range(3)]
        for i, val in zip(selected, props):
            padded[i] = val
        return torch.stack(padded)

    def get_loss(props):
        tar = torch.tensor([v for v in target if v is not None], dtype=torch.float)
        idx = torch.tensor([i for i, v in enumerate(target) if v is not None], dtype=torch.long)
        return F.mse_loss(torch.index_select(props, 0, idx), tar)

    optimized = z.clone()
    optimized.requires_grad = True
    opt = torch.optim.Adam([optimized], lr=lr)
    best_loss, not_inc, best_z = 100, 0, optimized.clone()
    steps = 0
    for _ in range(max_iter):
        loss = get_loss(predict_props(optimized))
        opt.zero_grad()
        loss.backward()
        opt.step()
        steps += 1
        loss = loss.item()
        if keep is not None:
            keep.append(loss)
        if loss < best_loss:
            best_loss, not_inc, best_z = loss, 0, optimized.clone()
        else:
            not_inc += 1
            if not_inc > patience:
                break
    return best_z.detach(), best_loss, steps


def sample_pieces(p, d, z, piece_atoms, max_atom_num, temperature, uniforms, keep=None):
    """The piece-sampling loop of `VAEPieceDecoder.inference` (`src/modules/vae_piece_decoder.py:134-158`)
    for ONE latent `z` [latent].  `torch.multinomial` (:149) is replaced by inversion of the cumulative
    distribution at `uniforms[step]` (its own random stream is not reproducible outside torch);
    everything else follows the loop line by line.  `keep` (a list) receives every step's probability
    vector.  Parity status: PINNED by `tests/golden/inference.pt`, which holds what the reference's own
    `inference` computed (probabilities handed to `torch.multinomial` at every step, the pieces drawn from
    them by the same inversion rule) with RDKit's molecule bookkeeping replaced by recording stand-ins
    (`tools/make_fixtures.py`)."""
    pad_idx, end_idx = d['num_piece'] - 2, d['num_piece'] - 1
    emb = p['decoder.piece_embedding.weight']
    cur = emb[end_idx].view(1, 1, -1)
    hidden = F.linear(z.view(1, -1), p['decoder.latent_to_rnn_hidden.weight'], p['decoder.latent_to_rnn_hidden.bias'])
    piece_ids, cur_id, atom_num = [], None, 0
    while cur_id != end_idx:
        hidden = gru_sequence(cur, hidden, p)[:, 0]
        output = F.linear(hidden, p['decoder.to_vocab.weight'], p['decoder.to_vocab.bias']).squeeze().clone()
        output[pad_idx] = float('-inf')
        if len(piece_ids) == 0:
            output[end_idx] = float('-inf')
        probs = torch.softmax(output / temperature, dim=-1)
        if keep is not None:
            keep.append(probs.detach().clone())
        cdf = torch.cumsum(probs.double(), dim=0)
        target = float(uniforms[len(piece_ids)]) * float(cdf[-1])
        cur_id = int(torch.searchsorted(cdf, torch.tensor(target, dtype=torch.double), right=True))
        cur_id = min(cur_id, d['num_piece'] - 1)
        cur = emb[cur_id].view(1, 1, -1)
        atom_num += piece_atoms[cur_id]
        piece_ids.append(cur_id)
        if atom_num > max_atom_num:
            break
        if len(piece_ids) == d['max_pos']:
            break
    return piece_ids[:-1]


def score_links(p, d, z, x, x_pieces, x_pos, edge_index, edge_attr, groups):
    """The link-scoring part of `VAEPieceDecoder.inference` (`src/modules/vae_piece_decoder.py:203-238`) for
    ONE set of fragments: `edge_index` [[a, b], ...] holds the in-piece bonds in both directions in the
    order the loop appends them (:189-196), `edge_attr` their bond types.  Returns
    (pred_edge_index [E,2], softmax(pred_edge) [E,4]).  Parity status: PINNED by `tests/golden/inference.pt`
    (the reference's own scorer output for the fragment sets its sampling loop produced)."""
    z = z.view(1, -1)
    node_x = torch.cat([p['decoder.atom_embedding.weight'][torch.tensor(x, dtype=torch.long)],
                        p['decoder.piece_embedding.weight'][torch.tensor(x_pieces, dtype=torch.long)],
                        p['decoder.pos_embedding.weight'][torch.tensor(x_pos, dtype=torch.long)]], dim=-1)
    if len(edge_index) == 0:
        ei = torch.zeros(2, 0, dtype=torch.long)
    else:
        ei = torch.tensor(edge_index, dtype=torch.long).t().contiguous()
    ea = F.one_hot(torch.tensor(edge_attr, dtype=torch.long), num_classes=d['num_bond'])
    node_embedding = embed_node(node_x, ei, ea, p, 'decoder.graph_embedding.', d['t_dec'])[-1]
    n = len(x)
    edge_select = torch.triu(torch.ones(n, n, dtype=torch.long))
    for group in groups:
        for i in range(len(group)):
            for j in range(i, len(group)):
                m, k = group[i], group[j]
                edge_select[m][k] = edge_select[k][m] = 0
    edge_select = edge_select.unsqueeze(0).bool()
    node_embedding = node_embedding.unsqueeze(0)
    src = torch.repeat_interleave(node_embedding, n, dim=1).view(1, n, n, -1)[edge_select]
    dst = torch.repeat_interleave(node_embedding, n, dim=0).view(1, n, n, -1)[edge_select]
    lat = torch.repeat_interleave(z, n ** 2, dim=0).view(1, n, n, -1)[edge_select]
    h = torch.cat([src, dst, lat], dim=-1)
    pred_edge_index = torch.nonzero(edge_select.squeeze(0))
    if h.shape[0] == 0:
        return pred_edge_index, torch.zeros(0, d['num_bond'])
    for k in (0, 2):
        h = torch.relu(F.linear(h, p[f'decoder.edge_predictor.0.seq.{k}.weight'], p[f'decoder.edge_predictor.0.seq.{k}.bias']))
    h = F.linear(h, p['decoder.edge_predictor.0.seq.4.weight'], p['decoder.edge_predictor.0.seq.4.bias'])
    pred_edge = F.linear(h, p['decoder.edge_predictor.1.weight'], p['decoder.edge_predictor.1.bias'])
    return pred_edge_index, torch.softmax(pred_edge, dim=-1)


def philox4x32_10(c, k0, k1):
    """Philox-4x32-10 (Salmon et al., SC'11; constants of the Random123 library): `c` four uint64 arrays holding
    32-bit counter words, `k0`, `k1` the key words.  Returns the four output words."""
    import numpy as np
    m32 = np.uint64(0xFFFFFFFF)
    c = [np.asarray(w, dtype=np.uint64) & m32 for w in c]
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c[0], np.uint64(0xCD9E8D57) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & m32, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & m32]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & m32, (k1 + np.uint64(0xBB67AE85)) & m32
    return c


def normal_rows(seed, draw, first_row, rows, width):
    """Restatement of the library's counter-based epsilon draw (`psvae_normal_rows`): element (r, c) is a
    function of (seed, draw, first_row + r, c) only.  Philox-4x32-10 with counter (c // 4, row low word, row
    high word, draw low word) and key (seed low word, seed high word ^ draw high word); outputs (x0, x1),
    (x2, x3) become columns 4g..4g+3 by Box-Muller over u = ((x >> 8) + 0.5) * 2^-24.  It stands where the
    reference calls `torch.randn_like` (`src/modules/vae_piece_decoder.py:61`); the reference holds no vector
    for it (its draw comes from torch's global generator), the generator itself is pinned by Random123's
    known-answer vectors in the tests."""
    import numpy as np
    groups = (width + 3) // 4
    r, g = np.meshgrid(np.arange(rows, dtype=np.uint64), np.arange(groups, dtype=np.uint64), indexing='ij')
    row = r + np.uint64(first_row)
    c = philox4x32_10([g, row & np.uint64(0xFFFFFFFF), row >> np.uint64(32), np.full_like(g, draw & 0xFFFFFFFF)],
                      seed & 0xFFFFFFFF, ((seed >> 32) ^ (draw >> 32)) & 0xFFFFFFFF)
    out = np.zeros((rows, groups * 4), dtype=np.float64)
    for h in range(2):
        u0 = ((c[2 * h] >> np.uint64(8)).astype(np.float64) + 0.5) * 2.0 ** -24
        u1 = ((c[2 * h + 1] >> np.uint64(8)).astype(np.float64) + 0.5) * 2.0 ** -24
        rad = np.sqrt(-2.0 * np.log(u0))
        out[:, 2 * h::4] = rad * np.cos(2.0 * np.pi * u1)
        out[:, 2 * h + 1::4] = rad * np.sin(2.0 * np.pi * u1)
    return torch.from_numpy(out[:, :width].astype(np.float32))
