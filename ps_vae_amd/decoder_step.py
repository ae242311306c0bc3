"""`VAEPieceDecoder.forward` called on its own (`src/modules/vae_piece_decoder.py:71-126`):
the fused step in decoder-only mode.  The caller supplies the graph conditions; the
step runs the decoder-side GINE stack over the edges it is given, the latent head,
the GRU piece branch and the link scorer, and returns `z` and the four losses.
Gradients flow to `conds` and (through the returned `z`) back from whatever the
caller does with `z`; the decoder's own parameters get their `.grad` accumulated
when backward runs, like any torch module."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .data import pack_native

_DEC_PREFIX = 'decoder.'


def _decoder_fields():
    """psvae_params fields that belong to the decoder -> parameter name inside VAEPieceDecoder."""
    return {f: k[len(_DEC_PREFIX):] for f, k in _lib.PARAM_KEYS.items() if k.startswith(_DEC_PREFIX)}


def make_dims(dec):
    d = _lib.Dims()
    tok = dec.tokenizer
    ge = dec.graph_embedding
    d.num_piece, d.num_atom, d.num_bond = tok.num_piece_type(), tok.num_atom_type(), ge.num_edge_type
    d.max_pos = dec.pos_embedding.num_embeddings
    d.d_atom, d.d_piece, d.d_pos = (dec.atom_embedding.embedding_dim, dec.piece_embedding.embedding_dim,
                                    dec.pos_embedding.embedding_dim)
    d.hidden, d.graph = ge.node_trans.out_features, dec.W_mean.in_features
    d.latent, d.rnn = dec.latent_dim, dec.rnn.hidden_size
    d.pred_hidden, d.n_props = 4, 1          # the property head does not run in this mode
    d.t_enc = d.t_dec = ge.t
    d.pad_idx = tok.pad_idx()
    d.eps_enc = d.eps_dec = float(ge.conv.eps)
    return d


def pack_decoder_batch(dec, x, x_pieces, x_pos, edge_index, edge_attr, pieces, edge_select, golden_edge):
    """Packed index blob from the tensors `forward` receives.  Padding atoms are those whose id is the
    tokenizer's atom pad index; every edge passed in is an edge of the decoder-side graph."""
    tok = dec.tokenizer
    xn = x.detach().cpu().numpy()
    B, N = xn.shape
    lens = (xn != tok.atom_pad_idx()).sum(axis=1).astype(np.int32)
    ei = edge_index.detach().cpu().numpy()
    ea = edge_attr.detach().cpu().numpy()
    ne = ei.shape[1] if ei.size else 0
    e_type = (ea.argmax(axis=1) if ea.ndim == 2 else ea).astype(np.int32) if ne else np.zeros(0, np.int32)
    pg, pm, pn = np.nonzero(edge_select.detach().cpu().numpy())
    pc = pieces.detach().cpu().numpy()
    props = np.zeros((B, 1), dtype=np.float32)
    nb = pack_native(lens, N, pc.shape[1], xn, x_pieces.detach().cpu().numpy(), x_pos.detach().cpu().numpy(),
                     ei[0] if ne else np.zeros(0, np.int32), ei[1] if ne else np.zeros(0, np.int32), e_type,
                     np.arange(ne, dtype=np.int32), pg, pm, pn, golden_edge.detach().cpu().numpy(), pc, props, tok,
                     dec.pos_embedding.num_embeddings)
    return nb


def _params_struct(tensors):
    s = _lib.Params()
    for f, t in tensors.items():
        setattr(s, f, t.data_ptr())
    return s


class _DecoderStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, conds, trigger, dec, nb, eps):
        L = _lib.lib()
        named = dict(dec.named_parameters())
        fields = {f: named[k].detach().contiguous() for f, k in _decoder_fields().items()}
        params = _params_struct(fields)
        dims = make_dims(dec)
        conds_c = conds.detach().to(torch.float32).contiguous()
        bs = nb.struct(eps)
        bs.conds = conds_c.data_ptr()
        nbytes = L.psvae_workspace_bytes(C.byref(dims), C.byref(bs))
        if nbytes == 0:
            raise RuntimeError('psvae_workspace_bytes rejected the decoder/batch sizes')
        ws = torch.empty(nbytes, dtype=torch.uint8, device=conds.device)
        losses = torch.empty(6, dtype=torch.float32, device=conds.device)
        stream = torch.cuda.current_stream().cuda_stream
        w3 = (C.c_float * 3)(0.0, 1.0, 1.0)     # total = rec + kl (not used by the module surface)
        _lib.check(L.psvae_step_forward(C.byref(dims), C.byref(params), C.byref(bs), ws.data_ptr(), nbytes,
                                        C.byref(w3), losses.data_ptr(), stream), 'psvae_step_forward (decoder only)')
        off = L.psvae_workspace_offset(C.byref(dims), C.byref(bs), b'z')
        B, Z = nb.sizes['B'], dims.latent
        z = ws.view(torch.float32)[off:off + B * Z].view(B, Z).clone()
        ctx.dec, ctx.nb, ctx.eps, ctx.ws, ctx.nbytes, ctx.w3 = dec, nb, eps, ws, nbytes, w3
        ctx.fields, ctx.dims, ctx.conds = fields, dims, conds_c
        dec._last_step = (ws, nb, dims)
        bs.conds = None
        return losses, z

    @staticmethod
    def backward(ctx, g_losses, g_z):
        L = _lib.lib()
        dec, nb, dims = ctx.dec, ctx.nb, ctx.dims
        gtensors = {f: torch.zeros_like(t) for f, t in ctx.fields.items()}
        params, grads = _params_struct(ctx.fields), _params_struct(gtensors)
        d_conds = torch.empty_like(ctx.conds)
        g = g_losses.contiguous()
        gz = g_z.to(torch.float32).contiguous() if g_z is not None else None
        bs = nb.struct(ctx.eps)
        bs.conds, bs.d_conds, bs.d_z = ctx.conds.data_ptr(), d_conds.data_ptr(), (gz.data_ptr() if gz is not None else None)
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(L.psvae_step_backward(C.byref(dims), C.byref(params), C.byref(bs), ctx.ws.data_ptr(), ctx.nbytes,
                                         C.byref(ctx.w3), g.data_ptr(), C.byref(grads), stream),
                   'psvae_step_backward (decoder only)')
        bs.conds = bs.d_conds = bs.d_z = None
        named = dict(dec.named_parameters())
        for f, k in _decoder_fields().items():
            p = named[k]
            if p.requires_grad:
                p.grad = gtensors[f] if p.grad is None else p.grad + gtensors[f]
        ctx.ws = None
        return d_conds, g.new_zeros(()), None, None, None


def decoder_forward(dec, x, x_pieces, x_pos, edge_index, edge_attr, pieces, conds, edge_select, golden_edge,
                    return_accu=False, eps=None):
    if not conds.is_cuda:
        raise RuntimeError('VAEPieceDecoder runs on the GPU only (no CPU implementation exists)')
    nb = pack_decoder_batch(dec, x, x_pieces, x_pos, edge_index, edge_attr, pieces, edge_select, golden_edge).to(conds.device)
    B = nb.sizes['B']
    if eps is None:
        eps = torch.randn(B, dec.latent_dim, dtype=torch.float32, device=conds.device)
    eps = eps.to(conds.device, torch.float32).contiguous()
    trigger = torch.zeros((), device=conds.device, requires_grad=True)   # keeps the graph alive when conds needs no grad
    losses, z = _DecoderStepFn.apply(conds, trigger, dec, nb, eps)
    out = (losses[1], losses[2], losses[3], losses[4])
    if return_accu:
        return z, (out, accuracies(dec))
    return z, out


def accuracies(dec):
    """`src/modules/vae_piece_decoder.py:120-124` from the activations of the last decoder-only step."""
    L = _lib.lib()
    ws, nb, dims = dec._last_step
    flat = ws.view(torch.float32)

    def view(name, rows, cols, ld):
        off = L.psvae_workspace_offset(C.byref(dims), C.byref(nb.struct()), name.encode())
        return flat[off:off + rows * ld].view(rows, ld)[:, :cols]

    B, Ls, S, V2 = nb.sizes['B'], nb.sizes['L'], nb.sizes['S'], dims.num_piece
    T = Ls - 1
    logits = view('logits', B * T, V2, (V2 + 3) // 4 * 4)
    off = nb.offsets[_lib.BLOB_FIELDS.index('pieces')]
    gold = nb.blob[off:off + B * Ls].view(B, Ls)[:, 1:].reshape(-1).long()
    not_pad = gold != dec.tokenizer.pad_idx()
    piece_accu = ((logits.argmax(-1) == gold) & not_pad).sum().item() / max(int(not_pad.sum().item()), 1)
    pair_logits = view('pair_logits', S, 4, 4)
    goff = nb.offsets[_lib.BLOB_FIELDS.index('pair_gold')]
    edge_accu = (pair_logits.argmax(-1) == nb.blob[goff:goff + S].long()).sum().item() / max(S, 1)
    return piece_accu, edge_accu
