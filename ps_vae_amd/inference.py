"""Batched front half of generation: the two GPU-shaped stages of
`VAEPieceDecoder.inference` (`src/modules/vae_piece_decoder.py:128-241`) for many
latent vectors at once.

    pieces = sample_pieces(model, z, max_atom_num=60)           # :134-158, one GRU decode for all z
    frags  = [fragments_of(p) for p in pieces]                   # RDKit: piece SMILES -> atoms, in-piece bonds
    links  = score_links(model, z, frags)                        # :203-238, one scorer pass for all pairs
    for f, l in zip(frags, links):
        for i in candidate_order(l, add_edge_th=0.5): ...        # :239-255, valence / cycle checks (RDKit)

The reference runs this loop one molecule at a time; what needs RDKit (expanding a
piece into atoms, valence and ring checks, assembling the molecule) stays with
the caller, the rest runs through `libpsvae_hip.so`.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .data import pack_native

# `cnt_atom` counts the characters of a piece SMILES that are keys of the reference's valence table
# (src/utils/chem_utils.py:14,158-163); its two-letter keys (Br, Cl) can never equal one character.
_ATOM_CHARS = frozenset('BCFINOPS')


def cnt_atom(smi):
    return sum(1 for c in smi if c in _ATOM_CHARS)


def piece_atom_counts(tokenizer):
    """Atoms the sampling loop charges for each token (`vae_piece_decoder.py:152`); 0 for `<pad>` and `<s>`."""
    return [cnt_atom(tokenizer.idx_to_piece(i)) for i in range(tokenizer.num_piece_type())]


def sample_pieces(model, z, max_atom_num, temperature=1.0, uniforms=None, piece_atoms=None, check_every=8,
                  generator=None):
    """Piece-id sequences (end token dropped) for every row of `z` [Bz, latent].

    `uniforms` [Bz, max_pos] in [0, 1) drive the draws (one per sequence and step, by inversion of the
    cumulative distribution); by default they come from `torch.rand` (optionally a `generator`).  The
    stream of `torch.multinomial`, which the reference calls, cannot be reproduced; the distribution
    sampled from is the same."""
    L = _lib.lib()
    model._sync_flat()
    dev = model._flat.device
    if dev.type != 'cuda':
        raise RuntimeError('sample_pieces runs on the GPU only (no CPU implementation exists)')
    z = z.to(dev, torch.float32).contiguous()
    if z.dim() == 1:
        z = z.unsqueeze(0)
    Bz, steps = z.shape[0], model._dims.max_pos
    if uniforms is None:
        uniforms = torch.rand(Bz, steps, device=dev, generator=generator)
    uniforms = uniforms.to(dev, torch.float32).contiguous()
    if tuple(uniforms.shape) != (Bz, steps):
        raise ValueError(f'uniforms must be [{Bz}, {steps}]')
    if piece_atoms is None:
        piece_atoms = piece_atom_counts(model.tokenizer)
    atoms = torch.as_tensor(piece_atoms, dtype=torch.int32).to(dev)
    if atoms.numel() != model._dims.num_piece:
        raise ValueError('piece_atoms needs one entry per token (pieces, <pad>, <s>)')
    ids = torch.zeros(Bz, steps, dtype=torch.int32, device=dev)
    lengths = torch.zeros(Bz, dtype=torch.int32, device=dev)
    nbytes = L.psvae_sample_pieces_workspace_bytes(C.byref(model._dims), Bz)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(L.psvae_sample_pieces(C.byref(model._dims), C.byref(model._params_struct), z.data_ptr(), Bz,
                                     atoms.data_ptr(), int(max_atom_num), float(temperature), uniforms.data_ptr(),
                                     ids.data_ptr(), lengths.data_ptr(), int(check_every), ws.data_ptr(), nbytes,
                                     torch.cuda.current_stream().cuda_stream), 'psvae_sample_pieces')
    ids, lengths = ids.cpu().numpy(), lengths.cpu().numpy()
    return [[int(v) for v in ids[b, :lengths[b]]] for b in range(Bz)]


def candidate_pairs(groups, n):
    """Atom pairs (m < n) of different pieces in row-major order: the nonzeros of the reference's
    upper-triangular `edge_select` (`vae_piece_decoder.py:215-222,236`)."""
    gid = np.full(n, -1, dtype=np.int64)
    for g, atoms in enumerate(groups):
        gid[np.asarray(atoms, dtype=np.int64)] = g
    m, k = np.triu_indices(n, 1)
    keep = gid[m] != gid[k]
    return m[keep].astype(np.int32), k[keep].astype(np.int32)


def pack_fragments(fragments, tokenizer, max_pos):
    """Host-side packing of assembled fragments into the kernels' index blob.  Each fragment is a dict
    with `x` (atom ids), `x_pieces`, `x_pos`, `bonds` ([[a, b], ...] in-piece bonds, each once),
    `bond_types` and `groups` (atom indices of every piece).  Returns (NativeBatch, pairs per fragment)."""
    B = len(fragments)
    lens = np.asarray([len(f['x']) for f in fragments], dtype=np.int32)
    if lens.min() < 1:
        raise ValueError('every fragment set needs at least one atom')
    N = int(lens.max())
    x = np.full((B, N), tokenizer.atom_pad_idx(), dtype=np.int32)
    xp = np.full((B, N), tokenizer.pad_idx(), dtype=np.int32)
    xq = np.zeros((B, N), dtype=np.int32)
    e_src, e_dst, e_type, pg, pm, pn, pairs = [], [], [], [], [], [], []
    for i, f in enumerate(fragments):
        k = lens[i]
        x[i, :k], xp[i, :k], xq[i, :k] = f['x'], f['x_pieces'], f['x_pos']
        if len(f['bonds']):
            b = np.asarray(f['bonds'], dtype=np.int64).reshape(-1, 2)
            t = np.asarray(f['bond_types'], dtype=np.int64)
            s = np.concatenate([b[:, 0], b[:, 1]])
            d = np.concatenate([b[:, 1], b[:, 0]])
            order = np.lexsort((d, s))          # by source, then target: the collate's edge order
            e_src.append(s[order] + i * N); e_dst.append(d[order] + i * N); e_type.append(np.concatenate([t, t])[order])
        m, q = candidate_pairs(f['groups'], int(k))
        pairs.append(np.stack([m, q], axis=1))
        pg.append(np.full(len(m), i, dtype=np.int32)); pm.append(m); pn.append(q)
    cat = lambda parts: np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, np.int32)
    e_src, e_dst, e_type, pg, pm, pn = cat(e_src), cat(e_dst), cat(e_type), cat(pg), cat(pm), cat(pn)
    end = tokenizer.end_idx()
    pieces = np.full((B, 2), end, dtype=np.int32)      # the piece branch does not run; a placeholder sequence
    nb = pack_native(lens, N, 2, x, xp, xq, e_src, e_dst, e_type, np.arange(len(e_src), dtype=np.int32), pg, pm, pn,
                     np.zeros(len(pg), np.int32), pieces, np.zeros((B, 1), np.float32), tokenizer, max_pos)
    return nb, pairs


def score_links(model, z, fragments):
    """For every fragment set: dict(pairs [E,2] int, probs [E,4], confidence [E], bond_type [E]) as numpy
    arrays: `pred_edge_index`, `softmax(pred_edge)` and its `max` of `vae_piece_decoder.py:233-238`."""
    L = _lib.lib()
    model._sync_flat()
    dev = model._flat.device
    if dev.type != 'cuda':
        raise RuntimeError('score_links runs on the GPU only (no CPU implementation exists)')
    z = z.to(dev, torch.float32).contiguous()
    if z.dim() == 1:
        z = z.unsqueeze(0)
    if z.shape[0] != len(fragments):
        raise ValueError('one latent vector per fragment set')
    nb, pairs = pack_fragments(fragments, model.tokenizer, model._dims.max_pos)
    nb = nb.to(dev)
    S = nb.sizes['S']
    bs = nb.struct()
    nbytes = L.psvae_workspace_bytes(C.byref(model._dims), C.byref(bs))
    if nbytes == 0:
        raise RuntimeError('unsupported layer sizes for the fused kernels')
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    probs = torch.zeros(max(S, 1), 4, dtype=torch.float32, device=dev)
    conf = torch.zeros(max(S, 1), dtype=torch.float32, device=dev)
    btype = torch.zeros(max(S, 1), dtype=torch.int32, device=dev)
    _lib.check(L.psvae_score_links(C.byref(model._dims), C.byref(model._params_struct), C.byref(bs), z.data_ptr(),
                                   ws.data_ptr(), nbytes, probs.data_ptr(), conf.data_ptr(), btype.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), 'psvae_score_links')
    probs, conf, btype = probs.cpu().numpy(), conf.cpu().numpy(), btype.cpu().numpy()
    out, o = [], 0
    for p in pairs:
        k = len(p)
        out.append({'pairs': p, 'probs': probs[o:o + k], 'confidence': conf[o:o + k], 'bond_type': btype[o:o + k]})
        o += k
    return out


def candidate_order(links, add_edge_th, none_bond=0):
    """Indices of the pairs worth trying, most confident first (`vae_piece_decoder.py:239-241`; Python's
    stable sort keeps the original order among equal confidences)."""
    conf, typ = links['confidence'], links['bond_type']
    possible = [i for i in range(len(conf)) if conf[i] >= add_edge_th and typ[i] != none_bond]
    return sorted(possible, key=lambda i: conf[i], reverse=True)
