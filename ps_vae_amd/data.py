"""Batching: the reference's `BPEMolDataset` stages 2 and 3 on top of the native
host collate (`psvae_collate_pairs` / `psvae_pack_batch` in libpsvae_hip.so).

`process_step3` returns the reference's batch dict (same keys, dtypes, ordering:
`src/data/bpe_dataset.py:142-155`) plus one extra key, `'_native'`, holding the
packed index blob the HIP kernels read.  A batch dict that lacks `'_native'`
(e.g. produced by other code) is packed on first use by `ensure_native`.
"""
import ctypes as C
import random

import numpy as np
import torch

from . import _lib

NONE_BOND = 0
NEG_RATE = 0.022  # src/data/bpe_dataset.py:129


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class NativeBatch:
    """Packed index blob + float side inputs of one batch (host or device)."""

    def __init__(self, blob, offsets, sizes, props_sel, denoms, local_denoms=None, graph_base=None):
        self.blob = blob              # int32 tensor
        self.offsets = offsets        # list of int (words)
        self.sizes = sizes            # dict B,n,E,E_in,S,L
        self.props = props_sel        # float32 [B, n_props]
        # float32 [4]: target tokens, S, B, B*n_props.  `denoms` is what the kernels divide the loss sums by: this
        # batch's own counts, or - written there by a data-parallel loop - the counts of the global batch;
        # `local_denoms` always keeps this batch's own counts (a resident batch can be re-reduced every epoch).
        self.denoms = denoms
        self.local_denoms = denoms.clone() if local_denoms is None else local_denoms
        # index of this batch's first graph inside the global batch it is a shard of (keys the epsilon draw);
        # None = not sharded by the loader
        self.graph_base = graph_base
        self._struct = None

    @property
    def device(self):
        return self.blob.device

    def to(self, device, non_blocking=True):
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:      # 'cuda' names the current device
            device = torch.device('cuda', torch.cuda.current_device())
        if self.blob.device == device:
            return self
        nb = NativeBatch(self.blob.to(device, non_blocking=non_blocking), self.offsets, self.sizes,
                         self.props.to(device, non_blocking=non_blocking),
                         self.denoms.to(device, non_blocking=non_blocking),
                         self.local_denoms.to(device, non_blocking=non_blocking), self.graph_base)
        return nb

    def pin_memory(self):
        return NativeBatch(self.blob.pin_memory(), self.offsets, self.sizes, self.props.pin_memory(),
                           self.denoms.pin_memory(), self.local_denoms.pin_memory(), self.graph_base)

    def field(self, name):
        """int32 view of one blob field (for tests)."""
        i = _lib.BLOB_FIELDS.index(name)
        end = self.offsets[i + 1] if i + 1 < len(self.offsets) else self.blob.numel()
        return self.blob[self.offsets[i]:end]

    def struct(self, eps=None, status=None):
        """ctypes psvae_batch pointing into this batch's (device) memory."""
        if self._struct is None:
            s = _lib.Batch()
            for k in ('B', 'n', 'E', 'E_in', 'S', 'L'):
                setattr(s, k, int(self.sizes[k]))
            base = self.blob.data_ptr()
            for name, off in zip(_lib.BLOB_FIELDS, self.offsets):
                setattr(s, name, base + 4 * off)
            s.props = self.props.data_ptr()
            s.denoms = self.denoms.data_ptr()
            self._struct = s
        self._struct.eps = eps.data_ptr() if eps is not None else None
        self._struct.status = status.data_ptr() if status is not None else None
        return self._struct


def pack_native(lens, N, L, x, x_pieces, x_pos, e_src, e_dst, e_type, in_piece, pair_g, pair_m, pair_n, pair_gold,
                pieces, props_sel, tokenizer, max_pos):
    """numpy int32 arrays in the reference's padded numbering -> NativeBatch (host)."""
    lib = _lib.lib()
    B = len(lens)
    nE, nI, nS = len(e_src), len(in_piece), len(pair_g)
    num_atom, num_piece = tokenizer.num_atom_type(), tokenizer.num_piece_type()
    cap = lib.psvae_pack_bound(B, N, nE, nS, L, num_atom, num_piece, max_pos)
    blob = np.zeros(cap, dtype=np.int32)
    offsets = np.zeros(len(_lib.BLOB_FIELDS), dtype=np.int64)
    words, n_nodes, n_tok = C.c_int64(0), C.c_int32(0), C.c_int32(0)
    a32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)
    lens, x, x_pieces, x_pos, pieces = a32(lens), a32(x), a32(x_pieces), a32(x_pos), a32(pieces)
    e_src, e_dst, e_type, in_piece = a32(e_src), a32(e_dst), a32(e_type), a32(in_piece)
    pair_g, pair_m, pair_n, pair_gold = a32(pair_g), a32(pair_m), a32(pair_n), a32(pair_gold)
    rc = lib.psvae_pack_batch(B, N, L, _ptr(lens), _ptr(x), _ptr(x_pieces), _ptr(x_pos), _ptr(e_src), _ptr(e_dst),
                              _ptr(e_type), nE, _ptr(in_piece), nI, _ptr(pair_g), _ptr(pair_m), _ptr(pair_n),
                              _ptr(pair_gold), nS, _ptr(pieces), num_atom, num_piece, max_pos, tokenizer.pad_idx(),
                              _ptr(blob), cap, _ptr(offsets), C.byref(words), C.byref(n_nodes), C.byref(n_tok))
    _lib.check(rc, 'psvae_pack_batch')
    props_sel = torch.as_tensor(np.ascontiguousarray(props_sel, dtype=np.float32))
    n_props = props_sel.shape[1] if props_sel.ndim == 2 else 1
    denoms = torch.tensor([float(n_tok.value), float(nS), float(B), float(B * n_props)], dtype=torch.float32)
    sizes = dict(B=B, n=n_nodes.value, E=nE, E_in=nI, S=nS, L=L)
    return NativeBatch(torch.from_numpy(blob[:words.value].copy()), [int(o) for o in offsets], sizes, props_sel, denoms)


class BPEMolDataset(torch.utils.data.Dataset):
    """Stages 2 and 3 of the reference dataset (`src/data/bpe_dataset.py:56-155`)
    over molecule records that stage 1 (RDKit) produced elsewhere."""

    selected_properties = (0, 2)   # qed, logp (src/train.py:90-91; src/evaluation/utils.py:122-134)
    max_pos = 50                   # src/train.py:101

    def __init__(self, records, tokenizer):
        super().__init__()
        self.data = records
        self.tokenizer = tokenizer

    @staticmethod
    def process_step1(mol, tokenizer):
        raise RuntimeError('stage 1 (RDKit molecule -> raw record) is a CPU preprocessing step outside this package '
                           '(reference: src/data/bpe_dataset.py:31-54)')

    @staticmethod
    def process_step2(data, tokenizer, lists=True):
        """Adjacency-matrix form (`bpe_dataset.py:56-84`).  Besides the reference's
        keys the result carries the two matrices as int8 arrays (`_ad`, `_cross`)
        so that stage 3 need not re-walk nested lists.  `lists=False` (what the
        dataset's `__getitem__` uses) leaves `ad_mat`, `edge_select`, `x_pieces`
        and `x_pos` as numpy arrays instead of converting them to nested Python
        lists, which is most of this stage's cost; they index the same way."""
        n = len(data['x'])
        ad = np.zeros((n, n), dtype=np.int8)
        if len(data['edge_attr']):
            ei = np.asarray(data['edge_index'], dtype=np.int64).reshape(-1, 2)
            et = np.asarray(data['edge_attr'], dtype=np.int8)
            # both directions of every bond in list order: later duplicates overwrite, as in the reference
            rows = np.stack([ei[:, 0], ei[:, 1]], axis=1).ravel()
            cols = np.stack([ei[:, 1], ei[:, 0]], axis=1).ravel()
            ad[rows, cols] = np.repeat(et, 2)
        pieces = data['pieces']
        gid = np.full(n, -1, dtype=np.int64)
        x_pieces = np.zeros(n, dtype=np.int64)
        x_pos = np.zeros(n, dtype=np.int64)
        groups = data['groups']
        glen = np.fromiter((len(g) for g in groups), dtype=np.int64, count=len(groups))
        if glen.sum():
            flat = np.fromiter((a for g in groups for a in g), dtype=np.int64, count=int(glen.sum()))
            pos_of = np.repeat(np.arange(len(groups), dtype=np.int64), glen)
            gid[flat] = pos_of                      # in list order: a later group overwrites, as the reference's loop does
            x_pieces[flat] = np.asarray(pieces, dtype=np.int64)[pos_of]
            x_pos[flat] = pos_of
        cross = (gid[:, None] != gid[None, :]) | (gid[:, None] < 0)     # an atom outside every group keeps its row of ones
        cross = cross.astype(np.int8)
        out = {'x': list(data['x']), 'props': data['props'], 'pieces': pieces, '_ad': ad, '_cross': cross}
        if lists:
            out.update(ad_mat=ad.tolist(), x_pieces=x_pieces.tolist(), x_pos=x_pos.tolist(), edge_select=cross.tolist())
        else:
            out.update(ad_mat=ad, x_pieces=x_pieces, x_pos=x_pos, edge_select=cross)
        return out

    @staticmethod
    def process_step3(data_list, tokenizer, device='cpu', all_pairs=False):
        """Collate (`bpe_dataset.py:87-155`).  The negative-pair sampler consumes
        Python's global `random` stream exactly like the reference: the generator
        state is handed to the native loop and written back afterwards.
        `all_pairs=True` keeps every cross-piece ordered pair (no sampling)."""
        B = len(data_list)
        lens = np.asarray([len(d['x']) for d in data_list], dtype=np.int32)
        N = int(lens.max())
        pad_piece, pad_atom = tokenizer.pad_idx(), tokenizer.atom_pad_idx()
        x = np.full((B, N), pad_atom, dtype=np.int64)
        xp = np.full((B, N), pad_piece, dtype=np.int64)
        xq = np.zeros((B, N), dtype=np.int64)
        ads, crosses = [], []
        for i, d in enumerate(data_list):
            k = lens[i]
            x[i, :k], xp[i, :k], xq[i, :k] = d['x'], d['x_pieces'], d['x_pos']
            ads.append(np.ascontiguousarray(d['_ad'] if '_ad' in d else np.asarray(d['ad_mat'], dtype=np.int8).reshape(k, k)).ravel())
            crosses.append(np.ascontiguousarray(d['_cross'] if '_cross' in d else np.asarray(d['edge_select'], dtype=np.int8).reshape(k, k)).ravel())
        ad = np.concatenate(ads) if ads else np.zeros(0, np.int8)
        cross = np.concatenate(crosses) if crosses else np.zeros(0, np.int8)
        L = max(len(d['pieces']) for d in data_list)
        pieces = np.full((B, L), pad_piece, dtype=np.int64)
        for i, d in enumerate(data_list):
            pieces[i, :len(d['pieces'])] = d['pieces']
        props = torch.tensor([d['props'] for d in data_list])
        return collate_matrices(lens, N, x, xp, xq, ad, cross, pieces, props, tokenizer, device, all_pairs, True)

    def __getitem__(self, idx):
        return self.process_step2(self.data[idx], self.tokenizer, lists=False)

    def __len__(self):
        return len(self.data)

    def collate_fn(self, data_list):
        return self.process_step3(data_list, self.tokenizer)


def collate_matrices(lens, N, x, xp, xq, ad, cross, pieces, props, tokenizer, device='cpu', all_pairs=False, full=True):
    """The walk of `process_step3` (`bpe_dataset.py:110-132`) over the concatenated bond-type / cross-piece
    matrices, then the packed index blob.  `full=False` leaves out the tensors of the reference's batch dict
    that only the reference's own modules read (`edge_index`, `edge_attr`, `edge_select`, `golden_edge`,
    `in_piece_edge_idx`): the kernels take everything from `'_native'`."""
    lib = _lib.lib()
    B = len(lens)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    total = int((lens.astype(np.int64) ** 2).sum())
    e_cap = int(np.count_nonzero(ad)) + 1
    p_cap = total + 1
    e_src, e_dst, e_type, in_piece = (np.zeros(e_cap, np.int32) for _ in range(4))
    pg, pm, pn, pgold = (np.zeros(p_cap, np.int32) for _ in range(4))
    ne, ni, npair = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    version, state, gauss = random.getstate()
    mt = np.asarray(state, dtype=np.uint32)
    rc = lib.psvae_collate_pairs(B, N, _ptr(lens), _ptr(ad), _ptr(cross), 2.0 if all_pairs else NEG_RATE, _ptr(mt),
                                 _ptr(e_src), _ptr(e_dst), _ptr(e_type), e_cap, C.byref(ne),
                                 _ptr(in_piece), C.byref(ni), _ptr(pg), _ptr(pm), _ptr(pn), _ptr(pgold), p_cap,
                                 C.byref(npair))
    _lib.check(rc, 'psvae_collate_pairs')
    if not all_pairs:
        random.setstate((version, tuple(mt.tolist()), gauss))
    ne, ni, npair = ne.value, ni.value, npair.value
    e_src, e_dst, e_type, in_piece = e_src[:ne], e_dst[:ne], e_type[:ne], in_piece[:ni]
    pg, pm, pn, pgold = pg[:npair], pm[:npair], pn[:npair], pgold[:npair]
    L = pieces.shape[1]
    mask = np.arange(N)[None, :] >= lens[:, None]
    props_sel = props.reshape(B, -1)[:, list(BPEMolDataset.selected_properties)].float().numpy()
    native = pack_native(lens, N, L, x, xp, xq, e_src, e_dst, e_type, in_piece, pg, pm, pn, pgold, pieces,
                         props_sel, tokenizer, BPEMolDataset.max_pos)
    as64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64))
    batch = {'x': as64(x), 'x_pieces': as64(xp), 'x_pos': as64(xq), 'atom_mask': torch.from_numpy(mask),
             'pieces': as64(pieces), 'props': props}
    if full:
        sel = np.zeros((B, N, N), dtype=bool)
        sel[pg, pm, pn] = True
        onehot = np.zeros((ne, 4), dtype=np.int64)
        onehot[np.arange(ne), e_type] = 1
        batch.update({
            'edge_index': torch.from_numpy(np.stack([e_src, e_dst]).astype(np.int64)) if ne else torch.zeros(2, 0, dtype=torch.long),
            'edge_attr': torch.from_numpy(onehot), 'edge_select': torch.from_numpy(sel),
            'golden_edge': torch.from_numpy(pgold.astype(np.int64)), 'in_piece_edge_idx': in_piece.tolist()})
    dev = torch.device(device)
    if dev.type != 'cpu':
        batch = to_device(batch, dev)
        native = native.to(dev)
    batch['_native'] = native
    return batch


class PackedMolDataset(torch.utils.data.Dataset):
    """The same batches as `BPEMolDataset` (bit for bit, including the draws from Python's `random`), for
    feeding a GPU whose step takes about a millisecond: the records are flattened into a few numpy arrays
    once, `__getitem__` hands out the index only, and `collate_fn` runs stage 2 and stage 3
    (`src/data/bpe_dataset.py:56-155`) for the whole batch in native code.  `full=True` also builds the
    tensors of the reference's batch dict that only the reference's own modules read."""

    def __init__(self, records, tokenizer, full=False):
        super().__init__()
        self.tokenizer, self.full = tokenizer, full
        M = len(records)
        n_at = np.fromiter((len(r['x']) for r in records), dtype=np.int64, count=M)
        n_bd = np.fromiter((len(r['edge_attr']) for r in records), dtype=np.int64, count=M)
        n_pc = np.fromiter((len(r['pieces']) for r in records), dtype=np.int64, count=M)
        self.atom_off = np.concatenate([[0], np.cumsum(n_at)]).astype(np.int64)
        self.bond_off = np.concatenate([[0], np.cumsum(n_bd)]).astype(np.int64)
        self.piece_off = np.concatenate([[0], np.cumsum(n_pc)]).astype(np.int64)
        tot = int(self.atom_off[-1])
        self.x_all = np.zeros(tot, np.int32)
        self.piece_of_atom = np.zeros(tot, np.int32)     # what process_step2 leaves in x_pieces (0 for an atom of no group)
        self.pos_of_atom = np.zeros(tot, np.int32)       # ... in x_pos
        self.group_of_atom = np.full(tot, -1, np.int32)
        self.bond_a = np.zeros(int(self.bond_off[-1]), np.int32)
        self.bond_b = np.zeros(int(self.bond_off[-1]), np.int32)
        self.bond_t = np.zeros(int(self.bond_off[-1]), np.int8)
        self.pieces_all = np.zeros(int(self.piece_off[-1]), np.int32)
        self.props = np.zeros((M, 3), np.float32)
        for m, r in enumerate(records):
            a0, b0, p0 = self.atom_off[m], self.bond_off[m], self.piece_off[m]
            self.x_all[a0:a0 + n_at[m]] = r['x']
            if n_bd[m]:
                ei = np.asarray(r['edge_index'], dtype=np.int32).reshape(-1, 2)
                self.bond_a[b0:b0 + n_bd[m]], self.bond_b[b0:b0 + n_bd[m]] = ei[:, 0], ei[:, 1]
                self.bond_t[b0:b0 + n_bd[m]] = r['edge_attr']
            self.pieces_all[p0:p0 + n_pc[m]] = r['pieces']
            for pos, group in enumerate(r['groups']):
                if len(group):
                    g = np.asarray(group, dtype=np.int64) + a0
                    self.group_of_atom[g] = pos
                    self.piece_of_atom[g] = r['pieces'][pos]
                    self.pos_of_atom[g] = pos
            self.props[m] = r['props']

    def __len__(self):
        return len(self.atom_off) - 1

    def __getitem__(self, idx):
        return int(idx)

    def collate_fn(self, indices, device='cpu', all_pairs=False):
        lib = _lib.lib()
        idx = np.ascontiguousarray(indices, dtype=np.int64)
        B = len(idx)
        n_at = self.atom_off[idx + 1] - self.atom_off[idx]
        N = int(n_at.max())
        L = int((self.piece_off[idx + 1] - self.piece_off[idx]).max())
        total = int((n_at ** 2).sum())
        lens = np.zeros(B, np.int32)
        x, xp, xq = (np.zeros((B, N), np.int32) for _ in range(3))
        pieces = np.zeros((B, L), np.int32)
        ad, cross = np.zeros(total, np.int8), np.zeros(total, np.int8)
        tok = self.tokenizer
        rc = lib.psvae_gather_molecules(B, _ptr(idx), _ptr(self.atom_off), _ptr(self.x_all), _ptr(self.piece_of_atom),
                                        _ptr(self.pos_of_atom), _ptr(self.group_of_atom), _ptr(self.bond_off), _ptr(self.bond_a),
                                        _ptr(self.bond_b), _ptr(self.bond_t), _ptr(self.piece_off), _ptr(self.pieces_all), N, L,
                                        tok.atom_pad_idx(), tok.pad_idx(), _ptr(lens), _ptr(x), _ptr(xp), _ptr(xq), _ptr(pieces),
                                        _ptr(ad), _ptr(cross), total)
        _lib.check(rc, 'psvae_gather_molecules')
        props = torch.from_numpy(self.props[idx])
        return collate_matrices(lens, N, x, xp, xq, ad, cross, pieces, props, tok, device, all_pairs, self.full)


def get_dataloader(records, tokenizer, batch_size, shuffle=False, num_workers=4, packed=False):
    """`get_dataloader` of the reference (`bpe_dataset.py:184-187`) over already
    preprocessed molecule records.  `packed=True` builds the same batches from a
    `PackedMolDataset` (about 20x less host time per batch; the batch dict then
    lacks the tensors only the reference's own modules read)."""
    dataset = PackedMolDataset(records, tokenizer) if packed else BPEMolDataset(records, tokenizer)
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle,
                                       collate_fn=dataset.collate_fn, num_workers=num_workers)


def to_device(batch, device, non_blocking=True):
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) or isinstance(v, NativeBatch):
            out[k] = v.to(device, non_blocking=non_blocking)
        else:
            out[k] = v
    return out


def ensure_native(batch, tokenizer, selected_properties=(0, 2), max_pos=50):
    """Pack a reference-format batch dict that has no `'_native'` entry."""
    nb = batch.get('_native')
    if nb is not None:
        return nb
    x = batch['x'].cpu().numpy()
    B, N = x.shape
    mask = batch['atom_mask'].cpu().numpy()
    lens = (~mask).sum(axis=1).astype(np.int32)
    ei = batch['edge_index'].cpu().numpy()
    ea = batch['edge_attr'].cpu().numpy()
    e_type = ea.argmax(axis=1).astype(np.int32) if ea.shape[0] else np.zeros(0, np.int32)
    sel = batch['edge_select'].cpu().numpy()
    pg, pm, pn = np.nonzero(sel)
    props = batch['props'].cpu().reshape(B, -1)[:, list(selected_properties)].float().numpy()
    pieces = batch['pieces'].cpu().numpy()
    nb = pack_native(lens, N, pieces.shape[1], x, batch['x_pieces'].cpu().numpy(), batch['x_pos'].cpu().numpy(),
                     ei[0] if ei.size else np.zeros(0, np.int32), ei[1] if ei.size else np.zeros(0, np.int32), e_type,
                     np.asarray(batch['in_piece_edge_idx'], dtype=np.int32), pg, pm, pn,
                     batch['golden_edge'].cpu().numpy(), pieces, props, tokenizer, max_pos)
    nb = nb.to(batch['x'].device)
    batch['_native'] = nb
    return nb
