"""Batches built on the GPU from a data set that lies flattened in device memory
(`psvae_collate_device`): stages 2 and 3 of the reference's dataset
(`src/data/bpe_dataset.py:56-155`) without a host pass per batch.

`DeviceMolDataset(records, tokenizer, device)` flattens the records once (what is static per
molecule: atoms, bonds as a CSR in local numbering, in-piece bonds, piece sequence, properties) and
uploads them; `collate(indices)` then needs one small upload (the molecule indices), eight kernel
launches and one 32-byte read-back (the number of selected pairs, which the step's launches are
sized by).  The result is a `NativeBatch` like the host collate's, field for field.

Negative pairs are drawn by a counter-based generator keyed by (seed, epoch, molecule, pair) instead
of Python's `random` stream, so this path is reproducible but not bit-identical to the reference's
sampling; `PackedMolDataset` / `BPEMolDataset` remain the bit-exact host paths.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .data import NEG_RATE, BPEMolDataset, NativeBatch, PackedMolDataset

MAX_ATOMS = 128     # per molecule, limit of the pair kernels


class DeviceMolDataset:
    def __init__(self, records, tokenizer, device, seed=2021, neg_rate=NEG_RATE):
        self.tokenizer, self.device = tokenizer, torch.device(device)
        self.seed, self.neg_rate = int(seed), float(neg_rate)
        flat = PackedMolDataset(records, tokenizer)          # atoms, groups, piece sequences, properties
        M = len(flat)
        self.n_at = (flat.atom_off[1:] - flat.atom_off[:-1]).astype(np.int64)
        self.n_pc = (flat.piece_off[1:] - flat.piece_off[:-1]).astype(np.int64)
        if M == 0 or self.n_at.max() > MAX_ATOMS or self.n_at.min() < 1:
            raise ValueError(f'molecules must have 1..{MAX_ATOMS} atoms for the device collate')
        rowptr, in_rowptr, nbr, typ, in_nbr, in_typ = [], [], [], [], [], []
        self.n_ed, self.n_in = np.zeros(M, np.int64), np.zeros(M, np.int64)
        for m in range(M):
            a0, n = int(flat.atom_off[m]), int(self.n_at[m])
            ad = np.zeros((n, n), np.int32)
            b0, b1 = int(flat.bond_off[m]), int(flat.bond_off[m + 1])
            if b1 > b0:     # both directions of every bond in list order: a later duplicate overwrites (bpe_dataset.py:62-66)
                u, v, t = flat.bond_a[b0:b1], flat.bond_b[b0:b1], flat.bond_t[b0:b1].astype(np.int32)
                ad[np.stack([u, v], 1).ravel(), np.stack([v, u], 1).ravel()] = np.repeat(t, 2)
            g = flat.group_of_atom[a0:a0 + n]
            same = (g[:, None] == g[None, :]) & (g[:, None] >= 0)        # edge_select == 0 (bpe_dataset.py:68-82)
            for mat, rp_list, nb_list, ty_list, cnt in ((ad, rowptr, nbr, typ, self.n_ed),
                                                        (np.where(same, ad, 0), in_rowptr, in_nbr, in_typ, self.n_in)):
                us, vs = np.nonzero(mat)                                     # row-major: neighbours ascending inside a row
                rp_list.append(np.concatenate([[0], np.cumsum(np.bincount(us, minlength=n))]).astype(np.int32))
                nb_list.append(vs.astype(np.int32)); ty_list.append(mat[us, vs].astype(np.int32))
                cnt[m] = len(us)
        cat = lambda parts: np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, np.int32)
        off32 = lambda counts: np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        if flat.atom_off[-1] + M >= 2 ** 31 or self.n_ed.sum() >= 2 ** 31 or flat.piece_off[-1] >= 2 ** 31:
            raise ValueError('data set too large for 32-bit offsets')
        host = dict(atom_off=flat.atom_off.astype(np.int32), atom_id=flat.x_all, piece_of_atom=flat.piece_of_atom,
                    pos_of_atom=flat.pos_of_atom, group_of_atom=flat.group_of_atom, edge_off=off32(self.n_ed),
                    rowptr_loc=cat(rowptr), nbr=cat(nbr), nbr_type=cat(typ), in_edge_off=off32(self.n_in),
                    in_rowptr_loc=cat(in_rowptr), in_nbr=cat(in_nbr), in_nbr_type=cat(in_typ),
                    piece_off=flat.piece_off.astype(np.int32), pieces=flat.pieces_all)
        pad1 = lambda a: a if a.size else np.zeros(1, a.dtype)               # no empty device arrays
        self._dev = {k: torch.from_numpy(np.ascontiguousarray(pad1(v), dtype=np.int32)).to(self.device) for k, v in host.items()}
        self._dev['props'] = torch.from_numpy(np.ascontiguousarray(flat.props, dtype=np.float32)).to(self.device)
        st = _lib.MolStore()
        st.n_mols = M
        for k, v in self._dev.items():
            setattr(st, k, v.data_ptr())
        st.n_props_all = flat.props.shape[1]
        self._store = st
        self._prop_cols = np.asarray(BPEMolDataset.selected_properties, dtype=np.int32)

    def __len__(self):
        return len(self.n_at)

    def sizes(self, indices):
        """(B, n, E, E_in, L, S_cap) of the batch `indices`, from the host-side counts."""
        idx = np.asarray(indices, dtype=np.int64)
        return (len(idx), int(self.n_at[idx].sum()), int(self.n_ed[idx].sum()), int(self.n_in[idx].sum()),
                int(self.n_pc[idx].max()), int((self.n_at[idx] ** 2).sum()))

    def collate(self, indices, epoch=0, all_pairs=False):
        """Batch dict `{'_native': NativeBatch}` of the molecules `indices`, built on the device."""
        L_ = _lib.lib()
        tok = self.tokenizer
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        B, n, E, E_in, L, S_cap = self.sizes(idx)
        na, npiece, mp = tok.num_atom_type(), tok.num_piece_type(), BPEMolDataset.max_pos
        offsets = (C.c_int64 * len(_lib.BLOB_FIELDS))()
        words = L_.psvae_collate_device_layout(B, n, E, E_in, S_cap, L, na, npiece, mp, offsets)
        if words < 0:
            raise RuntimeError('psvae_collate_device_layout rejected the batch sizes')
        dev = self.device
        blob = torch.zeros(words, dtype=torch.int32, device=dev)
        mol = torch.from_numpy(idx).to(dev, non_blocking=True)
        props = torch.empty(B, len(self._prop_cols), dtype=torch.float32, device=dev)
        denoms = torch.empty(4, dtype=torch.float32, device=dev)
        counts = torch.zeros(8, dtype=torch.int32, device=dev)
        nbytes = L_.psvae_collate_device_scratch_bytes(B, n, L, na, npiece, mp)
        scratch = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        _lib.check(L_.psvae_collate_device(C.byref(self._store), mol.data_ptr(), B, n, E, E_in, L, S_cap,
                                           2.0 if all_pairs else self.neg_rate, self.seed, int(epoch), na, npiece, mp,
                                           tok.pad_idx(), self._prop_cols.ctypes.data, len(self._prop_cols),
                                           blob.data_ptr(), words, props.data_ptr(), denoms.data_ptr(), counts.data_ptr(),
                                           scratch.data_ptr(), nbytes, torch.cuda.current_stream(dev).cuda_stream),
                   'psvae_collate_device')
        got = counts.cpu().tolist()          # the one read-back: S (and the consistency flags)
        if got[6] != 0:
            raise RuntimeError(f'psvae_collate_device: inconsistent sizes or keys (flags {got[6]:#x})')
        sizes = dict(B=B, n=n, E=E, E_in=E_in, S=got[0], L=L)
        return {'_native': NativeBatch(blob, [int(o) for o in offsets], sizes, props, denoms)}


class DeviceLoader:
    """Iterates over a `DeviceMolDataset` in batches, as `get_dataloader(...)` iterates over the host data sets
    (`src/data/bpe_dataset.py:184-187`): `shuffle` draws a new order every epoch (`set_epoch`, which also keys
    the negative-pair draws).  The next batch is built on a side stream while the caller's stream still runs
    the step of the previous one; the only host wait per batch is the read-back of the next batch's pair count."""

    def __init__(self, dataset, batch_size, shuffle=False, seed=0, all_pairs=False, drop_last=False):
        self.dataset, self.batch_size, self.shuffle, self.seed = dataset, int(batch_size), shuffle, int(seed)
        self.all_pairs, self.drop_last, self.epoch = all_pairs, drop_last, 0
        self._side = torch.cuda.Stream(device=dataset.device)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def batches(self):
        order = np.arange(len(self.dataset))
        if self.shuffle:
            order = np.random.default_rng([self.seed, self.epoch]).permutation(order)
        return [order[k * self.batch_size:(k + 1) * self.batch_size] for k in range(len(self))]

    def __iter__(self):
        main = torch.cuda.current_stream(self.dataset.device)
        pending = None
        self._side.wait_stream(main)            # the data set was uploaded on the caller's stream
        for idx in self.batches():
            with torch.cuda.stream(self._side):
                batch = self.dataset.collate(idx, epoch=self.epoch, all_pairs=self.all_pairs)
            nb = batch['_native']
            for t in (nb.blob, nb.props, nb.denoms):
                t.record_stream(main)           # built on the side stream, consumed on the caller's
            if pending is not None:
                yield pending
            pending = batch
            main.wait_stream(self._side)
        if pending is not None:
            yield pending
