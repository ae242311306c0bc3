#!/usr/bin/env python3
"""Throughput of the batched generation front half: piece sampling and link scoring for many latents."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_vae_amd import PSVAEModel, Tokenizer  # noqa: E402
from ps_vae_amd.inference import sample_pieces, score_links  # noqa: E402
from ps_vae_amd.nn_config import default_args, full_config  # noqa: E402
from ps_vae_amd.synth import make_molecules  # noqa: E402


def fragments(mols):
    out = []
    for mol in mols:
        groups = [g for g in mol['groups'] if len(g)]
        new_id, x, xp, xq, ng = {}, [], [], [], []
        for pos, (g, pid) in enumerate(zip(groups, mol['pieces'][1:-1])):
            ids = []
            for a in g:
                new_id[a] = len(x); ids.append(len(x)); x.append(mol['x'][a]); xp.append(pid); xq.append(pos + 1)
            ng.append(ids)
        gid = {a: i for i, g in enumerate(ng) for a in g}
        bonds, types = [], []
        for (a, b), t in zip(mol['edge_index'], mol['edge_attr']):
            a, b = new_id[a], new_id[b]
            if gid[a] == gid[b]:
                bonds.append([a, b]); types.append(t)
        out.append({'x': x, 'x_pieces': xp, 'x_pos': xq, 'bonds': bonds, 'bond_types': types, 'groups': ng})
    return out


def main():
    V = 300
    tok = Tokenizer.synthetic(V)
    torch.manual_seed(2021)
    model = PSVAEModel(full_config(default_args(), tok), tok).cuda()
    with torch.no_grad():
        model.decoder.to_vocab.bias[V + 1] += 3.0       # untrained weights: make sequences end (~8 pieces)
    atoms = [3] * V + [0, 0]
    for Bz in (256, 4096, 32768):
        z = torch.randn(Bz, 56, device='cuda')
        sample_pieces(model, z, 60, piece_atoms=atoms)
        torch.cuda.synchronize()
        t = time.perf_counter()
        seqs = sample_pieces(model, z, 60, piece_atoms=atoms)
        dt = time.perf_counter() - t
        print(f'sample_pieces Bz={Bz}: {dt * 1e3:.1f} ms, {Bz / dt:.0f} sequences/s, mean length {np.mean([len(s) for s in seqs]):.1f}',
              flush=True)
    for B in (128, 1024):
        frags = fragments(make_molecules(7, 'zinc', V, B))
        z = torch.randn(B, 56, device='cuda')
        score_links(model, z, frags)
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = score_links(model, z, frags)
        dt = time.perf_counter() - t
        pairs = sum(len(o['pairs']) for o in out)
        print(f'score_links B={B}: {dt * 1e3:.1f} ms incl. host packing and copies, {B / dt:.0f} molecules/s, {pairs} pairs', flush=True)


if __name__ == '__main__':
    main()
