#!/usr/bin/env python3
"""Timings of the two non-headline benchmark configurations on one GPU (device-resident, collated batches):
  * bulk latent encoding (`PSVAEModel.encode`, the deterministic part of `get_z`) in batches of 4096
    ZINC-shape graphs, with the library's per-kernel-class profile;
  * the stress shape: 60-atom graphs, 8 GINE iterations, hidden 512, every cross-piece pair scored, 32 graphs
    per batch, full training step.
Usage: config_bench.py [encode|stress|both]"""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_vae_amd import PSVAEModel, Tokenizer, _lib, default_args, full_config, synth  # noqa: E402
from ps_vae_amd.data import PackedMolDataset, to_device  # noqa: E402


def classes(raw, steps):
    rows = []
    for k, (ms, work, cnt) in sorted(raw.items(), key=lambda kv: -kv[1][0]):
        unit = 'TF' if k.startswith('gemm_') and 'reduce' not in k or k.startswith('gru') else 'GB/s'
        rate = work / (ms * 1e-3) / (1e12 if unit == 'TF' else 1e9) if ms > 0 else 0.0
        rows.append(f'    {k:20s} {ms / steps * 1e3:8.1f} us/step {cnt / steps:5.1f} launches  {rate:8.1f} {unit}')
    return '\n'.join(rows)


def encode(dev):
    V, B = 300, 4096
    tok = Tokenizer.synthetic(V)
    torch.manual_seed(2021)
    model = PSVAEModel(full_config(default_args(), tok), tok).to(dev)
    random.seed(1)
    ds = PackedMolDataset(synth.make_molecules(11, 'zinc', V, 2 * B), tok)
    t = time.perf_counter()
    batches = [ds.collate_fn(list(range(i * B, (i + 1) * B))) for i in range(2)]
    host = (time.perf_counter() - t) / 2
    batches = [to_device(b, dev) for b in batches]
    for b in batches:
        model.encode(b)
    torch.cuda.synchronize()
    n = 20
    t = time.perf_counter()
    for i in range(n):
        model.encode(batches[i % 2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    L = _lib.lib()
    L.psvae_profile_enable(1)
    for i in range(4):
        model.encode(batches[i % 2])
    torch.cuda.synchronize()
    raw = _lib.profile_collect()
    L.psvae_profile_enable(0)
    nn = batches[0]['_native'].sizes
    print(f'encode: {B} graphs ({nn["n"]} atoms, {nn["E"]} directed bonds) in {dt * 1e3:.2f} ms -> {B / dt:,.0f} graphs/s '
          f'(1 M graphs in {1e6 / (B / dt):.1f} s); host collate of such a batch (packed data set, one thread): {host * 1e3:.0f} ms')
    print(classes(raw, 4), flush=True)


def stress(dev):
    V, B = 300, 32
    tok = Tokenizer.synthetic(V)
    torch.manual_seed(2021)
    args = default_args(node_hidden_dim=512)
    cfg = full_config(args, tok)
    cfg['encoder']['t'] = 8
    cfg['vae_piece_decoder']['t'] = 8
    model = PSVAEModel(cfg, tok).to(dev)
    opt = model.configure_optimizers()
    ds = PackedMolDataset(synth.make_molecules(12, 'stress', V, 2 * B), tok)
    batches = [to_device(ds.collate_fn(list(range(i * B, (i + 1) * B)), all_pairs=True), dev) for i in range(2)]

    def step(i):
        opt.zero_grad()
        loss = model.training_step(batches[i % 2])
        loss.backward()
        opt.step()
        model.global_step += 1
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    n = 20
    t = time.perf_counter()
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    L = _lib.lib()
    L.psvae_profile_enable(1)
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    raw = _lib.profile_collect()
    L.psvae_profile_enable(0)
    nn = batches[0]['_native'].sizes
    print(f'stress: {B} graphs x 60 atoms, hidden 512, t = 8, {nn["S"]} pairs: {dt * 1e3:.2f} ms per training step -> {B / dt:,.0f} graphs/s')
    print(classes(raw, 4), flush=True)


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'both'
    dev = torch.device('cuda:0')
    if what in ('encode', 'both'):
        encode(dev)
    if what in ('stress', 'both'):
        stress(dev)
