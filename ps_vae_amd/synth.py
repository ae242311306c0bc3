"""Seeded synthetic molecule generator.

Produces molecule records in the raw form the reference's dataset caches after
its RDKit stage (`src/data/bpe_dataset.py:47-54`): atom ids, undirected bond
list, bond types, three property scores, the piece-id sequence with a leading
and trailing `<s>`, and the atom groups of every piece (the two `<s>` groups
are empty).  RDKit cannot run here, so graph shapes follow the statistics of
the shipped SMILES files instead: ZINC250K averages 23.2 heavy atoms and 1.075
bonds per atom (max 38 atoms); QM9 8.8 atoms and 1.07 bonds per atom (max 9).
"""
from dataclasses import dataclass

import numpy as np

NUM_ATOM_TYPES = 10  # real atom ids 0..9; id 10 is the atom padding id
_ATOM_P = np.array([0.005, 0.01, 0.72, 0.015, 0.02, 0.002, 0.12, 0.10, 0.003, 0.005])
_ATOM_P = _ATOM_P / _ATOM_P.sum()
_BOND_P = np.array([0.80, 0.18, 0.02])


@dataclass(frozen=True)
class Shape:
    name: str
    mean_atoms: float
    sd_atoms: float
    min_atoms: int
    max_atoms: int
    bonds_per_atom: float
    all_pairs: bool = False


QM9 = Shape('qm9', 8.8, 0.5, 4, 9, 1.07)
ZINC = Shape('zinc', 23.2, 4.5, 6, 38, 1.075)
STRESS = Shape('stress', 60.0, 0.0, 60, 60, 1.10, all_pairs=True)
SHAPES = {s.name: s for s in (QM9, ZINC, STRESS)}


def _random_molecule_graph(rng, n, bonds_per_atom):
    """Random connected graph: a spanning tree with degree <= 4, then
    ring-closing chords until the bond count reaches n * bonds_per_atom."""
    deg = np.zeros(n, dtype=np.int64)
    bonds = []
    have = set()
    for v in range(1, n):
        cand = [u for u in range(v) if deg[u] < 4]
        # favour recent atoms so the tree is chain-like, as molecules are
        w = np.array([1.0 + 3.0 * (u >= v - 3) for u in cand])
        u = cand[int(rng.choice(len(cand), p=w / w.sum()))]
        bonds.append((u, v))
        have.add((u, v))
        deg[u] += 1
        deg[v] += 1
    target = max(n - 1, int(round(n * bonds_per_atom)))
    tries = 0
    while len(bonds) < target and tries < 50 * n:
        tries += 1
        a = int(rng.integers(0, n))
        span = int(rng.integers(2, 7))
        b = a + span
        if b >= n:
            continue
        if deg[a] >= 4 or deg[b] >= 4 or (a, b) in have:
            continue
        bonds.append((a, b))
        have.add((a, b))
        deg[a] += 1
        deg[b] += 1
    return bonds


def _partition_into_pieces(rng, n, bonds):
    """Cut the atoms into connected groups of size ~ 1 + Poisson(2.5)."""
    nbrs = [[] for _ in range(n)]
    for a, b in bonds:
        nbrs[a].append(b)
        nbrs[b].append(a)
    unassigned = set(range(n))
    groups = []
    while unassigned:
        seed = min(unassigned)
        want = 1 + int(rng.poisson(2.5))
        group = [seed]
        unassigned.discard(seed)
        frontier = [seed]
        while frontier and len(group) < want:
            cur = frontier.pop(0)
            for nb in nbrs[cur]:
                if nb in unassigned and len(group) < want:
                    unassigned.discard(nb)
                    group.append(nb)
                    frontier.append(nb)
        groups.append(sorted(group))
    return groups


def make_molecule(rng, shape: Shape, vocab_size: int, n_atoms=None):
    """One raw molecule record (the dict of `bpe_dataset.py:47-54`)."""
    if n_atoms is None:
        if shape.sd_atoms > 0:
            n_atoms = int(round(rng.normal(shape.mean_atoms, shape.sd_atoms)))
        else:
            n_atoms = int(shape.mean_atoms)
        n_atoms = min(max(n_atoms, shape.min_atoms), shape.max_atoms)
    n = n_atoms
    bonds = _random_molecule_graph(rng, n, shape.bonds_per_atom) if n > 1 else []
    x = [int(v) for v in rng.choice(NUM_ATOM_TYPES, size=n, p=_ATOM_P)]
    edge_attr = [int(v) + 1 for v in rng.choice(3, size=len(bonds), p=_BOND_P)]
    groups = _partition_into_pieces(rng, n, bonds)
    order = rng.permutation(len(groups))
    groups = [groups[i] for i in order]
    piece_ids = [int(v) for v in rng.integers(0, vocab_size, size=len(groups))]
    end_idx = vocab_size + 1
    return {
        'x': x,
        'edge_index': [[int(a), int(b)] for a, b in bonds],
        'edge_attr': edge_attr,
        'props': [float(v) for v in rng.uniform(0.0, 1.0, size=3)],
        'pieces': [end_idx] + piece_ids + [end_idx],
        'groups': [[]] + [list(map(int, g)) for g in groups] + [[]],
    }


def make_molecules(seed, shape, vocab_size, count):
    if isinstance(shape, str):
        shape = SHAPES[shape]
    rng = np.random.default_rng(seed)
    return [make_molecule(rng, shape, vocab_size) for _ in range(count)]
