"""Vocabulary surface consumed by the training step.

Mirrors the accessor surface of the reference `Tokenizer`
(`src/data/mol_bpe.py:156-173,224-249`) and `GeneralVocab`
(`src/utils/chem_utils.py:14-63`).  Reads both vocabulary file formats:
  * src format: lines `smiles \t n_atoms \t frequency`           (mol_bpe.py:158-166)
  * ps  format: the same, preceded by one JSON line {"kekulize": bool}  (ps/mol_bpe.py:174-197)
Piece index = line order; `<pad>` = V and `<s>` = V + 1 follow.

Principal-subgraph mining and greedy tokenisation of molecules need RDKit and
stay a CPU preprocessing step outside this package.
"""
import json

ATOMS = ['B', 'Br', 'C', 'Cl', 'F', 'I', 'N', 'O', 'P', 'S']
BONDS = [None, 'SINGLE', 'DOUBLE', 'TRIPLE']


class GeneralVocab:
    def __init__(self, atom_special=None, bond_special=None):
        self.idx2atom = list(ATOMS) + list(atom_special or [])
        self.atom2idx = {a: i for i, a in enumerate(self.idx2atom)}
        self.idx2bond = list(BONDS) + list(bond_special or [])
        self.bond2idx = {b: i for i, b in enumerate(self.idx2bond)}
        self.atom_special = list(atom_special or [])
        self.bond_special = list(bond_special or [])

    def idx_to_atom(self, idx):
        return self.idx2atom[idx]

    def atom_to_idx(self, atom):
        return self.atom2idx[atom]

    def idx_to_bond(self, idx):
        return self.idx2bond[idx]

    def bond_to_idx(self, bond):
        # accepts None, the names above, or an RDKit BondType (matched by name)
        if bond is None or bond in self.bond2idx:
            return self.bond2idx[bond]
        return self.bond2idx[getattr(bond, 'name', str(bond))]

    def bond_idx_to_valence(self, idx):
        return {'SINGLE': 1, 'DOUBLE': 2, 'TRIPLE': 3}.get(self.idx2bond[idx], -1)

    def num_atom_type(self):
        return len(self.idx2atom)

    def num_bond_type(self):
        return len(self.idx2bond)


class Tokenizer:
    def __init__(self, vocab_path=None, pieces=None):
        """`vocab_path`: a vocabulary file of either format; or pass `pieces`,
        a list of (smiles, n_atoms, freq), directly."""
        self.kekulize = None        # the ps format's header flag; None for the header-less src format
        if pieces is None:
            with open(vocab_path, 'r') as fin:
                lines = fin.read().strip().split('\n')
            if lines and lines[0].lstrip().startswith('{'):
                self.kekulize = bool(json.loads(lines[0]).get('kekulize', False))
                lines = lines[1:]
            pieces = []
            for line in lines:
                smi, n_atoms, freq = line.strip().split('\t')
                pieces.append((smi, int(n_atoms), int(freq)))
        self.vocab_dict, self.idx2piece, self.piece2idx = {}, [], {}
        for smi, n_atoms, freq in pieces:
            self.vocab_dict[smi] = (n_atoms, freq)
            self.piece2idx[smi] = len(self.idx2piece)
            self.idx2piece.append(smi)
        self.pad, self.end = '<pad>', '<s>'
        for smi in (self.pad, self.end):
            self.piece2idx[smi] = len(self.idx2piece)
            self.idx2piece.append(smi)
        self.atom_pad = '<pad>'
        self.chem_vocab = GeneralVocab(atom_special=[self.atom_pad])
        # names the stand-alone miner's Tokenizer uses for the same tables (ps/mol_bpe.py:185-204)
        self.idx2subgraph, self.subgraph2idx = self.idx2piece, self.piece2idx
        self.bond_start = '<bstart>'
        self.max_num_nodes = max([n for _, n, _ in pieces], default=0) + 2     # + start, padding

    @classmethod
    def synthetic(cls, vocab_size):
        """A vocabulary of `vocab_size` placeholder pieces (for synthetic data)."""
        return cls(pieces=[(f'P{i}', 1, 0) for i in range(vocab_size)])

    def tokenize(self, mol, return_idx=False):
        raise RuntimeError('molecule tokenisation needs RDKit and is a CPU preprocessing step outside this package '
                           '(reference: src/data/mol_bpe.py:175-222)')

    __call__ = tokenize

    def idx_to_piece(self, idx):
        return self.idx2piece[idx]

    def piece_to_idx(self, piece):
        return self.piece2idx[piece]

    # the same two accessors under the names of ps/mol_bpe.py:245-249,260-261
    idx_to_subgraph, subgraph_to_idx = idx_to_piece, piece_to_idx

    def num_subgraph_type(self):
        return len(self.idx2piece)

    def atom_pos_pad_idx(self):
        return self.max_num_nodes - 1

    def atom_pos_start_idx(self):
        return self.max_num_nodes - 2

    def pad_idx(self):
        return self.piece2idx[self.pad]

    def end_idx(self):
        return self.piece2idx[self.end]

    def atom_pad_idx(self):
        return self.chem_vocab.atom_to_idx(self.atom_pad)

    def num_piece_type(self):
        return len(self.idx2piece)

    def num_atom_type(self):
        return self.chem_vocab.num_atom_type()

    def __len__(self):
        return len(self.idx2piece)
