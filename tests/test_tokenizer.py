"""`Tokenizer(vocab_path)` on the reference's own vocabulary file.

`tests/golden/vocab_ps.txt` is a copy of the data file `ps/output/vocab_temp.txt` of the reference (the
12-entry vocabulary its authors mined from `ps/test/test_temp.txt`): the stand-alone miner's format, one JSON
header line and then `smiles \\t atoms \\t frequency` (`ps/mol_bpe.py:174-197`).  `vocab_src.txt` is the same
without the header: the format `src/data/mol_bpe.py:151-166` writes and reads.  The sizes asserted here are what
`src/utils/nn_utils.py:180-206` turns into layer shapes (`num_piece_type` rows of the piece table and of
`to_vocab`, `pad_idx` as the ignored class of the piece loss)."""
import os

import pytest

from ps_vae_amd import PSVAEModel, Tokenizer, default_args, full_config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
PIECES = ['C=CC', 'C=C', 'B', 'Br', 'C', 'Cl', 'F', 'I', 'N', 'O', 'P', 'S']


@pytest.mark.parametrize('name,kekulize', [('vocab_ps.txt', False), ('vocab_src.txt', None)])
def test_reads_both_vocabulary_formats(name, kekulize):
    tok = Tokenizer(os.path.join(GOLDEN, name))
    assert tok.kekulize is kekulize
    assert tok.num_piece_type() == 14 and len(tok) == 14
    assert tok.pad_idx() == 12 and tok.end_idx() == 13
    assert [tok.idx_to_piece(i) for i in range(12)] == PIECES          # index = line order
    assert tok.idx_to_piece(12) == '<pad>' and tok.idx_to_piece(13) == '<s>'
    assert all(tok.piece_to_idx(p) == i for i, p in enumerate(PIECES))
    assert tok.vocab_dict['C=CC'] == (3, 3) and tok.vocab_dict['C'] == (1, 12)
    # atom level: 10 elements + <pad>, 4 bond classes (src/utils/chem_utils.py:14-31)
    assert tok.num_atom_type() == 11 and tok.atom_pad_idx() == 10
    assert tok.chem_vocab.num_bond_type() == 4 and tok.chem_vocab.atom_to_idx('C') == 2
    # the miner's names for the same tables (ps/mol_bpe.py:185-204,245-261)
    assert tok.num_subgraph_type() == 14 and tok.idx_to_subgraph(0) == 'C=CC' and tok.subgraph_to_idx('<s>') == 13
    assert tok.max_num_nodes == 3 + 2 and tok.atom_pos_pad_idx() == 4 and tok.atom_pos_start_idx() == 3


def test_both_formats_give_the_same_model_shapes():
    a, b = Tokenizer(os.path.join(GOLDEN, 'vocab_ps.txt')), Tokenizer(os.path.join(GOLDEN, 'vocab_src.txt'))
    assert a.idx2piece == b.idx2piece and a.vocab_dict == b.vocab_dict
    model = PSVAEModel(full_config(default_args(), a), a)
    sd = model.state_dict()
    assert tuple(sd['decoder.piece_embedding.weight'].shape) == (14, 100)
    assert tuple(sd['decoder.to_vocab.weight'].shape) == (14, 200)
    assert model._dims.pad_idx == 12 and model._dims.num_piece == 14


def test_tokenising_molecules_is_left_to_rdkit():
    tok = Tokenizer(os.path.join(GOLDEN, 'vocab_ps.txt'))
    with pytest.raises(RuntimeError, match='RDKit'):
        tok('C=CC')
