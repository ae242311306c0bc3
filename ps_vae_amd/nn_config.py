"""Constructor-argument builders with the reference's names, flags and defaults
(`src/utils/nn_utils.py:139-206`; defaults from `src/train.py:57-111`)."""
import argparse

PROPS = ['qed', 'sa', 'logp']  # src/evaluation/utils.py


def map_prop_to_idx(props):
    """`src/evaluation/utils.py:122-134`: property names -> column indices."""
    idxs = [PROPS.index(p) for p in props]
    idxs.sort()
    return idxs


def default_args(**over):
    """The training CLI's defaults as a namespace (`src/train.py:66-110`, with
    the values `scripts/train.sh` passes for alpha / beta / step_beta)."""
    a = argparse.Namespace(
        batch_size=32, lr=1e-3, alpha=0.1, beta=0.0, step_beta=0.002, max_beta=0.01, kl_warmup=0,
        kl_anneal_iter=1000, epochs=6, patience=3, grad_clip=10.0, props=['qed', 'logp'],
        predictor_hidden_dim=200, node_hidden_dim=300, graph_embedding_dim=400, latent_dim=56, max_pos=50,
        atom_embedding_dim=50, piece_embedding_dim=100, pos_embedding_dim=50, piece_hidden_dim=200)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def common_config(args):
    return {'lr': args.lr, 'alpha': args.alpha, 'beta': args.beta, 'max_beta': args.max_beta,
            'step_beta': args.step_beta, 'kl_warmup': args.kl_warmup, 'kl_anneal_iter': args.kl_anneal_iter,
            'selected_properties': map_prop_to_idx(args.props)}


def predictor_config(args):
    return {'predictor': {'dim_feature': args.latent_dim, 'dim_hidden': args.predictor_hidden_dim,
                          'num_property': len(args.props)}}


def encoder_config(args, vocab):
    dim_in = vocab.num_atom_type()
    if hasattr(vocab, 'num_formal_charge'):
        dim_in += vocab.num_formal_charge()
    return {'encoder': {'dim_in': dim_in, 'num_edge_type': vocab.num_bond_type(),
                        'dim_hidden': args.node_hidden_dim, 'dim_out': args.graph_embedding_dim}}


def ps_vae_config(args, tokenizer):
    chem_vocab = tokenizer.chem_vocab
    atom_dim = args.atom_embedding_dim + args.piece_embedding_dim + args.pos_embedding_dim
    return {
        'encoder': {'dim_in': atom_dim, 'num_edge_type': chem_vocab.num_bond_type(),
                    'dim_hidden': args.node_hidden_dim, 'dim_out': args.graph_embedding_dim},
        'vae_piece_decoder': {
            'atom_embedding_dim': args.atom_embedding_dim, 'piece_embedding_dim': args.piece_embedding_dim,
            'max_pos': args.max_pos, 'pos_embedding_dim': args.pos_embedding_dim,
            'piece_hidden_dim': args.piece_hidden_dim, 'node_hidden_dim': args.node_hidden_dim,
            'num_edge_type': chem_vocab.num_bond_type(), 'cond_dim': args.graph_embedding_dim,
            'latent_dim': args.latent_dim, 'tokenizer': tokenizer, 't': 4},
    }


def full_config(args, tokenizer):
    """What `src/train.py:130-132` assembles before building the model."""
    config = {**common_config(args), **encoder_config(args, tokenizer.chem_vocab), **predictor_config(args)}
    config.update(ps_vae_config(args, tokenizer))
    return config
