"""ctypes binding of libpsvae_hip.so (declarations mirror include/psvae.h).

There is no fallback: if the shared library is missing or does not load, every
entry point raises, so a silent CPU path can never stand in for the HIP one.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libpsvae_hip.so')

i32, i64, f32, vp, sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class Dims(C.Structure):
    _fields_ = [(k, i32) for k in (
        'num_piece', 'num_atom', 'num_bond', 'max_pos', 'd_atom', 'd_piece', 'd_pos', 'hidden', 'graph', 'latent',
        'rnn', 'pred_hidden', 'n_props', 't_enc', 't_dec', 'pad_idx')] + [('eps_enc', f32), ('eps_dec', f32)]


PARAM_FIELDS = (
    'pred_w0', 'pred_b0', 'pred_w1', 'pred_b1', 'pred_wo', 'pred_bo',
    'enc_node_w', 'enc_node_b', 'enc_edge_w', 'enc_edge_b', 'enc_w1', 'enc_b1', 'enc_w2', 'enc_b2',
    'enc_lin_w', 'enc_lin_b', 'atom_emb', 'piece_emb', 'pos_emb', 'l2h_w', 'l2h_b',
    'gru_wih', 'gru_whh', 'gru_bih', 'gru_bhh', 'vocab_w', 'vocab_b',
    'dec_node_w', 'dec_node_b', 'dec_edge_w', 'dec_edge_b', 'dec_w1', 'dec_b1', 'dec_w2', 'dec_b2',
    'sc_w0', 'sc_b0', 'sc_w1', 'sc_b1', 'sc_w2', 'sc_b2', 'sc_w3', 'sc_b3',
    'mean_w', 'mean_b', 'lvar_w', 'lvar_b')

# struct field -> state-dict key of the reference model
PARAM_KEYS = {
    'pred_w0': 'predictor.mlp.0.weight', 'pred_b0': 'predictor.mlp.0.bias',
    'pred_w1': 'predictor.mlp.2.weight', 'pred_b1': 'predictor.mlp.2.bias',
    'pred_wo': 'predictor.output.weight', 'pred_bo': 'predictor.output.bias',
    'enc_node_w': 'encoder.node_trans.weight', 'enc_node_b': 'encoder.node_trans.bias',
    'enc_edge_w': 'encoder.edge_trans.weight', 'enc_edge_b': 'encoder.edge_trans.bias',
    'enc_w1': 'encoder.conv.nn.seq.0.weight', 'enc_b1': 'encoder.conv.nn.seq.0.bias',
    'enc_w2': 'encoder.conv.nn.seq.2.weight', 'enc_b2': 'encoder.conv.nn.seq.2.bias',
    'enc_lin_w': 'encoder.linear.weight', 'enc_lin_b': 'encoder.linear.bias',
    'atom_emb': 'decoder.atom_embedding.weight', 'piece_emb': 'decoder.piece_embedding.weight',
    'pos_emb': 'decoder.pos_embedding.weight',
    'l2h_w': 'decoder.latent_to_rnn_hidden.weight', 'l2h_b': 'decoder.latent_to_rnn_hidden.bias',
    'gru_wih': 'decoder.rnn.weight_ih_l0', 'gru_whh': 'decoder.rnn.weight_hh_l0',
    'gru_bih': 'decoder.rnn.bias_ih_l0', 'gru_bhh': 'decoder.rnn.bias_hh_l0',
    'vocab_w': 'decoder.to_vocab.weight', 'vocab_b': 'decoder.to_vocab.bias',
    'dec_node_w': 'decoder.graph_embedding.node_trans.weight', 'dec_node_b': 'decoder.graph_embedding.node_trans.bias',
    'dec_edge_w': 'decoder.graph_embedding.edge_trans.weight', 'dec_edge_b': 'decoder.graph_embedding.edge_trans.bias',
    'dec_w1': 'decoder.graph_embedding.conv.nn.seq.0.weight', 'dec_b1': 'decoder.graph_embedding.conv.nn.seq.0.bias',
    'dec_w2': 'decoder.graph_embedding.conv.nn.seq.2.weight', 'dec_b2': 'decoder.graph_embedding.conv.nn.seq.2.bias',
    'sc_w0': 'decoder.edge_predictor.0.seq.0.weight', 'sc_b0': 'decoder.edge_predictor.0.seq.0.bias',
    'sc_w1': 'decoder.edge_predictor.0.seq.2.weight', 'sc_b1': 'decoder.edge_predictor.0.seq.2.bias',
    'sc_w2': 'decoder.edge_predictor.0.seq.4.weight', 'sc_b2': 'decoder.edge_predictor.0.seq.4.bias',
    'sc_w3': 'decoder.edge_predictor.1.weight', 'sc_b3': 'decoder.edge_predictor.1.bias',
    'mean_w': 'decoder.W_mean.weight', 'mean_b': 'decoder.W_mean.bias',
    'lvar_w': 'decoder.W_log_var.weight', 'lvar_b': 'decoder.W_log_var.bias',
}


class Params(C.Structure):
    _fields_ = [(k, vp) for k in PARAM_FIELDS]


BATCH_PTRS = (
    'atom_id', 'piece_id', 'pos_id', 'node_graph', 'graph_off',
    'rowptr', 'col', 'etype', 'rowptr_t', 'col_t', 'etype_t',
    'in_rowptr', 'in_col', 'in_etype', 'in_rowptr_t', 'in_col_t', 'in_etype_t',
    'pair_src', 'pair_dst', 'pair_graph', 'pair_gold',
    'pair_by_src_ptr', 'pair_by_src_idx', 'pair_by_dst_ptr', 'pair_by_dst_idx', 'pair_graph_off', 'pieces',
    'atom_rows_ptr', 'atom_rows_idx', 'piece_rows_ptr', 'piece_rows_idx', 'pos_rows_ptr', 'pos_rows_idx',
    'seq_rows_ptr', 'seq_rows_idx')
# index-blob field order (enum PSVAE_BLOB_* of the header) - identical to BATCH_PTRS
BLOB_FIELDS = BATCH_PTRS


class Adam(C.Structure):
    _fields_ = ([(k, vp) for k in ('params', 'grads', 'exp_avg', 'exp_avg_sq')] + [('count', i64), ('partials', vp)] +
                [('max_norm', f32)] + [(k, C.c_double) for k in ('lr', 'beta1', 'beta2', 'eps', 'bias1', 'bias2')] + [('grad_norm_out', vp)])


class EpsDraw(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('draw', C.c_uint64), ('first_row', C.c_int64)]


class Batch(C.Structure):
    _fields_ = ([(k, i32) for k in ('B', 'n', 'E', 'E_in', 'S', 'L')] + [(k, vp) for k in BATCH_PTRS] +
                [('props', vp), ('eps', vp), ('denoms', vp), ('conds', vp), ('d_conds', vp), ('d_z', vp), ('status', vp)])


class MolStore(C.Structure):
    _fields_ = ([('n_mols', i64)] + [(k, vp) for k in (
        'atom_off', 'atom_id', 'piece_of_atom', 'pos_of_atom', 'group_of_atom', 'edge_off', 'rowptr_loc', 'nbr', 'nbr_type',
        'in_edge_off', 'in_rowptr_loc', 'in_nbr', 'in_nbr_type', 'piece_off', 'pieces', 'props')] + [('n_props_all', i32)])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} is missing: build it with `python -m ps_vae_amd.build` '
                           '(there is no CPU fallback for the HIP kernels)')
    # torch ships its own HIP runtime (libamdhip64); it must be the copy already mapped when this
    # library is loaded, otherwise two runtimes coexist and torch's streams mean nothing to ours.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    L.psvae_strerror.restype = C.c_char_p
    L.psvae_strerror.argtypes = [C.c_int]
    L.psvae_abi_version.restype = C.c_int
    L.psvae_option_set.argtypes = [C.c_char_p, C.c_int]
    L.psvae_option_get.argtypes = [C.c_char_p, P(C.c_int)]
    L.psvae_option_name.restype = C.c_char_p
    L.psvae_option_name.argtypes = [C.c_int]
    L.psvae_workspace_bytes.restype = sz
    L.psvae_workspace_bytes.argtypes = [P(Dims), P(Batch)]
    L.psvae_workspace_offset.restype = i64
    L.psvae_workspace_offset.argtypes = [P(Dims), P(Batch), C.c_char_p]
    L.psvae_step_forward.argtypes = [P(Dims), P(Params), P(Batch), vp, sz, P(f32 * 3), vp, vp]
    L.psvae_step_backward.argtypes = [P(Dims), P(Params), P(Batch), vp, sz, P(f32 * 3), vp, P(Params), vp]
    L.psvae_step_backward_signal.argtypes = [P(Dims), P(Params), P(Batch), vp, sz, P(f32 * 3), vp, P(Params), vp, vp, vp]
    L.psvae_train_step.argtypes = [P(Dims), P(Params), P(Batch), vp, sz, P(f32 * 3), vp, vp, P(Params), P(Adam), P(EpsDraw), vp, vp, vp]
    L.psvae_streams_init.argtypes = [vp]
    L.psvae_encode.argtypes = [P(Dims), P(Params), P(Batch), vp, sz, vp, vp, vp]
    L.psvae_sample_pieces_workspace_bytes.restype = sz
    L.psvae_sample_pieces_workspace_bytes.argtypes = [P(Dims), C.c_int]
    L.psvae_sample_pieces.argtypes = [P(Dims), P(Params), vp, C.c_int, vp, C.c_int, f32, vp, vp, vp, C.c_int, vp, sz, vp]
    L.psvae_score_links.argtypes = [P(Dims), P(Params), P(Batch), vp, vp, sz, vp, vp, vp, vp]
    L.psvae_normal_rows.argtypes = [C.c_uint64, C.c_uint64, i64, C.c_int, C.c_int, vp, vp]
    L.psvae_adam_partials.restype = sz
    L.psvae_adam_partials.argtypes = [i64]
    L.psvae_clip_adam.argtypes = [vp, vp, vp, vp, i64, vp, f32] + [C.c_double] * 6 + [vp, vp]
    L.psvae_linear_scratch_bytes.restype = sz
    L.psvae_linear_scratch_bytes.argtypes = [C.c_int] * 4
    L.psvae_linear_forward.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, sz, vp]
    L.psvae_linear_backward_input.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]
    L.psvae_linear_backward_weight.argtypes = [vp, C.c_int, i64, vp, C.c_int, i64, C.c_int, vp, vp, C.c_int, C.c_int,
                                               C.c_int, vp, sz, vp]
    L.psvae_noalloc_scratch_bytes.restype = sz
    L.psvae_linear_backward_weight_noalloc.argtypes = [vp, C.c_int, i64, vp, C.c_int, i64, C.c_int, vp, vp, C.c_int, C.c_int,
                                                       C.c_int, vp]
    L.psvae_gine_aggregate_forward.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, f32, vp, C.c_int, C.c_int, C.c_int, vp]
    L.psvae_gine_layer_scratch_bytes.restype = sz
    L.psvae_gine_layer_scratch_bytes.argtypes = [C.c_int]
    L.psvae_gine_layer_forward.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, sz, vp]
    L.psvae_gine_layer_backward.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, sz, vp]
    L.psvae_scorer_backward.argtypes = [vp] * 9 + [C.c_int] * 3 + [vp, sz, vp]
    L.psvae_scorer_scratch_bytes.restype = sz
    L.psvae_scorer_scratch_bytes.argtypes = [C.c_int, C.c_int]
    L.psvae_scorer_forward.argtypes = [vp, C.c_int] + [vp] * 20 + [C.c_int] * 3 + [vp, sz, vp]
    L.psvae_gine_aggregate_backward.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, f32, vp, C.c_int, vp, vp,
                                                C.c_int, vp, vp, vp, C.c_int, C.c_int, vp]
    L.psvae_gine_edge_grad_scratch.restype = sz
    L.psvae_gine_edge_grad_scratch.argtypes = [C.c_int, C.c_int]
    L.psvae_segment_pool_forward.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.psvae_gru_exchange_bytes.restype = C.c_size_t
    L.psvae_gru_exchange_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    L.psvae_gru_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]
    L.psvae_gru_forward_stamped.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp, vp]
    L.psvae_gru_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]
    L.psvae_softmax_ce_forward.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.psvae_collate_pairs.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_double, vp, vp, vp, vp, i64, P(i64), vp, P(i64),
                                      vp, vp, vp, vp, i64, P(i64)]
    L.psvae_gather_molecules.argtypes = [C.c_int] + [vp] * 12 + [C.c_int] * 4 + [vp] * 7 + [i64]
    L.psvae_pack_bound.restype = i64
    L.psvae_pack_bound.argtypes = [C.c_int, C.c_int, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int]
    L.psvae_pack_batch.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, vp,
                                   i64, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, vp, P(i64), P(i32), P(i32)]
    L.psvae_collate_device_layout.restype = i64
    L.psvae_collate_device_layout.argtypes = [C.c_int, C.c_int, i64, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.psvae_collate_device_scratch_bytes.restype = sz
    L.psvae_collate_device_scratch_bytes.argtypes = [C.c_int] * 6
    L.psvae_collate_device.argtypes = [P(MolStore), vp, C.c_int, C.c_int, i64, i64, C.c_int, i64, C.c_double, C.c_uint64,
                                       C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, i64, vp, vp, vp,
                                       vp, sz, vp]
    L.psvae_profile_enable.argtypes = [C.c_int]
    L.psvae_profile_num_classes.restype = C.c_int
    L.psvae_profile_class_name.restype = C.c_char_p
    L.psvae_profile_class_name.argtypes = [C.c_int]
    L.psvae_profile_collect.argtypes = [vp, vp, vp]
    L.psvae_profile_event_overhead_ms.restype = C.c_double
    L.psvae_profile_event_overhead_ms.argtypes = [vp]
    if L.psvae_abi_version() != 7:
        raise RuntimeError('libpsvae_hip.so ABI version mismatch; rebuild it')
    _lib = L
    return L


EXPORTS = (
    'psvae_strerror', 'psvae_abi_version', 'psvae_option_set', 'psvae_option_get', 'psvae_option_name',
    'psvae_gine_layer_backward', 'psvae_scorer_backward', 'psvae_workspace_bytes', 'psvae_workspace_offset', 'psvae_step_forward', 'psvae_train_step',
    'psvae_step_backward', 'psvae_step_backward_signal', 'psvae_streams_init', 'psvae_encode', 'psvae_sample_pieces_workspace_bytes', 'psvae_sample_pieces',
    'psvae_score_links', 'psvae_normal_rows', 'psvae_adam_partials', 'psvae_clip_adam', 'psvae_linear_scratch_bytes',
    'psvae_linear_forward', 'psvae_linear_backward_input', 'psvae_linear_backward_weight',
    'psvae_noalloc_scratch_bytes', 'psvae_linear_backward_weight_noalloc',
    'psvae_gine_aggregate_forward', 'psvae_scorer_scratch_bytes', 'psvae_scorer_forward', 'psvae_gine_layer_scratch_bytes', 'psvae_gine_layer_forward', 'psvae_gine_aggregate_backward', 'psvae_gine_edge_grad_scratch',
    'psvae_segment_pool_forward',
    'psvae_gru_exchange_bytes', 'psvae_gru_forward', 'psvae_gru_forward_stamped', 'psvae_gru_backward', 'psvae_softmax_ce_forward', 'psvae_collate_pairs', 'psvae_gather_molecules',
    'psvae_pack_bound', 'psvae_pack_batch', 'psvae_collate_device_layout', 'psvae_collate_device_scratch_bytes',
    'psvae_collate_device', 'psvae_profile_enable', 'psvae_profile_num_classes',
    'psvae_profile_class_name', 'psvae_profile_collect', 'psvae_profile_event_overhead_ms')


def profile_collect():
    """{class name: (milliseconds, work, launches)} since the last call; clears the log."""
    import numpy as np
    L = lib()
    k = L.psvae_profile_num_classes()
    ms, work, cnt = np.zeros(k), np.zeros(k), np.zeros(k, dtype=np.int64)
    check(L.psvae_profile_collect(ms.ctypes.data, work.ctypes.data, cnt.ctypes.data), 'psvae_profile_collect')
    return {L.psvae_profile_class_name(i).decode(): (float(ms[i]), float(work[i]), int(cnt[i])) for i in range(k) if cnt[i]}


def set_option(name, value):
    """Force a kernel-selection option (`include/psvae.h`: psvae_option_set); -1 gives the choice back to the library.
    Returns the previous value."""
    L = lib()
    old = C.c_int(0)
    check(L.psvae_option_get(name.encode(), C.byref(old)), f'psvae_option_get({name})')
    check(L.psvae_option_set(name.encode(), int(value)), f'psvae_option_set({name})')
    return old.value


def option_names():
    L, out, i = lib(), [], 0
    while True:
        n = L.psvae_option_name(i)
        if n is None:
            return out
        out.append(n.decode())
        i += 1


def check(rc, what='psvae call'):
    if rc != 0:
        raise RuntimeError(f'{what} failed: {lib().psvae_strerror(rc).decode()} (code {rc})')
