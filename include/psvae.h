/*
 * psvae.h - C ABI of libpsvae_hip.so: the PS-VAE encoder/decoder training step
 * as hand-written gfx950 (MI355X) kernels.
 *
 * The reference has no FFI of its own: its hot path is the Python nn.Module
 * surface of PS-VAE-main/src/modules/ (encoder, vae_piece_decoder, predictor, common_nn) and src/pl_models/ps_vae_model.py.
 * Every entry point below therefore cites the reference function whose
 * arithmetic it replaces; the Python package ps_vae_amd binds them with ctypes
 * underneath modules that keep the reference's signatures.
 *
 * Conventions
 *  - All device pointers are plain HIP device addresses (e.g. tensor.data_ptr()).
 *    Float tensors are fp32, index tensors int32, row-major, contiguous unless a
 *    leading dimension ("ld", in elements) is passed.
 *  - The caller owns every buffer, including the workspace; the library
 *    allocates no device memory.  Its only state is, per device, two side
 *    streams and a set of events created on the first step call (see
 *    psvae_step_forward) and shared by all step calls on that device.  Host
 *    threads that call psvae_step_forward / _backward / _backward_signal /
 *    psvae_train_step for the same device at the same time are serialised by the
 *    library (a mutex per device held while a call enqueues), and calls on
 *    different streams get correct dependencies (every wait belongs to the record
 *    made by the same call); what stays the caller's duty is memory: two calls
 *    that share a workspace must be ordered on the GPU (same stream, or an
 *    event).  The other entry points have no such state.  Besides that the
 *    library holds, per device, one 32 MiB device array
 *    (psvae_linear_backward_weight_noalloc).
 *  - Work is enqueued on the hipStream_t passed as `stream` (a void* here so the
 *    header needs no HIP include); no entry point synchronises the device.
 *  - Return value: 0 = success; negative = PSVAE_E_* argument error detected on
 *    the host before anything was launched; positive = the hipError_t of a
 *    failed launch.  psvae_strerror() describes either.
 */
#ifndef PSVAE_H
#define PSVAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSVAE_ABI_VERSION 7

enum {
    PSVAE_OK = 0,
    PSVAE_E_NULL = -1,      /* a required pointer is NULL */
    PSVAE_E_SHAPE = -2,     /* a size is negative / inconsistent / unsupported */
    PSVAE_E_ALIGN = -3,     /* pointer or leading dimension breaks an alignment rule */
    PSVAE_E_WORKSPACE = -4, /* workspace too small (see psvae_workspace_bytes) */
    PSVAE_E_CAPACITY = -5,  /* host output buffer too small */
    PSVAE_E_STATIC_LIMIT = -6 /* problem needs more scratch than the no-allocation entry point's static array holds */
};

const char* psvae_strerror(int code);
int psvae_abi_version(void);

/* ---- kernel-selection options (no counterpart in the reference) -------------
 * Where the library chooses between two forms of a kernel by problem size, the choice can be forced: value -1 =
 * the library's own choice (the default), otherwise as listed.  An option is read on every call, so it can be
 * changed between calls of one process (the tests run the same step under every setting); at load time each
 * option takes the value of the environment variable PSVAE_<NAME IN CAPITALS> if that is set.  Results under any
 * setting satisfy the same parity bounds; with the same setting two runs are bitwise identical.
 *   streams        0: the step's independent branches stay on the caller's stream (no internal side streams)
 *   fused_layer    0: a GINE application runs as aggregation + two GEMM launches instead of one fused launch
 *   fused_bwd      0 / 1: backward of the GINE MLPs as two GEMMs / as one launch
 *   fused_scorer   0: link scorer as gather + GEMM chain; 1: one launch, forward and backward; 2: forward only
 *   split_bf16     0: the fused kernels' products on the fp32 matrix instruction instead of the bf16 instruction
 *                  over operands cut into three bf16 pieces
 *   split_rows     32 / 64: nodes per workgroup of the cut-operand GINE kernels
 *   split_gemm     0 / 1: 64-tile weight gradients on the fp32 / on the cut-operand kernel
 *   wide_gemm      0 / 1: the 160-tile weight-gradient kernel never / wherever the reduction is long enough
 *   gru_multi_cu   0 / 1: single-CU / multi-CU recurrence
 *   gru_tiles      1 / 2 / 4: sequence tiles a workgroup of the multi-CU recurrence steps together
 *   gru_spin_ticks bound of a wait inside the multi-CU recurrence in ticks of 10 ns (default 2 s); 0: give up at
 *                  once, which exercises the redo on the single-CU kernels
 * psvae_option_set / _get return PSVAE_E_SHAPE for an unknown name; psvae_option_name(i) enumerates the names
 * (NULL past the last). */
int psvae_option_set(const char* name, int value);
int psvae_option_get(const char* name, int* value);
const char* psvae_option_name(int index);

/* ---- model sizes (src/utils/nn_utils.py:180-206, src/train.py:92-110) ---- */
typedef struct psvae_dims {
    int32_t num_piece;   /* V + 2  (Tokenizer.num_piece_type, src/data/mol_bpe.py:239) */
    int32_t num_atom;    /* 11 */
    int32_t num_bond;    /* 4; the kernels require exactly 4 */
    int32_t max_pos;     /* rows of pos_embedding */
    int32_t d_atom, d_piece, d_pos; /* embedding widths; node width = their sum */
    int32_t hidden;      /* GINE hidden H (multiple of 4) */
    int32_t graph;       /* encoder.linear output G */
    int32_t latent;      /* Z */
    int32_t rnn;         /* GRU hidden R */
    int32_t pred_hidden;
    int32_t n_props;
    int32_t t_enc, t_dec; /* GINE iterations of encoder / decoder-side encoder */
    int32_t pad_idx;     /* ignore_index of the piece loss (= V) */
    float eps_enc, eps_dec; /* GINEConv eps buffers (conv.eps) */
} psvae_dims;

/* ---- parameters: one device pointer per state-dict entry ------------------
 * Names in comments are the reference's state-dict keys.  The same struct type
 * carries the gradient buffers (same shapes) for psvae_step_backward.
 * Optional fast path: when enc_node_w/dec_node_w (and their biases) are
 * adjacent in memory the two node_trans products run as one grouped launch;
 * likewise W_mean/W_log_var.  Adjacency is detected, never required. */
typedef struct psvae_params {
    float *pred_w0, *pred_b0, *pred_w1, *pred_b1, *pred_wo, *pred_bo; /* predictor.mlp.{0,2}, predictor.output */
    float *enc_node_w, *enc_node_b;   /* encoder.node_trans   [H, Din] */
    float *enc_edge_w, *enc_edge_b;   /* encoder.edge_trans   [H, 4]   */
    float *enc_w1, *enc_b1, *enc_w2, *enc_b2; /* encoder.conv.nn.seq.{0,2} [H,H] */
    float *enc_lin_w, *enc_lin_b;     /* encoder.linear [G, t_enc*H] */
    float *atom_emb, *piece_emb, *pos_emb; /* decoder.{atom,piece,pos}_embedding.weight */
    float *l2h_w, *l2h_b;             /* decoder.latent_to_rnn_hidden [R, Z] */
    float *gru_wih, *gru_whh, *gru_bih, *gru_bhh; /* decoder.rnn.*_l0 ([3R,Dp],[3R,R],[3R],[3R]) */
    float *vocab_w, *vocab_b;         /* decoder.to_vocab [V+2, R] */
    float *dec_node_w, *dec_node_b, *dec_edge_w, *dec_edge_b; /* decoder.graph_embedding.* */
    float *dec_w1, *dec_b1, *dec_w2, *dec_b2;
    float *sc_w0, *sc_b0, *sc_w1, *sc_b1, *sc_w2, *sc_b2; /* decoder.edge_predictor.0.seq.{0,2,4} */
    float *sc_w3, *sc_b3;             /* decoder.edge_predictor.1 [4, 2H+Z] */
    float *mean_w, *mean_b, *lvar_w, *lvar_b; /* decoder.W_mean / W_log_var [Z, G] */
} psvae_params;

/* ---- one packed batch ------------------------------------------------------
 * Packed layout: the n real atoms of the batch, graph after graph (padding rows
 * of the reference's [B,N] layout carry no edges, are masked out of pooling and
 * are never selected as pairs, so dropping them is exact).  Built on the host by
 * psvae_pack_batch from the index arrays of process_step3
 * (src/data/bpe_dataset.py:87-155). */
typedef struct psvae_batch {
    int32_t B, n, E, E_in, S, L; /* graphs, atoms, directed bonds, in-piece bonds, selected pairs, padded sequence length */
    const int32_t *atom_id, *piece_id, *pos_id; /* [n] */
    const int32_t *node_graph;   /* [n]   graph of each atom */
    const int32_t *graph_off;    /* [B+1] first atom of each graph */
    /* full bond graph: CSR by target (rows = target, entries = sources in edge order) and by source */
    const int32_t *rowptr, *col, *etype;       /* [n+1], [E], [E] */
    const int32_t *rowptr_t, *col_t, *etype_t; /* transpose */
    /* in-piece bond graph (edge_index[:, in_piece_edge_idx], src/pl_models/ps_vae_model.py:45-48) */
    const int32_t *in_rowptr, *in_col, *in_etype;
    const int32_t *in_rowptr_t, *in_col_t, *in_etype_t;
    /* selected ordered atom pairs in row-major (graph, m, n) order = order of golden_edge */
    const int32_t *pair_src, *pair_dst, *pair_graph, *pair_gold; /* [S] */
    const int32_t *pair_by_src_ptr, *pair_by_src_idx; /* [n+1], [S] pairs grouped by src atom */
    const int32_t *pair_by_dst_ptr, *pair_by_dst_idx; /* [n+1], [S] */
    const int32_t *pair_graph_off;                    /* [B+1] */
    const int32_t *pieces;       /* [B, L] piece ids incl. leading/trailing <s>, padded with pad_idx */
    /* inverse lists for the deterministic embedding-table gradients */
    const int32_t *atom_rows_ptr, *atom_rows_idx;   /* [num_atom+1], [n] atoms using each table row */
    const int32_t *piece_rows_ptr, *piece_rows_idx; /* [num_piece+1], [n] */
    const int32_t *pos_rows_ptr, *pos_rows_idx;     /* [max_pos+1], [n] */
    const int32_t *seq_rows_ptr, *seq_rows_idx;     /* [num_piece+1], [B*(L-1)] GRU input tokens per table row */
    const float *props;          /* [B, n_props] the selected property targets */
    const float *eps;            /* [B, Z] the N(0,1) draw of rsample */
    const float *denoms;         /* [4] loss denominators: non-pad target tokens, S, B, B*n_props.
                                    Data-parallel callers all-reduce these so every rank
                                    divides by the global-batch counts. */
    /* Decoder-only mode = VAEPieceDecoder.forward called on its own (src/modules/vae_piece_decoder.py:71-126):
     * when `conds` is set the step takes the graph conditions [B, G] from it instead of running the encoder
     * (the in_* arrays then describe the edges the caller passed), skips the property head (loss 0) and
     * needs none of the encoder / predictor fields of psvae_params; backward writes dL/dconds into d_conds
     * and adds d_z (dL/dz from the caller's use of the returned z, may be NULL) to the latent gradient. */
    const float *conds;          /* [B, G] or NULL */
    float *d_conds;              /* [B, G], backward output in decoder-only mode */
    const float *d_z;            /* [B, Z] or NULL */
    /* Optional status words (device, 8 x int32, caller-zeroed; NULL = not reported).  Word 1 counts the recurrences
     * (forward or backward) that were redone on the single-CU kernels because a workgroup of the multi-CU GRU
     * recurrence stopped waiting for its partners (its waits are bounded at 2 s so that a GPU which cannot keep the
     * group resident cannot hang the process; the redo is queued behind every multi-CU launch and does nothing
     * otherwise): the step's outputs are valid either way, a non-zero count says that the step took seconds.
     * Word 0 is reserved for "outputs invalid" and is not set by any kernel of this version.  The library only
     * ever adds to the words, so one buffer passed to every step accumulates over a whole run. */
    int32_t *status;
} psvae_batch;

/* ---- whole training step ---------------------------------------------------
 * psvae_step_forward  = PSVAEModel.forward (src/pl_models/ps_vae_model.py:33-58)
 *   incl. VAEPieceDecoder.forward (src/modules/vae_piece_decoder.py:71-117),
 *   Encoder.embed_node/embed_graph (src/modules/encoder.py:53-69), Predictor.forward
 *   (src/modules/predictor.py:19-21).  losses[0..5] (device) = pred, rec, piece, edge, kl, total,
 *   where total = w[0]*pred + w[1]*rec + w[2]*kl is PSVAEModel.weighted_loss
 *   (ps_vae_model.py:71-74) for the HOST array loss_weights = {1-alpha, alpha, beta}
 *   (NULL: total = 0).
 * psvae_step_backward = what autograd derives from it; loss_grads[0..5] are the
 *   upstream gradients of those six outputs as DEVICE floats; pass the same
 *   loss_weights.  Gradients are written (not accumulated) into `grads`.
 *   Independent branches of the step run on two internal side streams that fork
 *   from and join back into `stream` inside each call (option streams = 0 disables).  The two unused parameters of the
 *   decoder-side encoder (its `linear`) have no field and get no gradient.
 * The workspace holds the saved activations: pass the same buffer, dims and
 * batch to forward and backward. */
size_t psvae_workspace_bytes(const psvae_dims* dims, const psvae_batch* batch_sizes);
int psvae_step_forward(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch,
                       void* workspace, size_t workspace_bytes, const float* loss_weights, float* losses,
                       void* stream);
int psvae_step_backward(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch,
                        void* workspace, size_t workspace_bytes, const float* loss_weights,
                        const float* loss_grads, const psvae_params* grads, void* stream);
/* psvae_step_backward that also tells a data-parallel caller when gradients may leave (no counterpart in the
 * reference, which has no working multi-GPU path): early_event_1 / early_event_2 (hipEvent_t, either may be
 * NULL) are recorded on the internal streams once the gradients of, respectively, the piece branch (to_vocab,
 * GRU, latent_to_rnn_hidden) and the property head + link scorer + W_mean/W_log_var + encoder.linear are
 * final, i.e. while the GINE stacks' backward is still running.  A caller that keeps the gradient tensors in
 * that order in one buffer can start the all-reduce of that range on another stream after waiting for both
 * events, and of the rest after the call's work on `stream`. */
int psvae_step_backward_signal(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch,
                               void* workspace, size_t workspace_bytes, const float* loss_weights,
                               const float* loss_grads, const psvae_params* grads, void* stream,
                               void* early_event_1, void* early_event_2);
/* One optimiser step in ONE call: psvae_step_forward, psvae_step_backward_signal with `loss_grads` (normally the
 * constant [0,0,0,0,0,1]: the gradient of the weighted total) and, when `opt` is given, psvae_clip_adam over the flat
 * buffers it names.  This is the body of Lightning's training loop for this model (src/pl_models/ps_vae_model.py:76-88
 * `training_step` -> loss.backward() -> src/train.py:47 gradient clipping -> ps_vae_model.py:123-127 Adam) with no
 * host code between the three parts.  Every result is, bit for bit, that of the three calls; the order in which the
 * work is queued is not: while the recurrence runs on the multi-CU kernels the call queues forward and backward as one
 * sequence (no join of the side streams between them, the six loss values written late in backward), which the
 * separate entry points cannot do because each must leave its results complete on the caller's stream.  A
 * data-parallel caller passes opt = NULL, reduces the gradients and calls psvae_clip_adam itself.  `grads` must
 * describe the same memory as opt->grads (tensor by tensor). */
typedef struct psvae_adam {
    float *params, *grads, *exp_avg, *exp_avg_sq;    /* flat fp32 buffers of `count` elements */
    int64_t count;
    float* partials;                                  /* psvae_adam_partials(count) floats */
    float max_norm;
    double lr, beta1, beta2, eps, bias1, bias2;       /* as psvae_clip_adam takes them */
    float* grad_norm_out;                             /* device, may be NULL */
} psvae_adam;
/* `eps_draw` (or NULL): batch->eps is an unfilled [B, Z] buffer and the call itself draws rsample's epsilon into it -
 * psvae_normal_rows(seed, draw, first_row, B, Z, batch->eps) - on one of its side streams beside the first kernels,
 * instead of the caller launching that draw in front of the step. */
typedef struct psvae_eps_draw {
    uint64_t seed, draw;     /* as psvae_normal_rows takes them */
    int64_t first_row;       /* index of the batch's first graph in the global batch */
} psvae_eps_draw;
int psvae_train_step(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch, void* workspace,
                     size_t workspace_bytes, const float* loss_weights, float* losses, const float* loss_grads,
                     const psvae_params* grads, const psvae_adam* opt, const psvae_eps_draw* eps_draw,
                     void* early_event_1, void* early_event_2, void* stream);
/* Creates the current device's two internal side streams now rather than inside the first step call, and checks
 * (with two 100 us probe kernels per pair, timed from the host) that they and `stream` -- the stream the step
 * calls will be given, NULL for the default stream -- run side by side; a side stream that does not is replaced
 * by a newly created one, up to eight times.  No counterpart in the reference.  HIP deals the streams of a process
 * onto a small number of hardware queues in creation order and streams that share a queue serialise, so a process
 * that also creates communicator (RCCL) or other library streams should call this once per device; otherwise
 * optional (the first step call creates the side streams without the check).  Synchronises `stream`.  Returns 0
 * also when the side streams are disabled (option streams = 0) or already exist.  Also looks up the device address
 * of the static scratch array of psvae_linear_backward_weight_noalloc on this device, so that a caller who must not
 * make runtime calls other than launches (HIP graph capture) can use that entry point from its first call on. */
int psvae_streams_init(void* stream);
/* Offsets (in floats) of named activations inside the workspace, for tests and
 * the module-level API.  Returns -1 for an unknown name. */
int64_t psvae_workspace_offset(const psvae_dims* dims, const psvae_batch* batch_sizes, const char* name);

/* Encoder-only latent encoding = PSVAEModel.get_z (ps_vae_model.py:160-170)
 * without the sampling: writes mean and log-variance [B, Z] each. */
int psvae_encode(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch,
                 void* workspace, size_t workspace_bytes, float* z_mean, float* z_log_var, void* stream);

/* ---- batched front half of generation: the two GPU-shaped stages of
 * VAEPieceDecoder.inference (src/modules/vae_piece_decoder.py:128-241) for many latents at once.
 * Turning piece ids into atoms and assembling the molecule (RDKit) stays with the caller.
 *
 * psvae_sample_pieces (:134-158): for each of the Bz latents z [Bz, latent] sample the piece sequence from the
 *   GRU: start token <s>, h0 = latent_to_rnn_hidden(z); every step masks <pad> (and <s> while the
 *   sequence is empty), draws from softmax(logits / temperature), adds piece_atoms[id] (DEVICE int32
 *   [num_piece]: atoms of each piece, 0 for <pad> and <s>) to the atom count, and stops on <s>, when the
 *   count exceeds max_atom_num or at max_pos pieces; the last token drawn is dropped in all three cases.
 *   The draw of sequence b at step t is the class whose cumulative-probability interval (classes in
 *   index order) holds uniforms[b*max_pos + t] in [0,1) (DEVICE float [Bz, max_pos]): the caller owns
 *   the random stream (torch.multinomial's own stream cannot be reproduced).
 *   piece_ids [Bz, max_pos] int32 receives the tokens (entries at and after lengths[b] are
 *   unspecified), lengths [Bz] the kept count.  check_every > 0: every that many steps the number of
 *   running sequences is copied to the host (this SYNCHRONISES the stream) and the loop ends early when
 *   it is 0; 0 = always max_pos steps, no synchronisation.
 * psvae_score_links (:203-238): `batch` holds the fragments (atoms, piece/position ids, in-piece bonds as
 *   in_rowptr/in_col/in_etype) and the candidate pairs; runs embed_atom, the decoder-side GINE stack and the
 *   scorer, then softmax over the 4 bond classes.  z [B, latent].  Outputs per pair (any may be NULL):
 *   probs [S,4], confidence [S] = max probability, bond_type [S] = its class (lowest index on ties).
 *   Workspace: psvae_workspace_bytes(dims, batch). */
size_t psvae_sample_pieces_workspace_bytes(const psvae_dims* dims, int Bz);
int psvae_sample_pieces(const psvae_dims* dims, const psvae_params* params, const float* z, int Bz,
                        const int32_t* piece_atoms, int max_atom_num, float temperature, const float* uniforms,
                        int32_t* piece_ids, int32_t* lengths, int check_every, void* workspace,
                        size_t workspace_bytes, void* stream);
int psvae_score_links(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch, const float* z,
                      void* workspace, size_t workspace_bytes, float* probs, float* confidence, int32_t* bond_type,
                      void* stream);

/* ---- rsample's epsilon (src/modules/vae_piece_decoder.py:61: torch.randn_like) as a counter-based draw:
 * out[r, c] (rows x width, contiguous) = the N(0,1) number that belongs to (seed, draw, first_row + r, c).
 * Philox-4x32-10 with counter (c/4, row low, row high, draw low) and key (seed low, seed high ^ draw high); the
 * four outputs become columns 4(c/4)..+3 by two Box-Muller transforms over 24-bit uniforms ((x>>8)+0.5)*2^-24.
 * `first_row` is the GLOBAL index of the first graph, `draw` counts the forward passes: a data-parallel run then
 * samples exactly what one process samples for the undivided batch, whatever the number of GPUs.  (The
 * reference's draw comes from torch's global generator and depends on how graphs are dealt to processes.) */
int psvae_normal_rows(uint64_t seed, uint64_t draw, int64_t first_row, int rows, int width, float* out, void* stream);

/* ---- optimiser: clip_grad_norm_(10) + Adam (src/train.py:47; ps_vae_model.py:123-127)
 * over flat fp32 buffers of `count` elements.  `partials` is scratch of
 * psvae_adam_partials(count) floats; grad_norm_out (device, may be NULL) gets
 * the pre-clip L2 norm.  bias1/bias2 = 1-beta1^t, 1-beta2^t.  The hyper-parameters are doubles, as torch holds
 * them: 1 - beta, lr / bias1 and sqrt(bias2) are formed in double and rounded to fp32 once, like torch.optim.Adam's
 * scalars (1.f - 0.999f is 1.3e-5 away from 0.001, enough to show in a loss curve after one step). */
size_t psvae_adam_partials(int64_t count);
int psvae_clip_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t count,
                    float* partials, float max_norm, double lr, double beta1, double beta2, double adam_eps,
                    double bias1, double bias2, float* grad_norm_out, void* stream);

/* ---- opt-in launch timer (benchmarking aid; no counterpart in the reference) ----
 * While enabled, every kernel launch of the library is bracketed by hipEvents on
 * its launch stream and tagged with a kernel class and its algorithmic work
 * (FLOPs for the gemm_* classes and the GRU, bytes for the others).
 * psvae_profile_collect waits for the recorded launches and ADDS duration (ms),
 * work and launch count per class into three arrays of
 * psvae_profile_num_classes() entries, then clears the log.  Off by default;
 * not thread-safe. */
int psvae_profile_enable(int on);
int psvae_profile_num_classes(void);
const char* psvae_profile_class_name(int kernel_class);
int psvae_profile_collect(double* ms, double* work, int64_t* launches);
/* Median time (ms) a hipEvent pair measures around an EMPTY kernel on `stream` (64 samples; synchronises the
 * stream): what the bracketing itself adds to every duration above.  Negative on failure. */
double psvae_profile_event_overhead_ms(void* stream);

/* ---- single operators (same kernels as the step; used by the module-level API
 * and the per-layer parity tests) -------------------------------------------- */
/* nn.Linear (+ReLU) forward: y[M,N] = act(x[M,K] w[N,K]^T + b). src/modules/common_nn.py:15-23 */
int psvae_linear_forward(const float* x, int ldx, const float* w, const float* b, float* y, int ldy,
                         int M, int N, int K, int relu, void* scratch, size_t scratch_bytes, void* stream);
/* dx[M,K] = (dy[M,N] w[N,K]) * (mask > 0 if mask) ; accumulate adds into dx */
int psvae_linear_backward_input(const float* dy, int lddy, const float* w, float* dx, int lddx,
                                int M, int N, int K, const float* mask, int ldmask, int accumulate, void* stream);
/* dw[N,K] = sum over nseg segments of dy_s^T x_s, db[N] = column sums of dy (db may be NULL) */
int psvae_linear_backward_weight(const float* dy, int lddy, int64_t seg_stride_dy, const float* x, int ldx,
                                 int64_t seg_stride_x, int nseg, float* dw, float* db, int M, int N, int K,
                                 void* scratch, size_t scratch_bytes, void* stream);
size_t psvae_linear_scratch_bytes(int M, int N, int K, int nseg);
/* psvae_linear_backward_weight without a scratch argument and without any allocation.  No counterpart in the
 * reference: it is for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  The slab partial sums of the split reduction go to a static device
 * array of psvae_noalloc_scratch_bytes() = 32 MiB that every GPU the library runs on holds until the process
 * ends; a problem whose planned split needs more (nseg * splits * N * (K+1) floats, K+1 rounded up to 16 where the
 * 160-tile kernel takes the product) is refused with
 * PSVAE_E_STATIC_LIMIT.  Same kernels and same results, bit for bit, as psvae_linear_backward_weight given a
 * scratch buffer of psvae_linear_scratch_bytes.  All calls on one GPU share the array: calls that overlap there
 * (two streams, two host threads) race; calls on one stream do not. */
size_t psvae_noalloc_scratch_bytes(void);
int psvae_linear_backward_weight_noalloc(const float* dy, int lddy, int64_t seg_stride_dy, const float* x, int ldx,
                                         int64_t seg_stride_x, int nseg, float* dw, float* db, int M, int N, int K,
                                         void* stream);
/* GINEConv message + aggregation (torch_geometric GINEConv as used at src/modules/encoder.py:48,58):
 * out[i] = (1+eps) x[i] + sum_{e in row i} relu(x[col[e]] + edge_w[:,etype[e]] + edge_b) */
int psvae_gine_aggregate_forward(const float* x, int ldx, const int32_t* rowptr, const int32_t* col,
                                 const int32_t* etype, const float* edge_w, const float* edge_b, float one_plus_eps,
                                 float* out, int ldo, int n, int H, void* stream);
/* One whole GINE application (src/modules/encoder.py:57-59: `x = self.conv(x, edge_index, edge_attr)` with the
 * two-layer MLP of src/modules/common_nn.py:8-23) in ONE launch:
 *   agg = the aggregation above;  hid = relu(agg w1^T + b1);  y = hid w2^T + b2      (w1, w2 [H,H] row-major)
 * The aggregated rows are built straight into the first product's operand image in LDS and the hidden rows stay
 * there for the second product: neither is read back from memory.  agg and hid ([n,H] each, both or neither)
 * receive copies for a backward pass; NULL = inference.  H a multiple of 4, 16 <= H <= 512, ldx % 4 == 0.
 * `scratch` (psvae_gine_layer_scratch_bytes(H) bytes, 16-byte aligned) receives the two weight matrices re-laid
 * for the matrix cores' operand fragments (the whole-step entry points do that once per step).
 * Arithmetic: fp32 in, fp32 out, fp32 accumulation.  The two products run on the bf16 matrix instruction over
 * operands cut exactly into three bf16 pieces each (x = h + m + l), six piece products per product; the result is
 * at least as close to the exact one as a chain of fp32 multiply-adds (option split_bf16 = 0 selects the fp32
 * matrix instruction instead). */
size_t psvae_gine_layer_scratch_bytes(int H);
int psvae_gine_layer_forward(const float* x, int ldx, const int32_t* rowptr, const int32_t* col, const int32_t* etype,
                             const float* edge_w, const float* edge_b, float one_plus_eps, const float* w1,
                             const float* b1, const float* w2, const float* b2, float* agg, float* hid, float* y,
                             int ldy, int n, int H, void* scratch, size_t scratch_bytes, void* stream);
/* Backward of the two Linear layers of that GINE application (what autograd derives from the MLP of
 * src/modules/common_nn.py:8-23 as used at src/modules/encoder.py:48,58) in ONE launch, given g = dL/dy [n,H] and
 * the saved hid [n,H]:   d_hid = (g w2) * [hid > 0];   d_agg = d_hid w1      (all matrices contiguous, pitch H).
 * d_hid is the operand of the first layer's weight gradient (psvae_linear_backward_weight(d_hid, agg)), d_agg the
 * input of psvae_gine_aggregate_backward.  Same sizes, scratch and arithmetic as the forward entry point. */
int psvae_gine_layer_backward(const float* g, const float* hid, const float* w1, const float* w2, float* d_hid,
                              float* d_agg, int n, int H, void* scratch, size_t scratch_bytes, void* stream);
/* The link scorer's forward pass (src/modules/vae_piece_decoder.py:97-117) in ONE launch: for each of the S selected
 * pairs, pair_in = [emb[src] | emb[dst] | z[graph]] (P = 2H + Z columns), h1 = relu(pair_in w0^T + b0),
 * h2 = relu(h1 w1^T + b1), h3 = h2 w2^T + b2 (the MLP(P, P/2, P, ReLU, 3) of :36-45), logits = h3 w3^T + b3 (4 bond
 * classes), lse = logsumexp(logits), row_loss = lse - logits[gold] (the rows of the edge cross-entropy, :53,117).
 * The pair rows are gathered straight into LDS and the activations stay there between the layers; pair_in [S,P],
 * h1/h2 [S,P/2], h3 [S,P], logits [S,4], row_loss/lse [S] receive the copies a backward pass reads.  Supported for
 * 320 < P/2 <= 352 (H = 300, Z = 56 gives 328; up to 384 with option split_bf16 = 0: what fits the LDS), H % 4 ==
 * Z % 4 == lde % 4 == 0 and P % 8 == 0; scratch: psvae_scorer_scratch_bytes (0 = unsupported size) bytes, 16-byte
 * aligned, for the re-laid weights.  The whole-step entry points run other sizes as gather + GEMM chain. */
size_t psvae_scorer_scratch_bytes(int H, int Z);
int psvae_scorer_forward(const float* emb, int lde, const float* z, const int32_t* pair_src, const int32_t* pair_dst,
                         const int32_t* pair_graph, const int32_t* pair_gold, const float* w0, const float* b0,
                         const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                         const float* b3, float* pair_in, float* h1, float* h2, float* h3, float* logits,
                         float* row_loss, float* lse, int S, int H, int Z, void* scratch, size_t scratch_bytes,
                         void* stream);
/* Backward of the scorer's three-layer MLP (src/modules/vae_piece_decoder.py:36-45 applied at :110) in ONE launch,
 * given d_h3 = dL/dh3 [S,P] and the saved h2, h1 [S,P/2]:
 *   d_h2 = (d_h3 w2) * [h2 > 0];   d_h1 = (d_h2 w1) * [h1 > 0];   d_pair_in = d_h1 w0        (contiguous rows)
 * d_h2 / d_h1 are the operands of the weight gradients, d_pair_in [S,P] is what gets scattered back to the atoms
 * and the latent.  Sizes, scratch and arithmetic as psvae_scorer_forward. */
int psvae_scorer_backward(const float* d_h3, const float* h2, const float* h1, const float* w0, const float* w1,
                          const float* w2, float* d_h2, float* d_h1, float* d_pair_in, int S, int H, int Z,
                          void* scratch, size_t scratch_bytes, void* stream);
/* dx[j] = (1+eps) da[j] + sum_{e in rowT j} da[colT[e]] * (x[j] + edge(etypeT[e]) > 0)  (+ pooled-gradient broadcast).
 * The same products summed per bond type are the gradient of edge_trans: when d_edge_w [H,4] is given the call
 * also writes it and d_edge_b [H], using `partial`, scratch of psvae_gine_edge_grad_scratch(n, H) floats. */
size_t psvae_gine_edge_grad_scratch(int n, int H);
int psvae_gine_aggregate_backward(const float* da, int ldda, const float* x, int ldx, const int32_t* rowptr_t,
                                  const int32_t* col_t, const int32_t* etype_t, const float* edge_w,
                                  const float* edge_b, float one_plus_eps, const float* dpool, int ldp,
                                  const int32_t* node_graph, float* dx, int lddx, float* d_edge_w, float* d_edge_b,
                                  float* partial, int n, int H, void* stream);
/* graph sum-pool: pooled[b] = sum of rows graph_off[b]..graph_off[b+1]. src/modules/encoder.py:64-67 */
int psvae_segment_pool_forward(const float* x, int ldx, const int32_t* graph_off, float* pooled, int ldp,
                               int B, int width, void* stream);
/* nn.GRU(batch_first, 1 layer) over T steps given gi = x W_ih^T + b_ih. src/modules/vae_piece_decoder.py:29,89
 * `exchange`: device scratch of psvae_gru_exchange_bytes(B,T,R) bytes, 16-byte aligned, through which the
 * workgroups that share a sequence tile hand each other the hidden state every step (multi-CU recurrence,
 * 16 <= R <= 256).  NULL selects the single-CU kernels (one workgroup per 16 sequences), which the library also takes
 * by itself for more than 768 sequences, where the multi-CU kernels' waiting workgroups cost the kernels of other
 * streams more than the shorter recurrence gains (then the buffer is left untouched).  Two calls that use
 * the same exchange buffer must not overlap. */
size_t psvae_gru_exchange_bytes(int B, int T, int R);
int psvae_gru_forward(const float* gi, const float* h0, const float* whh, const float* bhh, float* hseq,
                      float* hprev, float* gates, int B, int T, int R, void* exchange, size_t exchange_bytes,
                      void* stream);
int psvae_gru_backward(const float* dhseq, const float* hprev, const float* gates, const float* whh,
                       float* dgi, float* dgh, float* dh0, int B, int T, int R, void* exchange,
                       size_t exchange_bytes, void* stream);
/* psvae_gru_forward on the multi-CU kernel with time stamps, for profiling (no counterpart in the reference): thread 0
 * of the first workgroup writes the GPU's 100 MHz wall clock to stamps[t][0..3] (device memory, T x 4 uint64) at the
 * top of time step t, when the operands of h_t have arrived from the other workgroups, after the K-split partial sums
 * went through LDS, and when the gates are done and h_{t+1} is published.  Outputs as psvae_gru_forward.  Only the
 * kernel form taken for R <= 208 and one sequence tile per workgroup (PSVAE_E_SHAPE otherwise). */
int psvae_gru_forward_stamped(const float* gi, const float* h0, const float* whh, const float* bhh, float* hseq,
                              float* hprev, float* gates, int B, int T, int R, void* exchange, size_t exchange_bytes,
                              uint64_t* stamps, void* stream);
/* CrossEntropyLoss rows (ignore_index): row_loss, lse per row. src/modules/vae_piece_decoder.py:53-54 */
int psvae_softmax_ce_forward(const float* logits, int ld, const int32_t* target, int target_stride,
                             int target_offset, int rows_per_group, int rows, int classes, int ignore_index,
                             float* row_loss, float* lse, void* stream);

/* ---- host-side batching (bit-exact integer work, no GPU involved) ----------
 * psvae_collate_pairs = the edge / pair enumeration loop of process_step3
 * (src/data/bpe_dataset.py:110-132).  `ad` and `cross` are the molecules' dense
 * bond-type and cross-piece matrices (n_i*n_i bytes each, concatenated).
 * `mt_state` is CPython's `random.getstate()[1]` (624 words + index): the
 * function consumes draws exactly where the reference calls random() and
 * leaves the advanced state in place for `random.setstate`.
 * Outputs use the reference's padded numbering (atom a of graph i = a + N*i). */
int psvae_collate_pairs(int B, int N, const int32_t* n_atoms, const int8_t* ad, const int8_t* cross,
                        double neg_rate, uint32_t* mt_state,
                        int32_t* e_src, int32_t* e_dst, int32_t* e_type, int64_t e_cap, int64_t* n_edges,
                        int32_t* in_piece_idx, int64_t* n_in_piece,
                        int32_t* pair_graph, int32_t* pair_m, int32_t* pair_n, int32_t* pair_gold,
                        int64_t pair_cap, int64_t* n_pairs);
/* Stage 2 of the dataset (process_step2, src/data/bpe_dataset.py:56-84) for B molecules `idx` of a data set
 * flattened beforehand: atoms of molecule m are atom_off[m]..atom_off[m+1] of x_all / piece_of_atom /
 * pos_of_atom / group_of_atom (group = index into the record's `groups`, -1 for none), its bonds
 * bond_off[m]..bond_off[m+1] of bond_a/bond_b/bond_t (local atom numbers, list order: a later duplicate
 * overwrites), its piece sequence piece_off[m]..piece_off[m+1] of pieces_all.  Writes what stage 3 consumes:
 * n_atoms [B], the padded x / x_pieces / x_pos [B,N] and pieces [B,L], and the concatenated dense
 * bond-type and cross-piece matrices (sum of n_i*n_i bytes each; matrix_bytes = their capacity). */
int psvae_gather_molecules(int B, const int64_t* idx, const int64_t* atom_off, const int32_t* x_all,
                           const int32_t* piece_of_atom, const int32_t* pos_of_atom, const int32_t* group_of_atom,
                           const int64_t* bond_off, const int32_t* bond_a, const int32_t* bond_b, const int8_t* bond_t,
                           const int64_t* piece_off, const int32_t* pieces_all, int N, int L, int atom_pad,
                           int piece_pad, int32_t* n_atoms, int32_t* x, int32_t* x_pieces, int32_t* x_pos,
                           int32_t* pieces, int8_t* ad, int8_t* cross, int64_t matrix_bytes);
/* Build the packed index blob of psvae_batch from padded-numbering arrays.
 * `blob` receives int32 words; `offsets[PSVAE_BLOB_FIELDS]` the start of each
 * field inside it (field order = PSVAE_BLOB_* below).  Returns words used via
 * *blob_words; PSVAE_E_CAPACITY when blob_cap is too small. */
enum {
    PSVAE_BLOB_ATOM_ID = 0, PSVAE_BLOB_PIECE_ID, PSVAE_BLOB_POS_ID, PSVAE_BLOB_NODE_GRAPH, PSVAE_BLOB_GRAPH_OFF,
    PSVAE_BLOB_ROWPTR, PSVAE_BLOB_COL, PSVAE_BLOB_ETYPE, PSVAE_BLOB_ROWPTR_T, PSVAE_BLOB_COL_T, PSVAE_BLOB_ETYPE_T,
    PSVAE_BLOB_IN_ROWPTR, PSVAE_BLOB_IN_COL, PSVAE_BLOB_IN_ETYPE, PSVAE_BLOB_IN_ROWPTR_T, PSVAE_BLOB_IN_COL_T,
    PSVAE_BLOB_IN_ETYPE_T, PSVAE_BLOB_PAIR_SRC, PSVAE_BLOB_PAIR_DST, PSVAE_BLOB_PAIR_GRAPH, PSVAE_BLOB_PAIR_GOLD,
    PSVAE_BLOB_PAIR_BY_SRC_PTR, PSVAE_BLOB_PAIR_BY_SRC_IDX, PSVAE_BLOB_PAIR_BY_DST_PTR, PSVAE_BLOB_PAIR_BY_DST_IDX,
    PSVAE_BLOB_PAIR_GRAPH_OFF, PSVAE_BLOB_PIECES, PSVAE_BLOB_ATOM_ROWS_PTR, PSVAE_BLOB_ATOM_ROWS_IDX,
    PSVAE_BLOB_PIECE_ROWS_PTR, PSVAE_BLOB_PIECE_ROWS_IDX, PSVAE_BLOB_POS_ROWS_PTR, PSVAE_BLOB_POS_ROWS_IDX,
    PSVAE_BLOB_SEQ_ROWS_PTR, PSVAE_BLOB_SEQ_ROWS_IDX, PSVAE_BLOB_FIELDS
};
int64_t psvae_pack_bound(int B, int N, int64_t n_edges, int64_t n_pairs, int L, int num_atom, int num_piece, int max_pos);
int psvae_pack_batch(int B, int N, int L, const int32_t* n_atoms,
                     const int32_t* x, const int32_t* x_pieces, const int32_t* x_pos, /* [B,N] padded */
                     const int32_t* e_src, const int32_t* e_dst, const int32_t* e_type, int64_t n_edges,
                     const int32_t* in_piece_idx, int64_t n_in_piece,
                     const int32_t* pair_graph, const int32_t* pair_m, const int32_t* pair_n,
                     const int32_t* pair_gold, int64_t n_pairs,
                     const int32_t* pieces, /* [B,L] */
                     int num_atom, int num_piece, int max_pos, int pad_idx,
                     int32_t* blob, int64_t blob_cap, int64_t* offsets, int64_t* blob_words,
                     int32_t* n_nodes_out, int32_t* n_target_tokens_out);


/* ---- batches built on the GPU ---------------------------------------------------
 * Stages 2 and 3 of the reference's dataset (process_step2 / process_step3,
 * src/data/bpe_dataset.py:56-155) for a data set that lies flattened in DEVICE memory: the result is the same
 * index blob psvae_pack_batch writes (same fields, same order inside every list), built by kernels on `stream`.
 * The store holds what is static per molecule m (all int32, local atom numbers 0..n_m-1):
 *   atoms  atom_off[m]..atom_off[m+1] of atom_id / piece_of_atom / pos_of_atom / group_of_atom (what
 *          process_step2 leaves in x / x_pieces / x_pos; group = index into the record's `groups`, -1 for none);
 *   bonds  as a CSR: row bounds rowptr_loc[atom_off[m] + m + a] (n_m + 1 of them), entries
 *          edge_off[m]..edge_off[m+1] of nbr (neighbour, ascending inside a row) / nbr_type.  A molecule's
 *          adjacency matrix is symmetric (bpe_dataset.py:62-66), so this one CSR is both the by-target and the
 *          by-source CSR of the batch; in_* is the same for the bonds inside a piece (edge_select == 0);
 *   pieces piece_off[m]..piece_off[m+1] of `pieces` (with the leading and trailing <s>);
 *   props  [n_mols, n_props_all].
 * Negative pairs (bpe_dataset.py:129: every ordered cross-piece pair without a bond is kept with probability
 * 0.022) are drawn from a counter-based generator keyed by (seed, epoch, molecule index, pair), NOT from the
 * reference's MT19937 stream: a molecule's pairs then depend neither on the batch it lands in nor on the number
 * of GPUs.  neg_rate >= 1 keeps every cross-piece pair; the host path psvae_collate_pairs stays the bit-exact one.
 * The caller supplies the sizes that follow from the store (n atoms, E directed bonds, E_in in-piece bonds,
 * L longest piece sequence: sums / maximum over mol_idx) and S_cap, an upper bound of the selected pairs that
 * fixes the blob layout (sum of n_m^2 always suffices); psvae_collate_device_layout gives the field offsets
 * (words) and the blob size for them.  counts_out (device, 8 words) receives S, the number of non-pad target
 * tokens, n, E, E_in, L as the kernels found them, and in word 6 a bit set of inconsistencies (0 = none);
 * read S from it before calling the step.  Molecules of more than 128 atoms are not supported here. */
typedef struct psvae_mol_store {
    int64_t n_mols;
    const int32_t *atom_off, *atom_id, *piece_of_atom, *pos_of_atom, *group_of_atom;
    const int32_t *edge_off, *rowptr_loc, *nbr, *nbr_type;
    const int32_t *in_edge_off, *in_rowptr_loc, *in_nbr, *in_nbr_type;
    const int32_t *piece_off, *pieces;
    const float *props;
    int32_t n_props_all;
} psvae_mol_store;
int64_t psvae_collate_device_layout(int B, int n, int64_t E, int64_t E_in, int64_t S_cap, int L, int num_atom,
                                    int num_piece, int max_pos, int64_t* offsets /* [PSVAE_BLOB_FIELDS] */);
size_t psvae_collate_device_scratch_bytes(int B, int n, int L, int num_atom, int num_piece, int max_pos);
int psvae_collate_device(const psvae_mol_store* store, const int32_t* mol_idx /* device [B] */, int B, int n, int64_t E,
                         int64_t E_in, int L, int64_t S_cap, double neg_rate, uint64_t seed, uint64_t epoch,
                         int num_atom, int num_piece, int max_pos, int pad_idx,
                         const int32_t* prop_cols /* host */, int n_prop_cols,
                         int32_t* blob, int64_t blob_words, float* props_out /* [B, n_prop_cols] */,
                         float* denoms_out /* [4] */, int32_t* counts_out /* [8] */, void* scratch, size_t scratch_bytes,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PSVAE_H */
