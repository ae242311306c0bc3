// Internal declarations shared by the HIP translation units of libpsvae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/psvae.h"

#define PSVAE_LAUNCH_CHECK()                                   \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return (int)e__;                \
    } while (0)
#define PSVAE_TRY(expr)                                        \
    do {                                                       \
        int rc__ = (expr);                                     \
        if (rc__ != 0) return rc__;                            \
    } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// ---------------------------------------------------------------- kernel-selection options (capi.hip)
// Every choice between two forms of a kernel that the library makes by size can be forced through
// psvae_option_set(name, value) (or, at load time, the environment variable PSVAE_<NAME>); -1 = the library's own
// choice.  Read on every call, so a test can flip them inside one process.
enum Option {
    OPT_STREAMS = 0,     // 0: everything on the caller's stream
    OPT_FUSED_LAYER,     // 0: a GINE application as aggregate + two GEMM launches
    OPT_FUSED_BWD,       // 0 / 1: the GINE MLPs' backward as two GEMMs / one launch
    OPT_FUSED_SCORER,    // 0: never one launch; 1: forward and backward always; 2: forward always
    OPT_SPLIT_BF16,      // 0: fused kernels on the fp32 matrix instruction
    OPT_SPLIT_ROWS,      // 32 / 64 nodes per workgroup of the cut-operand GINE kernels
    OPT_SPLIT_GEMM,      // 0 / 1: 64-tile weight gradients on the fp32 / the cut-operand kernel
    OPT_WIDE_GEMM,       // 0 / 1: 160-tile weight-gradient kernel never / wherever the shape allows
    OPT_GRU_MULTI_CU,    // 0 / 1: single-CU / multi-CU recurrence
    OPT_GRU_TILES,       // 1 / 2 / 4 sequence tiles per workgroup of the multi-CU recurrence
    OPT_GRU_SPIN_TICKS,  // bound of a multi-CU recurrence wait in 10 ns ticks; 0: give up at once (tests the redo)
    OPT_COUNT
};
int opt(int which);

// ---------------------------------------------------------------- launch timer (profiler.hip)
enum KernelClass {
    KC_GEMM_FWD = 0, KC_GEMM_DX, KC_GEMM_DW, KC_GEMM_REDUCE, KC_AGG_FWD, KC_AGG_BWD, KC_ETAB, KC_EMBED, KC_TABLE_GRAD,
    KC_POOL, KC_PAIR_GATHER, KC_PAIR_SCATTER, KC_GRU_FWD, KC_GRU_BWD, KC_CE, KC_HEAD, KC_ADAM, KC_GINE_LAYER, KC_SCORER, KC_GINE_LAYER_BWD, KC_COUNT
};
struct ProfScope {
    ProfScope(hipStream_t s, int kc, double work);
    ~ProfScope();
    hipStream_t s_; int idx_;
};

// ---------------------------------------------------------------- GEMM family
// C[i][j] (+)= epilogue( sum_r A(i,r) * B(j,r) ), fp32 in / fp32 accumulate on
// v_mfma_f32_32x32x2_f32.  A(i,r) = A[i*lda + r] or, when a_t, A[r*lda + i]
// (same for B).  grid.z enumerates nseg segments x splits reduction slices.
struct GemmDesc {
    const float* A = nullptr; const float* B = nullptr; float* C = nullptr;
    int M = 0, N = 0, R = 0;
    int lda = 0, ldb = 0, ldc = 0;
    bool a_t = false, b_t = false;
    const float* bias = nullptr;   // per column j
    int relu = 0;
    const float* mask = nullptr; int ldmask = 0;  // C = mask > 0 ? C : 0
    int accumulate = 0;            // C += result
    int ones_col = -1;             // column j of B that reads as 1.0 (bias-gradient trick); counts inside N
    float* c_extra = nullptr;      // destination of column ones_col (length M)
    // segments (grouped or summed)
    int nseg = 1;
    int64_t seg_a = 0, seg_b = 0, seg_c = 0, seg_bias = 0, seg_mask = 0;
    bool sum_segments = false;     // true: all segments add into one C (weight gradients)
    // scratch for split reductions
    float* scratch = nullptr; size_t scratch_floats = 0;
    // `slabs_out` (plain products only: no mask, ReLU, segments): when the reduction is split, the slab sum is left to
    // the caller's next kernel - the slabs stay in `scratch`, bias not added, and *slabs_out says where; when it is not
    // split, C is written as usual and slabs_out->nz = 0.
    struct GemmSlabs* slabs_out = nullptr;
};
struct GemmSlabs { const float* slabs; int nz; long long stride; int ld; };
int gemm_f32(hipStream_t s, const GemmDesc& d);
int noalloc_scratch_init();     // resolves the static scratch array of the no-allocation entry point on the current device
size_t gemm_scratch_floats(int M, int N, int R, int nseg, bool sum_segments);

// Linear-layer helpers on top of gemm_f32
int linear_fwd(hipStream_t s, const float* x, int ldx, const float* w, const float* b, float* y, int ldy,
               int M, int N, int K, int relu, float* scratch, size_t scratch_floats);
int linear_bwd_x(hipStream_t s, const float* dy, int lddy, const float* w, float* dx, int lddx,
                 int M, int N, int K, const float* mask, int ldmask, int accumulate);
int linear_bwd_w(hipStream_t s, const float* dy, int lddy, int64_t seg_dy, const float* x, int ldx, int64_t seg_x,
                 int nseg, float* dw, float* db, int M, int N, int K, float* scratch, size_t scratch_floats);

// ------------------------------------------------------------ graph kernels
struct AggGroup {  // one GINE stack (encoder or decoder-side encoder)
    const float* x; float* out;           // layer input / aggregated output
    const int32_t *rowptr, *col, *etype;
    const float *edge_w, *edge_b;
    float ope;                            // 1 + eps
    int E;                                // edge count (work accounting only)
};
int gine_agg_fwd(hipStream_t s, const AggGroup* g, int ngroups, int ldx, int ldo, int n, int H);
struct AggBwdGroup {
    const float* da;        // may be null: only the pooled-gradient broadcast is written
    const float* x;         // the layer's input (for the ReLU mask)
    float* dx;
    const int32_t *rowptr_t, *col_t, *etype_t;
    const float *edge_w, *edge_b;
    float ope;
    const float* dpool;     // [B, ldp] slice start, or null
    int E;                  // edge count (work accounting only)
    float* partial;         // null, or [gine_agg_bwd_blocks(n,H)][4][H]: per-block sums for the edge-table gradient
};
int gine_agg_bwd_blocks(int n, int H);
int gine_agg_bwd(hipStream_t s, const AggBwdGroup* g, int ngroups, int ldda, int ldx, int lddx, int ldp,
                 const int32_t* node_graph, int n, int H);
// d_edge_w [H,4], d_edge_b [H] = ordered sum of np partials [4][H] written by gine_agg_bwd;
// seg_partial: scratch of ETAB_SEGMENTS * 4 * H floats for the first level of the sum
constexpr int ETAB_SEGMENTS = 32;
struct EtabFinalGroup { const float* partial; float* seg_partial; float* d_edge_w; float* d_edge_b; };
int gine_etab_final(hipStream_t s, const EtabFinalGroup* g, int ngroups, int np, int H);

// One GINE application in one launch: aggregation (as the producer of the first product's operand) -> Linear +
// ReLU -> Linear; agg / hid ([n, H] each) are written only when `save` (backward needs them), never re-read.
struct FusedLayerGroup {
    const float* x;                               // layer input [n, ldx]
    const int32_t *rowptr, *col, *etype;          // CSR by target
    const float *edge_w, *edge_b;
    const float *w1p, *b1, *w2p, *b2;             // conv.nn.seq.{0,2}: weights PACKED by fused_pack, biases
    float *agg, *hid;                             // saved for backward, or null
    float* y;                                     // [n, ldy]
    float ope;                                    // 1 + eps
    int E;                                        // edge count (work accounting only)
};
bool gine_layer_fused_supported(int H, int ldx, int ldy);
// Backward of the two Linear layers of a GINE application in one launch: dh = (g w2) * [hid > 0], da = dh w1, with
// w2tp / w1tp packed by fused_pack with `transposed` set (all matrices [n, H] / [H, H]).
// `dpool` (or null; cut-operand kernels only): the rows of g are not read but ARE the pooled-output gradient of the row's
// graph, dpool[node_graph[row] * ldp + column] - the top of a stack whose output went through the graph sum - and the
// kernel writes them to g on the way (the weight gradients read them there).
struct FusedLayerBwdGroup { const float *g, *hid, *w2tp, *w1tp; float *dh, *da; const float* dpool = nullptr; };
int gine_layer_bwd(hipStream_t s, const FusedLayerBwdGroup* g, int ngroups, int n, int H, const int32_t* node_graph = nullptr,
                   int ldp = 0);
// Weights as the fused kernels read them: a [N][K] matrix (row pitch ld; `transposed`: stored as [K][N]) re-laid as
// [ceil(K/16)][4][Npad][4] so that a wave's operand fragment is four 256-byte runs; fused_pack_floats = its size.
// `split`: three bf16 planes per 32 reduction indices instead (the form the GINE kernels take when gine_layer_split(H) says so).
struct FusedPack { const float* src; float* dst; int N, K, ld; bool transposed; bool split = false; };
bool gine_layer_split(int H);
size_t fused_pack_floats(int N, int K);
int fused_pack(hipStream_t s, const FusedPack* d, int count);   // up to 8 matrices in one launch
int gine_layer_fwd(hipStream_t s, const FusedLayerGroup* g, int ngroups, int ldx, int ldy, int n, int H, bool save);

// The link scorer's forward pass in one launch: pair gather -> MLP(P, P/2, P, ReLU, 3) -> Linear(P, 4) + cross-entropy
// rows.  `packed`: scorer_pack_floats floats written by scorer_pack (once per step).  pin [S,P], s1/s2 [S,P/2],
// s3 [S,P], logits [S,4], row_loss/lse [S] are the copies backward reads.
struct ScorerFwd {
    const float* emb; int lde; const float* z;
    const int32_t *src, *dst, *graph, *gold;
    const float* packed; const float *b0, *b1, *b2, *w3, *b3;
    float *pin, *s1, *s2, *s3, *logits, *row_loss, *lse;
    int S, H, Z;
};
bool scorer_fused_supported(int H, int Z, int lde);
size_t scorer_pack_floats(int H, int Z);
int scorer_pack(hipStream_t s, const float* w0, const float* w1, const float* w2, float* packed, int H, int Z);
int scorer_fwd_fused(hipStream_t s, const ScorerFwd& f);
// Backward of the scorer's three Linear layers in one launch, given ds3 = dL/ds3 [S,P] (written by the head's backward):
// ds2, ds1 [S,P/2] and dpin [S,P].  `packed`: scorer_pack_floats floats written by scorer_pack_bwd.
struct ScorerBwd { const float *ds3, *s2, *s1, *packed; float *ds2, *ds1, *dpin; int S, H, Z; };
int scorer_pack_bwd(hipStream_t s, const float* w0, const float* w1, const float* w2, float* packed, int H, int Z);
int scorer_bwd_fused(hipStream_t s, const ScorerBwd& f);

int embed3_gather(hipStream_t s, const int32_t* a, const int32_t* p, const int32_t* q, const float* ta,
                  const float* tp, const float* tq, int da, int dp, int dq, float* out, int n);
int rows_gather(hipStream_t s, const int32_t* idx, int idx_stride_outer, int inner, int idx_ld, const float* table,
                int width, float* out, int rows);
// dTable[row] = sum over list 1 of src1[idx, off1 .. off1+D) + sum over list 2 of src2[idx, off2 ..)
struct TableGradDesc {
    float* dtable; int rows, D;
    const int32_t *ptr1, *idx1; const float* src1; int ld1, off1;
    const int32_t *ptr2, *idx2; const float* src2; int ld2, off2;   // ptr2 may be null
};
size_t table_grads_scratch(const TableGradDesc* t, int ntables);
int table_grads(hipStream_t s, const TableGradDesc* t, int ntables, float* partial, size_t partial_floats);
// x: slices [rows, ldx] lying slice_stride floats apart, slice k -> output columns [k*slice_width, (k+1)*slice_width)
int segment_pool_fwd(hipStream_t s, const float* x, int ldx, int64_t slice_stride, int slice_width,
                     const int32_t* graph_off, float* pooled, int ldp, int B, int width);
int pair_gather(hipStream_t s, const float* emb, int lde, const float* z, const int32_t* src, const int32_t* dst,
                const int32_t* graph, float* out, int S, int H, int Z);
int pair_scatter(hipStream_t s, const float* dpin, int P, const int32_t* by_src_ptr, const int32_t* by_src_idx,
                 const int32_t* by_dst_ptr, const int32_t* by_dst_idx, const int32_t* pair_graph_off, float* demb,
                 int ldde, float* dz, int dz_accumulate, int n, int B, int H, int Z);

// ------------------------------------------------------------ heads / losses
// Upstream gradients of the four loss terms [pred, piece, edge, kl], derived inside the consuming kernels from
// the gradient g[6] (DEVICE) of the step's outputs [pred, rec, piece, edge, kl, total] and the loss weights:
// total = w_pred pred + w_rec rec + w_kl kl, rec = piece + edge.
struct GScale { const float* g; float w_pred, w_rec, w_kl; };
#ifdef __HIPCC__
__device__ __forceinline__ float gscale_at(const GScale& s, int k) {
    const float g5 = s.g[5];
    return k == 0 ? s.g[0] + g5 * s.w_pred : (k == 1 ? s.g[2] + s.g[1] + g5 * s.w_rec : (k == 2 ? s.g[3] + s.g[1] + g5 * s.w_rec : s.g[4] + g5 * s.w_kl));
}
#endif
// Linear layers over a handful of rows (latent_to_rnn_hidden, W_mean/W_log_var): a wave per output instead of MFMA tiles
int small_linear_fwd(hipStream_t s, const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int N, int K);
int small_linear_bwd_x(hipStream_t s, const float* dy, int lddy, const float* w, float* dx, int lddx, int M, int N, int K);
// [W_mean; W_log_var] (one [2Z, G] matrix) fused with rsample (forward: mulv, z, KL rows) and with its input
// gradient (backward: d_mulv, d_cond): one launch each instead of two on the step's dependent chain
bool latent_head_supported(const float* cond, int ldc, const float* w, int Z, int G);
int latent_head_fwd_slabs(hipStream_t s, const GemmSlabs& sl, const float* cond_bias, float* cond, int ldc, const float* w,
                          const float* bias, const float* eps, float* mulv, float* z, float* klrow, int B, int Z, int G);
int latent_head_fwd(hipStream_t s, const float* cond, int ldc, const float* w, const float* bias, const float* eps,
                    float* mulv, float* z, float* klrow, int B, int Z, int G);
int latent_head_bwd(hipStream_t s, const float* mulv, const float* eps, const float* dz0, const float* dz1, const float* dz2,
                    GScale gscale, const float* denoms, const float* w, float* dmulv, float* dcond, int ldd, int B, int Z, int G);
int vae_head_fwd(hipStream_t s, const float* mulv, const float* eps, float* z, float* klrow, int B, int Z);
int vae_head_bwd(hipStream_t s, const float* mulv, const float* eps, const float* dz0, const float* dz1, const float* dz2,
                 GScale gscale,
                 const float* denoms, float* dmulv, int B, int Z);
int softmax_ce_fwd(hipStream_t s, const float* logits, int ld, const int32_t* target, int tstride, int toffset,
                   int rows_per_group, int rows, int classes, int ignore_index, float* row_loss, float* lse);
bool softmax_ce_fused_supported(int classes);
int softmax_ce_fwd_bwd(hipStream_t s, float* logits_inout, int ld, const int32_t* target, int tstride, int toffset,
                       int rows_per_group, int rows, int classes, int ignore_index, float* row_loss, float* lse,
                       GScale gscale, int gidx, const float* denoms, int didx);
int softmax_ce_bwd(hipStream_t s, float* logits_inout, int ld, const int32_t* target, int tstride, int toffset,
                   int rows_per_group, int rows, int classes, int ignore_index, const float* lse,
                   GScale gscale, int gidx, const float* denoms, int didx);
// Linear(P -> 4) + cross-entropy of the link scorer in one launch each way (wave per row, weights in registers).
// forward: logits [S,4], lse, row loss; backward: logits <- dL/dlogits in place, dx [S, lddx] = dlogits W.
bool pair_head_supported(const float* x, int ldx, const float* w, const float* dx, int lddx, int P);
int pair_head_fwd(hipStream_t s, const float* x, int ldx, const float* w, const float* bias, const int32_t* gold,
                  float* logits, float* row_loss, float* lse, int S, int P);
int pair_head_bwd(hipStream_t s, float* logits_inout, const float* lse, const int32_t* gold, const float* w, GScale gscale,
                  const float* denoms, float* dx, int lddx, int S, int P);
int mse_fwd(hipStream_t s, const float* pred, int ldp, const float* gold, float* row, int B, int P);
int mse_bwd(hipStream_t s, const float* pred, int ldp, const float* gold, GScale gscale, const float* denoms,
            float* dpred, int B, int P);
int losses_finalize(hipStream_t s, const float* mse_row, int B, const float* piece_row, int rows_p,
                    const float* edge_row, int S, const float* klrow, const float* denoms, const float* w3, float* losses);
int fill_zero(hipStream_t s, float* p, int64_t count);
// softmax over the 4 bond classes of S pairs: probabilities [S,4], the largest one and its class (any may be null)
int link_softmax(hipStream_t s, const float* logits, float* probs, float* conf, int32_t* type, int S);

// ------------------------------------------------------------ GRU
// `exch` (gru_exchange_bytes, 16-byte aligned, or NULL) enables the multi-CU recurrence; `status` (device, or
// NULL): word 0 is set to 1 when a workgroup of that recurrence stopped waiting for its partners
size_t gru_exchange_bytes(int B, int T, int R);
bool gru_multi_cu_selected(int B, int R);
int gru_fwd_stamped(hipStream_t s, const float* gi, const float* h0, const float* whh, const float* bhh, float* hseq,
                    float* hprev, float* gates, int B, int T, int R, void* exch, size_t exch_bytes, unsigned long long* stamps);
int gru_fwd(hipStream_t s, const float* gi, const float* h0, const float* whh, const float* bhh, float* hseq,
            float* hprev, float* gates, int B, int T, int R, void* exch, size_t exch_bytes, bool exch_clean = false,
            int32_t* status = nullptr);
int gru_bwd(hipStream_t s, const float* dhseq, const float* hprev, const float* gates, const float* whh,
            float* dgi, float* dgh, float* dh0, int B, int T, int R, void* exch, size_t exch_bytes, bool exch_clean = false,
            int32_t* status = nullptr);
// zero the exchange buffer ahead of time (then pass exch_clean = true)
int gru_exchange_clear(hipStream_t s, void* exch, size_t exch_bytes, int B, int T, int R);

// ------------------------------------------------------------ optimiser
int clip_adam(hipStream_t s, float* p, float* g, float* m, float* v, int64_t count, float* partials,
              float max_norm, double lr, double b1, double b2, double eps, double bias1, double bias2, float* norm_out);
