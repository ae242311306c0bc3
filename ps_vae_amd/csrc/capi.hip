// extern "C" surface for the single operators (thin argument checks around the
// same launchers the whole-step entry points use).
#include "psvae_internal.h"
#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>

extern "C" const char* psvae_strerror(int code) {
    switch (code) {
        case PSVAE_OK: return "ok";
        case PSVAE_E_NULL: return "psvae: a required pointer is NULL";
        case PSVAE_E_SHAPE: return "psvae: a size argument is negative, inconsistent or unsupported";
        case PSVAE_E_ALIGN: return "psvae: pointer or leading dimension violates an alignment rule (16 bytes / multiple of 4)";
        case PSVAE_E_WORKSPACE: return "psvae: workspace or scratch buffer too small";
        case PSVAE_E_CAPACITY: return "psvae: host output buffer too small";
        case PSVAE_E_STATIC_LIMIT: return "psvae: the problem needs more scratch than the 32 MiB static array of the no-allocation entry point holds; use the variant that takes a scratch buffer";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "psvae: unknown error code";
}

extern "C" int psvae_abi_version(void) { return PSVAE_ABI_VERSION; }

// ---- kernel-selection options: one table, filled from the environment (PSVAE_<NAME>) the first time it is read
namespace {
const char* const option_names[OPT_COUNT] = {"streams", "fused_layer", "fused_bwd", "fused_scorer", "split_bf16", "split_rows",
                                             "split_gemm", "wide_gemm", "gru_multi_cu", "gru_tiles", "gru_spin_ticks"};
int option_values[OPT_COUNT];
std::once_flag option_once;

void options_from_env() {
    for (int i = 0; i < OPT_COUNT; ++i) {
        char var[64] = "PSVAE_";
        size_t k = strlen(var);
        for (const char* c = option_names[i]; *c && k + 1 < sizeof(var); ++c) var[k++] = (char)toupper((unsigned char)*c);
        var[k] = 0;
        const char* e = getenv(var);
        option_values[i] = (e && *e) ? atoi(e) : -1;
    }
}
int option_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(name, option_names[i]) == 0) return i;
    return -1;
}
}  // namespace

int opt(int which) {
    std::call_once(option_once, options_from_env);
    return option_values[which];
}

extern "C" int psvae_option_set(const char* name, int value) {
    const int i = option_index(name);
    if (i < 0) return PSVAE_E_SHAPE;
    std::call_once(option_once, options_from_env);
    option_values[i] = value < 0 ? -1 : value;
    return 0;
}

extern "C" int psvae_option_get(const char* name, int* value) {
    const int i = option_index(name);
    if (i < 0) return PSVAE_E_SHAPE;
    if (!value) return PSVAE_E_NULL;
    *value = opt(i);
    return 0;
}

extern "C" const char* psvae_option_name(int index) { return index >= 0 && index < OPT_COUNT ? option_names[index] : nullptr; }

extern "C" int psvae_clip_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t count,
                               float* partials, float max_norm, double lr, double beta1, double beta2, double adam_eps,
                               double bias1, double bias2, float* grad_norm_out, void* stream) {
    return clip_adam((hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, count, partials, max_norm, lr, beta1, beta2,
                     adam_eps, bias1, bias2, grad_norm_out);
}

extern "C" size_t psvae_linear_scratch_bytes(int M, int N, int K, int nseg) {
    size_t a = gemm_scratch_floats(M, N, K, 1, false);          // forward
    size_t b = gemm_scratch_floats(N, K + 1, M, nseg, true);     // weight gradient with the bias column ...
    size_t c = gemm_scratch_floats(N, K, M, nseg, true);         // ... and without it (another tile count, another split)
    if (c > b) b = c;
    return (a > b ? a : b) * sizeof(float) + 256;
}

extern "C" int psvae_linear_forward(const float* x, int ldx, const float* w, const float* b, float* y, int ldy,
                                    int M, int N, int K, int relu, void* scratch, size_t scratch_bytes, void* stream) {
    if (!x || !w || !y) return PSVAE_E_NULL;
    if (M < 0 || N < 1 || K < 1 || ldx < K || ldy < N) return PSVAE_E_SHAPE;
    return linear_fwd((hipStream_t)stream, x, ldx, w, b, y, ldy, M, N, K, relu, (float*)scratch, scratch_bytes / sizeof(float));
}

extern "C" int psvae_linear_backward_input(const float* dy, int lddy, const float* w, float* dx, int lddx,
                                           int M, int N, int K, const float* mask, int ldmask, int accumulate, void* stream) {
    if (!dy || !w || !dx) return PSVAE_E_NULL;
    if (M < 0 || N < 1 || K < 1 || lddy < N || lddx < K || (mask && ldmask < K)) return PSVAE_E_SHAPE;
    return linear_bwd_x((hipStream_t)stream, dy, lddy, w, dx, lddx, M, N, K, mask, ldmask, accumulate);
}

extern "C" int psvae_linear_backward_weight(const float* dy, int lddy, int64_t seg_stride_dy, const float* x, int ldx,
                                            int64_t seg_stride_x, int nseg, float* dw, float* db, int M, int N, int K,
                                            void* scratch, size_t scratch_bytes, void* stream) {
    if (!dy || !x || !dw) return PSVAE_E_NULL;
    if (M < 0 || N < 1 || K < 1 || nseg < 1 || lddy < N || ldx < K) return PSVAE_E_SHAPE;
    return linear_bwd_w((hipStream_t)stream, dy, lddy, seg_stride_dy, x, ldx, seg_stride_x, nseg, dw, db, M, N, K,
                        (float*)scratch, scratch_bytes / sizeof(float));
}

extern "C" int psvae_gine_aggregate_forward(const float* x, int ldx, const int32_t* rowptr, const int32_t* col,
                                            const int32_t* etype, const float* edge_w, const float* edge_b,
                                            float one_plus_eps, float* out, int ldo, int n, int H, void* stream) {
    if (!x || !rowptr || !edge_w || !edge_b || !out) return PSVAE_E_NULL;
    if (n < 0 || H < 4 || ldx < H || ldo < H) return PSVAE_E_SHAPE;
    AggGroup g{x, out, rowptr, col, etype, edge_w, edge_b, one_plus_eps, 0};
    return gine_agg_fwd((hipStream_t)stream, &g, 1, ldx, ldo, n, H);
}

extern "C" size_t psvae_gine_layer_scratch_bytes(int H) { return 2 * fused_pack_floats(H, H) * sizeof(float); }

extern "C" int psvae_gine_layer_forward(const float* x, int ldx, const int32_t* rowptr, const int32_t* col,
                                        const int32_t* etype, const float* edge_w, const float* edge_b, float one_plus_eps,
                                        const float* w1, const float* b1, const float* w2, const float* b2, float* agg,
                                        float* hid, float* y, int ldy, int n, int H, void* scratch, size_t scratch_bytes,
                                        void* stream) {
    if (!x || !rowptr || !edge_w || !edge_b || !w1 || !b1 || !w2 || !b2 || !y || !scratch) return PSVAE_E_NULL;
    if ((agg == nullptr) != (hid == nullptr)) return PSVAE_E_NULL;
    if (n < 0 || H < 4 || ldx < H || ldy < H) return PSVAE_E_SHAPE;
    if (!gine_layer_fused_supported(H, ldx, ldy)) return PSVAE_E_SHAPE;
    if (!aligned16(scratch)) return PSVAE_E_ALIGN;
    if (scratch_bytes < psvae_gine_layer_scratch_bytes(H)) return PSVAE_E_WORKSPACE;
    float* w1p = (float*)scratch;
    float* w2p = w1p + fused_pack_floats(H, H);
    const FusedPack pk[2] = {{w1, w1p, H, H, H, false, gine_layer_split(H)}, {w2, w2p, H, H, H, false, gine_layer_split(H)}};
    PSVAE_TRY(fused_pack((hipStream_t)stream, pk, 2));
    FusedLayerGroup g{x, rowptr, col ? col : rowptr, etype ? etype : rowptr, edge_w, edge_b, w1p, b1, w2p, b2, agg, hid, y,
                      one_plus_eps, 0};
    return gine_layer_fwd((hipStream_t)stream, &g, 1, ldx, ldy, n, H, agg != nullptr);
}

extern "C" int psvae_gine_layer_backward(const float* g, const float* hid, const float* w1, const float* w2, float* d_hid,
                                         float* d_agg, int n, int H, void* scratch, size_t scratch_bytes, void* stream) {
    if (!g || !hid || !w1 || !w2 || !d_hid || !d_agg || !scratch) return PSVAE_E_NULL;
    if (n < 0 || H < 4) return PSVAE_E_SHAPE;
    if (!gine_layer_fused_supported(H, H, H)) return PSVAE_E_SHAPE;
    if (!aligned16(scratch)) return PSVAE_E_ALIGN;
    if (scratch_bytes < psvae_gine_layer_scratch_bytes(H)) return PSVAE_E_WORKSPACE;
    float* w1tp = (float*)scratch;
    float* w2tp = w1tp + fused_pack_floats(H, H);
    const FusedPack pk[2] = {{w1, w1tp, H, H, H, true, gine_layer_split(H)}, {w2, w2tp, H, H, H, true, gine_layer_split(H)}};
    PSVAE_TRY(fused_pack((hipStream_t)stream, pk, 2));
    FusedLayerBwdGroup grp{g, hid, w2tp, w1tp, d_hid, d_agg};
    return gine_layer_bwd((hipStream_t)stream, &grp, 1, n, H);
}

extern "C" size_t psvae_scorer_scratch_bytes(int H, int Z) {
    if (!scorer_fused_supported(H, Z, H)) return 0;
    return scorer_pack_floats(H, Z) * sizeof(float);
}

extern "C" int psvae_scorer_forward(const float* emb, int lde, const float* z, const int32_t* pair_src, const int32_t* pair_dst,
                                    const int32_t* pair_graph, const int32_t* pair_gold, const float* w0, const float* b0,
                                    const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                    const float* b3, float* pair_in, float* h1, float* h2, float* h3, float* logits,
                                    float* row_loss, float* lse, int S, int H, int Z, void* scratch, size_t scratch_bytes,
                                    void* stream) {
    if (!w0 || !w1 || !w2 || !scratch) return PSVAE_E_NULL;
    if (S < 0 || lde < H) return PSVAE_E_SHAPE;
    if (!scorer_fused_supported(H, Z, lde)) return PSVAE_E_SHAPE;
    if (!aligned16(scratch)) return PSVAE_E_ALIGN;
    if (scratch_bytes < psvae_scorer_scratch_bytes(H, Z)) return PSVAE_E_WORKSPACE;
    PSVAE_TRY(scorer_pack((hipStream_t)stream, w0, w1, w2, (float*)scratch, H, Z));
    ScorerFwd f{emb, lde, z, pair_src, pair_dst, pair_graph, pair_gold, (const float*)scratch, b0, b1, b2, w3, b3,
                pair_in, h1, h2, h3, logits, row_loss, lse, S, H, Z};
    return scorer_fwd_fused((hipStream_t)stream, f);
}

extern "C" int psvae_scorer_backward(const float* d_h3, const float* h2, const float* h1, const float* w0, const float* w1,
                                     const float* w2, float* d_h2, float* d_h1, float* d_pair_in, int S, int H, int Z,
                                     void* scratch, size_t scratch_bytes, void* stream) {
    if (!w0 || !w1 || !w2 || !scratch) return PSVAE_E_NULL;
    if (S < 0) return PSVAE_E_SHAPE;
    if (!scorer_fused_supported(H, Z, H)) return PSVAE_E_SHAPE;
    if (!aligned16(scratch)) return PSVAE_E_ALIGN;
    if (scratch_bytes < psvae_scorer_scratch_bytes(H, Z)) return PSVAE_E_WORKSPACE;
    PSVAE_TRY(scorer_pack_bwd((hipStream_t)stream, w0, w1, w2, (float*)scratch, H, Z));
    ScorerBwd b{d_h3, h2, h1, (const float*)scratch, d_h2, d_h1, d_pair_in, S, H, Z};
    return scorer_bwd_fused((hipStream_t)stream, b);
}

extern "C" size_t psvae_gine_edge_grad_scratch(int n, int H) {
    if (n < 0 || H < 4 || H % 4) return 0;
    return ((size_t)gine_agg_bwd_blocks(n, H) + ETAB_SEGMENTS) * 4 * H;
}

extern "C" int psvae_gine_aggregate_backward(const float* da, int ldda, const float* x, int ldx, const int32_t* rowptr_t,
                                             const int32_t* col_t, const int32_t* etype_t, const float* edge_w,
                                             const float* edge_b, float one_plus_eps, const float* dpool, int ldp,
                                             const int32_t* node_graph, float* dx, int lddx, float* d_edge_w,
                                             float* d_edge_b, float* partial, int n, int H, void* stream) {
    if (!dx || (da && (!x || !rowptr_t || !edge_w || !edge_b))) return PSVAE_E_NULL;
    if (d_edge_w && (!da || !d_edge_b || !partial)) return PSVAE_E_NULL;
    if (n < 0 || H < 4 || lddx < H) return PSVAE_E_SHAPE;
    AggBwdGroup g{da, x, dx, rowptr_t, col_t, etype_t, edge_w, edge_b, one_plus_eps, dpool, 0, d_edge_w ? partial : nullptr};
    PSVAE_TRY(gine_agg_bwd((hipStream_t)stream, &g, 1, da ? ldda : H, da ? ldx : H, lddx, dpool ? ldp : H, node_graph, n, H));
    if (d_edge_w && n > 0) {
        EtabFinalGroup e{partial, partial + (size_t)gine_agg_bwd_blocks(n, H) * 4 * H, d_edge_w, d_edge_b};
        return gine_etab_final((hipStream_t)stream, &e, 1, gine_agg_bwd_blocks(n, H), H);
    }
    return 0;
}

extern "C" int psvae_segment_pool_forward(const float* x, int ldx, const int32_t* graph_off, float* pooled, int ldp,
                                          int B, int width, void* stream) {
    if (!x || !graph_off || !pooled) return PSVAE_E_NULL;
    if (B < 0 || width < 4 || ldx < width || ldp < width) return PSVAE_E_SHAPE;
    return segment_pool_fwd((hipStream_t)stream, x, ldx, 0, width, graph_off, pooled, ldp, B, width);
}

extern "C" size_t psvae_gru_exchange_bytes(int B, int T, int R) {
    if (B < 0 || T < 0 || R < 4) return 0;
    return gru_exchange_bytes(B, T, R);
}

extern "C" int psvae_gru_forward(const float* gi, const float* h0, const float* whh, const float* bhh, float* hseq,
                                 float* hprev, float* gates, int B, int T, int R, void* exchange,
                                 size_t exchange_bytes, void* stream) {
    if (!gi || !h0 || !whh || !bhh || !hseq || !hprev || !gates) return PSVAE_E_NULL;
    if (B < 0 || T < 0 || R < 4) return PSVAE_E_SHAPE;
    return gru_fwd((hipStream_t)stream, gi, h0, whh, bhh, hseq, hprev, gates, B, T, R, exchange, exchange_bytes);
}

extern "C" int psvae_gru_forward_stamped(const float* gi, const float* h0, const float* whh, const float* bhh, float* hseq,
                                         float* hprev, float* gates, int B, int T, int R, void* exchange,
                                         size_t exchange_bytes, uint64_t* stamps, void* stream) {
    if (!gi || !h0 || !whh || !bhh || !hseq || !hprev || !gates || !exchange || !stamps) return PSVAE_E_NULL;
    if (B < 0 || T < 0 || R < 4) return PSVAE_E_SHAPE;
    return gru_fwd_stamped((hipStream_t)stream, gi, h0, whh, bhh, hseq, hprev, gates, B, T, R, exchange, exchange_bytes,
                           (unsigned long long*)stamps);
}

extern "C" int psvae_gru_backward(const float* dhseq, const float* hprev, const float* gates, const float* whh,
                                  float* dgi, float* dgh, float* dh0, int B, int T, int R, void* exchange,
                                  size_t exchange_bytes, void* stream) {
    if (!dhseq || !hprev || !gates || !whh || !dgi || !dgh || !dh0) return PSVAE_E_NULL;
    if (B < 0 || T < 0 || R < 4) return PSVAE_E_SHAPE;
    return gru_bwd((hipStream_t)stream, dhseq, hprev, gates, whh, dgi, dgh, dh0, B, T, R, exchange, exchange_bytes);
}

extern "C" int psvae_softmax_ce_forward(const float* logits, int ld, const int32_t* target, int target_stride,
                                        int target_offset, int rows_per_group, int rows, int classes, int ignore_index,
                                        float* row_loss, float* lse, void* stream) {
    if (!logits || !target || !row_loss || !lse) return PSVAE_E_NULL;
    if (rows < 0 || classes < 1 || ld < classes) return PSVAE_E_SHAPE;
    return softmax_ce_fwd((hipStream_t)stream, logits, ld, target, target_stride, target_offset, rows_per_group, rows, classes,
                          ignore_index, row_loss, lse);
}
