// Host-side sequencing of the whole training step: one call enqueues every
// kernel of PSVAEModel.forward (or its backward) on the caller's stream.
//
// Packed layout: all node tensors have n rows (real atoms only).  Each GINE
// stack keeps its node_trans output and its t layer outputs side by side in one
// [n, (t+1)*H] matrix ("layer stack"): slice l is the input of layer l, slices
// 1..t are the reference's all_x, slice t its last-layer embedding.  When both
// stacks run the same number of iterations, every per-layer launch handles the
// encoder and the decoder-side encoder together (grid dimension = group).
#include "psvae_internal.h"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <mutex>

namespace {

struct Sz {
    int B, n, S, L, T, BT;
    int Da, Dp, Dq, Din, H, G, Z, R, V2, P, P2, Hp, NP, te, td, NA, MP;
    int ldl, ldpred;
};

Sz sizes(const psvae_dims& d, const psvae_batch& b) {
    Sz s;
    s.B = b.B; s.n = b.n; s.S = b.S; s.L = b.L; s.T = b.L > 0 ? b.L - 1 : 0; s.BT = s.B * s.T;
    s.Da = d.d_atom; s.Dp = d.d_piece; s.Dq = d.d_pos; s.Din = s.Da + s.Dp + s.Dq;
    s.H = d.hidden; s.G = d.graph; s.Z = d.latent; s.R = d.rnn;
    s.V2 = d.num_piece; s.P = 2 * d.hidden + d.latent; s.P2 = s.P / 2; s.Hp = d.pred_hidden; s.NP = d.n_props;
    s.te = d.t_enc; s.td = d.t_dec; s.NA = d.num_atom; s.MP = d.max_pos;
    s.ldl = (int)round_up(s.V2, 4); s.ldpred = (int)round_up(s.NP, 4);
    return s;
}

int check_dims(const psvae_dims* d, const psvae_batch* b) {
    if (!d || !b) return PSVAE_E_NULL;
    if (d->num_bond != 4) return PSVAE_E_SHAPE;
    if (d->hidden <= 0 || d->hidden % 4 || d->latent <= 0 || d->latent % 4 || d->rnn <= 0 || d->rnn % 4) return PSVAE_E_ALIGN;
    if (d->t_enc < 1 || d->t_dec < 1 || d->graph < 1 || d->pred_hidden < 1 || d->n_props < 1) return PSVAE_E_SHAPE;
    if (d->d_atom < 1 || d->d_piece < 1 || d->d_pos < 1 || d->num_piece < 3) return PSVAE_E_SHAPE;
    if (b->B < 1 || b->n < 1 || b->E < 0 || b->E_in < 0 || b->S < 0 || b->L < 2) return PSVAE_E_SHAPE;
    return 0;
}

struct WS {
    // forward (saved for backward)
    float *X0, *XE, *XD, *AE, *AD, *HE, *HD, *pooled, *cond, *mulv, *z, *klrow;
    float *semb, *gi, *h0, *hseq, *hprev, *gates, *logits, *rowloss_p, *lse_p;
    float *pin, *s1, *s2, *s3, *slog, *rowloss_e, *lse_e, *q1, *q2, *pred, *mse_row, *losses;
    // backward
    float *dpred, *dq2, *dq1, *dz, *dz_piece, *dz_pair, *gscale, *dhseq, *dgi, *dgh, *dh0, *dsemb, *ds3, *ds2, *ds1, *dpin;
    float *dmulv, *dcond, *dpooled, *GE, *GD, *G0, *DHE, *DHD, *DAE, *DAD, *dX0, *etab_part, *wpack, *spack;
    float *tg_part; size_t tg_part_floats;
    float *scratch, *scratch1, *scratch2; size_t scratch_floats;
    float* gru_exch; size_t gru_exch_bytes;
    size_t total_floats;
};

struct Field { const char* name; float* WS::*member; size_t count; };

// One table drives both the size query and the pointer carve-up.
size_t carve(const Sz& s, float* base, WS* w, const char* want, int64_t* want_off) {
    const size_t n = s.n, B = s.B, S = s.S, BT = s.BT, H = s.H;
    const size_t tmax = s.te > s.td ? s.te : s.td;
    size_t maxw = (size_t)s.H * (s.H + 1);
    auto upw = [&](size_t rows, size_t cols) { if (rows * (cols + 1) > maxw) maxw = rows * (cols + 1); };
    upw(s.H, s.Din); upw(2 * s.H, s.Din); upw(s.G, s.te * s.H); upw(s.V2, s.R); upw(3 * s.R, s.R); upw(3 * s.R, s.Dp);
    upw(s.P2, s.P); upw(s.P, s.P2); upw(s.P2, s.P2); upw(2 * s.Z, s.G); upw(s.Hp, s.Hp); upw(s.Hp, s.Z); upw(s.R, s.Z);
    const size_t scratch = ((size_t)6 << 20) + tmax * maxw;
    const size_t chunks = (size_t)gine_agg_bwd_blocks(s.n, s.H);
    const int dmax = s.Da > s.Dp ? (s.Da > s.Dq ? s.Da : s.Dq) : (s.Dp > s.Dq ? s.Dp : s.Dq);
    const size_t tg_part = (size_t)(s.NA + s.V2 + s.MP) * 32 * dmax;
    const size_t gru_exch_bytes = gru_exchange_bytes(s.B, s.T, s.R);
    const Field f[] = {
        {"x0", &WS::X0, n * s.Din},
        {"enc_stack", &WS::XE, n * (s.te + 1) * H}, {"dec_stack", &WS::XD, n * (s.td + 1) * H},
        {"enc_agg", &WS::AE, n * s.te * H}, {"dec_agg", &WS::AD, n * s.td * H},
        {"enc_hid", &WS::HE, n * s.te * H}, {"dec_hid", &WS::HD, n * s.td * H},
        {"pooled", &WS::pooled, B * s.te * H}, {"cond", &WS::cond, B * s.G}, {"mulv", &WS::mulv, B * 2 * s.Z},
        {"z", &WS::z, B * s.Z}, {"klrow", &WS::klrow, B},
        {"seq_emb", &WS::semb, BT * s.Dp}, {"gi", &WS::gi, BT * 3 * s.R}, {"h0", &WS::h0, B * s.R},
        {"rnn_out", &WS::hseq, BT * s.R}, {"rnn_prev", &WS::hprev, BT * s.R}, {"gates", &WS::gates, BT * 4 * s.R},
        {"logits", &WS::logits, BT * s.ldl}, {"piece_rowloss", &WS::rowloss_p, BT}, {"piece_lse", &WS::lse_p, BT},
        {"pair_in", &WS::pin, S * s.P}, {"pair_h1", &WS::s1, S * s.P2}, {"pair_h2", &WS::s2, S * s.P2},
        {"pair_h3", &WS::s3, S * s.P}, {"pair_logits", &WS::slog, S * 4}, {"edge_rowloss", &WS::rowloss_e, S},
        {"edge_lse", &WS::lse_e, S}, {"pred_h1", &WS::q1, B * s.Hp}, {"pred_h2", &WS::q2, B * s.Hp},
        {"pred", &WS::pred, B * s.ldpred}, {"mse_row", &WS::mse_row, B}, {"losses", &WS::losses, 8},
        {"d_pred", &WS::dpred, B * s.ldpred}, {"d_pred_h2", &WS::dq2, B * s.Hp}, {"d_pred_h1", &WS::dq1, B * s.Hp},
        {"d_z", &WS::dz, B * s.Z}, {"d_z_piece", &WS::dz_piece, B * s.Z}, {"d_z_pair", &WS::dz_pair, B * s.Z}, {"gscale", &WS::gscale, 8}, {"d_rnn_out", &WS::dhseq, BT * s.R}, {"d_gi", &WS::dgi, BT * 3 * s.R},
        {"d_gh", &WS::dgh, BT * 3 * s.R}, {"d_h0", &WS::dh0, B * s.R}, {"d_seq_emb", &WS::dsemb, BT * s.Dp},
        {"d_pair_h3", &WS::ds3, S * s.P}, {"d_pair_h2", &WS::ds2, S * s.P2}, {"d_pair_h1", &WS::ds1, S * s.P2},
        {"d_pair_in", &WS::dpin, S * s.P}, {"d_mulv", &WS::dmulv, B * 2 * s.Z}, {"d_cond", &WS::dcond, B * s.G},
        {"d_pooled", &WS::dpooled, B * s.te * H},
        {"enc_g", &WS::GE, n * (s.te + 1) * H}, {"dec_g", &WS::GD, n * (s.td + 1) * H}, {"g0", &WS::G0, n * 2 * H},
        {"enc_dhid", &WS::DHE, n * s.te * H}, {"dec_dhid", &WS::DHD, n * s.td * H},
        {"enc_dagg", &WS::DAE, n * s.te * H}, {"dec_dagg", &WS::DAD, n * s.td * H},
        {"d_x0", &WS::dX0, n * s.Din}, {"etab_part", &WS::etab_part, 2 * (tmax * chunks + ETAB_SEGMENTS) * 4 * H},
        {"table_grad_part", &WS::tg_part, tg_part}, {"gru_exchange", &WS::gru_exch, gru_exch_bytes / 4},
        {"packed_weights", &WS::wpack, 8 * fused_pack_floats(s.H, s.H)},
        {"packed_scorer", &WS::spack, 2 * scorer_pack_floats(s.H, s.Z)},
        {"scratch", &WS::scratch, scratch}, {"scratch1", &WS::scratch1, scratch}, {"scratch2", &WS::scratch2, scratch},
    };
    size_t off = 0;
    for (const Field& e : f) {
        if (w) w->*(e.member) = base + off;
        if (want && want_off && strcmp(want, e.name) == 0) *want_off = (int64_t)off;
        off += (size_t)round_up((int64_t)(e.count ? e.count : 1), 64);
    }
    if (w) { w->gru_exch_bytes = gru_exch_bytes; w->scratch_floats = scratch; w->tg_part_floats = tg_part; w->total_floats = off; }
    return off;
}

struct Stack {  // one GINE stack
    float *X, *A, *Hid, *Gr, *DH, *DA;
    const int32_t *rowptr, *col, *etype, *rowptr_t, *col_t, *etype_t;
    const float *node_w, *node_b, *edge_w, *edge_b, *w1, *b1, *w2, *b2;
    float *w1p, *w2p;     // conv.nn weights re-laid for the fused layer kernel (workspace, written once per call)
    float *w1tp, *w2tp;   // the same in their transposed role (input gradients), written once per backward call
    float *g_node_w, *g_node_b, *g_edge_w, *g_edge_b, *g_w1, *g_b1, *g_w2, *g_b2;
    float ope; int t; int E;
};

Stack enc_stack(const psvae_dims& d, const psvae_params& p, const psvae_params* g, const psvae_batch& b, const WS& w) {
    Stack k{};
    k.X = w.XE; k.A = w.AE; k.Hid = w.HE; k.Gr = w.GE; k.DH = w.DHE; k.DA = w.DAE;
    k.rowptr = b.rowptr; k.col = b.col; k.etype = b.etype; k.rowptr_t = b.rowptr_t; k.col_t = b.col_t; k.etype_t = b.etype_t;
    k.node_w = p.enc_node_w; k.node_b = p.enc_node_b; k.edge_w = p.enc_edge_w; k.edge_b = p.enc_edge_b;
    k.w1 = p.enc_w1; k.b1 = p.enc_b1; k.w2 = p.enc_w2; k.b2 = p.enc_b2;
    k.w1p = w.wpack; k.w2p = w.wpack + fused_pack_floats(d.hidden, d.hidden);
    k.w1tp = w.wpack + 4 * fused_pack_floats(d.hidden, d.hidden); k.w2tp = w.wpack + 5 * fused_pack_floats(d.hidden, d.hidden);
    if (g) { k.g_node_w = g->enc_node_w; k.g_node_b = g->enc_node_b; k.g_edge_w = g->enc_edge_w; k.g_edge_b = g->enc_edge_b;
             k.g_w1 = g->enc_w1; k.g_b1 = g->enc_b1; k.g_w2 = g->enc_w2; k.g_b2 = g->enc_b2; }
    k.ope = 1.0f + d.eps_enc; k.t = d.t_enc; k.E = b.E;
    return k;
}
Stack dec_stack(const psvae_dims& d, const psvae_params& p, const psvae_params* g, const psvae_batch& b, const WS& w) {
    Stack k{};
    k.X = w.XD; k.A = w.AD; k.Hid = w.HD; k.Gr = w.GD; k.DH = w.DHD; k.DA = w.DAD;
    k.rowptr = b.in_rowptr; k.col = b.in_col; k.etype = b.in_etype;
    k.rowptr_t = b.in_rowptr_t; k.col_t = b.in_col_t; k.etype_t = b.in_etype_t;
    k.node_w = p.dec_node_w; k.node_b = p.dec_node_b; k.edge_w = p.dec_edge_w; k.edge_b = p.dec_edge_b;
    k.w1 = p.dec_w1; k.b1 = p.dec_b1; k.w2 = p.dec_w2; k.b2 = p.dec_b2;
    k.w1p = w.wpack + 2 * fused_pack_floats(d.hidden, d.hidden); k.w2p = w.wpack + 3 * fused_pack_floats(d.hidden, d.hidden);
    k.w1tp = w.wpack + 6 * fused_pack_floats(d.hidden, d.hidden); k.w2tp = w.wpack + 7 * fused_pack_floats(d.hidden, d.hidden);
    if (g) { k.g_node_w = g->dec_node_w; k.g_node_b = g->dec_node_b; k.g_edge_w = g->dec_edge_w; k.g_edge_b = g->dec_edge_b;
             k.g_w1 = g->dec_w1; k.g_b1 = g->dec_b1; k.g_w2 = g->dec_w2; k.g_b2 = g->dec_b2; }
    k.ope = 1.0f + d.eps_dec; k.t = d.t_dec; k.E = b.E_in;
    return k;
}

// Forward of 1 or 2 stacks with the same t: node_trans, then t x (aggregate, Linear+ReLU, Linear).
// `packed` (or null): an event after which the stacks' packed weights are in place (written by the caller on
// another stream); without it they are packed here.
bool fused_layers_on() { return opt(OPT_FUSED_LAYER) != 0; }
// The link scorer as one launch per direction (32 pairs per workgroup, one workgroup per CU) instead of the pair
// gather, three GEMM launches and the head.  Forward alone: pays while all its workgroups fit the GPU in one round
// (128 graphs: 135 workgroups, and the CUs it leaves free go to the recurrence beside it; +1.4 % on the fp32
// instruction), loses at 256 - 384 graphs (a second, nearly empty round).  Forward and backward together, on the
// bf16 instruction over cut operands (gine_layer_split): -1.4 % at 384 graphs, +2.5 % at 512, +4.9 % at 768, +6.1 %
// at 1024; at 128 graphs the one-launch backward still loses 1.2 % (its one-per-CU workgroups keep the recurrence's
// backward, which needs its workgroups resident together, waiting for CUs).  Default: forward fused up to 240
// workgroups; both from 480 workgroups up when the cut-operand kernels are on.
// Option fused_scorer: 0 = never, 1 = forward and backward always, 2 = forward always.
int fused_scorer_mode() { return fused_layers_on() ? opt(OPT_FUSED_SCORER) : 0; }
bool fused_scorer_fwd_on(int S, int H) {
    const int m = fused_scorer_mode(), wgs = (S + 31) / 32;
    return m == 1 || m == 2 || (m < 0 && S > 0 && (wgs <= 240 || (gine_layer_split(H) && wgs >= 480)));
}
bool fused_scorer_bwd_on(int S, int H) {
    const int m = fused_scorer_mode();
    return m == 1 || (m < 0 && gine_layer_split(H) && (S + 31) / 32 >= 480);
}
// The fused backward of the GINE MLPs.  On the fp32 matrix instruction it pays once a launch holds more row tiles
// than the GPU has compute units (measured: -1.2 % at 128 graphs per step, +0.6 % at 256, +2.2 % at 512); on the
// bf16 instruction over cut operands (gine_layer_split) its two products take half the time and it pays at every
// size measured (+3 % at 128 graphs per step).  Option fused_bwd = 0 / 1 forces it off / on.
bool fused_bwd_on(int n, int ng, int H) {
    const int forced = opt(OPT_FUSED_BWD);
    if (!fused_layers_on()) return false;
    if (forced >= 0) return forced != 0;
    return gine_layer_split(H) || (long long)((n + 31) / 32) * ng >= 320;
}
int stacks_forward(hipStream_t st, const Sz& s, const WS& w, const Stack* k, int ng, bool save = true, hipEvent_t packed = nullptr) {
    const int n = s.n, H = s.H, t = k[0].t;
    const size_t nH = (size_t)n * H;   // the layer stack is slice-major: [t+1][n][H]
    const int64_t dX = ng == 2 ? k[1].X - k[0].X : 0, dA = ng == 2 ? k[1].A - k[0].A : 0, dHid = ng == 2 ? k[1].Hid - k[0].Hid : 0;
    {   // slice 0 = node_trans(x0)
        GemmDesc d;
        d.A = w.X0; d.lda = s.Din; d.B = k[0].node_w; d.ldb = s.Din; d.C = k[0].X; d.ldc = H;
        d.M = n; d.N = H; d.R = s.Din; d.bias = k[0].node_b; d.nseg = ng;
        if (ng == 2) { d.seg_b = k[1].node_w - k[0].node_w; d.seg_bias = k[1].node_b - k[0].node_b; d.seg_c = dX; }
        d.scratch = w.scratch; d.scratch_floats = w.scratch_floats;
        PSVAE_TRY(gemm_f32(st, d));
    }
    // one launch per GINE application where the fused kernel takes the size (option fused_layer = 0: three launches)
    const bool fused = fused_layers_on() && gine_layer_fused_supported(H, H, H);
    if (fused && packed) {
        if (hipStreamWaitEvent(st, packed, 0) != hipSuccess) return (int)hipGetLastError();
    } else if (fused) {    // the weights in the matrix cores' fragment order, once for all t applications
        FusedPack pk[4];
        for (int q = 0; q < ng; ++q) {
            pk[2 * q] = FusedPack{k[q].w1, k[q].w1p, H, H, H, false, gine_layer_split(H)};
            pk[2 * q + 1] = FusedPack{k[q].w2, k[q].w2p, H, H, H, false, gine_layer_split(H)};
        }
        PSVAE_TRY(fused_pack(st, pk, 2 * ng));
    }
    for (int l = 0; fused && l < t; ++l) {
        FusedLayerGroup fg[2];
        for (int q = 0; q < ng; ++q)
            fg[q] = FusedLayerGroup{k[q].X + (size_t)l * nH, k[q].rowptr, k[q].col, k[q].etype, k[q].edge_w, k[q].edge_b,
                                    k[q].w1p, k[q].b1, k[q].w2p, k[q].b2, save ? k[q].A + (size_t)l * nH : nullptr,
                                    save ? k[q].Hid + (size_t)l * nH : nullptr, k[q].X + (size_t)(l + 1) * nH, k[q].ope, k[q].E};
        PSVAE_TRY(gine_layer_fwd(st, fg, ng, H, H, n, H, save));
        if (l == t - 1) return 0;
    }
    for (int l = 0; l < t; ++l) {
        AggGroup ag[2];
        for (int q = 0; q < ng; ++q)
            ag[q] = AggGroup{k[q].X + (size_t)l * nH, k[q].A + (size_t)l * nH, k[q].rowptr, k[q].col, k[q].etype,
                             k[q].edge_w, k[q].edge_b, k[q].ope, k[q].E};
        PSVAE_TRY(gine_agg_fwd(st, ag, ng, H, H, n, H));
        GemmDesc d1;
        d1.A = k[0].A + (size_t)l * n * H; d1.lda = H; d1.B = k[0].w1; d1.ldb = H; d1.C = k[0].Hid + (size_t)l * n * H; d1.ldc = H;
        d1.M = n; d1.N = H; d1.R = H; d1.bias = k[0].b1; d1.relu = 1; d1.nseg = ng;
        if (ng == 2) { d1.seg_a = dA; d1.seg_b = k[1].w1 - k[0].w1; d1.seg_bias = k[1].b1 - k[0].b1; d1.seg_c = dHid; }
        d1.scratch = w.scratch; d1.scratch_floats = w.scratch_floats;
        PSVAE_TRY(gemm_f32(st, d1));
        GemmDesc d2;
        d2.A = k[0].Hid + (size_t)l * n * H; d2.lda = H; d2.B = k[0].w2; d2.ldb = H; d2.C = k[0].X + (size_t)(l + 1) * nH; d2.ldc = H;
        d2.M = n; d2.N = H; d2.R = H; d2.bias = k[0].b2; d2.nseg = ng;
        if (ng == 2) { d2.seg_a = dHid; d2.seg_b = k[1].w2 - k[0].w2; d2.seg_bias = k[1].b2 - k[0].b2; d2.seg_c = dX; }
        d2.scratch = w.scratch; d2.scratch_floats = w.scratch_floats;
        PSVAE_TRY(gemm_f32(st, d2));
    }
    return 0;
}

// per-stack region of the edge-table partials: [t][blocks][4][H]
static float* etab_region(const Sz& s, const WS& w, int slot) {
    const size_t tmax = s.te > s.td ? s.te : s.td;
    return w.etab_part + (size_t)slot * (tmax * gine_agg_bwd_blocks(s.n, s.H) + ETAB_SEGMENTS) * 4 * s.H;
}

// Backward of 1 or 2 stacks with the same t.  On entry Gr slice t (= [n,H] block t of
// Gr) holds dL/dx_t.  dpool[q] (or null) is the pooled-output gradient [B, t*H]
// whose slice l-1 is broadcast into dL/dx_l.  Leaves dL/dx_0 of stack q in
// w.G0[:, q*H:(q+1)*H] and writes the weight gradients of conv.nn and edge_trans.
// `top_from_pool`: dL/dx_t of the stacks that have a dpool is not in Gr yet; the fused backward of the top layer takes it
// from dpool's last slice itself (and writes it to Gr).
int stacks_backward(hipStream_t st, const Sz& s, const WS& w, const psvae_batch& b, const Stack* k, int ng,
                    const float* const* dpool, int g0_slot0, int etab_slot0, bool top_from_pool = false) {
    const int n = s.n, H = s.H, t = k[0].t;
    const size_t nH = (size_t)n * H, nblk = (size_t)gine_agg_bwd_blocks(n, H);
    const int64_t dG = ng == 2 ? k[1].Gr - k[0].Gr : 0, dDH = ng == 2 ? k[1].DH - k[0].DH : 0;
    const int64_t dDA = ng == 2 ? k[1].DA - k[0].DA : 0, dHid = ng == 2 ? k[1].Hid - k[0].Hid : 0;
    const bool fused = fused_bwd_on(n, ng, H) && gine_layer_fused_supported(H, H, H);
    for (int l = t - 1; l >= 0; --l) {
        if (fused) {    // dHid_l = (G_{l+1} W2) * (Hid_l > 0) and dAgg_l = dHid_l W1 in one launch
            FusedLayerBwdGroup fg[2];
            for (int q = 0; q < ng; ++q)
                fg[q] = FusedLayerBwdGroup{k[q].Gr + (size_t)(l + 1) * nH, k[q].Hid + (size_t)l * nH, k[q].w2tp, k[q].w1tp,
                                           k[q].DH + (size_t)l * nH, k[q].DA + (size_t)l * nH,
                                           (top_from_pool && l == t - 1 && dpool && dpool[q]) ? dpool[q] + (size_t)(t - 1) * H : nullptr};
            PSVAE_TRY(gine_layer_bwd(st, fg, ng, n, H, b.node_graph, t * H));
        } else {
        GemmDesc d2;  // dHid_l = (G_{l+1} W2) * (Hid_l > 0)
        d2.A = k[0].Gr + (size_t)(l + 1) * nH; d2.lda = H; d2.B = k[0].w2; d2.ldb = H; d2.b_t = true;
        d2.C = k[0].DH + (size_t)l * nH; d2.ldc = H; d2.M = n; d2.N = H; d2.R = H;
        d2.mask = k[0].Hid + (size_t)l * nH; d2.ldmask = H; d2.nseg = ng;
        if (ng == 2) { d2.seg_a = dG; d2.seg_b = k[1].w2 - k[0].w2; d2.seg_c = dDH; d2.seg_mask = dHid; }
        d2.scratch = w.scratch; d2.scratch_floats = w.scratch_floats;
        PSVAE_TRY(gemm_f32(st, d2));
        GemmDesc d1;  // dAgg_l = dHid_l W1
        d1.A = k[0].DH + (size_t)l * nH; d1.lda = H; d1.B = k[0].w1; d1.ldb = H; d1.b_t = true;
        d1.C = k[0].DA + (size_t)l * nH; d1.ldc = H; d1.M = n; d1.N = H; d1.R = H; d1.nseg = ng;
        if (ng == 2) { d1.seg_a = dDH; d1.seg_b = k[1].w1 - k[0].w1; d1.seg_c = dDA; }
        d1.scratch = w.scratch; d1.scratch_floats = w.scratch_floats;
        PSVAE_TRY(gemm_f32(st, d1));
        }
        AggBwdGroup ag[2];
        for (int q = 0; q < ng; ++q) {
            float* dx = l > 0 ? k[q].Gr + (size_t)l * nH : w.G0 + (size_t)(g0_slot0 + q) * H;
            const float* dp = (l > 0 && dpool && dpool[q]) ? dpool[q] + (size_t)(l - 1) * H : nullptr;
            ag[q] = AggBwdGroup{k[q].DA + (size_t)l * nH, k[q].X + (size_t)l * nH, dx, k[q].rowptr_t, k[q].col_t,
                                k[q].etype_t, k[q].edge_w, k[q].edge_b, k[q].ope, dp, k[q].E,
                                etab_region(s, w, etab_slot0 + q) + (size_t)l * nblk * 4 * H};
        }
        PSVAE_TRY(gine_agg_bwd(st, ag, ng, H, H, l > 0 ? H : 2 * H, t * H, b.node_graph, n, H));
    }
    return 0;
}

// edge_trans gradients of 1 or 2 stacks with the same t, from the partials stacks_backward left behind
int stacks_edge_grads(hipStream_t st, const Sz& s, const WS& w, const Stack* k, int ng, int slot0) {
    EtabFinalGroup eg[2];
    const size_t np = (size_t)k[0].t * gine_agg_bwd_blocks(s.n, s.H);
    for (int q = 0; q < ng; ++q) {
        float* region = etab_region(s, w, slot0 + q);
        eg[q] = EtabFinalGroup{region, region + np * 4 * s.H, k[q].g_edge_w, k[q].g_edge_b};
    }
    return gine_etab_final(st, eg, ng, (int)np, s.H);
}

// Side streams ("lanes") for the branches of the step that do not depend on each other: the
// piece/GRU branch and the property head run beside the link scorer, and weight-gradient
// products run beside the input-gradient chain.  Fork and join are event waits on the
// caller's stream, so from outside the call still behaves like work on that one stream.
// Option streams = 0 runs everything on the caller's stream.
struct Lanes { hipStream_t s1, s2; hipEvent_t ev[17]; };
std::mutex lanes_mu;
Lanes* lanes_per_device[64] = {};
bool lanes_unavailable = false;      // stream or event creation failed once: stay on the caller's stream

bool create_lane(hipStream_t* s) {
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking) == hipSuccess;
}

// One wave that keeps a compute unit busy for a fixed time (constant 100 MHz counter), for lanes_overlap().
__global__ void lane_probe_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}
// Do kernels on streams `a` and `b` run side by side?  HIP deals a process's streams onto a few hardware
// queues and two streams that landed on the same queue take turns.  Two 100 us probes are launched, one on
// each stream, and timed from the host: side by side they take little more than 100 us, in turn 200 us.
bool lanes_overlap(hipStream_t a, hipStream_t b) {
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return true;
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(lane_probe_kernel, dim3(1), dim3(64), 0, a, 10000LL);
        hipLaunchKernelGGL(lane_probe_kernel, dim3(1), dim3(64), 0, b, 10000LL);
        if (hipGetLastError() != hipSuccess) return true;        // no probe, no verdict: keep the streams as they are
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return true;
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    return best < 165.0;
}

Lanes* get_lanes(bool probe = false, hipStream_t caller = nullptr) {
    std::lock_guard<std::mutex> lock(lanes_mu);
    if (opt(OPT_STREAMS) == 0 || lanes_unavailable) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!lanes_per_device[dev]) {
        Lanes* l = new Lanes();
        bool ok = create_lane(&l->s1) && create_lane(&l->s2);
        // With `probe`, a lane that shares its hardware queue with the caller's stream or with the other lane is
        // replaced by a newly created stream (which gets the next queue) until all three run side by side; the
        // rejected streams are kept until the search is over so that their queues stay taken.
        hipStream_t rejected[16]; int nrej = 0;
        for (int attempt = 0; ok && probe && attempt < 8; ++attempt) {
            if (!lanes_overlap(caller, l->s1)) { rejected[nrej++] = l->s1; ok = create_lane(&l->s1); continue; }
            if (!lanes_overlap(caller, l->s2) || !lanes_overlap(l->s1, l->s2)) { rejected[nrej++] = l->s2; ok = create_lane(&l->s2); continue; }
            break;
        }
        for (int i = 0; i < nrej; ++i) (void)hipStreamDestroy(rejected[i]);
        for (int i = 0; ok && i < 17; ++i) ok = hipEventCreateWithFlags(&l->ev[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) { delete l; lanes_unavailable = true; return nullptr; }
        lanes_per_device[dev] = l;
    }
    return lanes_per_device[dev];
}
// The lanes and their events are shared by every step call on a device: a call records an event and waits for it a few
// lines later, so two host threads inside step calls for the same device at the same time would wait for each other's
// records.  Each step entry point holds the device's mutex while it enqueues (uncontended: some 20 ns per call).
std::mutex step_mu[65];
struct StepLock {
    std::unique_lock<std::mutex> held;
    StepLock() {
        int dev = 64;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 64;
        held = std::unique_lock<std::mutex>(step_mu[dev]);
    }
};

// Creates the calling device's lanes now instead of inside the first step call, and makes sure that they and
// `stream` (the stream the step calls will be given) sit on different hardware queues.  HIP deals a process's
// streams onto a few hardware queues (four by default) in the order they are created, and two streams that
// share a queue run one after the other; which queue a stream gets depends on every other stream the process
// (communicator, libraries) has created before.
extern "C" int psvae_streams_init(void* stream) {
    PSVAE_TRY(noalloc_scratch_init());
    Lanes* L = get_lanes(true, (hipStream_t)stream);
    if (!L) return 0;       // lanes disabled (option streams = 0) or unavailable: every call stays on the caller's stream
    if (hipStreamSynchronize(L->s1) != hipSuccess || hipStreamSynchronize(L->s2) != hipSuccess)
        return (int)hipGetLastError();
    return 0;
}
// `to` waits for everything enqueued so far on `from`
int hand_over(hipStream_t from, hipStream_t to, hipEvent_t ev) {
    if (from == to) return 0;
    if (hipEventRecord(ev, from) != hipSuccess) return (int)hipGetLastError();
    if (hipStreamWaitEvent(to, ev, 0) != hipSuccess) return (int)hipGetLastError();
    return 0;
}

__global__ void split_mulv_kernel(const float* __restrict__ mulv, float* __restrict__ mean, float* __restrict__ lvar, int B, int Z) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * Z) return;
    const int b = t / Z, k = t - b * Z;
    mean[t] = mulv[(size_t)b * 2 * Z + k];
    lvar[t] = -fabsf(mulv[(size_t)b * 2 * Z + Z + k]);
}

// Linear layer over one row per graph: the wave-per-output kernel while the output is small (its cost is a
// few microseconds of latency), the tiled MFMA product beyond.
constexpr long long SMALL_LINEAR_OUTPUTS = 1 << 16;
int rows_linear_fwd(hipStream_t st, const float* x, int ldx, const float* wt, const float* bias, float* y, int ldy,
                    int M, int N, int K, float* scratch, size_t scratch_floats) {
    if ((long long)M * N <= SMALL_LINEAR_OUTPUTS && K % 4 == 0 && ldx % 4 == 0 && aligned16(x) && aligned16(wt))
        return small_linear_fwd(st, x, ldx, wt, bias, y, ldy, M, N, K);
    return linear_fwd(st, x, ldx, wt, bias, y, ldy, M, N, K, 0, scratch, scratch_floats);
}

// Encoder half shared by the training step and psvae_encode: x0 .. mulv.
int encoder_to_mulv(hipStream_t st, const psvae_dims& d, const psvae_params& p, const psvae_batch& b, const Sz& s,
                    const WS& w, bool with_decoder_stack, bool with_rsample, hipEvent_t packed = nullptr) {
    PSVAE_TRY(embed3_gather(st, b.atom_id, b.piece_id, b.pos_id, p.atom_emb, p.piece_emb, p.pos_emb, s.Da, s.Dp, s.Dq, w.X0, s.n));
    const float* cond = b.conds ? b.conds : w.cond;
    Stack k[2] = {enc_stack(d, p, nullptr, b, w), dec_stack(d, p, nullptr, b, w)};
    if (b.conds) {   // decoder-only: the conditions come from the caller, only the decoder-side stack runs
        PSVAE_TRY(stacks_forward(st, s, w, &k[1], 1, with_rsample, packed));
    } else if (with_decoder_stack && s.te == s.td) {
        PSVAE_TRY(stacks_forward(st, s, w, k, 2, with_rsample, packed));
    } else {
        PSVAE_TRY(stacks_forward(st, s, w, &k[0], 1, with_rsample, packed));
        if (with_decoder_stack) PSVAE_TRY(stacks_forward(st, s, w, &k[1], 1, with_rsample, packed));
    }
    const bool adjacent = p.lvar_w == p.mean_w + (size_t)s.Z * s.G && p.lvar_b == p.mean_b + s.Z;
    const bool fused_head = with_rsample && adjacent && latent_head_supported(cond, s.G, p.mean_w, s.Z, s.G);
    GemmSlabs sl{nullptr, 0, 0, 0};
    if (!b.conds) {
        PSVAE_TRY(segment_pool_fwd(st, w.XE + (size_t)s.n * s.H, s.H, (int64_t)s.n * s.H, s.H, b.graph_off, w.pooled, s.te * s.H, s.B, s.te * s.H));
        // encoder.linear; in front of the fused head a split reduction leaves its slabs to the head's kernel
        GemmDesc g;
        g.A = w.pooled; g.lda = s.te * s.H; g.B = p.enc_lin_w; g.ldb = s.te * s.H; g.C = w.cond; g.ldc = s.G;
        g.M = s.B; g.N = s.G; g.R = s.te * s.H; g.bias = p.enc_lin_b;
        g.scratch = w.scratch; g.scratch_floats = w.scratch_floats;
        if (fused_head && s.G <= 1024) g.slabs_out = &sl;
        PSVAE_TRY(gemm_f32(st, g));
    }
    if (fused_head) {    // mulv, z and the KL rows at once
        if (sl.nz > 0) return latent_head_fwd_slabs(st, sl, p.enc_lin_b, w.cond, s.G, p.mean_w, p.mean_b, b.eps, w.mulv, w.z, w.klrow, s.B, s.Z, s.G);
        return latent_head_fwd(st, cond, s.G, p.mean_w, p.mean_b, b.eps, w.mulv, w.z, w.klrow, s.B, s.Z, s.G);
    }
    if (adjacent) {
        PSVAE_TRY(rows_linear_fwd(st, cond, s.G, p.mean_w, p.mean_b, w.mulv, 2 * s.Z, s.B, 2 * s.Z, s.G, w.scratch, w.scratch_floats));
    } else {
        PSVAE_TRY(rows_linear_fwd(st, cond, s.G, p.mean_w, p.mean_b, w.mulv, 2 * s.Z, s.B, s.Z, s.G, w.scratch, w.scratch_floats));
        PSVAE_TRY(rows_linear_fwd(st, cond, s.G, p.lvar_w, p.lvar_b, w.mulv + s.Z, 2 * s.Z, s.B, s.Z, s.G, w.scratch, w.scratch_floats));
    }
    if (with_rsample) PSVAE_TRY(vae_head_fwd(st, w.mulv, b.eps, w.z, w.klrow, s.B, s.Z));
    return 0;
}

}  // namespace

extern "C" size_t psvae_workspace_bytes(const psvae_dims* dims, const psvae_batch* bs) {
    if (check_dims(dims, bs) != 0) return 0;
    const Sz s = sizes(*dims, *bs);
    return carve(s, nullptr, nullptr, nullptr, nullptr) * sizeof(float);
}

extern "C" int64_t psvae_workspace_offset(const psvae_dims* dims, const psvae_batch* bs, const char* name) {
    if (check_dims(dims, bs) != 0 || !name) return -1;
    const Sz s = sizes(*dims, *bs);
    int64_t off = -1;
    carve(s, nullptr, nullptr, name, &off);
    return off;
}

namespace {
// The weights backward's fused kernels read, in their transposed role and the matrix cores' fragment order: the GINE
// MLPs of the stacks that run the one-launch backward, the link scorer's where its one-launch backward runs.
// `*any` tells whether anything was queued.
int backward_packs(hipStream_t on, const psvae_dims& d, const psvae_params& p, const psvae_batch& b, const Sz& s, const WS& w, bool* any) {
    const bool dec_only = b.conds != nullptr;
    const bool fused_gine = fused_bwd_on(s.n, (dec_only || s.te != s.td) ? 1 : 2, s.H) && gine_layer_fused_supported(s.H, s.H, s.H);
    if (fused_gine) {
        const Stack kk[2] = {enc_stack(d, p, nullptr, b, w), dec_stack(d, p, nullptr, b, w)};
        FusedPack pk[4];
        int cnt = 0;
        for (int q = dec_only ? 1 : 0; q < 2; ++q) {
            pk[cnt++] = FusedPack{kk[q].w1, kk[q].w1tp, s.H, s.H, s.H, true, gine_layer_split(s.H)};
            pk[cnt++] = FusedPack{kk[q].w2, kk[q].w2tp, s.H, s.H, s.H, true, gine_layer_split(s.H)};
        }
        PSVAE_TRY(fused_pack(on, pk, cnt));
    }
    const bool fused_scorer = fused_scorer_bwd_on(s.S, s.H) && scorer_fused_supported(s.H, s.Z, s.H) && pair_head_supported(w.s3, s.P, p.sc_w3, w.ds3, s.P, s.P);
    if (fused_scorer) PSVAE_TRY(scorer_pack_bwd(on, p.sc_w0, p.sc_w1, p.sc_w2, w.spack + scorer_pack_floats(s.H, s.Z), s.H, s.Z));
    *any = fused_gine || fused_scorer;
    return 0;
}

// `for_training` (psvae_train_step): backward follows inside the same library call, so its packed weights are laid out
// here as well, beside the forward ones and off every dependent chain.
// `draw` (psvae_train_step, or null): batch->eps is not filled yet; its rows are drawn here, on lane 1 in front of the
// weight packs, instead of by a launch of the caller's in front of the whole step.
int step_forward(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch, void* workspace,
                 size_t workspace_bytes, const float* loss_weights, float* losses, void* stream, bool for_training,
                 const psvae_eps_draw* draw = nullptr, const float* loss_grads = nullptr) {
    PSVAE_TRY(check_dims(dims, batch));
    if (!params || !workspace || !losses) return PSVAE_E_NULL;
    if (!aligned16(workspace)) return PSVAE_E_ALIGN;
    const psvae_dims& d = *dims; const psvae_params& p = *params; const psvae_batch& b = *batch;
    const Sz s = sizes(d, b);
    WS w;
    if (carve(s, (float*)workspace, &w, nullptr, nullptr) * sizeof(float) > workspace_bytes) return PSVAE_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    StepLock one_enqueuer_per_device;
    Lanes* L = get_lanes();
    hipStream_t s1 = L ? L->s1 : st, s2 = L ? L->s2 : st;
    const size_t scf = w.scratch_floats;

    // lane 1 starts with the part of the piece branch that needs no latent: token embeddings and
    // the GRU input products for all steps (src/modules/vae_piece_decoder.py:82,87-89)
    if (L) PSVAE_TRY(hand_over(st, s1, L->ev[0]));
    // lane 1 first lays the weights of the fused kernels out in the matrix cores' fragment order (GINE MLPs of both
    // stacks, link scorer): off the dependent chain, which needs them only after the embeddings and node_trans.  (On
    // lane 1, which has just been handed over to, rather than on lane 2: a hand-over - an event record plus a stream
    // wait - costs the host 10-13 us, as much as four kernel launches.)
    const bool dec_only = b.conds != nullptr;
    const bool fused_gine = fused_layers_on() && gine_layer_fused_supported(s.H, s.H, s.H);
    const bool fused_scorer = fused_scorer_fwd_on(s.S, s.H) && scorer_fused_supported(s.H, s.Z, s.H) && pair_head_supported(w.s3, s.P, p.sc_w3, w.ds3, s.P, s.P);
    hipEvent_t packed = nullptr;
    if (draw && !b.eps) return PSVAE_E_NULL;
    if (draw) PSVAE_TRY(psvae_normal_rows(draw->seed, draw->draw, draw->first_row, s.B, s.Z, const_cast<float*>(b.eps), L ? s1 : st));
    if (L && (fused_gine || fused_scorer || for_training)) {
        if (fused_gine) {
            const Stack k[2] = {enc_stack(d, p, nullptr, b, w), dec_stack(d, p, nullptr, b, w)};
            FusedPack pk[4];
            int cnt = 0;
            for (int q = dec_only ? 1 : 0; q < 2; ++q) {
                pk[cnt++] = FusedPack{k[q].w1, k[q].w1p, s.H, s.H, s.H, false, gine_layer_split(s.H)};
                pk[cnt++] = FusedPack{k[q].w2, k[q].w2p, s.H, s.H, s.H, false, gine_layer_split(s.H)};
            }
            PSVAE_TRY(fused_pack(s1, pk, cnt));
        }
        if (fused_scorer) PSVAE_TRY(scorer_pack(s1, p.sc_w0, p.sc_w1, p.sc_w2, w.spack, s.H, s.Z));
        if (fused_gine || fused_scorer || draw) {
            if (hipEventRecord(L->ev[6], s1) != hipSuccess) return (int)hipGetLastError();
            packed = L->ev[6];
        }
        if (draw && !fused_gine) {     // nothing in front of the latent head waits for that event by itself
            if (hipStreamWaitEvent(st, packed, 0) != hipSuccess) return (int)hipGetLastError();
            packed = nullptr;
        }
        bool any = false;
        if (for_training) PSVAE_TRY(backward_packs(s1, d, p, b, s, w, &any));    // (reach backward through the join below)
    } else {
        if (fused_scorer) PSVAE_TRY(scorer_pack(st, p.sc_w0, p.sc_w1, p.sc_w2, w.spack, s.H, s.Z));
        bool any = false;
        if (for_training) PSVAE_TRY(backward_packs(st, d, p, b, s, w, &any));
    }
    PSVAE_TRY(rows_gather(s1, b.pieces, 0, s.T, s.L, p.piece_emb, s.Dp, w.semb, s.BT));
    PSVAE_TRY(linear_fwd(s1, w.semb, s.Dp, p.gru_wih, p.gru_bih, w.gi, 3 * s.R, s.BT, 3 * s.R, s.Dp, 0, w.scratch1, scf));
    PSVAE_TRY(gru_exchange_clear(s1, w.gru_exch, w.gru_exch_bytes, s.B, s.T, s.R));   // long before z arrives
    // main: encoder and decoder-side GINE stacks, pooling, latent
    PSVAE_TRY(encoder_to_mulv(st, d, p, b, s, w, true, true, fused_gine ? packed : nullptr));
    if (L) { PSVAE_TRY(hand_over(st, s1, L->ev[1])); PSVAE_TRY(hand_over(st, s2, L->ev[2])); }
    // (the scorer, the longest of the three branches, is queued first)
    // main: link scorer (src/modules/vae_piece_decoder.py:97-110)
    if (fused_scorer) {     // gather, the three Linear layers, Linear(P -> 4) and the cross-entropy rows in one launch
        if (packed && !fused_gine && hipStreamWaitEvent(st, packed, 0) != hipSuccess) return (int)hipGetLastError();
        ScorerFwd f{w.XD + (size_t)s.td * s.n * s.H, s.H, w.z, b.pair_src, b.pair_dst, b.pair_graph, b.pair_gold, w.spack,
                    p.sc_b0, p.sc_b1, p.sc_b2, p.sc_w3, p.sc_b3, w.pin, w.s1, w.s2, w.s3, w.slog, w.rowloss_e, w.lse_e,
                    s.S, s.H, s.Z};
        PSVAE_TRY(scorer_fwd_fused(st, f));     // (the packed weights were waited for before the GINE layers, or packed on this stream)
    } else {
    PSVAE_TRY(pair_gather(st, w.XD + (size_t)s.td * s.n * s.H, s.H, w.z, b.pair_src, b.pair_dst, b.pair_graph, w.pin, s.S, s.H, s.Z));
    PSVAE_TRY(linear_fwd(st, w.pin, s.P, p.sc_w0, p.sc_b0, w.s1, s.P2, s.S, s.P2, s.P, 1, w.scratch, scf));
    PSVAE_TRY(linear_fwd(st, w.s1, s.P2, p.sc_w1, p.sc_b1, w.s2, s.P2, s.S, s.P2, s.P2, 1, w.scratch, scf));
    PSVAE_TRY(linear_fwd(st, w.s2, s.P2, p.sc_w2, p.sc_b2, w.s3, s.P, s.S, s.P, s.P2, 0, w.scratch, scf));
    if (pair_head_supported(w.s3, s.P, p.sc_w3, w.ds3, s.P, s.P)) {
        PSVAE_TRY(pair_head_fwd(st, w.s3, s.P, p.sc_w3, p.sc_b3, b.pair_gold, w.slog, w.rowloss_e, w.lse_e, s.S, s.P));
    } else {
        PSVAE_TRY(linear_fwd(st, w.s3, s.P, p.sc_w3, p.sc_b3, w.slog, 4, s.S, 4, s.P, 0, w.scratch, scf));
        PSVAE_TRY(softmax_ce_fwd(st, w.slog, 4, b.pair_gold, 0, 0, s.S > 0 ? s.S : 1, s.S, 4, -1, w.rowloss_e, w.lse_e));
    }
    }
    // lane 1: GRU + vocabulary logits + piece loss rows
    PSVAE_TRY(rows_linear_fwd(s1, w.z, s.Z, p.l2h_w, p.l2h_b, w.h0, s.R, s.B, s.R, s.Z, w.scratch1, scf));
    PSVAE_TRY(gru_fwd(s1, w.gi, w.h0, p.gru_whh, p.gru_bhh, w.hseq, w.hprev, w.gates, s.B, s.T, s.R, w.gru_exch, w.gru_exch_bytes, true, b.status));
    PSVAE_TRY(linear_fwd(s1, w.hseq, s.R, p.vocab_w, p.vocab_b, w.logits, s.ldl, s.BT, s.V2, s.R, 0, w.scratch1, scf));
    if (for_training && loss_grads && softmax_ce_fused_supported(s.V2)) {
        // the piece loss's rows and, over the logits, their gradient in one launch (backward then starts at to_vocab)
        const GScale gs{loss_grads, loss_weights ? loss_weights[0] : 0.f, loss_weights ? loss_weights[1] : 0.f,
                        loss_weights ? loss_weights[2] : 0.f};
        PSVAE_TRY(softmax_ce_fwd_bwd(s1, w.logits, s.ldl, b.pieces, s.L, 1, s.T, s.BT, s.V2, d.pad_idx, w.rowloss_p, w.lse_p, gs, 1, b.denoms, 0));
    } else {
        PSVAE_TRY(softmax_ce_fwd(s1, w.logits, s.ldl, b.pieces, s.L, 1, s.T, s.BT, s.V2, d.pad_idx, w.rowloss_p, w.lse_p));
    }
    // lane 2: property head (src/modules/predictor.py:19-21) + squared error rows
    if (!dec_only) {
        PSVAE_TRY(linear_fwd(s2, w.z, s.Z, p.pred_w0, p.pred_b0, w.q1, s.Hp, s.B, s.Hp, s.Z, 1, w.scratch2, scf));
        PSVAE_TRY(linear_fwd(s2, w.q1, s.Hp, p.pred_w1, p.pred_b1, w.q2, s.Hp, s.B, s.Hp, s.Hp, 1, w.scratch2, scf));
        PSVAE_TRY(linear_fwd(s2, w.q2, s.Hp, p.pred_wo, p.pred_bo, w.pred, s.ldpred, s.B, s.NP, s.Hp, 0, w.scratch2, scf));
        PSVAE_TRY(mse_fwd(s2, w.pred, s.ldpred, b.props, w.mse_row, s.B, s.NP));
    } else {
        PSVAE_TRY(fill_zero(s2, w.mse_row, s.B));   // no property head: its loss reads 0
    }
    // (for_training: no join here.  Backward's kernels take nothing from the six loss values - their upstream gradients
    //  come from the loss weights and denominators - so the caller's stream goes straight on into backward's link-scorer
    //  chain, the lanes into backward's piece branch and property head, and backward forms the loss values on a lane
    //  once it has waited for all three streams' forward work anyway.)
    if (L && !for_training) { PSVAE_TRY(hand_over(s1, st, L->ev[3])); PSVAE_TRY(hand_over(s2, st, L->ev[4])); }
    // the recurrence's exchange buffer is left zeroed for backward's recurrence, behind the join: nothing waits for it
    PSVAE_TRY(gru_exchange_clear(s1, w.gru_exch, w.gru_exch_bytes, s.B, s.T, s.R));
    if (for_training) return 0;
    // straight into the caller's buffer (a separate 24-byte copy would be one more dependent launch)
    return losses_finalize(st, w.mse_row, s.B, w.rowloss_p, s.BT, w.rowloss_e, s.S, w.klrow, b.denoms, loss_weights, losses);
}
}  // namespace

extern "C" int psvae_step_forward(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch,
                                  void* workspace, size_t workspace_bytes, const float* loss_weights, float* losses,
                                  void* stream) {
    return step_forward(dims, params, batch, workspace, workspace_bytes, loss_weights, losses, stream, false);
}

namespace {
// `after_forward` (psvae_train_step): the forward pass of the same batch was queued by the same library call just before,
// without its closing join and without forming the loss values (`losses_out`: done here).  Its packed weights for
// backward are in place, and the lanes need no hand-over from the caller's stream to start: what their first kernels
// read (logits, predictions, the loss gradients the caller prepared before the call) was written on the lanes
// themselves or before the lanes last waited for the caller's stream.
int step_backward(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch,
                  void* workspace, size_t workspace_bytes, const float* loss_weights,
                  const float* loss_grads, const psvae_params* grads, void* stream, hipEvent_t early1, hipEvent_t early2,
                  bool after_forward = false, float* losses_out = nullptr) {
    PSVAE_TRY(check_dims(dims, batch));
    if (!params || !workspace || !loss_grads || !grads) return PSVAE_E_NULL;
    const psvae_dims& d = *dims; const psvae_params& p = *params; const psvae_params& g = *grads; const psvae_batch& b = *batch;
    const Sz s = sizes(d, b);
    WS w;
    if (carve(s, (float*)workspace, &w, nullptr, nullptr) * sizeof(float) > workspace_bytes) return PSVAE_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    StepLock one_enqueuer_per_device;
    Lanes* L = get_lanes();
    hipStream_t s1 = L ? L->s1 : st, s2 = L ? L->s2 : st;
    const size_t scf = w.scratch_floats;
    float *sc = w.scratch, *sc1 = w.scratch1, *sc2 = w.scratch2;
    const size_t nH = (size_t)s.n * s.H;

    // upstream gradients of [pred, piece, edge, kl]: derived inside the kernels that use them
    const GScale gs{loss_grads, loss_weights ? loss_weights[0] : 0.f, loss_weights ? loss_weights[1] : 0.f,
                    loss_weights ? loss_weights[2] : 0.f};
    if (L && !after_forward) { PSVAE_TRY(hand_over(st, s1, L->ev[0])); PSVAE_TRY(hand_over(st, s2, L->ev[1])); }
    // lane 2 first: the GINE MLP weights in their transposed role, laid out for the fused backward kernel (the
    // dependent chain reaches the GINE stacks hundreds of microseconds later)
    const bool dec_only_ = b.conds != nullptr;
    const bool fused_gine = fused_bwd_on(s.n, (dec_only_ || s.te != s.td) ? 1 : 2, s.H) && gine_layer_fused_supported(s.H, s.H, s.H);
    const bool fused_scorer = fused_scorer_bwd_on(s.S, s.H) && scorer_fused_supported(s.H, s.Z, s.H) && pair_head_supported(w.s3, s.P, p.sc_w3, w.ds3, s.P, s.P);
    float* spack_bwd = w.spack + scorer_pack_floats(s.H, s.Z);
    bool packs_pending = false;      // the caller's stream has to wait for event 15 before it uses the packed weights
    if (!after_forward) {
        PSVAE_TRY(backward_packs(s2, d, p, b, s, w, &packs_pending));
        if (L && packs_pending && hipEventRecord(L->ev[15], s2) != hipSuccess) return (int)hipGetLastError();
    }
    // (the recurrence's exchange buffer was zeroed by the forward call, behind its own recurrence, and backward's
    //  recurrence zeroes it again behind itself: no launch in front of the recurrence, no hand-over for it)
    // --- lane 1: piece branch (cross-entropy, to_vocab, GRU, latent_to_rnn_hidden)
    if (!(after_forward && softmax_ce_fused_supported(s.V2)))     // (after_forward: the forward pass left the gradient in place)
        PSVAE_TRY(softmax_ce_bwd(s1, w.logits, s.ldl, b.pieces, s.L, 1, s.T, s.BT, s.V2, d.pad_idx, w.lse_p, gs, 1, b.denoms, 0));
    PSVAE_TRY(linear_bwd_x(s1, w.logits, s.ldl, p.vocab_w, w.dhseq, s.R, s.BT, s.V2, s.R, nullptr, 0, 0));
    PSVAE_TRY(gru_bwd(s1, w.dhseq, w.hprev, w.gates, p.gru_whh, w.dgi, w.dgh, w.dh0, s.B, s.T, s.R, w.gru_exch, w.gru_exch_bytes, true, b.status));
    if ((long long)s.B * s.Z <= SMALL_LINEAR_OUTPUTS) PSVAE_TRY(small_linear_bwd_x(s1, w.dh0, s.R, p.l2h_w, w.dz_piece, s.Z, s.B, s.R, s.Z));
    else PSVAE_TRY(linear_bwd_x(s1, w.dh0, s.R, p.l2h_w, w.dz_piece, s.Z, s.B, s.R, s.Z, nullptr, 0, 0));
    // (dz_piece is what the caller's stream waits for in front of the latent head; the piece-embedding gradient the
    //  next product feeds is not needed before the embedding tables, at the end of the step)
    if (L && hipEventRecord(L->ev[2], s1) != hipSuccess) return (int)hipGetLastError();   // dz_piece ready
    PSVAE_TRY(gru_exchange_clear(s1, w.gru_exch, w.gru_exch_bytes, s.B, s.T, s.R));      // clean for a second backward on this forward
    PSVAE_TRY(linear_bwd_x(s1, w.dgi, 3 * s.R, p.gru_wih, w.dsemb, s.Dp, s.BT, 3 * s.R, s.Dp, nullptr, 0, 0));
    if (L && hipEventRecord(L->ev[16], s1) != hipSuccess) return (int)hipGetLastError();  // dsemb ready
    PSVAE_TRY(linear_bwd_w(s1, w.logits, s.ldl, 0, w.hseq, s.R, 0, 1, g.vocab_w, g.vocab_b, s.BT, s.V2, s.R, sc1, scf));
    PSVAE_TRY(linear_bwd_w(s1, w.dgh, 3 * s.R, 0, w.hprev, s.R, 0, 1, g.gru_whh, g.gru_bhh, s.BT, 3 * s.R, s.R, sc1, scf));
    PSVAE_TRY(linear_bwd_w(s1, w.dgi, 3 * s.R, 0, w.semb, s.Dp, 0, 1, g.gru_wih, g.gru_bih, s.BT, 3 * s.R, s.Dp, sc1, scf));
    PSVAE_TRY(linear_bwd_w(s1, w.dh0, s.R, 0, w.z, s.Z, 0, 1, g.l2h_w, g.l2h_b, s.B, s.R, s.Z, sc1, scf));
    if (early1 && hipEventRecord(early1, s1) != hipSuccess) return (int)hipGetLastError();   // piece-branch gradients final
    // --- lane 2: property head
    const bool dec_only = b.conds != nullptr;
    if (dec_only && !b.d_conds) return PSVAE_E_NULL;
    const float* cond = dec_only ? b.conds : w.cond;
    if (!dec_only) {
        PSVAE_TRY(mse_bwd(s2, w.pred, s.ldpred, b.props, gs, b.denoms, w.dpred, s.B, s.NP));
        PSVAE_TRY(linear_bwd_x(s2, w.dpred, s.ldpred, p.pred_wo, w.dq2, s.Hp, s.B, s.NP, s.Hp, w.q2, s.Hp, 0));
        PSVAE_TRY(linear_bwd_x(s2, w.dq2, s.Hp, p.pred_w1, w.dq1, s.Hp, s.B, s.Hp, s.Hp, w.q1, s.Hp, 0));
        PSVAE_TRY(linear_bwd_x(s2, w.dq1, s.Hp, p.pred_w0, w.dz, s.Z, s.B, s.Hp, s.Z, nullptr, 0, 0));
    }
    if (L && hipEventRecord(L->ev[3], s2) != hipSuccess) return (int)hipGetLastError();   // dz (property head part) ready
    if (!dec_only) {
        PSVAE_TRY(linear_bwd_w(s2, w.dpred, s.ldpred, 0, w.q2, s.Hp, 0, 1, g.pred_wo, g.pred_bo, s.B, s.NP, s.Hp, sc2, scf));
        PSVAE_TRY(linear_bwd_w(s2, w.dq2, s.Hp, 0, w.q1, s.Hp, 0, 1, g.pred_w1, g.pred_b1, s.B, s.Hp, s.Hp, sc2, scf));
        PSVAE_TRY(linear_bwd_w(s2, w.dq1, s.Hp, 0, w.z, s.Z, 0, 1, g.pred_w0, g.pred_b0, s.B, s.Hp, s.Z, sc2, scf));
    }
    // --- main: link scorer input-gradient chain; its weight gradients follow on lane 2
    const bool fused_pair_head = pair_head_supported(w.s3, s.P, p.sc_w3, w.ds3, s.P, s.P);
    if (fused_pair_head) PSVAE_TRY(pair_head_bwd(st, w.slog, w.lse_e, b.pair_gold, p.sc_w3, gs, b.denoms, w.ds3, s.P, s.S, s.P));
    else PSVAE_TRY(softmax_ce_bwd(st, w.slog, 4, b.pair_gold, 0, 0, s.S > 0 ? s.S : 1, s.S, 4, -1, w.lse_e, gs, 2, b.denoms, 1));
    if (fused_scorer) {
        // the three input-gradient products of the scorer's MLP in one launch; lane 2 takes the weight gradients
        if (L) PSVAE_TRY(hand_over(st, s2, L->ev[4]));
        PSVAE_TRY(linear_bwd_w(s2, w.slog, 4, 0, w.s3, s.P, 0, 1, g.sc_w3, g.sc_b3, s.S, 4, s.P, sc2, scf));
        PSVAE_TRY(linear_bwd_w(s2, w.ds3, s.P, 0, w.s2, s.P2, 0, 1, g.sc_w2, g.sc_b2, s.S, s.P, s.P2, sc2, scf));
        if (L && packs_pending && hipStreamWaitEvent(st, L->ev[15], 0) != hipSuccess) return (int)hipGetLastError();
        ScorerBwd sb{w.ds3, w.s2, w.s1, spack_bwd, w.ds2, w.ds1, w.dpin, s.S, s.H, s.Z};
        PSVAE_TRY(scorer_bwd_fused(st, sb));
        if (L) PSVAE_TRY(hand_over(st, s2, L->ev[5]));
        PSVAE_TRY(linear_bwd_w(s2, w.ds2, s.P2, 0, w.s1, s.P2, 0, 1, g.sc_w1, g.sc_b1, s.S, s.P2, s.P2, sc2, scf));
        PSVAE_TRY(linear_bwd_w(s2, w.ds1, s.P2, 0, w.pin, s.P, 0, 1, g.sc_w0, g.sc_b0, s.S, s.P2, s.P, sc2, scf));
    } else {
    // (With few pairs the four weight gradients are queued behind the whole input-gradient chain instead of each
    //  behind its own link: the chain, which the caller's stream waits for, then runs beside fewer competing
    //  workgroups, and three of four hand-overs go - +0.5 % at 128 graphs, -0.7 % at 256.)
    const bool dw_late = s.S <= 6144 && L;
    if (!dw_late) {
        if (L) PSVAE_TRY(hand_over(st, s2, L->ev[4]));
        PSVAE_TRY(linear_bwd_w(s2, w.slog, 4, 0, w.s3, s.P, 0, 1, g.sc_w3, g.sc_b3, s.S, 4, s.P, sc2, scf));
    }
    if (!fused_pair_head) PSVAE_TRY(linear_bwd_x(st, w.slog, 4, p.sc_w3, w.ds3, s.P, s.S, 4, s.P, nullptr, 0, 0));
    if (dw_late) {
        PSVAE_TRY(linear_bwd_x(st, w.ds3, s.P, p.sc_w2, w.ds2, s.P2, s.S, s.P, s.P2, w.s2, s.P2, 0));
        PSVAE_TRY(linear_bwd_x(st, w.ds2, s.P2, p.sc_w1, w.ds1, s.P2, s.S, s.P2, s.P2, w.s1, s.P2, 0));
        PSVAE_TRY(linear_bwd_x(st, w.ds1, s.P2, p.sc_w0, w.dpin, s.P, s.S, s.P2, s.P, nullptr, 0, 0));
        PSVAE_TRY(hand_over(st, s2, L->ev[7]));      // one hand-over for all four weight gradients
        PSVAE_TRY(linear_bwd_w(s2, w.slog, 4, 0, w.s3, s.P, 0, 1, g.sc_w3, g.sc_b3, s.S, 4, s.P, sc2, scf));
        PSVAE_TRY(linear_bwd_w(s2, w.ds3, s.P, 0, w.s2, s.P2, 0, 1, g.sc_w2, g.sc_b2, s.S, s.P, s.P2, sc2, scf));
        PSVAE_TRY(linear_bwd_w(s2, w.ds2, s.P2, 0, w.s1, s.P2, 0, 1, g.sc_w1, g.sc_b1, s.S, s.P2, s.P2, sc2, scf));
        PSVAE_TRY(linear_bwd_w(s2, w.ds1, s.P2, 0, w.pin, s.P, 0, 1, g.sc_w0, g.sc_b0, s.S, s.P2, s.P, sc2, scf));
    } else {
    if (L) PSVAE_TRY(hand_over(st, s2, L->ev[5]));
    PSVAE_TRY(linear_bwd_w(s2, w.ds3, s.P, 0, w.s2, s.P2, 0, 1, g.sc_w2, g.sc_b2, s.S, s.P, s.P2, sc2, scf));
    PSVAE_TRY(linear_bwd_x(st, w.ds3, s.P, p.sc_w2, w.ds2, s.P2, s.S, s.P, s.P2, w.s2, s.P2, 0));
    if (L) PSVAE_TRY(hand_over(st, s2, L->ev[6]));
    PSVAE_TRY(linear_bwd_w(s2, w.ds2, s.P2, 0, w.s1, s.P2, 0, 1, g.sc_w1, g.sc_b1, s.S, s.P2, s.P2, sc2, scf));
    PSVAE_TRY(linear_bwd_x(st, w.ds2, s.P2, p.sc_w1, w.ds1, s.P2, s.S, s.P2, s.P2, w.s1, s.P2, 0));
    if (L) PSVAE_TRY(hand_over(st, s2, L->ev[7]));
    PSVAE_TRY(linear_bwd_w(s2, w.ds1, s.P2, 0, w.pin, s.P, 0, 1, g.sc_w0, g.sc_b0, s.S, s.P2, s.P, sc2, scf));
    PSVAE_TRY(linear_bwd_x(st, w.ds1, s.P2, p.sc_w0, w.dpin, s.P, s.S, s.P2, s.P, nullptr, 0, 0));
    }
    }
    PSVAE_TRY(pair_scatter(st, w.dpin, s.P, b.pair_by_src_ptr, b.pair_by_src_idx, b.pair_by_dst_ptr, b.pair_by_dst_idx,
                           b.pair_graph_off, w.GD + (size_t)s.td * nH, s.H, w.dz_pair, 0, s.n, s.B, s.H, s.Z));
    // --- latent head and encoder.linear: needs the three parts of dL/dz
    if (L && (hipStreamWaitEvent(st, L->ev[2], 0) != hipSuccess || hipStreamWaitEvent(st, L->ev[3], 0) != hipSuccess))
        return (int)hipGetLastError();
    float* dcond = dec_only ? b.d_conds : w.dcond;
    const bool head_adjacent = p.lvar_w == p.mean_w + (size_t)s.Z * s.G && p.lvar_b == p.mean_b + s.Z &&
                               g.lvar_w == g.mean_w + (size_t)s.Z * s.G && g.lvar_b == g.mean_b + s.Z;
    const bool fused_head = head_adjacent && 2 * s.Z <= 1024;    // d_mulv and d_cond in one launch
    const float* dz_a = dec_only ? w.dz_piece : w.dz;
    const float* dz_b = dec_only ? w.dz_pair : w.dz_piece;
    const float* dz_c = dec_only ? b.d_z : w.dz_pair;
    if (fused_head) PSVAE_TRY(latent_head_bwd(st, w.mulv, b.eps, dz_a, dz_b, dz_c, gs, b.denoms, p.mean_w, w.dmulv, dcond, s.G, s.B, s.Z, s.G));
    else PSVAE_TRY(vae_head_bwd(st, w.mulv, b.eps, dz_a, dz_b, dz_c, gs, b.denoms, w.dmulv, s.B, s.Z));
    // (the weight gradients of these small layers go to lane 2: nothing on the chain waits for them)
    if (L) PSVAE_TRY(hand_over(st, s2, L->ev[12]));
    // The loss values the forward pass of psvae_train_step left open: by now the caller's stream has waited for lane 1's
    // piece branch, and lane 2, which has just waited for the caller's stream, holds the property head's rows itself.
    if (losses_out) PSVAE_TRY(losses_finalize(s2, w.mse_row, s.B, w.rowloss_p, s.BT, w.rowloss_e, s.S, w.klrow, b.denoms, loss_weights, losses_out));
    if (head_adjacent) {
        PSVAE_TRY(linear_bwd_w(s2, w.dmulv, 2 * s.Z, 0, cond, s.G, 0, 1, g.mean_w, g.mean_b, s.B, 2 * s.Z, s.G, sc2, scf));
        if (!fused_head) PSVAE_TRY(linear_bwd_x(st, w.dmulv, 2 * s.Z, p.mean_w, dcond, s.G, s.B, 2 * s.Z, s.G, nullptr, 0, 0));
    } else {
        PSVAE_TRY(linear_bwd_w(s2, w.dmulv, 2 * s.Z, 0, cond, s.G, 0, 1, g.mean_w, g.mean_b, s.B, s.Z, s.G, sc2, scf));
        PSVAE_TRY(linear_bwd_w(s2, w.dmulv + s.Z, 2 * s.Z, 0, cond, s.G, 0, 1, g.lvar_w, g.lvar_b, s.B, s.Z, s.G, sc2, scf));
        PSVAE_TRY(linear_bwd_x(st, w.dmulv, 2 * s.Z, p.mean_w, dcond, s.G, s.B, s.Z, s.G, nullptr, 0, 0));
        PSVAE_TRY(linear_bwd_x(st, w.dmulv + s.Z, 2 * s.Z, p.lvar_w, dcond, s.G, s.B, s.Z, s.G, nullptr, 0, 1));
    }
    const int TH = s.te * s.H;
    if (!dec_only) {
        // (with the fused head d_cond was final before the hand-over above; otherwise it has just been written)
        if (L && !fused_head) PSVAE_TRY(hand_over(st, s2, L->ev[13]));
        PSVAE_TRY(linear_bwd_w(s2, w.dcond, s.G, 0, w.pooled, TH, 0, 1, g.enc_lin_w, g.enc_lin_b, s.B, s.G, TH, sc2, scf));
        PSVAE_TRY(linear_bwd_x(st, w.dcond, s.G, p.enc_lin_w, w.dpooled, TH, s.B, s.G, TH, nullptr, 0, 0));
    }
    // predictor, scorer, latent head and encoder.linear gradients final (all queued on lane 2 by now)
    if (early2 && hipEventRecord(early2, s2) != hipSuccess) return (int)hipGetLastError();
    // --- GINE stacks: dL/dx_te of the encoder is the broadcast of the last pooled slice
    Stack k[2] = {enc_stack(d, p, &g, b, w), dec_stack(d, p, &g, b, w)};
    // (with the cut-operand fused backward the top layer's launch forms the broadcast itself: one launch fewer on the chain)
    const bool top_from_pool = !dec_only && fused_gine && gine_layer_split(s.H);
    if (!dec_only && !top_from_pool) {
        AggBwdGroup top{nullptr, nullptr, w.GE + (size_t)s.te * nH, nullptr, nullptr, nullptr, nullptr, nullptr, 1.f,
                        w.dpooled + (size_t)(s.te - 1) * s.H, 0, nullptr};
        PSVAE_TRY(gine_agg_bwd(st, &top, 1, s.H, s.H, s.H, TH, b.node_graph, s.n, s.H));
    }
    const float* dpool[2] = {w.dpooled, nullptr};
    if (L && fused_gine && packs_pending && hipStreamWaitEvent(st, L->ev[15], 0) != hipSuccess) return (int)hipGetLastError();
    if (dec_only) {
        PSVAE_TRY(stacks_backward(st, s, w, b, &k[1], 1, &dpool[1], 1, 1));
    } else if (s.te == s.td) {
        PSVAE_TRY(stacks_backward(st, s, w, b, k, 2, dpool, 0, 0, top_from_pool));
    } else {
        PSVAE_TRY(stacks_backward(st, s, w, b, &k[0], 1, &dpool[0], 0, 0, top_from_pool));
        PSVAE_TRY(stacks_backward(st, s, w, b, &k[1], 1, &dpool[1], 1, 1));
    }
    // weight gradients of the shared GINE MLPs (sum over the t applications): encoder on lane 1,
    // decoder-side encoder on lane 2, beside node_trans / edge_trans / embedding tables on main
    if (L) { PSVAE_TRY(hand_over(st, s1, L->ev[8])); PSVAE_TRY(hand_over(st, s2, L->ev[9])); }
    // (the edge-table sums - two short launches of a few workgroups - go to lane 2 in front of its products while the
    //  batch is small: the caller's stream, which still has node_trans and the embedding tables to do, is then the
    //  last of the three to finish, +1.3 % at 128 graphs per step; at 1024 graphs the second of the two launches waits
    //  behind the long-running weight-gradient workgroups wherever it is queued, and in front of lane 2's products
    //  it delays them: -0.6 %.)
    hipStream_t se = s.n <= 12288 ? s2 : st;
    if (dec_only) {
        PSVAE_TRY(stacks_edge_grads(se, s, w, &k[1], 1, 1));
    } else if (s.te == s.td) {
        PSVAE_TRY(stacks_edge_grads(se, s, w, k, 2, 0));
    } else {
        PSVAE_TRY(stacks_edge_grads(se, s, w, &k[0], 1, 0));
        PSVAE_TRY(stacks_edge_grads(se, s, w, &k[1], 1, 1));
    }
    for (int q = dec_only ? 1 : 0; q < 2; ++q) {
        hipStream_t sq = q == 0 ? s1 : s2;
        float* scq = q == 0 ? sc1 : sc2;
        PSVAE_TRY(linear_bwd_w(sq, k[q].Gr + nH, s.H, (int64_t)nH, k[q].Hid, s.H, (int64_t)nH, k[q].t, k[q].g_w2, k[q].g_b2,
                               s.n, s.H, s.H, scq, scf));
        PSVAE_TRY(linear_bwd_w(sq, k[q].DH, s.H, (int64_t)nH, k[q].A, s.H, (int64_t)nH, k[q].t, k[q].g_w1, k[q].g_b1,
                               s.n, s.H, s.H, scq, scf));
    }
    // --- node_trans of both stacks from G0 = [dL/dx0_enc | dL/dx0_dec]  ([n, 2H])
    const bool node_adjacent = p.dec_node_w == p.enc_node_w + (size_t)s.H * s.Din && g.dec_node_w == g.enc_node_w + (size_t)s.H * s.Din &&
                               g.dec_node_b == g.enc_node_b + s.H;
    if (dec_only) {
        PSVAE_TRY(linear_bwd_x(st, w.G0 + s.H, 2 * s.H, p.dec_node_w, w.dX0, s.Din, s.n, s.H, s.Din, nullptr, 0, 0));
        PSVAE_TRY(linear_bwd_w(st, w.G0 + s.H, 2 * s.H, 0, w.X0, s.Din, 0, 1, g.dec_node_w, g.dec_node_b, s.n, s.H, s.Din, sc, scf));
    } else if (node_adjacent) {
        PSVAE_TRY(linear_bwd_x(st, w.G0, 2 * s.H, p.enc_node_w, w.dX0, s.Din, s.n, 2 * s.H, s.Din, nullptr, 0, 0));
        PSVAE_TRY(linear_bwd_w(st, w.G0, 2 * s.H, 0, w.X0, s.Din, 0, 1, g.enc_node_w, g.enc_node_b, s.n, 2 * s.H, s.Din, sc, scf));
    } else {
        PSVAE_TRY(linear_bwd_x(st, w.G0, 2 * s.H, p.enc_node_w, w.dX0, s.Din, s.n, s.H, s.Din, nullptr, 0, 0));
        PSVAE_TRY(linear_bwd_x(st, w.G0 + s.H, 2 * s.H, p.dec_node_w, w.dX0, s.Din, s.n, s.H, s.Din, nullptr, 0, 1));
        PSVAE_TRY(linear_bwd_w(st, w.G0, 2 * s.H, 0, w.X0, s.Din, 0, 1, g.enc_node_w, g.enc_node_b, s.n, s.H, s.Din, sc, scf));
        PSVAE_TRY(linear_bwd_w(st, w.G0 + s.H, 2 * s.H, 0, w.X0, s.Din, 0, 1, g.dec_node_w, g.dec_node_b, s.n, s.H, s.Din, sc, scf));
    }
    // --- embedding tables (piece table also receives the GRU-input gradient)
    const TableGradDesc tg[3] = {
        {g.atom_emb, d.num_atom, s.Da, b.atom_rows_ptr, b.atom_rows_idx, w.dX0, s.Din, 0, nullptr, nullptr, nullptr, 0, 0},
        {g.piece_emb, d.num_piece, s.Dp, b.piece_rows_ptr, b.piece_rows_idx, w.dX0, s.Din, s.Da,
         b.seq_rows_ptr, b.seq_rows_idx, w.dsemb, s.Dp, 0},
        {g.pos_emb, d.max_pos, s.Dq, b.pos_rows_ptr, b.pos_rows_idx, w.dX0, s.Din, s.Da + s.Dp, nullptr, nullptr, nullptr, 0, 0}};
    if (L && hipStreamWaitEvent(st, L->ev[16], 0) != hipSuccess) return (int)hipGetLastError();
    PSVAE_TRY(table_grads(st, tg, 3, w.tg_part, w.tg_part_floats));
    if (L) { PSVAE_TRY(hand_over(s1, st, L->ev[10])); PSVAE_TRY(hand_over(s2, st, L->ev[11])); }
    return 0;
}
}  // namespace

extern "C" int psvae_step_backward(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch,
                                   void* workspace, size_t workspace_bytes, const float* loss_weights,
                                   const float* loss_grads, const psvae_params* grads, void* stream) {
    return step_backward(dims, params, batch, workspace, workspace_bytes, loss_weights, loss_grads, grads, stream, nullptr, nullptr);
}

extern "C" int psvae_step_backward_signal(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch,
                                          void* workspace, size_t workspace_bytes, const float* loss_weights,
                                          const float* loss_grads, const psvae_params* grads, void* stream,
                                          void* early_event_1, void* early_event_2) {
    return step_backward(dims, params, batch, workspace, workspace_bytes, loss_weights, loss_grads, grads, stream,
                         (hipEvent_t)early_event_1, (hipEvent_t)early_event_2);
}

extern "C" int psvae_train_step(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch, void* workspace,
                                size_t workspace_bytes, const float* loss_weights, float* losses, const float* loss_grads,
                                const psvae_params* grads, const psvae_adam* opt, const psvae_eps_draw* eps_draw,
                                void* early_event_1, void* early_event_2, void* stream) {
    PSVAE_TRY(check_dims(dims, batch));
    // Forward and backward are queued as one sequence - no join between them, the lanes running on from forward into
    // backward, the loss values formed on a lane late in backward - while the recurrence runs on the short multi-CU
    // kernels (measured: +3.5 % at 128 graphs, +1 % at 256 and 512).  With the long single-CU recurrences of larger
    // batches the caller's stream would start backward's one-per-CU link-scorer launch beside the forward recurrence and
    // starve it: -3.5 % at 1024 graphs; there the call is the plain sequence of forward, backward and optimiser.
    if (!gru_multi_cu_selected(batch->B, dims->rnn)) {
        if (eps_draw) {
            if (!batch->eps) return PSVAE_E_NULL;
            PSVAE_TRY(psvae_normal_rows(eps_draw->seed, eps_draw->draw, eps_draw->first_row, batch->B, dims->latent, const_cast<float*>(batch->eps), stream));
        }
        PSVAE_TRY(step_forward(dims, params, batch, workspace, workspace_bytes, loss_weights, losses, stream, false, nullptr));
        PSVAE_TRY(step_backward(dims, params, batch, workspace, workspace_bytes, loss_weights, loss_grads, grads, stream,
                                (hipEvent_t)early_event_1, (hipEvent_t)early_event_2, false, nullptr));
    } else {
    if (!loss_grads) return PSVAE_E_NULL;
    PSVAE_TRY(step_forward(dims, params, batch, workspace, workspace_bytes, loss_weights, losses, stream, true, eps_draw, loss_grads));
    PSVAE_TRY(step_backward(dims, params, batch, workspace, workspace_bytes, loss_weights, loss_grads, grads, stream,
                            (hipEvent_t)early_event_1, (hipEvent_t)early_event_2, true, losses));
    }
    if (opt) {
        if (!opt->params || !opt->grads || !opt->exp_avg || !opt->exp_avg_sq || !opt->partials) return PSVAE_E_NULL;
        PSVAE_TRY(psvae_clip_adam(opt->params, opt->grads, opt->exp_avg, opt->exp_avg_sq, opt->count, opt->partials, opt->max_norm,
                                  opt->lr, opt->beta1, opt->beta2, opt->eps, opt->bias1, opt->bias2, opt->grad_norm_out, stream));
    }
    return 0;
}

extern "C" int psvae_encode(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch,
                            void* workspace, size_t workspace_bytes, float* z_mean, float* z_log_var, void* stream) {
    PSVAE_TRY(check_dims(dims, batch));
    if (!params || !workspace || !z_mean || !z_log_var) return PSVAE_E_NULL;
    const Sz s = sizes(*dims, *batch);
    WS w;
    if (carve(s, (float*)workspace, &w, nullptr, nullptr) * sizeof(float) > workspace_bytes) return PSVAE_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    PSVAE_TRY(encoder_to_mulv(st, *dims, *params, *batch, s, w, false, false));
    hipLaunchKernelGGL(split_mulv_kernel, dim3((s.B * s.Z + 255) / 256), dim3(256), 0, st, w.mulv, z_mean, z_log_var, s.B, s.Z);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

// Link scores of assembled fragments (src/modules/vae_piece_decoder.py:203-238): atom embeddings, the
// decoder-side GINE stack over the in-piece bonds, then the scorer over the batch's candidate pairs.
extern "C" int psvae_score_links(const psvae_dims* dims, const psvae_params* params, const psvae_batch* batch,
                                 const float* z, void* workspace, size_t workspace_bytes, float* probs, float* confidence,
                                 int32_t* bond_type, void* stream) {
    PSVAE_TRY(check_dims(dims, batch));
    if (!params || !workspace || !z) return PSVAE_E_NULL;
    if (!aligned16(workspace) || !aligned16(z) || (probs && !aligned16(probs))) return PSVAE_E_ALIGN;
    const psvae_dims& d = *dims; const psvae_params& p = *params; const psvae_batch& b = *batch;
    const Sz s = sizes(d, b);
    WS w;
    if (carve(s, (float*)workspace, &w, nullptr, nullptr) * sizeof(float) > workspace_bytes) return PSVAE_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t scf = w.scratch_floats;
    PSVAE_TRY(embed3_gather(st, b.atom_id, b.piece_id, b.pos_id, p.atom_emb, p.piece_emb, p.pos_emb, s.Da, s.Dp, s.Dq, w.X0, s.n));
    const Stack k = dec_stack(d, p, nullptr, b, w);
    PSVAE_TRY(stacks_forward(st, s, w, &k, 1, false));
    if (s.S == 0) return 0;
    PSVAE_TRY(pair_gather(st, w.XD + (size_t)s.td * s.n * s.H, s.H, z, b.pair_src, b.pair_dst, b.pair_graph, w.pin, s.S, s.H, s.Z));
    PSVAE_TRY(linear_fwd(st, w.pin, s.P, p.sc_w0, p.sc_b0, w.s1, s.P2, s.S, s.P2, s.P, 1, w.scratch, scf));
    PSVAE_TRY(linear_fwd(st, w.s1, s.P2, p.sc_w1, p.sc_b1, w.s2, s.P2, s.S, s.P2, s.P2, 1, w.scratch, scf));
    PSVAE_TRY(linear_fwd(st, w.s2, s.P2, p.sc_w2, p.sc_b2, w.s3, s.P, s.S, s.P, s.P2, 0, w.scratch, scf));
    PSVAE_TRY(linear_fwd(st, w.s3, s.P, p.sc_w3, p.sc_b3, w.slog, 4, s.S, 4, s.P, 0, w.scratch, scf));
    return link_softmax(st, w.slog, probs, confidence, bond_type, s.S);
}
