// One GINE application as ONE launch (src/modules/encoder.py:57-59 with torch_geometric's GINEConv and the
// two-layer MLP of src/modules/common_nn.py:8-23):
//     agg_i = (1+eps) x_i + sum_{e: dst=i} relu(x_src(e) + edge(type_e))        message + CSR segment-sum
//     hid   = relu(agg W1^T + b1)                                                Linear + ReLU
//     y     = hid W2^T + b2                                                      Linear
// A workgroup owns 32 consecutive nodes.  It builds their aggregated rows straight into an LDS image [32][K]
// (the A operand of the first product; also streamed to memory when backward needs it), multiplies the image with
// W1, writes the activated result back into the same LDS image, multiplies that with W2 and stores the rows of y.
// Neither agg nor hid is ever read back from memory by the forward pass.
//
// The products run on v_mfma_f32_16x16x4_f32 (exact fp32).  The A operand stays in LDS for the whole reduction; the
// B operand (the weights) goes from L2 straight into registers, from a PACKED copy that pack_weights() writes
// once per step: Wp[kb][q][c][j] = W[c][16 kb + 4 q + j] (zero beyond the matrix), so that lane (column c = l & 15,
// quarter q = l >> 4) fetches ONE float4 whose element j feeds the j-th MFMA of the block (the A operand takes the
// same k: one ds_read_b128 of the image row), and the 16 lanes of a quarter read 256 contiguous bytes.  (Read from
// the row-major matrix, the same fragment is 64 separate 16-byte requests per instruction, 16 rows apart, and the
// address path, not the matrix pipe, sets the pace: 40-55 % MFMA utilisation measured.)  A block of 16 reduction
// indices costs a wave 2 LDS reads and CT global loads for 8 CT MFMAs.  Nothing is staged cooperatively, hence no
// barrier inside a product: the four waves (each owning CT 16-column tiles of all 32 rows) run independently, with
// the weights prefetched two blocks ahead.  Image pitch = 8 mod 64 floats: conflict-free ds_read_b128 for this
// lane map (rows two apart land on disjoint 16-byte bank slots inside every 16-lane service group).
#include "psvae_internal.h"
#include "fused_device.h"
#include <algorithm>

namespace {

constexpr int FL_ROWS = 32, FL_RT = 2;

// Packed weights: [KB][4][Npad][4] floats, KB = ceil(K / 16) blocks, Npad = 64 CT columns (CT 16-column tiles for
// each of the four waves).  `transposed`: the source holds W^T ([K][N] row-major, i.e. the product is x W instead
// of x W^T: the input-gradient form).
struct PackOne { const float* src; float* dst; int N, K, ld, transposed, Npad, KB; };
struct PackArgs { PackOne m[8]; int count; };

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs a) {
    const PackOne& m = a.m[blockIdx.y];
    const int total = m.KB * 4 * m.Npad;           // float4 slots
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int c = t % m.Npad, kbq = t / m.Npad, k = kbq * 4;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = c < m.N && k + j < m.K;
            const size_t idx = m.transposed ? (size_t)min(k + j, m.K - 1) * m.ld + min(c, m.N - 1)
                                            : (size_t)min(c, m.N - 1) * m.ld + min(k + j, m.K - 1);
            const float x = m.src[idx];
            v[j] = ok ? x : 0.f;
        }
        *reinterpret_cast<float4*>(m.dst + (size_t)t * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// acc[rt][ct] += img[rows of tile rt][0 .. 16 KB) . W[columns of tile ct][same k], W packed as above.  RT 16-row
// tiles per workgroup (image rows rt * 16 + (l & 15)), CT 16-column tiles per wave.
// The weights of a product's first two blocks, requested ahead of the phase that precedes the product (they depend
// on nothing the kernel computes): their L2 round trip then runs under that phase instead of in front of the
// product's first MFMA.
template <int CT>
__device__ __forceinline__ void fl_prefetch(const float* __restrict__ Wp, int KB, int wave, int lane, float4 (&h0)[CT], float4 (&h1)[CT]) {
    constexpr int Npad = 64 * CT;
    const float* wlane = Wp + ((size_t)(lane >> 4) * Npad + wave * CT * 16 + (lane & 15)) * 4;
    const float* p1 = wlane + (size_t)min(1, KB - 1) * (16 * Npad);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) h0[ct] = fl_ld4(wlane + ct * 64);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) h1[ct] = fl_ld4(p1 + ct * 64);
}

template <int CT, int RT, bool PRE = false>
__device__ __forceinline__ void fl_product(const float* img, int P, const float* __restrict__ Wp, int KB, int wave, int lane,
                                           floatx4 (&acc)[RT][CT], float4 (&b0)[CT], float4 (&b1)[CT]) {
    // (Every row's reduction runs in the same order wherever the row sits in the batch, so a graph's results do not
    // depend on its neighbours.  Starting each workgroup at a different block of the reduction, to keep the
    // workgroups from asking for the same lines of the matrix at the same moment, was measured and gains nothing.)
    const int nl = lane & 15, kq = lane >> 4;
    constexpr int Npad = 64 * CT;
    const float* wlane = Wp + ((size_t)kq * Npad + wave * CT * 16 + nl) * 4;      // + kb * 16 * Npad + ct * 64
    const float* arow = img + nl * P + 4 * kq;                                      // + rt * 16 * P + kb * 16
    float4 b2[CT], a0[RT], a1[RT], a2[RT];
    // blocks past the last one are clamped to it (fetched, never used)
#define FL_LOADB(B_, KB_)                                                                          \
    {                                                                                              \
        const float* p_ = wlane + (size_t)min((KB_), KB - 1) * (16 * Npad);                        \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) B_[ct] = fl_ld4(p_ + ct * 64);           \
    }
#define FL_READA(A_, KB_)                                                                                   \
    {                                                                                                       \
        const int o_ = min((KB_), KB - 1) * 16;                                                             \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                   \
            A_[rt] = *reinterpret_cast<const float4*>(arow + rt * 16 * P + o_);                             \
    }
#define FL_MFMA(A_, B_)                                                                                         \
    {                                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                                 \
                const float bv = j == 0 ? B_[ct].x : (j == 1 ? B_[ct].y : (j == 2 ? B_[ct].z : B_[ct].w));      \
                _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                             \
                    const float av = j == 0 ? A_[rt].x : (j == 1 ? A_[rt].y : (j == 2 ? A_[rt].z : A_[rt].w));  \
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[rt][ct], 0, 0, 0);           \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    }
    // Stage s (registers a_s, b_s) holds block kb = s (mod 3).  Every block's weights are requested two blocks
    // ahead and its image rows one block ahead; the scheduling barriers keep the requests in program order, so that
    // each wait is for the oldest outstanding group only.
    if (!PRE) {
        FL_LOADB(b0, 0)
        __builtin_amdgcn_sched_barrier(0);
        FL_LOADB(b1, 1)
    }
    FL_READA(a0, 0)
    __builtin_amdgcn_sched_barrier(0);
    for (int kb = 0; kb < KB; kb += 3) {        // (the guards are uniform and hold no loads)
        FL_LOADB(b2, kb + 2)
        FL_READA(a1, kb + 1)
        __builtin_amdgcn_sched_barrier(0);
        FL_MFMA(a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        FL_LOADB(b0, kb + 3)
        FL_READA(a2, kb + 2)
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < KB) FL_MFMA(a1, b1)
        __builtin_amdgcn_sched_barrier(0);
        FL_LOADB(b1, kb + 4)
        FL_READA(a0, kb + 3)
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 2 < KB) FL_MFMA(a2, b2)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef FL_READA
#undef FL_MFMA
#undef FL_LOADB
}

template <int CT, int RT>
__device__ __forceinline__ void fl_product(const float* img, int P, const float* __restrict__ Wp, int KB, int wave, int lane,
                                           floatx4 (&acc)[RT][CT]) {
    float4 b0[CT], b1[CT];
    fl_product<CT, RT, false>(img, P, Wp, KB, wave, lane, acc, b0, b1);
}

// What a product leaves in the accumulators, plus bias (and ReLU), written to columns col0 + c of an LDS image
// and/or of global rows row0 + r (< nrows).  An accumulator holds rows 4 (l >> 4) + i, column l & 15 of its tile.
// ZERO_PAD: image columns from N up to the wave tiles' end are written as zeros (they are the next product's
// reduction padding); otherwise columns >= N are left alone.
template <int CT, int RT, bool RELU, bool ZERO_PAD>
__device__ __forceinline__ void fl_store(const floatx4 (&acc)[RT][CT], const float (&bias)[CT], int N, float* img, int P, int col0,
                                         float* gout, size_t ldg, int row0, int nrows, int wave, int lane) {
    const int nl = lane & 15, kq = lane >> 4;
    const bool full = row0 + 16 * RT <= nrows;      // uniform: every row of the tile exists (all tiles but the last)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = (wave * CT + ct) * 16 + nl;
        const bool cok = c < N;
        float v[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = acc[rt][ct][i] + bias[ct];
                if (RELU) x = fmaxf(x, 0.f);
                v[rt][i] = cok ? x : 0.f;
            }
        // one predicate per column tile instead of one per element
        if (img && (cok || (ZERO_PAD && col0 + c < P))) {
            float* q = img + (4 * kq) * P + col0 + c;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) q[(rt * 16 + i) * P] = v[rt][i];
        }
        if (gout && cok) {
            float* q = gout + (size_t)(row0 + 4 * kq) * ldg + col0 + c;
            if (full) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) q[(size_t)(rt * 16 + i) * ldg] = v[rt][i];
            } else {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (row0 + rt * 16 + 4 * kq + i < nrows) q[(size_t)(rt * 16 + i) * ldg] = v[rt][i];
            }
        }
    }
}

struct FLGroup {
    const float* x;                               // layer input [n, ldx]
    const int32_t *rowptr, *col, *etype;          // CSR by target, sources in edge order
    const float *edge_w, *edge_b;                 // edge_trans [H,4], [H]
    const float *w1p, *b1, *w2p, *b2;             // conv.nn.seq.{0,2}: packed weights, biases
    float *agg, *hid;                             // [n, H] each, saved for backward; may be null
    float* y;                                     // [n, ldy]
    float ope;
};
struct FLArgs { FLGroup g[2]; int n, H, P, ldx, ldy; };

// SAVE: agg and hid are also written to memory (training); without it nothing but y leaves the workgroup.
template <int CT, bool SAVE>
__global__ __launch_bounds__(FL_THREADS) void gine_layer_fwd_kernel(const FLArgs a) {
    extern __shared__ __attribute__((aligned(16))) float fl_lds[];
    __shared__ int s_e0[FL_ROWS], s_deg[FL_ROWS], s_nb[FL_ROWS][FL_NB], s_ty[FL_ROWS][FL_NB];
    const FLGroup& g = a.g[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, P = a.P, H4 = H >> 2, n = a.n, KB = (H + 15) >> 4;
    float* img = fl_lds;                   // [32][P]
    float* etab = fl_lds + FL_ROWS * P;    // [5][H]: edge term per bond type, row 4 = -inf ("no edge")
    // workgroups go to the 8 XCDs round-robin: neighbouring row tiles (which share neighbour rows) are renumbered
    // onto one XCD so that a row is fetched into one L2
    const int per_xcd = gridDim.x / 8, bx = blockIdx.x;
    const int tile = bx < per_xcd * 8 ? (bx % 8) * per_xcd + bx / 8 : bx;
    const int row0 = tile * FL_ROWS;
    const float* __restrict__ x = g.x;
    // the biases of this lane's columns, requested now (used after the products)
    float bias1[CT], bias2[CT];
    fl_bias<CT>(g.b1, H, wave, lane, bias1);
    fl_bias<CT>(g.b2, H, wave, lane, bias2);

    // ---- phase 1: the tile's CSR rows (one thread per node), then the edge table, then the aggregation
    if (tid < FL_ROWS) {
        const int i = min(row0 + tid, n - 1);
        const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
        const int deg = row0 + tid < n ? e1 - e0 : 0;
        s_e0[tid] = e0; s_deg[tid] = deg;
        const int E = g.rowptr[n], emax = max(E - 1, 0);
        const int32_t* __restrict__ colp = E > 0 ? g.col : g.rowptr;        // a batch without edges: any readable words
        const int32_t* __restrict__ typ = E > 0 ? g.etype : g.rowptr;
#pragma unroll
        for (int k = 0; k < FL_NB; ++k) {
            const int e = min(e0 + k, emax);
            const int j = colp[e], t = typ[e] & 3;
            s_nb[tid][k] = k < deg ? j : i;         // surplus slots: own row (a valid address), type 4 => message 0
            s_ty[tid][k] = k < deg ? t : 4;
        }
    }
    for (int k = tid; k < H; k += FL_THREADS) {
        const float4 wk = fl_ld4(g.edge_w + (size_t)k * 4);
        const float bk = g.edge_b[k];
        etab[k] = wk.x + bk; etab[H + k] = wk.y + bk; etab[2 * H + k] = wk.z + bk; etab[3 * H + k] = wk.w + bk;
        etab[4 * H + k] = -INFINITY;
    }
    {   // columns H .. (pitch) of the image read as zeros in the products
        const int padw = P - H;
        for (int k = tid; k < FL_ROWS * padw; k += FL_THREADS) img[(k / padw) * P + H + (k % padw)] = 0.f;
    }
    __syncthreads();
    float4 hb0[CT], hb1[CT];
    fl_prefetch<CT>(g.w1p, KB, wave, lane, hb0, hb1);
    {
        const int items = FL_ROWS * H4;
        // two items per round: 2 x (own row + 4 neighbour rows) = 10 independent 16-byte loads in flight per thread
        for (int it = tid; it < items; it += 2 * FL_THREADS) {
            float4 acc2[2], own2[2];
            int r2[2], c2[2];
            float4 xs[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = min(it + u * FL_THREADS, items - 1);
                const int r = t / H4, c = (t - r * H4) * 4;
                r2[u] = r; c2[u] = c;
                const int i = min(row0 + r, n - 1);
                own2[u] = fl_ld4(x + (size_t)i * a.ldx + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) xs[u][k] = fl_ld4(x + (size_t)s_nb[r][k] * a.ldx + c);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = r2[u], c = c2[u];
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 e = *reinterpret_cast<const float4*>(etab + s_ty[r][k] * H + c);
                    acc.x += fmaxf(xs[u][k].x + e.x, 0.f); acc.y += fmaxf(xs[u][k].y + e.y, 0.f);
                    acc.z += fmaxf(xs[u][k].z + e.z, 0.f); acc.w += fmaxf(xs[u][k].w + e.w, 0.f);
                }
                const int deg = s_deg[r];
                for (int k = 4; k < deg; ++k) {          // the rare node with more than four bonds, in edge order
                    int j, t;
                    if (k < FL_NB) { j = s_nb[r][k]; t = s_ty[r][k]; }
                    else { const int e = s_e0[r] + k; j = g.col[e]; t = g.etype[e] & 3; }
                    const float4 xv = fl_ld4(x + (size_t)j * a.ldx + c);
                    const float4 e = *reinterpret_cast<const float4*>(etab + t * H + c);
                    acc.x += fmaxf(xv.x + e.x, 0.f); acc.y += fmaxf(xv.y + e.y, 0.f);
                    acc.z += fmaxf(xv.z + e.z, 0.f); acc.w += fmaxf(xv.w + e.w, 0.f);
                }
                acc2[u] = make_float4(acc.x + g.ope * own2[u].x, acc.y + g.ope * own2[u].y, acc.z + g.ope * own2[u].z,
                                      acc.w + g.ope * own2[u].w);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (it + u * FL_THREADS >= items) continue;
                const int r = r2[u], c = c2[u];
                const bool live = row0 + r < n;
                const float4 v = live ? acc2[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(img + r * P + c) = v;
                if (SAVE && live) *reinterpret_cast<float4*>(g.agg + (size_t)(row0 + r) * H + c) = v;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: hid = relu(agg W1^T + b1), back into the image
    floatx4 acc[FL_RT][CT];
    fl_zero<CT, FL_RT>(acc);
    fl_product<CT, FL_RT, true>(img, P, g.w1p, KB, wave, lane, acc, hb0, hb1);
    fl_prefetch<CT>(g.w2p, KB, wave, lane, hb0, hb1);
    __syncthreads();            // every wave is done reading the aggregated image
    fl_store<CT, FL_RT, true, true>(acc, bias1, H, img, P, 0, SAVE ? g.hid : nullptr, H, row0, n, wave, lane);
    __syncthreads();

    // ---- phase 3: y = hid W2^T + b2
    fl_zero<CT, FL_RT>(acc);
    fl_product<CT, FL_RT, true>(img, P, g.w2p, KB, wave, lane, acc, hb0, hb1);
    fl_store<CT, FL_RT, false, false>(acc, bias2, H, nullptr, 0, 0, g.y, a.ldy, row0, n, wave, lane);
}

// Backward through a ReLU: v = acc where the saved forward activation `act` (global, [nrows, ldact]) is positive,
// else 0; written to the LDS image (with zero padding up to the wave tiles' end) and to global rows.  The gate
// values are requested first, then `pre_write_sync` lets the caller's barrier ("everyone is done reading the image
// that is about to be overwritten") sit between the requests and their use.
template <int CT, int RT, bool SYNC>
__device__ __forceinline__ void fl_store_gated(const floatx4 (&acc)[RT][CT], const float* __restrict__ act, size_t ldact, int N,
                                               float* img, int P, float* gout, size_t ldg, int row0, int nrows, int wave, int lane) {
    const int nl = lane & 15, kq = lane >> 4;
    float gate[RT][CT][4];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = min((wave * CT + ct) * 16 + nl, N - 1);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) gate[rt][ct][i] = act[(size_t)min(row0 + rt * 16 + 4 * kq + i, nrows - 1) * ldact + c];
    }
    if (SYNC) __syncthreads();
    const bool full = row0 + 16 * RT <= nrows;      // uniform
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = (wave * CT + ct) * 16 + nl;
        const bool cok = c < N;
        float v[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[rt][i] = (cok && gate[rt][ct][i] > 0.f) ? acc[rt][ct][i] : 0.f;
        if (c < P) {
            float* q = img + (4 * kq) * P + c;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) q[(rt * 16 + i) * P] = v[rt][i];
        }
        if (cok) {
            float* q = gout + (size_t)(row0 + 4 * kq) * ldg + c;
            if (full) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) q[(size_t)(rt * 16 + i) * ldg] = v[rt][i];
            } else {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (row0 + rt * 16 + 4 * kq + i < nrows) q[(size_t)(rt * 16 + i) * ldg] = v[rt][i];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the two Linear layers of a GINE application as ONE launch:
//     dHid = (G W2) * [hid > 0]        G = dL/dy of the layer (rows of the tile loaded into the LDS image)
//     dAgg = dHid W1
// W2 and W1 arrive packed in their transposed role (reduction over the weight matrix's ROW index).  dHid (needed by
// the weight gradients of the first Linear) and dAgg (input of the aggregation's backward) go to memory; dHid never
// comes back from there.
// `dpool` (or null): the rows of G are not read from `g` but are the pooled-output gradient of each row's graph,
// dpool[node_graph[row] * ldp + column] (the top of the encoder stack: what the graph sum's backward broadcasts), and
// the kernel writes them to `g` on the way, where the weight gradients read them (cut-operand kernel only).
struct FLBGroup { const float *g, *hid, *w2tp, *w1tp; float *dh, *da; const float* dpool; };
struct FLBArgs { FLBGroup g[2]; int n, H, P; const int32_t* node_graph; int ldp; };

template <int CT>
__global__ __launch_bounds__(FL_THREADS) void gine_layer_bwd_kernel(const FLBArgs a) {
    extern __shared__ __attribute__((aligned(16))) float fl_lds[];
    const FLBGroup& g = a.g[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, P = a.P, H4 = H >> 2, n = a.n, KB = (H + 15) >> 4;
    float* img = fl_lds;
    const int per_xcd = gridDim.x / 8, bx = blockIdx.x;
    const int tile = bx < per_xcd * 8 ? (bx % 8) * per_xcd + bx / 8 : bx;
    const int row0 = tile * FL_ROWS;
    float4 hb0[CT], hb1[CT];
    fl_prefetch<CT>(g.w2tp, KB, wave, lane, hb0, hb1);
    {   // the tile's rows of G, all loads of a thread first
        const int items = FL_ROWS * H4;
        for (int it = tid; it < items; it += 4 * FL_THREADS) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = min(it + u * FL_THREADS, items - 1);
                const int r = t / H4, c = (t - r * H4) * 4;
                v[u] = fl_ld4(g.g + (size_t)min(row0 + r, n - 1) * H + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = it + u * FL_THREADS;
                if (t >= items) continue;
                const int r = t / H4, c = (t - r * H4) * 4;
                *reinterpret_cast<float4*>(img + r * P + c) = row0 + r < n ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const int padw = P - H;
        for (int k = tid; k < FL_ROWS * padw; k += FL_THREADS) img[(k / padw) * P + H + (k % padw)] = 0.f;
    }
    __syncthreads();
    floatx4 acc[FL_RT][CT];
    fl_zero<CT, FL_RT>(acc);
    fl_product<CT, FL_RT, true>(img, P, g.w2tp, KB, wave, lane, acc, hb0, hb1);
    fl_prefetch<CT>(g.w1tp, KB, wave, lane, hb0, hb1);
    fl_store_gated<CT, FL_RT, true>(acc, g.hid, H, H, img, P, g.dh, H, row0, n, wave, lane);   // (barrier inside: G is no longer read)
    __syncthreads();
    fl_zero<CT, FL_RT>(acc);
    fl_product<CT, FL_RT, true>(img, P, g.w1tp, KB, wave, lane, acc, hb0, hb1);
    float nobias[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) nobias[ct] = 0.f;
    fl_store<CT, FL_RT, false, false>(acc, nobias, H, nullptr, 0, 0, g.da, H, row0, n, wave, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// The link scorer's forward pass as ONE launch (src/modules/vae_piece_decoder.py:97-117):
//     pin  = [emb[src] | emb[dst] | z[graph]]                        pair input, gathered straight into LDS
//     s1   = relu(pin W0^T + b0);  s2 = relu(s1 W1^T + b1);  s3 = s2 W2^T + b2      MLP(P, P/2, P, ReLU, 3)
//     logits = s3 W3^T + b3  (4 bond classes), lse, row loss = lse - logits[gold]   Linear + cross-entropy rows
// A workgroup owns 16 RT consecutive pairs; the activations move between two LDS images and are copied to memory
// only for backward's use.  The last MLP layer (P outputs) runs as two halves of P/2 columns over the same image.
struct ScorerArgs {
    const float* emb; int lde;                 // decoder-side node embeddings [n, lde]
    const float* z;                            // [B, Z]
    const int32_t *src, *dst, *graph, *gold;   // [S]
    const float *w0p, *b0, *w1p, *b1, *w2p_lo, *w2p_hi, *b2, *w3, *b3;
    float *pin, *s1, *s2, *s3, *logits, *row_loss, *lse;
    int S, H, Z, P, P2, PA, PB;
};

template <int CT, int RT>
__global__ __launch_bounds__(FL_THREADS) void scorer_fwd_kernel(const ScorerArgs a) {
    constexpr int ROWS = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float fl_lds[];
    __shared__ int s_src[ROWS], s_dst[ROWS], s_gr[ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * ROWS, S = a.S, H = a.H, P = a.P, P2 = a.P2, PA = a.PA, PB = a.PB;
    float* imgA = fl_lds;              // [ROWS][PA]: pair input, later s3
    float* imgB = fl_lds + ROWS * PA;  // [ROWS][PB]: s1, then s2
    float bias0[CT], bias1[CT], bias2lo[CT], bias2hi[CT];
    fl_bias<CT>(a.b0, P2, wave, lane, bias0);
    fl_bias<CT>(a.b1, P2, wave, lane, bias1);
    fl_bias<CT>(a.b2, P2, wave, lane, bias2lo);
    fl_bias<CT>(a.b2 + P2, P2, wave, lane, bias2hi);
    if (tid < ROWS) {
        const int sidx = min(row0 + tid, S - 1);
        s_src[tid] = a.src[sidx]; s_dst[tid] = a.dst[sidx]; s_gr[tid] = a.graph[sidx];
    }
    for (int k = tid; k < ROWS * (PA - P); k += FL_THREADS) imgA[(k / (PA - P)) * PA + P + k % (PA - P)] = 0.f;
    for (int k = tid; k < ROWS * (PB - P2); k += FL_THREADS) imgB[(k / (PB - P2)) * PB + P2 + k % (PB - P2)] = 0.f;
    __syncthreads();
    {   // gather: four chunks per thread and round, all loads first
        const int P4 = P >> 2, items = ROWS * P4;
        for (int it = tid; it < items; it += 4 * FL_THREADS) {
            float4 v[4];
            int rr[4], cc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = min(it + u * FL_THREADS, items - 1);
                const int r = t / P4, c = (t - r * P4) * 4;
                rr[u] = r; cc[u] = c;
                const float* p = c < H ? a.emb + (size_t)s_src[r] * a.lde + c
                                       : (c < 2 * H ? a.emb + (size_t)s_dst[r] * a.lde + (c - H) : a.z + (size_t)s_gr[r] * a.Z + (c - 2 * H));
                v[u] = fl_ld4(p);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (it + u * FL_THREADS >= items) continue;
                const bool live = row0 + rr[u] < S;
                const float4 w = live ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(imgA + rr[u] * PA + cc[u]) = w;
                if (live) *reinterpret_cast<float4*>(a.pin + (size_t)(row0 + rr[u]) * P + cc[u]) = w;
            }
        }
    }
    __syncthreads();
    floatx4 acc[RT][CT];
    // s1
    fl_zero<CT, RT>(acc);
    fl_product<CT, RT>(imgA, PA, a.w0p, (P + 15) >> 4, wave, lane, acc);
    fl_store<CT, RT, true, true>(acc, bias0, P2, imgB, PB, 0, a.s1, P2, row0, S, wave, lane);
    __syncthreads();
    // s2 (in place)
    fl_zero<CT, RT>(acc);
    fl_product<CT, RT>(imgB, PB, a.w1p, (P2 + 15) >> 4, wave, lane, acc);
    __syncthreads();
    fl_store<CT, RT, true, true>(acc, bias1, P2, imgB, PB, 0, a.s2, P2, row0, S, wave, lane);
    __syncthreads();
    // s3, two halves, into the first image (the pair input is no longer needed)
    fl_zero<CT, RT>(acc);
    fl_product<CT, RT>(imgB, PB, a.w2p_lo, (P2 + 15) >> 4, wave, lane, acc);
    fl_store<CT, RT, false, false>(acc, bias2lo, P2, imgA, PA, 0, a.s3, P, row0, S, wave, lane);
    fl_zero<CT, RT>(acc);
    fl_product<CT, RT>(imgB, PB, a.w2p_hi, (P2 + 15) >> 4, wave, lane, acc);
    fl_store<CT, RT, false, false>(acc, bias2hi, P2, imgA, PA, P2, a.s3, P, row0, S, wave, lane);
    __syncthreads();
    // head: wave w takes rows 16h + 4w .. 16h + 4w+3 of every 16-row block h; lanes split the P columns in float4 chunks
    for (int hb = 0; hb < RT; ++hb) {
        const int rb = 16 * hb + 4 * wave;
        const int P4 = P >> 2;
        float part[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) part[i][c] = 0.f;
        for (int ch = lane; ch < P4; ch += 64) {
            float4 w[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) w[c] = fl_ld4(a.w3 + (size_t)c * P + ch * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 xv = *reinterpret_cast<const float4*>(imgA + (rb + i) * PA + ch * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) part[i][c] += ((xv.x * w[c].x + xv.y * w[c].y) + xv.z * w[c].z) + xv.w * w[c].w;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part[i][c] += __shfl_xor(part[i][c], o, 64);
        if (lane < 4) {
            const int r = row0 + rb + lane;
            if (r < S) {
                float l[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // (static indexing: lane picks its row with selects)
                    const float v = lane == 0 ? part[0][c] : (lane == 1 ? part[1][c] : (lane == 2 ? part[2][c] : part[3][c]));
                    l[c] = v + a.b3[c];
                }
                const float mx = fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3]));
                const float lse = mx + logf(((expf(l[0] - mx) + expf(l[1] - mx)) + expf(l[2] - mx)) + expf(l[3] - mx));
                const int tg = a.gold[r];
                reinterpret_cast<float4*>(a.logits)[r] = make_float4(l[0], l[1], l[2], l[3]);
                a.lse[r] = lse;
                a.row_loss[r] = lse - (tg == 0 ? l[0] : (tg == 1 ? l[1] : (tg == 2 ? l[2] : l[3])));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the link scorer's MLP as ONE launch (the head's backward has already turned the logits' gradient
// into ds3 = dL/ds3):   ds2 = (ds3 W2) * [s2 > 0];   ds1 = (ds2 W1) * [s1 > 0];   dpin = ds1 W0.
// 16 RT pairs per workgroup; ds2 / ds1 (operands of the weight gradients) and dpin (scattered to the atoms afterwards)
// go to memory.  The weights arrive packed in their transposed role, W0 as two halves of P/2 output columns.
struct ScorerBwdArgs {
    const float *ds3, *s2, *s1;
    const float *w2tp, *w1tp, *w0tp_lo, *w0tp_hi;
    float *ds2, *ds1, *dpin;
    int S, P, P2, PA, PB;
};

template <int CT, int RT>
__global__ __launch_bounds__(FL_THREADS) void scorer_bwd_kernel(const ScorerBwdArgs a) {
    constexpr int ROWS = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float fl_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * ROWS, S = a.S, P = a.P, P2 = a.P2, PA = a.PA, PB = a.PB;
    float* imgA = fl_lds;              // [ROWS][PA]: ds3
    float* imgB = fl_lds + ROWS * PA;  // [ROWS][PB]: ds2, then ds1
    {
        const int P4 = P >> 2, items = ROWS * P4;
        for (int it = tid; it < items; it += 4 * FL_THREADS) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = min(it + u * FL_THREADS, items - 1);
                const int r = t / P4, c = (t - r * P4) * 4;
                v[u] = fl_ld4(a.ds3 + (size_t)min(row0 + r, S - 1) * P + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = it + u * FL_THREADS;
                if (t >= items) continue;
                const int r = t / P4, c = (t - r * P4) * 4;
                *reinterpret_cast<float4*>(imgA + r * PA + c) = row0 + r < S ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        for (int k = tid; k < ROWS * (PA - P); k += FL_THREADS) imgA[(k / (PA - P)) * PA + P + k % (PA - P)] = 0.f;
        for (int k = tid; k < ROWS * (PB - P2); k += FL_THREADS) imgB[(k / (PB - P2)) * PB + P2 + k % (PB - P2)] = 0.f;
    }
    __syncthreads();
    floatx4 acc[RT][CT];
    fl_zero<CT, RT>(acc);
    fl_product<CT, RT>(imgA, PA, a.w2tp, (P + 15) >> 4, wave, lane, acc);
    fl_store_gated<CT, RT, false>(acc, a.s2, P2, P2, imgB, PB, a.ds2, P2, row0, S, wave, lane);
    __syncthreads();
    fl_zero<CT, RT>(acc);
    fl_product<CT, RT>(imgB, PB, a.w1tp, (P2 + 15) >> 4, wave, lane, acc);
    fl_store_gated<CT, RT, true>(acc, a.s1, P2, P2, imgB, PB, a.ds1, P2, row0, S, wave, lane);
    __syncthreads();
    float nobias[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) nobias[ct] = 0.f;
    fl_zero<CT, RT>(acc);
    fl_product<CT, RT>(imgB, PB, a.w0tp_lo, (P2 + 15) >> 4, wave, lane, acc);
    fl_store<CT, RT, false, false>(acc, nobias, P2, nullptr, 0, 0, a.dpin, P, row0, S, wave, lane);
    fl_zero<CT, RT>(acc);
    fl_product<CT, RT>(imgB, PB, a.w0tp_hi, (P2 + 15) >> 4, wave, lane, acc);
    fl_store<CT, RT, false, false>(acc, nobias, P2, nullptr, 0, P2, a.dpin, P, row0, S, wave, lane);
}

// ===============================================================================================================
// The same two kernels with the products on the bf16 matrix instruction (v_mfma_f32_16x16x32_bf16: 16 times the
// rate of the fp32 one) at fp32 accuracy: every fp32 operand is cut into three bf16 pieces x = h + m + l, each the
// round-to-nearest bf16 of what the pieces before it leave (|m| <= 2^-9 |x|, |l| <= 2^-18 |x|; the cut is EXACT:
// three 8-bit significands with signs of their own cover the 24 bits), and a product a.b is the sum of the six
// piece products down to 2^-18 of it  (hh) + (hm + mh) + (mm + hl + lh);  the three left out (ml, lm, ll) are
// below 2^-26 |a||b| and of either sign, a quarter of the rounding a chain of fp32 fused multiply-adds makes at
// every step.  Each
// piece product is exact in fp32 and an instruction adds 32 of them into the fp32 accumulator, so a reduction of
// length K rounds the accumulator 6 K / 32 times instead of K times.  Six instructions of 16 cycles replace eight
// of 32 for the same 16 x 16 x 32 block.
//
// Operands: the weights are cut once per step by pack_weights_split_kernel into fragment order
// [k / 32][piece][k quarter][column][8 bf16] (a lane's fragment is one 16-byte load, the 16 lanes of a quarter read
// 256 contiguous bytes); the activations are cut as they are written into the LDS image, which holds three planes
// [piece][row][pitch] of bf16 with a row pitch of an odd number of 16-byte units (conflict-free ds_read_b128 of
// the fragments: lane l reads 8 consecutive k of row l & 15 at k quarter l >> 4).
struct PackSplitOne { const float* src; char* dst; int N, K, ld, transposed, Npad, KB; };
struct PackSplitArgs { PackSplitOne m[8]; int count; };

__global__ __launch_bounds__(256) void pack_weights_split_kernel(const PackSplitArgs a) {
    const PackSplitOne& m = a.m[blockIdx.y];
    const int total = m.KB * 4 * m.Npad;           // fragments: (k block, k quarter, column)
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int c = t % m.Npad, kq = (t / m.Npad) & 3, kb = t / (4 * m.Npad), k = kb * 32 + kq * 8;
        unsigned h[8], mm[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = c < m.N && k + j < m.K;
            const size_t idx = m.transposed ? (size_t)min(k + j, m.K - 1) * m.ld + min(c, m.N - 1)
                                            : (size_t)min(c, m.N - 1) * m.ld + min(k + j, m.K - 1);
            const float x = m.src[idx];
            bf_cut(ok ? x : 0.f, h[j], mm[j], l[j]);
        }
        char* out = m.dst + (((size_t)kb * 3 * 4 + kq) * m.Npad + c) * 16;
        const size_t pstride = (size_t)4 * m.Npad * 16;
        *reinterpret_cast<uint4*>(out) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        *reinterpret_cast<uint4*>(out + pstride) = make_uint4(mm[0] | (mm[1] << 16), mm[2] | (mm[3] << 16), mm[4] | (mm[5] << 16), mm[6] | (mm[7] << 16));
        *reinterpret_cast<uint4*>(out + 2 * pstride) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
    }
}

// P in FLArgs / FLBArgs: the image's row pitch in BYTES for these kernels
// RT 16-row tiles per workgroup: 2 (32 nodes) while that already gives every compute unit a workgroup or more, 4
// (64 nodes: a streamed weight block feeds twice the matrix instructions) for launches of several rounds.
template <int CT, int RT, bool SAVE>
__global__ __launch_bounds__(FL_THREADS) void gine_layer_fwd_split_kernel(const FLArgs a) {
    constexpr int ROWS = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float fl_lds[];
    __shared__ int s_e0[ROWS], s_deg[ROWS], s_nb[ROWS][FL_NB], s_ty[ROWS][FL_NB];
    const FLGroup& g = a.g[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, PB = a.P, H4 = H >> 2, n = a.n, KB = (H + 31) >> 5, kpad = KB * 32;
    const int plane = ROWS * PB;
    char* img = reinterpret_cast<char*>(fl_lds);                    // [3][32][PB bytes]
    float* etab = reinterpret_cast<float*>(img + 3 * plane);        // [5][H]
    const int per_xcd = gridDim.x / 8, bx = blockIdx.x;
    const int tile = bx < per_xcd * 8 ? (bx % 8) * per_xcd + bx / 8 : bx;
    const int row0 = tile * ROWS;
    const float* __restrict__ x = g.x;
    const char* w1p = reinterpret_cast<const char*>(g.w1p);
    const char* w2p = reinterpret_cast<const char*>(g.w2p);
    float bias1[CT], bias2[CT];
    fl_bias<CT>(g.b1, H, wave, lane, bias1);
    fl_bias<CT>(g.b2, H, wave, lane, bias2);

    if (tid < ROWS) {
        const int i = min(row0 + tid, n - 1);
        const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
        const int deg = row0 + tid < n ? e1 - e0 : 0;
        s_e0[tid] = e0; s_deg[tid] = deg;
        const int E = g.rowptr[n], emax = max(E - 1, 0);
        const int32_t* __restrict__ colp = E > 0 ? g.col : g.rowptr;
        const int32_t* __restrict__ typ = E > 0 ? g.etype : g.rowptr;
#pragma unroll
        for (int k = 0; k < FL_NB; ++k) {
            const int e = min(e0 + k, emax);
            const int j = colp[e], t = typ[e] & 3;
            s_nb[tid][k] = k < deg ? j : i;
            s_ty[tid][k] = k < deg ? t : 4;
        }
    }
    for (int k = tid; k < H; k += FL_THREADS) {
        const float4 wk = fl_ld4(g.edge_w + (size_t)k * 4);
        const float bk = g.edge_b[k];
        etab[k] = wk.x + bk; etab[H + k] = wk.y + bk; etab[2 * H + k] = wk.z + bk; etab[3 * H + k] = wk.w + bk;
        etab[4 * H + k] = -INFINITY;
    }
    {   // columns H .. kpad of every plane read as zeros in the products
        const int padw = kpad - H;
        for (int k = tid; k < 3 * ROWS * padw; k += FL_THREADS) {
            const int p = k / (ROWS * padw), q = k - p * (ROWS * padw);
            *reinterpret_cast<unsigned short*>(img + p * plane + (q / padw) * PB + 2 * (H + q % padw)) = 0;
        }
    }
    __syncthreads();
    float4 hb0[3][CT], hb1[3][CT];
    flb_prefetch<CT>(w1p, KB, wave, lane, hb0, hb1);
    {
        const int items = ROWS * H4;
        for (int it = tid; it < items; it += 2 * FL_THREADS) {
            float4 acc2[2], own2[2];
            int r2[2], c2[2];
            float4 xs[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = min(it + u * FL_THREADS, items - 1);
                const int r = t / H4, c = (t - r * H4) * 4;
                r2[u] = r; c2[u] = c;
                const int i = min(row0 + r, n - 1);
                own2[u] = fl_ld4(x + (size_t)i * a.ldx + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) xs[u][k] = fl_ld4(x + (size_t)s_nb[r][k] * a.ldx + c);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = r2[u], c = c2[u];
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 e = *reinterpret_cast<const float4*>(etab + s_ty[r][k] * H + c);
                    acc.x += fmaxf(xs[u][k].x + e.x, 0.f); acc.y += fmaxf(xs[u][k].y + e.y, 0.f);
                    acc.z += fmaxf(xs[u][k].z + e.z, 0.f); acc.w += fmaxf(xs[u][k].w + e.w, 0.f);
                }
                const int deg = s_deg[r];
                for (int k = 4; k < deg; ++k) {
                    int j, t;
                    if (k < FL_NB) { j = s_nb[r][k]; t = s_ty[r][k]; }
                    else { const int e = s_e0[r] + k; j = g.col[e]; t = g.etype[e] & 3; }
                    const float4 xv = fl_ld4(x + (size_t)j * a.ldx + c);
                    const float4 e = *reinterpret_cast<const float4*>(etab + t * H + c);
                    acc.x += fmaxf(xv.x + e.x, 0.f); acc.y += fmaxf(xv.y + e.y, 0.f);
                    acc.z += fmaxf(xv.z + e.z, 0.f); acc.w += fmaxf(xv.w + e.w, 0.f);
                }
                acc2[u] = make_float4(acc.x + g.ope * own2[u].x, acc.y + g.ope * own2[u].y, acc.z + g.ope * own2[u].z,
                                      acc.w + g.ope * own2[u].w);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (it + u * FL_THREADS >= items) continue;
                const int r = r2[u], c = c2[u];
                const bool live = row0 + r < n;
                const float4 v = live ? acc2[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                bf_put4(img, plane, r * PB + 2 * c, v);
                if (SAVE && live) *reinterpret_cast<float4*>(g.agg + (size_t)(row0 + r) * H + c) = v;
            }
        }
    }
    __syncthreads();

    floatx4 acc[RT][CT];
    fl_zero<CT, RT>(acc);
    flb_product<CT, RT>(img, PB, plane, w1p, KB, wave, lane, acc, hb0, hb1);
    flb_prefetch<CT>(w2p, KB, wave, lane, hb0, hb1);
    __syncthreads();
    flb_store<CT, RT, true, true>(acc, bias1, H, img, PB, plane, kpad, SAVE ? g.hid : nullptr, H, row0, n, wave, lane);
    __syncthreads();

    fl_zero<CT, RT>(acc);
    flb_product<CT, RT>(img, PB, plane, w2p, KB, wave, lane, acc, hb0, hb1);
    flb_store<CT, RT, false, false>(acc, bias2, H, nullptr, 0, 0, 0, g.y, a.ldy, row0, n, wave, lane);
}

template <int CT, int RT>
__global__ __launch_bounds__(FL_THREADS) void gine_layer_bwd_split_kernel(const FLBArgs a) {
    constexpr int ROWS = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float fl_lds[];
    const FLBGroup& g = a.g[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, PB = a.P, H4 = H >> 2, n = a.n, KB = (H + 31) >> 5, kpad = KB * 32;
    const int plane = ROWS * PB;
    char* img = reinterpret_cast<char*>(fl_lds);
    const char* w2tp = reinterpret_cast<const char*>(g.w2tp);
    const char* w1tp = reinterpret_cast<const char*>(g.w1tp);
    const int per_xcd = gridDim.x / 8, bx = blockIdx.x;
    const int tile = bx < per_xcd * 8 ? (bx % 8) * per_xcd + bx / 8 : bx;
    const int row0 = tile * ROWS;
    float4 hb0[3][CT], hb1[3][CT];
    flb_prefetch<CT>(w2tp, KB, wave, lane, hb0, hb1);
    {
        const int items = ROWS * H4;
        for (int it = tid; it < items; it += 4 * FL_THREADS) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = min(it + u * FL_THREADS, items - 1);
                const int r = t / H4, c = (t - r * H4) * 4, row = min(row0 + r, n - 1);
                v[u] = g.dpool ? fl_ld4(g.dpool + (size_t)a.node_graph[row] * a.ldp + c) : fl_ld4(g.g + (size_t)row * H + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = it + u * FL_THREADS;
                if (t >= items) continue;
                const int r = t / H4, c = (t - r * H4) * 4;
                const bool live = row0 + r < n;
                bf_put4(img, plane, r * PB + 2 * c, live ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f));
                if (g.dpool && live) *reinterpret_cast<float4*>(const_cast<float*>(g.g) + (size_t)(row0 + r) * H + c) = v[u];
            }
        }
        const int padw = kpad - H;
        for (int k = tid; k < 3 * ROWS * padw; k += FL_THREADS) {
            const int p = k / (ROWS * padw), q = k - p * (ROWS * padw);
            *reinterpret_cast<unsigned short*>(img + p * plane + (q / padw) * PB + 2 * (H + q % padw)) = 0;
        }
    }
    __syncthreads();
    floatx4 acc[RT][CT];
    fl_zero<CT, RT>(acc);
    flb_product<CT, RT>(img, PB, plane, w2tp, KB, wave, lane, acc, hb0, hb1);
    flb_prefetch<CT>(w1tp, KB, wave, lane, hb0, hb1);
    flb_store_gated<CT, RT, true>(acc, g.hid, H, H, img, PB, plane, kpad, g.dh, H, row0, n, wave, lane);
    __syncthreads();
    fl_zero<CT, RT>(acc);
    flb_product<CT, RT>(img, PB, plane, w1tp, KB, wave, lane, acc, hb0, hb1);
    float nobias[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) nobias[ct] = 0.f;
    flb_store<CT, RT, false, false>(acc, nobias, H, nullptr, 0, 0, 0, g.da, H, row0, n, wave, lane);
}

// The link scorer's two kernels on cut operands.  ScorerArgs / ScorerBwdArgs as above with PA, PB = the split
// images' row pitches in BYTES for reductions over P and P/2 and P3 = the pitch in floats of the fp32 image of s3
// (which only the head reads).  The LDS regions are reused as the data in them dies: the pair input's planes
// (reduction over P: 130 KB at 32 pairs) make way for s1's planes and later for s3; s2's planes sit behind them.
template <int CT, int RT>
__global__ __launch_bounds__(FL_THREADS) void scorer_fwd_split_kernel(const ScorerArgs a, int P3, int off2) {
    constexpr int ROWS = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float fl_lds[];
    __shared__ int s_src[ROWS], s_dst[ROWS], s_gr[ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * ROWS, S = a.S, H = a.H, P = a.P, P2 = a.P2, PA = a.PA, PB = a.PB;
    const int KBA = (P + 31) >> 5, KBB = (P2 + 31) >> 5, kpadA = KBA * 32, kpadB = KBB * 32;
    const int planeA = ROWS * PA, planeB = ROWS * PB;
    char* base = reinterpret_cast<char*>(fl_lds);
    char* img1 = base;                 // pair input planes, then s1 planes
    char* img2 = base + off2;          // s2 planes
    float* img3 = fl_lds;              // s3 as fp32 [ROWS][P3]
    const char* w0p = reinterpret_cast<const char*>(a.w0p);
    const char* w1p = reinterpret_cast<const char*>(a.w1p);
    const char* w2lo = reinterpret_cast<const char*>(a.w2p_lo);
    const char* w2hi = reinterpret_cast<const char*>(a.w2p_hi);
    float bias0[CT], bias1[CT], bias2lo[CT], bias2hi[CT];
    fl_bias<CT>(a.b0, P2, wave, lane, bias0);
    fl_bias<CT>(a.b1, P2, wave, lane, bias1);
    fl_bias<CT>(a.b2, P2, wave, lane, bias2lo);
    fl_bias<CT>(a.b2 + P2, P2, wave, lane, bias2hi);
    if (tid < ROWS) {
        const int sidx = min(row0 + tid, S - 1);
        s_src[tid] = a.src[sidx]; s_dst[tid] = a.dst[sidx]; s_gr[tid] = a.graph[sidx];
    }
    {
        const int padw = kpadA - P;
        for (int k = tid; k < 3 * ROWS * padw; k += FL_THREADS) {
            const int p = k / (ROWS * padw), q = k - p * (ROWS * padw);
            *reinterpret_cast<unsigned short*>(img1 + p * planeA + (q / padw) * PA + 2 * (P + q % padw)) = 0;
        }
    }
    __syncthreads();
    float4 hb0[3][CT], hb1[3][CT];
    flb_prefetch<CT>(w0p, KBA, wave, lane, hb0, hb1);
    {   // gather: four chunks per thread and round, all loads first
        const int P4 = P >> 2, items = ROWS * P4;
        for (int it = tid; it < items; it += 4 * FL_THREADS) {
            float4 v[4];
            int rr[4], cc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = min(it + u * FL_THREADS, items - 1);
                const int r = t / P4, c = (t - r * P4) * 4;
                rr[u] = r; cc[u] = c;
                const float* p = c < H ? a.emb + (size_t)s_src[r] * a.lde + c
                                       : (c < 2 * H ? a.emb + (size_t)s_dst[r] * a.lde + (c - H) : a.z + (size_t)s_gr[r] * a.Z + (c - 2 * H));
                v[u] = fl_ld4(p);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (it + u * FL_THREADS >= items) continue;
                const bool live = row0 + rr[u] < S;
                const float4 w = live ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                bf_put4(img1, planeA, rr[u] * PA + 2 * cc[u], w);
                if (live) *reinterpret_cast<float4*>(a.pin + (size_t)(row0 + rr[u]) * P + cc[u]) = w;
            }
        }
    }
    __syncthreads();
    floatx4 acc[RT][CT];
    // s1
    fl_zero<CT, RT>(acc);
    flb_product<CT, RT>(img1, PA, planeA, w0p, KBA, wave, lane, acc, hb0, hb1);
    flb_prefetch<CT>(w1p, KBB, wave, lane, hb0, hb1);
    __syncthreads();            // the pair input is dead: its region takes s1
    flb_store<CT, RT, true, true>(acc, bias0, P2, img1, PB, planeB, kpadB, a.s1, P2, row0, S, wave, lane);
    __syncthreads();
    // s2, behind s1
    fl_zero<CT, RT>(acc);
    flb_product<CT, RT>(img1, PB, planeB, w1p, KBB, wave, lane, acc, hb0, hb1);
    flb_prefetch<CT>(w2lo, KBB, wave, lane, hb0, hb1);
    flb_store<CT, RT, true, true>(acc, bias1, P2, img2, PB, planeB, kpadB, a.s2, P2, row0, S, wave, lane);
    __syncthreads();            // s1 is dead: its region takes s3 (fp32)
    fl_zero<CT, RT>(acc);
    flb_product<CT, RT>(img2, PB, planeB, w2lo, KBB, wave, lane, acc, hb0, hb1);
    flb_prefetch<CT>(w2hi, KBB, wave, lane, hb0, hb1);
    fl_store<CT, RT, false, false>(acc, bias2lo, P2, img3, P3, 0, a.s3, P, row0, S, wave, lane);
    fl_zero<CT, RT>(acc);
    flb_product<CT, RT>(img2, PB, planeB, w2hi, KBB, wave, lane, acc, hb0, hb1);
    fl_store<CT, RT, false, false>(acc, bias2hi, P2, img3, P3, P2, a.s3, P, row0, S, wave, lane);
    __syncthreads();
    // head: wave w takes rows 16h + 4w .. 16h + 4w+3 of every 16-row block h; lanes split the P columns in float4 chunks
    for (int hb = 0; hb < RT; ++hb) {
        const int rb = 16 * hb + 4 * wave;
        const int P4 = P >> 2;
        float part[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) part[i][c] = 0.f;
        for (int ch = lane; ch < P4; ch += 64) {
            float4 w[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) w[c] = fl_ld4(a.w3 + (size_t)c * P + ch * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 xv = *reinterpret_cast<const float4*>(img3 + (rb + i) * P3 + ch * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) part[i][c] += ((xv.x * w[c].x + xv.y * w[c].y) + xv.z * w[c].z) + xv.w * w[c].w;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part[i][c] += __shfl_xor(part[i][c], o, 64);
        if (lane < 4) {
            const int r = row0 + rb + lane;
            if (r < S) {
                float l[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = lane == 0 ? part[0][c] : (lane == 1 ? part[1][c] : (lane == 2 ? part[2][c] : part[3][c]));
                    l[c] = v + a.b3[c];
                }
                const float mx = fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3]));
                const float lse = mx + logf(((expf(l[0] - mx) + expf(l[1] - mx)) + expf(l[2] - mx)) + expf(l[3] - mx));
                const int tg = a.gold[r];
                reinterpret_cast<float4*>(a.logits)[r] = make_float4(l[0], l[1], l[2], l[3]);
                a.lse[r] = lse;
                a.row_loss[r] = lse - (tg == 0 ? l[0] : (tg == 1 ? l[1] : (tg == 2 ? l[2] : l[3])));
            }
        }
    }
}

template <int CT, int RT>
__global__ __launch_bounds__(FL_THREADS) void scorer_bwd_split_kernel(const ScorerBwdArgs a, int off2) {
    constexpr int ROWS = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float fl_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * ROWS, S = a.S, P = a.P, P2 = a.P2, PA = a.PA, PB = a.PB;
    const int KBA = (P + 31) >> 5, KBB = (P2 + 31) >> 5, kpadA = KBA * 32, kpadB = KBB * 32;
    const int planeA = ROWS * PA, planeB = ROWS * PB;
    char* img1 = reinterpret_cast<char*>(fl_lds);          // ds3 planes, then ds2 planes
    char* img2 = img1 + off2;                              // ds1 planes
    const char* w2tp = reinterpret_cast<const char*>(a.w2tp);
    const char* w1tp = reinterpret_cast<const char*>(a.w1tp);
    const char* w0lo = reinterpret_cast<const char*>(a.w0tp_lo);
    const char* w0hi = reinterpret_cast<const char*>(a.w0tp_hi);
    float4 hb0[3][CT], hb1[3][CT];
    flb_prefetch<CT>(w2tp, KBA, wave, lane, hb0, hb1);
    {
        const int P4 = P >> 2, items = ROWS * P4;
        for (int it = tid; it < items; it += 4 * FL_THREADS) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = min(it + u * FL_THREADS, items - 1);
                const int r = t / P4, c = (t - r * P4) * 4;
                v[u] = fl_ld4(a.ds3 + (size_t)min(row0 + r, S - 1) * P + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = it + u * FL_THREADS;
                if (t >= items) continue;
                const int r = t / P4, c = (t - r * P4) * 4;
                bf_put4(img1, planeA, r * PA + 2 * c, row0 + r < S ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f));
            }
        }
        const int padw = kpadA - P;
        for (int k = tid; k < 3 * ROWS * padw; k += FL_THREADS) {
            const int p = k / (ROWS * padw), q = k - p * (ROWS * padw);
            *reinterpret_cast<unsigned short*>(img1 + p * planeA + (q / padw) * PA + 2 * (P + q % padw)) = 0;
        }
    }
    __syncthreads();
    floatx4 acc[RT][CT];
    fl_zero<CT, RT>(acc);
    flb_product<CT, RT>(img1, PA, planeA, w2tp, KBA, wave, lane, acc, hb0, hb1);
    flb_prefetch<CT>(w1tp, KBB, wave, lane, hb0, hb1);
    flb_store_gated<CT, RT, true>(acc, a.s2, P2, P2, img1, PB, planeB, kpadB, a.ds2, P2, row0, S, wave, lane);   // (barrier inside: ds3 is dead)
    __syncthreads();
    fl_zero<CT, RT>(acc);
    flb_product<CT, RT>(img1, PB, planeB, w1tp, KBB, wave, lane, acc, hb0, hb1);
    flb_prefetch<CT>(w0lo, KBB, wave, lane, hb0, hb1);
    flb_store_gated<CT, RT, false>(acc, a.s1, P2, P2, img2, PB, planeB, kpadB, a.ds1, P2, row0, S, wave, lane);  // behind ds2
    __syncthreads();
    float nobias[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) nobias[ct] = 0.f;
    fl_zero<CT, RT>(acc);
    flb_product<CT, RT>(img2, PB, planeB, w0lo, KBB, wave, lane, acc, hb0, hb1);
    flb_prefetch<CT>(w0hi, KBB, wave, lane, hb0, hb1);
    fl_store<CT, RT, false, false>(acc, nobias, P2, nullptr, 0, 0, a.dpin, P, row0, S, wave, lane);
    fl_zero<CT, RT>(acc);
    flb_product<CT, RT>(img2, PB, planeB, w0hi, KBB, wave, lane, acc, hb0, hb1);
    fl_store<CT, RT, false, false>(acc, nobias, P2, nullptr, 0, P2, a.dpin, P, row0, S, wave, lane);
}


int fl_pitch(int K) {       // smallest pitch >= K rounded up to 16 with pitch = 8 mod 64
    const int kp = (K + 15) / 16 * 16;
    return (kp - 8 + 63) / 64 * 64 + 8;
}

}  // namespace

static int fused_ct(int N) {          // the kernels' template value for N output columns: CT 16-column tiles per wave
    const int ct = ((N + 15) / 16 + 3) / 4;
    return ct <= 5 ? 5 : (ct <= 6 ? 6 : 8);
}

// Nodes per workgroup of the cut-operand kernels: 64 once that still leaves every compute unit a round and a half
// of workgroups (and the three planes of 64 rows fit the LDS), 32 below.  Option split_rows = 32 / 64 forces.
static int split_rows(int n, int ng, int H, size_t extra_lds) {
    const int forced = opt(OPT_SPLIT_ROWS);
    const bool fits = (size_t)3 * 64 * flb_pitch_bytes(H) + extra_lds <= 156 * 1024;
    if (forced == 32 || !fits) return 32;
    if (forced == 64) return 64;
    // one workgroup per compute unit at a time (registers): rounds of 256; a 64-node workgroup takes twice as long
    // as a 32-node one and uses the weights it streams twice as well (measured: 64 wins at 256 and at 1024 graphs
    // per step, loses at 128 and 512)
    const long long r32 = ((long long)((n + 31) / 32) * ng + 255) / 256, r64 = 2 * (((long long)((n + 63) / 64) * ng + 255) / 256);
    return r64 <= r32 ? 64 : 32;
}

bool gine_layer_fused_supported(int H, int ldx, int ldy) {
    return H % 4 == 0 && H >= 16 && H <= 512 && ldx % 4 == 0 && ldy >= H;
}

// (room for either form of the packed matrix: fp32 fragments, or three bf16 planes per 32 reduction indices)
size_t fused_pack_floats(int N, int K) {
    const size_t npad = (size_t)64 * fused_ct(N);
    return std::max((size_t)((K + 15) / 16) * 16 * npad, (size_t)((K + 31) / 32) * 48 * npad);
}

// Do the fused GINE kernels run their products on the bf16 matrix instruction over three-way cut operands (fp32
// accuracy, see above)?  Option split_bf16 = 0 keeps the fp32 instruction.  Whoever packs the weights for
// gine_layer_fwd / gine_layer_bwd must ask this and set FusedPack::split accordingly.
// Raises a kernel's dynamic-LDS limit when a launch needs more than was granted so far: once per kernel instantiation
// and size instead of a runtime call in front of every launch (one process drives one GPU).
template <auto Kernel>
static int grant_lds(size_t lds) {
    static size_t granted = 48 * 1024;
    if (lds <= granted) return 0;
    if (hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return (int)hipGetLastError();
    granted = lds;
    return 0;
}

bool gine_layer_split(int H) {
    return opt(OPT_SPLIT_BF16) != 0 && H >= 16 && H <= 512;
}

int fused_pack(hipStream_t s, const FusedPack* d, int count) {
    if (count < 1 || count > 8) return PSVAE_E_SHAPE;
    if (d[0].split) {
        PackSplitArgs b{};
        b.count = count;
        int most = 0;
        for (int i = 0; i < count; ++i) {
            if (!d[i].split) return PSVAE_E_SHAPE;          // one form per launch
            if (!d[i].src || !d[i].dst) return PSVAE_E_NULL;
            if (d[i].N < 1 || d[i].K < 1 || d[i].N > 512 || !aligned16(d[i].dst)) return PSVAE_E_SHAPE;
            b.m[i] = PackSplitOne{d[i].src, reinterpret_cast<char*>(d[i].dst), d[i].N, d[i].K, d[i].ld, d[i].transposed ? 1 : 0,
                                  64 * fused_ct(d[i].N), (d[i].K + 31) / 32};
            most = std::max(most, b.m[i].KB * 4 * b.m[i].Npad);
        }
        ProfScope prof(s, KC_HEAD, 0);
        hipLaunchKernelGGL(pack_weights_split_kernel, dim3((most + 255) / 256, count), dim3(256), 0, s, b);
        PSVAE_LAUNCH_CHECK();
        return 0;
    }
    PackArgs a{};
    a.count = count;
    int most = 0;
    for (int i = 0; i < count; ++i) {
        if (d[i].split) return PSVAE_E_SHAPE;
        if (!d[i].src || !d[i].dst) return PSVAE_E_NULL;
        if (d[i].N < 1 || d[i].K < 1 || d[i].N > 512 || !aligned16(d[i].dst)) return PSVAE_E_SHAPE;
        a.m[i] = PackOne{d[i].src, d[i].dst, d[i].N, d[i].K, d[i].ld, d[i].transposed ? 1 : 0, 64 * fused_ct(d[i].N), (d[i].K + 15) / 16};
        most = std::max(most, a.m[i].KB * 4 * a.m[i].Npad);
    }
    ProfScope prof(s, KC_HEAD, 0);
    hipLaunchKernelGGL(pack_weights_kernel, dim3((most + 255) / 256, count), dim3(256), 0, s, a);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int gine_layer_fwd(hipStream_t s, const FusedLayerGroup* grp, int ng, int ldx, int ldy, int n, int H, bool save) {
    if (n <= 0) return 0;
    if (ng < 1 || ng > 2) return PSVAE_E_SHAPE;
    if (!gine_layer_fused_supported(H, ldx, ldy)) return PSVAE_E_SHAPE;
    FLArgs a{};
    double flops = 0;
    for (int q = 0; q < ng; ++q) {
        const FusedLayerGroup& f = grp[q];
        if (!f.x || !f.rowptr || !f.col || !f.etype || !f.edge_w || !f.edge_b || !f.w1p || !f.b1 || !f.w2p || !f.b2 || !f.y)
            return PSVAE_E_NULL;
        if (save && (!f.agg || !f.hid)) return PSVAE_E_NULL;
        if (!aligned16(f.x) || !aligned16(f.w1p) || !aligned16(f.w2p) || !aligned16(f.edge_w) || (save && !aligned16(f.agg)))
            return PSVAE_E_ALIGN;
        a.g[q] = FLGroup{f.x, f.rowptr, f.col, f.etype, f.edge_w, f.edge_b, f.w1p, f.b1, f.w2p, f.b2, f.agg, f.hid, f.y, f.ope};
        flops += 4.0 * n * H * H;
    }
    const bool split = gine_layer_split(H);
    a.n = n; a.H = H; a.P = split ? flb_pitch_bytes(H) : fl_pitch(H); a.ldx = ldx; a.ldy = ldy;
    const int rows = split ? split_rows(n, ng, H, 5 * (size_t)H * sizeof(float)) : FL_ROWS;
    const size_t lds = split ? (size_t)3 * rows * a.P + 5 * (size_t)H * sizeof(float)
                             : ((size_t)FL_ROWS * a.P + 5 * (size_t)H) * sizeof(float);
    const int CT = fused_ct(H);
    const dim3 grid((n + rows - 1) / rows, ng);
    ProfScope prof(s, KC_GINE_LAYER, flops);
#define FLS_LAUNCH2(CT_, RT_, SAVE_)                                                                                     \
    do {                                                                                                                 \
        PSVAE_TRY((grant_lds<gine_layer_fwd_split_kernel<CT_, RT_, SAVE_>>(lds)));                                       \
        hipLaunchKernelGGL((gine_layer_fwd_split_kernel<CT_, RT_, SAVE_>), grid, dim3(FL_THREADS), lds, s, a);           \
    } while (0)
#define FLS_LAUNCH(CT_)                                                                                                  \
    do {                                                                                                                 \
        if (rows == 64) { if (save) FLS_LAUNCH2(CT_, 4, true); else FLS_LAUNCH2(CT_, 4, false); }                        \
        else { if (save) FLS_LAUNCH2(CT_, 2, true); else FLS_LAUNCH2(CT_, 2, false); }                                   \
    } while (0)
    if (split) {
        if (CT == 5) FLS_LAUNCH(5);
        else if (CT == 6) FLS_LAUNCH(6);
        else FLS_LAUNCH(8);
        PSVAE_LAUNCH_CHECK();
        return 0;
    }
#undef FLS_LAUNCH
#undef FLS_LAUNCH2
#define FL_LAUNCH(CT_)                                                                                                   \
    do {                                                                                                                 \
        if (save) {                                                                                                      \
            if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)gine_layer_fwd_kernel<CT_, true>,                    \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
                return (int)hipGetLastError();                                                                           \
            hipLaunchKernelGGL((gine_layer_fwd_kernel<CT_, true>), grid, dim3(FL_THREADS), lds, s, a);                   \
        } else {                                                                                                         \
            if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)gine_layer_fwd_kernel<CT_, false>,                   \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
                return (int)hipGetLastError();                                                                           \
            hipLaunchKernelGGL((gine_layer_fwd_kernel<CT_, false>), grid, dim3(FL_THREADS), lds, s, a);                  \
        }                                                                                                                \
    } while (0)
    if (CT == 5) FL_LAUNCH(5);
    else if (CT == 6) FL_LAUNCH(6);
    else FL_LAUNCH(8);
#undef FL_LAUNCH
    PSVAE_LAUNCH_CHECK();
    return 0;
}

// LDS the one-launch scorer kernels need for 32 pairs per workgroup, in the form the products currently take
// (cut operands: three bf16 planes per image, regions reused as their contents die; fp32: two images side by side),
// and the most a workgroup may ask for.  One place, so that the predicate, the launchers and the size queries agree.
struct ScorerLds { size_t fwd, bwd, cap; int PA, PB, P3; size_t off2_fwd, off2_bwd; };
static ScorerLds scorer_lds(int H, int Z) {
    const int P = 2 * H + Z, P2 = P / 2;
    ScorerLds l{};
    if (gine_layer_split(H)) {
        l.PA = flb_pitch_bytes(P); l.PB = flb_pitch_bytes(P2); l.P3 = P + 4;
        const size_t planes2 = (size_t)3 * 32 * l.PB;
        l.off2_fwd = (std::max(planes2, (size_t)32 * l.P3 * sizeof(float)) + 255) / 256 * 256;
        l.off2_bwd = (planes2 + 255) / 256 * 256;
        l.fwd = std::max((size_t)3 * 32 * l.PA, l.off2_fwd + planes2);
        l.bwd = std::max((size_t)3 * 32 * l.PA, l.off2_bwd + planes2);
        l.cap = 159 * 1024;
    } else {
        l.PA = fl_pitch(P); l.PB = fl_pitch(P2);
        l.fwd = l.bwd = (size_t)32 * (l.PA + l.PB) * sizeof(float);
        l.cap = 156 * 1024;
    }
    return l;
}

// Sizes the one-launch scorer kernels take: the template instance covers 320 < P/2 <= 384 output columns, and both
// directions' images must fit the LDS (with cut operands that ends at P/2 = 352; H = 300, Z = 56 gives 328).
// Everything else runs as pair gather + GEMM chain + head.
bool scorer_fused_supported(int H, int Z, int lde) {
    const int P = 2 * H + Z;
    if (!(H % 4 == 0 && Z % 4 == 0 && lde % 4 == 0 && P % 8 == 0 && P / 2 > 320 && P / 2 <= 384)) return false;
    const ScorerLds l = scorer_lds(H, Z);
    return l.fwd <= l.cap && l.bwd <= l.cap;
}

// packed-weight floats the fused scorer needs (w0, w1, two halves of w2), in that order
size_t scorer_pack_floats(int H, int Z) {
    const int P = 2 * H + Z, P2 = P / 2;
    return fused_pack_floats(P2, P) + 3 * fused_pack_floats(P2, P2);
}

int scorer_pack(hipStream_t s, const float* w0, const float* w1, const float* w2, float* packed, int H, int Z) {
    const int P = 2 * H + Z, P2 = P / 2;
    float* w0p = packed;
    float* w1p = w0p + fused_pack_floats(P2, P);
    float* w2lo = w1p + fused_pack_floats(P2, P2);
    float* w2hi = w2lo + fused_pack_floats(P2, P2);
    const bool sp = gine_layer_split(H);
    const FusedPack pk[4] = {{w0, w0p, P2, P, P, false, sp}, {w1, w1p, P2, P2, P2, false, sp}, {w2, w2lo, P2, P2, P2, false, sp},
                             {w2 + (size_t)P2 * P2, w2hi, P2, P2, P2, false, sp}};
    return fused_pack(s, pk, 4);
}

int scorer_fwd_fused(hipStream_t s, const ScorerFwd& f) {
    if (f.S <= 0) return 0;
    if (!scorer_fused_supported(f.H, f.Z, f.lde)) return PSVAE_E_SHAPE;
    const int P = 2 * f.H + f.Z, P2 = P / 2;
    if (!f.emb || !f.z || !f.src || !f.dst || !f.graph || !f.gold || !f.packed || !f.b0 || !f.b1 || !f.b2 || !f.w3 || !f.b3 ||
        !f.pin || !f.s1 || !f.s2 || !f.s3 || !f.logits || !f.row_loss || !f.lse) return PSVAE_E_NULL;
    if (!aligned16(f.emb) || !aligned16(f.z) || !aligned16(f.packed) || !aligned16(f.w3) || !aligned16(f.pin) || !aligned16(f.logits))
        return PSVAE_E_ALIGN;
    ScorerArgs a{};
    a.emb = f.emb; a.lde = f.lde; a.z = f.z; a.src = f.src; a.dst = f.dst; a.graph = f.graph; a.gold = f.gold;
    a.w0p = f.packed; a.w1p = a.w0p + fused_pack_floats(P2, P); a.w2p_lo = a.w1p + fused_pack_floats(P2, P2);
    a.w2p_hi = a.w2p_lo + fused_pack_floats(P2, P2);
    a.b0 = f.b0; a.b1 = f.b1; a.b2 = f.b2; a.w3 = f.w3; a.b3 = f.b3;
    a.pin = f.pin; a.s1 = f.s1; a.s2 = f.s2; a.s3 = f.s3; a.logits = f.logits; a.row_loss = f.row_loss; a.lse = f.lse;
    const ScorerLds sl = scorer_lds(f.H, f.Z);
    a.S = f.S; a.H = f.H; a.Z = f.Z; a.P = P; a.P2 = P2; a.PA = sl.PA; a.PB = sl.PB;
    // 32 pairs per workgroup (141 KB of LDS, one workgroup per CU): a streamed weight block then feeds twice the
    // matrix instructions of a 16-pair form (measured: 16 pairs ran slower than the GEMM chain everywhere)
    if (gine_layer_split(f.H)) {        // products on the bf16 matrix instruction over cut operands (weights packed to match)
        const int P3 = sl.P3;
        const size_t off2 = sl.off2_fwd, lds = sl.fwd;
        static bool split_attr_set = false;
        if (!split_attr_set) {
            if (hipFuncSetAttribute((const void*)scorer_fwd_split_kernel<6, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess)
                return (int)hipGetLastError();
            split_attr_set = true;
        }
        if (lds > 159 * 1024) return PSVAE_E_SHAPE;
        ProfScope prof(s, KC_SCORER, 2.0 * f.S * ((double)P * P2 + (double)P2 * P2 + (double)P2 * P + 4.0 * P));
        hipLaunchKernelGGL((scorer_fwd_split_kernel<6, 2>), dim3((f.S + 31) / 32), dim3(FL_THREADS), lds, s, a, P3, (int)off2);
        PSVAE_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = sl.fwd;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)scorer_fwd_kernel<6, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
            return (int)hipGetLastError();
        attr_set = true;
    }
    if (lds > 156 * 1024) return PSVAE_E_SHAPE;
    ProfScope prof(s, KC_SCORER, 2.0 * f.S * ((double)P * P2 + (double)P2 * P2 + (double)P2 * P + 4.0 * P));
    hipLaunchKernelGGL((scorer_fwd_kernel<6, 2>), dim3((f.S + 31) / 32), dim3(FL_THREADS), lds, s, a);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int gine_layer_bwd(hipStream_t s, const FusedLayerBwdGroup* grp, int ng, int n, int H, const int32_t* node_graph, int ldp) {
    if (n <= 0) return 0;
    if (ng < 1 || ng > 2) return PSVAE_E_SHAPE;
    if (!gine_layer_fused_supported(H, H, H)) return PSVAE_E_SHAPE;
    FLBArgs a{};
    for (int q = 0; q < ng; ++q) {
        const FusedLayerBwdGroup& f = grp[q];
        if (!f.g || !f.hid || !f.w2tp || !f.w1tp || !f.dh || !f.da) return PSVAE_E_NULL;
        if (!aligned16(f.g) || !aligned16(f.w2tp) || !aligned16(f.w1tp) || (f.dpool && !aligned16(f.dpool))) return PSVAE_E_ALIGN;
        if (f.dpool && (!node_graph || ldp % 4 || !gine_layer_split(H))) return f.dpool && !node_graph ? PSVAE_E_NULL : PSVAE_E_SHAPE;
        a.g[q] = FLBGroup{f.g, f.hid, f.w2tp, f.w1tp, f.dh, f.da, f.dpool};
    }
    a.node_graph = node_graph; a.ldp = ldp;
    const bool split = gine_layer_split(H);
    a.n = n; a.H = H; a.P = split ? flb_pitch_bytes(H) : fl_pitch(H);
    const int rows = split ? split_rows(n, ng, H, 0) : FL_ROWS;
    const size_t lds = split ? (size_t)3 * rows * a.P : (size_t)FL_ROWS * a.P * sizeof(float);
    const int CT = fused_ct(H);
    const dim3 grid((n + rows - 1) / rows, ng);
    ProfScope prof(s, KC_GINE_LAYER_BWD, 4.0 * n * H * H * ng);
#define FLBS_LAUNCH2(CT_, RT_)                                                                                        \
    do {                                                                                                              \
        PSVAE_TRY((grant_lds<gine_layer_bwd_split_kernel<CT_, RT_>>(lds)));                                           \
        hipLaunchKernelGGL((gine_layer_bwd_split_kernel<CT_, RT_>), grid, dim3(FL_THREADS), lds, s, a);               \
    } while (0)
#define FLBS_LAUNCH(CT_) do { if (rows == 64) FLBS_LAUNCH2(CT_, 4); else FLBS_LAUNCH2(CT_, 2); } while (0)
    if (split) {
        if (CT == 5) FLBS_LAUNCH(5);
        else if (CT == 6) FLBS_LAUNCH(6);
        else FLBS_LAUNCH(8);
        PSVAE_LAUNCH_CHECK();
        return 0;
    }
#undef FLBS_LAUNCH
#undef FLBS_LAUNCH2
#define FLB_LAUNCH(CT_)                                                                                               \
    do {                                                                                                              \
        if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)gine_layer_bwd_kernel<CT_>,                           \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return (int)hipGetLastError();                                                                            \
        hipLaunchKernelGGL((gine_layer_bwd_kernel<CT_>), grid, dim3(FL_THREADS), lds, s, a);                          \
    } while (0)
    if (CT == 5) FLB_LAUNCH(5);
    else if (CT == 6) FLB_LAUNCH(6);
    else FLB_LAUNCH(8);
#undef FLB_LAUNCH
    PSVAE_LAUNCH_CHECK();
    return 0;
}

// transposed-role packs for the scorer's backward: w2 [P, P2], w1 [P2, P2], w0 [P2, P] (two halves of P2 columns)
int scorer_pack_bwd(hipStream_t s, const float* w0, const float* w1, const float* w2, float* packed, int H, int Z) {
    const int P = 2 * H + Z, P2 = P / 2;
    float* w2tp = packed;
    float* w1tp = w2tp + fused_pack_floats(P2, P);
    float* w0lo = w1tp + fused_pack_floats(P2, P2);
    float* w0hi = w0lo + fused_pack_floats(P2, P2);
    const bool sp = gine_layer_split(H);
    const FusedPack pk[4] = {{w2, w2tp, P2, P, P2, true, sp}, {w1, w1tp, P2, P2, P2, true, sp}, {w0, w0lo, P2, P2, P, true, sp},
                             {w0 + P2, w0hi, P2, P2, P, true, sp}};
    return fused_pack(s, pk, 4);
}

int scorer_bwd_fused(hipStream_t s, const ScorerBwd& f) {
    if (f.S <= 0) return 0;
    if (!scorer_fused_supported(f.H, f.Z, f.H)) return PSVAE_E_SHAPE;
    const int P = 2 * f.H + f.Z, P2 = P / 2;
    if (!f.ds3 || !f.s2 || !f.s1 || !f.packed || !f.ds2 || !f.ds1 || !f.dpin) return PSVAE_E_NULL;
    if (!aligned16(f.ds3) || !aligned16(f.packed)) return PSVAE_E_ALIGN;
    ScorerBwdArgs a{};
    a.ds3 = f.ds3; a.s2 = f.s2; a.s1 = f.s1;
    a.w2tp = f.packed; a.w1tp = a.w2tp + fused_pack_floats(P2, P); a.w0tp_lo = a.w1tp + fused_pack_floats(P2, P2);
    a.w0tp_hi = a.w0tp_lo + fused_pack_floats(P2, P2);
    a.ds2 = f.ds2; a.ds1 = f.ds1; a.dpin = f.dpin;
    const ScorerLds sl = scorer_lds(f.H, f.Z);
    a.S = f.S; a.P = P; a.P2 = P2; a.PA = sl.PA; a.PB = sl.PB;
    if (gine_layer_split(f.H)) {
        const size_t off2 = sl.off2_bwd, lds = sl.bwd;
        static bool split_attr_set = false;
        if (!split_attr_set) {
            if (hipFuncSetAttribute((const void*)scorer_bwd_split_kernel<6, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess)
                return (int)hipGetLastError();
            split_attr_set = true;
        }
        if (lds > 159 * 1024) return PSVAE_E_SHAPE;
        ProfScope prof(s, KC_SCORER, 2.0 * f.S * ((double)P * P2 + (double)P2 * P2 + (double)P2 * P));
        hipLaunchKernelGGL((scorer_bwd_split_kernel<6, 2>), dim3((f.S + 31) / 32), dim3(FL_THREADS), lds, s, a, (int)off2);
        PSVAE_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = sl.bwd;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)scorer_bwd_kernel<6, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
            return (int)hipGetLastError();
        attr_set = true;
    }
    if (lds > 156 * 1024) return PSVAE_E_SHAPE;
    ProfScope prof(s, KC_SCORER, 2.0 * f.S * ((double)P * P2 + (double)P2 * P2 + (double)P2 * P));
    hipLaunchKernelGGL((scorer_bwd_kernel<6, 2>), dim3((f.S + 31) / 32), dim3(FL_THREADS), lds, s, a);
    PSVAE_LAUNCH_CHECK();
    return 0;
}
