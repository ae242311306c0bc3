// Memory-bound kernels of the message-passing path: GINE message construction +
// CSR segment-sum (forward and backward), edge-table gradient, embedding gather
// and table gradients, graph sum-pool, atom-pair gather / scatter.
//
// All reductions walk index lists in a fixed order (no atomics), so results are
// bitwise reproducible.  Feature rows are processed as float4 channel chunks:
// consecutive lanes take consecutive chunks of one row, so every global access
// is a coalesced 16-byte-per-lane stream; neighbour rows of a molecule lie
// within a few KB of each other and are served by L1/L2.
#include "psvae_internal.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Row (r, float column c) of a matrix with a row pitch of ldb BYTES, addressed as a uniform base plus a 32-bit
// byte offset: one 24-bit multiply-add per address and the scalar-base form of the load, instead of a
// 64-bit multiply and add per row.  Needs rows * pitch < 4 GB and rows, pitch < 2^24 (the launchers check).
__device__ __forceinline__ float4 ld4r(const float* base, int r, unsigned ldb, int c) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (__umul24((unsigned)r, ldb) + (unsigned)c * 4u));
}
// `stream`: the output is far larger than the caches and nobody re-reads it soon, so it is written with
// non-temporal stores that leave the L2 to the rows being read (+5 % at 95 k atoms); a small output stays in
// L2 for the GEMM that consumes it next.
template <bool STREAM = false>
__device__ __forceinline__ void st4r(float* base, int r, unsigned ldb, int c, float4 v) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* p = reinterpret_cast<f4*>(reinterpret_cast<char*>(base) + (__umul24((unsigned)r, ldb) + (unsigned)c * 4u));
    f4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    if (STREAM) __builtin_nontemporal_store(w, p); else *p = w;
}
constexpr long long AGG_STREAM_BYTES = 16LL << 20;   // outputs beyond this are written non-temporally
__device__ __forceinline__ int ldi(const int32_t* base, int k) {   // base[k], k >= 0, as base + 32-bit byte offset
    return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(base) + (unsigned)k * 4u);
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Edge term e[type][channel] = edge_w[channel*4 + type] + edge_b[channel] (edge_trans applied to a one-hot),
// staged in LDS: the block reads edge_w [H,4] and edge_b once, as coalesced rows, and leaves the 4 rows in
// `tab` (4*H floats); a thread then reads the 4 channels of an edge's type with one 16-byte LDS load.
// (Read straight from global memory, the [H,4] layout costs each wave four 64-byte-strided loads plus a
// chain of selects per edge, which made the aggregation kernels instruction-bound.)
__device__ __forceinline__ void stage_edge_tab(float* tab, const float* __restrict__ w, const float* __restrict__ b, int H) {
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        const float4 wk = ld4(w + (long long)k * 4);
        const float bk = b[k];
        tab[k] = wk.x + bk; tab[H + k] = wk.y + bk; tab[2 * H + k] = wk.z + bk; tab[3 * H + k] = wk.w + bk;
    }
    __syncthreads();
}

struct AggArgs { AggGroup g[2]; int ldx, ldo, n, H4, stream_out; };
// Blocks per group: one item per thread up to 4 blocks per CU; beyond that the blocks loop, and 3 per CU
// streamed fastest from HBM (more blocks in flight measured slower).
constexpr int AGG_FWD_MAX_BLOCKS = 1024, AGG_FWD_LOOP_BLOCKS = 768;

// Forward aggregation, software-pipelined over a thread's items (node, channel chunk).  An item needs three
// dependent memory round trips: its CSR row bounds, then the neighbour indices and bond types of its first
// four edges, then the neighbour rows.  The loop keeps three items in flight, one per stage, so that an
// iteration issues the row loads of item k, the index loads of item k+1 and the bound loads of item k+2
// together and pays one round trip instead of three.  Items past the end are clamped to the last one
// (valid addresses, nothing stored).  Every load is unconditional: slots past a row's end re-read its last
// edge (or edge 0 for an empty row) with bond type 4, whose table row is -inf, so that their message
// relu(x + -inf) is exactly 0.  Rows with more than four edges finish in a loop of their own; the sum
// keeps edge order throughout.
struct AggIdx { int j0, j1, j2, j3, t0, t1, t2, t3; };

__device__ __forceinline__ AggIdx agg_load_idx(const int32_t* __restrict__ col, const int32_t* __restrict__ etype, int e, int e1, int emax) {
    // edges e .. e+3 of a row ending at e1; emax = index of the last edge of the whole graph batch (>= 0)
    const int last = min(max(e1 - 1, e), emax);
    const int ea = min(e, last), eb = min(e + 1, last), ec = min(e + 2, last), ed = min(e + 3, last);
    AggIdx r;
    r.j0 = ldi(col, ea); r.j1 = ldi(col, eb); r.j2 = ldi(col, ec); r.j3 = ldi(col, ed);
    const int ta = ldi(etype, ea) & 3, tb = ldi(etype, eb) & 3, tc = ldi(etype, ec) & 3, td = ldi(etype, ed) & 3;
    r.t0 = e < e1 ? ta : 4; r.t1 = e + 1 < e1 ? tb : 4; r.t2 = e + 2 < e1 ? tc : 4; r.t3 = e + 3 < e1 ? td : 4;
    return r;
}

__device__ __forceinline__ float4 agg_msg4(float4 acc, const float* tab, int H, int c, const AggIdx& q,
                                           float4 x0, float4 x1, float4 x2, float4 x3) {
    const float4 e0 = *reinterpret_cast<const float4*>(tab + q.t0 * H + c);
    const float4 e1 = *reinterpret_cast<const float4*>(tab + q.t1 * H + c);
    const float4 e2 = *reinterpret_cast<const float4*>(tab + q.t2 * H + c);
    const float4 e3 = *reinterpret_cast<const float4*>(tab + q.t3 * H + c);
    acc.x += fmaxf(x0.x + e0.x, 0.f); acc.y += fmaxf(x0.y + e0.y, 0.f);
    acc.z += fmaxf(x0.z + e0.z, 0.f); acc.w += fmaxf(x0.w + e0.w, 0.f);
    acc.x += fmaxf(x1.x + e1.x, 0.f); acc.y += fmaxf(x1.y + e1.y, 0.f);
    acc.z += fmaxf(x1.z + e1.z, 0.f); acc.w += fmaxf(x1.w + e1.w, 0.f);
    acc.x += fmaxf(x2.x + e2.x, 0.f); acc.y += fmaxf(x2.y + e2.y, 0.f);
    acc.z += fmaxf(x2.z + e2.z, 0.f); acc.w += fmaxf(x2.w + e2.w, 0.f);
    acc.x += fmaxf(x3.x + e3.x, 0.f); acc.y += fmaxf(x3.y + e3.y, 0.f);
    acc.z += fmaxf(x3.z + e3.z, 0.f); acc.w += fmaxf(x3.w + e3.w, 0.f);
    return acc;
}

template <bool STREAM>
__global__ __launch_bounds__(256) void gine_agg_fwd_kernel(const AggArgs a) {
    extern __shared__ __attribute__((aligned(16))) float etab_lds[];
    const AggGroup& g = a.g[blockIdx.y];
    const int H = a.H4 * 4;
    // 32-bit item arithmetic (the launcher checks n * H/4 < 2^31)
    // All blocks sweep the items together, one window of gridDim.x * 256 items at a time (HBM sees one moving
    // front, as in a plain copy).  Inside a window, workgroups are dealt to the 8 XCDs round-robin by block id,
    // so blocks b, b+8, b+16 ... take neighbouring pieces: a node's neighbours (a few rows away) are then read
    // by the same XCD, which fetches a row from HBM once and serves the re-reads from its L2.
    const unsigned H4 = (unsigned)a.H4, stride = gridDim.x * blockDim.x;
    const unsigned total = (unsigned)a.n * H4;
    const unsigned per_xcd = gridDim.x / 8, bx = blockIdx.x;
    const unsigned logical = bx < per_xcd * 8 ? (bx % 8) * per_xcd + bx / 8 : bx;
    const unsigned tfirst = min(logical * blockDim.x + threadIdx.x, total - 1);
    const bool active = logical * blockDim.x + threadIdx.x < total;
    // the edge count is the last row bound; a batch without edges reads its (unused) indices from rowptr[0] = 0
    const int E = g.rowptr[a.n], emax = max(E - 1, 0);
    const int32_t* __restrict__ col = E > 0 ? g.col : g.rowptr;
    const int32_t* __restrict__ etype = E > 0 ? g.etype : g.rowptr;
    const float* __restrict__ x = g.x;
    const unsigned ldxb = (unsigned)a.ldx * 4u, ldob = (unsigned)a.ldo * 4u;
    // pipeline fill: bounds of items 0 and 1, indices of item 0
    int iA = (int)(tfirst / H4), cA = (int)(tfirst - (unsigned)iA * H4) * 4;
    int eA0 = ldi(g.rowptr, iA), eA1 = ldi(g.rowptr, iA + 1);
    const unsigned tB = min(tfirst + stride, total - 1);
    int iB = (int)(tB / H4), cB = (int)(tB - (unsigned)iB * H4) * 4;
    int eB0 = ldi(g.rowptr, iB), eB1 = ldi(g.rowptr, iB + 1);
    AggIdx qA = agg_load_idx(col, etype, eA0, eA1, emax);
    // the edge table is staged while those index loads are in flight (a short launch is a chain of latencies)
    for (int k = threadIdx.x; k < H; k += blockDim.x) etab_lds[4 * H + k] = -INFINITY;   // row of the "no edge" type
    stage_edge_tab(etab_lds, g.edge_w, g.edge_b, H);
    if (!active) return;
    for (unsigned t = tfirst; t < total; t += stride) {
        // stage 3 of item A: its own row and four neighbour rows
        const float4 xi = ld4r(x, iA, ldxb, cA);
        const float4 x0 = ld4r(x, qA.j0, ldxb, cA), x1 = ld4r(x, qA.j1, ldxb, cA);
        const float4 x2 = ld4r(x, qA.j2, ldxb, cA), x3 = ld4r(x, qA.j3, ldxb, cA);
        // stage 2 of item B, stage 1 of item C
        const AggIdx qB = agg_load_idx(col, etype, eB0, eB1, emax);
        const unsigned tC = min(t + 2 * stride, total - 1);
        const int iC = (int)(tC / H4), cC = (int)(tC - (unsigned)iC * H4) * 4;
        const int eC0 = ldi(g.rowptr, iC), eC1 = ldi(g.rowptr, iC + 1);
        float4 acc = agg_msg4(make_float4(0.f, 0.f, 0.f, 0.f), etab_lds, H, cA, qA, x0, x1, x2, x3);
        for (int e = eA0 + 4; e < eA1; e += 4) {   // the rare row with more than four edges
            const AggIdx q = agg_load_idx(col, etype, e, eA1, emax);
            const float4 y0 = ld4r(x, q.j0, ldxb, cA), y1 = ld4r(x, q.j1, ldxb, cA);
            const float4 y2 = ld4r(x, q.j2, ldxb, cA), y3 = ld4r(x, q.j3, ldxb, cA);
            acc = agg_msg4(acc, etab_lds, H, cA, q, y0, y1, y2, y3);
        }
        st4r<STREAM>(g.out, iA, ldob, cA,
                     make_float4(acc.x + g.ope * xi.x, acc.y + g.ope * xi.y, acc.z + g.ope * xi.z, acc.w + g.ope * xi.w));
        iA = iB; cA = cB; eA0 = eB0; eA1 = eB1; qA = qB;
        iB = iC; cB = cC; eB0 = eC0; eB1 = eC1;
    }
}

struct AggBwdArgs { AggBwdGroup g[2]; int ldda, ldx, lddx, ldp, n, H4, H4b, nps, passes, stream_out; const int32_t* node_graph; };

// gate and accumulate up to four out-edges of node j: acc += v, s_type += v with v = da[i] * (x[j] + e(type) > 0)
__device__ __forceinline__ void agg_bwd_msg4(float4& acc, float4& s0, float4& s1, float4& s2, float4& s3, const float* tab, int H,
                                             int c, const AggIdx& q, float4 xj, float4 d0, float4 d1, float4 d2, float4 d3) {
    const float4 dd[4] = {d0, d1, d2, d3};
    const int tt[4] = {q.t0, q.t1, q.t2, q.t3};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // an empty slot has type 4, whose table row is -inf: its gate is closed and v = 0
        const float4 ee = *reinterpret_cast<const float4*>(tab + tt[k] * H + c);
        const float4 v = make_float4(xj.x + ee.x > 0.f ? dd[k].x : 0.f, xj.y + ee.y > 0.f ? dd[k].y : 0.f,
                                     xj.z + ee.z > 0.f ? dd[k].z : 0.f, xj.w + ee.w > 0.f ? dd[k].w : 0.f);
        acc = add4(acc, v);
        // per-type sums: s_k += [type == k] * v, one fused multiply-add each (exactly s_k + v or s_k)
        const float m0 = tt[k] == 0 ? 1.f : 0.f, m1 = tt[k] == 1 ? 1.f : 0.f, m2 = tt[k] == 2 ? 1.f : 0.f, m3 = tt[k] == 3 ? 1.f : 0.f;
        s0 = make_float4(fmaf(m0, v.x, s0.x), fmaf(m0, v.y, s0.y), fmaf(m0, v.z, s0.z), fmaf(m0, v.w, s0.w));
        s1 = make_float4(fmaf(m1, v.x, s1.x), fmaf(m1, v.y, s1.y), fmaf(m1, v.z, s1.z), fmaf(m1, v.w, s1.w));
        s2 = make_float4(fmaf(m2, v.x, s2.x), fmaf(m2, v.y, s2.y), fmaf(m2, v.z, s2.z), fmaf(m2, v.w, s2.w));
        s3 = make_float4(fmaf(m3, v.x, s3.x), fmaf(m3, v.y, s3.y), fmaf(m3, v.z, s3.z), fmaf(m3, v.w, s3.w));
    }
}

// Backward of the aggregation, by SOURCE node: dx[j] = (1+eps) da[j] + sum over out-edges (j -> i) of
// da[i] * (x[j] + e(type) > 0)  (+ the pooled-gradient broadcast).  The same products, summed per bond
// type instead of per node, are the gradient of the edge table, so the kernel accumulates those on the
// way: a block owns passes * nps consecutive nodes (nps = 256 / H4b node slots, every thread keeps one
// channel chunk and walks the nodes slot, slot + nps, ...; rows wider than 256 chunks are split over
// blockIdx.z in pieces of H4b chunks), combines its slots in order through LDS and writes one [4][H]
// partial per node block.  A thread's walk is software-pipelined like the forward kernel's: the rows of
// node k, the out-edge indices of node k+1 and the CSR bounds of node k+2 are fetched together.
template <bool STREAM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void gine_agg_bwd_kernel(const AggBwdArgs a) {
    __shared__ float4 part[4][256];
    extern __shared__ __attribute__((aligned(16))) float etab_lds[];
    const AggBwdGroup& g = a.g[blockIdx.y];
    const int tid = threadIdx.x;
    const int slot = tid / a.H4b, cl = tid - slot * a.H4b, c4 = blockIdx.z * a.H4b + cl, c = c4 * 4;
    const bool lane_on = slot < a.nps && c4 < a.H4;
    const int H = a.H4 * 4;
    // workgroups are dealt to the 8 XCDs round-robin by block id: blocks b, b+8, b+16 ... take neighbouring node
    // ranges, so a range and the neighbour rows just outside it are read through the same XCD's L2
    const int per_xcd = gridDim.x / 8, bx = blockIdx.x;
    const int logical = bx < per_xcd * 8 ? (bx % 8) * per_xcd + bx / 8 : bx;
    const int node0 = logical * a.nps * a.passes, nend = min(a.n, node0 + a.nps * a.passes);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!g.da) {   // uniform over the block: only the pooled-gradient broadcast is written
        for (int j = node0 + slot; lane_on && j < nend; j += a.nps)
            st4(g.dx + (long long)j * a.lddx + c, g.dpool ? ld4(g.dpool + (long long)a.node_graph[j] * a.ldp + c) : zero);
        return;
    }
    float4 s0 = zero, s1 = zero, s2 = zero, s3 = zero;
    const bool active = lane_on && node0 + slot < nend;
    // pipeline fill first (every thread, from clamped node numbers), so that the staging of the edge table below
    // overlaps the latency of these index loads
    {
        // the edge count is the last row bound; a batch without edges reads its (unused) indices from rowptr_t[0] = 0
        const int E = g.rowptr_t[a.n], emax = max(E - 1, 0);
        const int32_t* __restrict__ col = E > 0 ? g.col_t : g.rowptr_t;
        const int32_t* __restrict__ etype = E > 0 ? g.etype_t : g.rowptr_t;
        const bool has_pool = g.dpool != nullptr;
        const int32_t* __restrict__ ngraph = has_pool ? a.node_graph : g.rowptr_t;   // without a pooled gradient: any readable word
        const float* __restrict__ pool = has_pool ? g.dpool : g.da;
        const float* __restrict__ x = g.x;
        const float* __restrict__ da = g.da;
        const unsigned ldxb = (unsigned)a.ldx * 4u, lddab = (unsigned)a.ldda * 4u, lddxb = (unsigned)a.lddx * 4u, ldpb = (unsigned)a.ldp * 4u;
        int jA = min(node0 + slot, nend - 1), jB = min(jA + a.nps, nend - 1);
        int eA0 = ldi(g.rowptr_t, jA), eA1 = ldi(g.rowptr_t, jA + 1), gA = has_pool ? ldi(ngraph, jA) : 0;
        int eB0 = ldi(g.rowptr_t, jB), eB1 = ldi(g.rowptr_t, jB + 1), gB = has_pool ? ldi(ngraph, jB) : 0;
        AggIdx qA = agg_load_idx(col, etype, eA0, eA1, emax);
        for (int k = tid; k < H; k += 256) etab_lds[4 * H + k] = -INFINITY;   // row of the "no edge" type
        stage_edge_tab(etab_lds, g.edge_w, g.edge_b, H);
        for (int j = active ? jA : nend; j < nend; j += a.nps) {
            // stage 3 of node A: its own rows, the rows of its four first out-neighbours, its graph's pooled gradient
            const float4 xj = ld4r(x, jA, ldxb, c), dj = ld4r(da, jA, lddab, c);
            const float4 d0 = ld4r(da, qA.j0, lddab, c), d1 = ld4r(da, qA.j1, lddab, c);
            const float4 d2 = ld4r(da, qA.j2, lddab, c), d3 = ld4r(da, qA.j3, lddab, c);
            const float4 dp = ld4r(pool, gA, ldpb, c);
            // stage 2 of node B, stage 1 of node C
            const AggIdx qB = agg_load_idx(col, etype, eB0, eB1, emax);
            const int jC = min(j + 2 * a.nps, nend - 1);
            const int eC0 = ldi(g.rowptr_t, jC), eC1 = ldi(g.rowptr_t, jC + 1), gC = has_pool ? ldi(ngraph, jC) : 0;
            float4 acc = zero;
            agg_bwd_msg4(acc, s0, s1, s2, s3, etab_lds, H, c, qA, xj, d0, d1, d2, d3);
            for (int e = eA0 + 4; e < eA1; e += 4) {   // the rare node with more than four out-edges
                const AggIdx q = agg_load_idx(col, etype, e, eA1, emax);
                const float4 y0 = ld4r(da, q.j0, lddab, c), y1 = ld4r(da, q.j1, lddab, c);
                const float4 y2 = ld4r(da, q.j2, lddab, c), y3 = ld4r(da, q.j3, lddab, c);
                agg_bwd_msg4(acc, s0, s1, s2, s3, etab_lds, H, c, q, xj, y0, y1, y2, y3);
            }
            acc = make_float4(acc.x + g.ope * dj.x, acc.y + g.ope * dj.y, acc.z + g.ope * dj.z, acc.w + g.ope * dj.w);
            if (has_pool) acc = add4(acc, dp);
            st4r<STREAM>(g.dx, jA, lddxb, c, acc);
            jA = jB; eA0 = eB0; eA1 = eB1; gA = gB; qA = qB;
            jB = jC; eB0 = eC0; eB1 = eC1; gB = gC;
        }
    }
    if (!g.partial) return;   // uniform over the block
    part[0][tid] = s0; part[1][tid] = s1; part[2][tid] = s2; part[3][tid] = s3;
    __syncthreads();
    if (tid < a.H4b && c4 < a.H4) {
        float* out = g.partial + (long long)logical * 16 * a.H4 + c;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            float4 v = part[tp][tid];
            for (int sl = 1; sl < a.nps; ++sl) v = add4(v, part[tp][sl * a.H4b + tid]);
            st4(out + tp * 4 * a.H4, v);
        }
    }
}

// Edge-table gradient from the per-block partials, as an ordered two-level sum.  One launch reduces
// `np` partials [4][H]: blockIdx.z takes a contiguous segment of them, its 16 thread rows take
// contiguous slices of the segment, rows are combined in order through LDS.  With `final` unset the
// block writes its [4][H] sum to out_partial[blockIdx.z]; with it set (one segment) it writes
// d_edge_w[c][type] and d_edge_b[c] = sum over types.
struct EtabArgs { EtabFinalGroup g[2]; int np, H4, final; };

__global__ __launch_bounds__(256) void gine_etab_reduce_kernel(const EtabArgs a) {
    __shared__ float4 part[16][4][16];
    const EtabFinalGroup& g = a.g[blockIdx.y];
    const int H = a.H4 * 4, tx = threadIdx.x, ty = threadIdx.y;
    const int c4 = blockIdx.x * 16 + tx, c = c4 * 4;
    const int seg = (a.np + gridDim.z - 1) / gridDim.z;
    const int q0 = min(a.np, (int)blockIdx.z * seg), q1 = min(a.np, q0 + seg);
    const int per = (q1 - q0 + 15) / 16;
    const int p0 = min(q1, q0 + ty * per), p1 = min(q1, p0 + per);
    const float* in = a.final ? g.seg_partial : g.partial;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (c4 < a.H4) {
#pragma unroll 4
        for (int p = p0; p < p1; ++p) {
            const float* src = in + (long long)p * 4 * H + c;
            s0 = add4(s0, ld4(src)); s1 = add4(s1, ld4(src + H)); s2 = add4(s2, ld4(src + 2 * H)); s3 = add4(s3, ld4(src + 3 * H));
        }
    }
    part[ty][0][tx] = s0; part[ty][1][tx] = s1; part[ty][2][tx] = s2; part[ty][3][tx] = s3;
    __syncthreads();
    if (ty < 4 && c4 < a.H4) {   // thread (tx, ty) adds up bond type ty, slices in order
        float4 v = part[0][ty][tx];
#pragma unroll
        for (int y = 1; y < 16; ++y) v = add4(v, part[y][ty][tx]);
        if (!a.final) st4(g.seg_partial + ((long long)blockIdx.z * 4 + ty) * H + c, v);
        part[0][ty][tx] = v;
    }
    if (!a.final) return;
    __syncthreads();
    if (ty == 0 && c4 < a.H4) {
        const float4 w0 = part[0][0][tx], w1 = part[0][1][tx], w2 = part[0][2][tx], w3 = part[0][3][tx];
        // d_edge_w is [H][4]: channel c+k holds (type0..3)
        st4(g.d_edge_w + (long long)c * 4, make_float4(w0.x, w1.x, w2.x, w3.x));
        st4(g.d_edge_w + (long long)(c + 1) * 4, make_float4(w0.y, w1.y, w2.y, w3.y));
        st4(g.d_edge_w + (long long)(c + 2) * 4, make_float4(w0.z, w1.z, w2.z, w3.z));
        st4(g.d_edge_w + (long long)(c + 3) * 4, make_float4(w0.w, w1.w, w2.w, w3.w));
        st4(g.d_edge_b + c, make_float4(((w0.x + w1.x) + w2.x) + w3.x, ((w0.y + w1.y) + w2.y) + w3.y,
                                        ((w0.z + w1.z) + w2.z) + w3.z, ((w0.w + w1.w) + w2.w) + w3.w));
    }
}

__global__ __launch_bounds__(256) void embed3_gather_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ p,
                                                            const int32_t* __restrict__ q, const float* __restrict__ ta,
                                                            const float* __restrict__ tp, const float* __restrict__ tq,
                                                            int da, int dp, int dq, float* __restrict__ out, int n) {
    const int D = da + dp + dq;
    const long long total = (long long)n * D;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(t / D), c = (int)(t - (long long)i * D);
        float v;
        if (c < da) v = ta[(long long)a[i] * da + c];
        else if (c < da + dp) v = tp[(long long)p[i] * dp + (c - da)];
        else v = tq[(long long)q[i] * dq + (c - da - dp)];
        out[t] = v;
    }
}

// out[r, :] = table[idx[(r / inner) * idx_ld + r % inner], :]
__global__ __launch_bounds__(256) void rows_gather_kernel(const int32_t* __restrict__ idx, int inner, int idx_ld,
                                                          const float* __restrict__ table, int width,
                                                          float* __restrict__ out, int rows) {
    const long long total = (long long)rows * width;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(t / width), c = (int)(t - (long long)r * width);
        const int id = idx[(long long)(r / inner) * idx_ld + (r % inner)];
        out[t] = table[(long long)id * width + c];
    }
}

// Embedding-table gradients for up to 3 tables in one launch.  Every table row owns up to
// two contribution lists (index lists built on the host).  Stage 1: block (row, segment)
// sums one of TG_SEG ordered segments of a list (threadIdx.y = quarter of the segment,
// combined in order through LDS) into partial[row][segment][:].  Stage 2 adds a row's
// non-empty segments in order.  No atomics: the sum order is fixed by the lists.
constexpr int TG_SEG = 16;
struct TableArgs { TableGradDesc t[3]; int row0[4]; int nt, dmax; float* partial; };

__device__ __forceinline__ int tg_find(const TableArgs& a, int grow) {
    int k = 0;
    while (k + 1 < a.nt && grow >= a.row0[k + 1]) ++k;
    return k;
}

__global__ __launch_bounds__(256) void table_grad_partial_kernel(const TableArgs a) {
    __shared__ float part[4][64];
    const int grow = blockIdx.x, yy = blockIdx.y, tx = threadIdx.x, ty = threadIdx.y;
    const TableGradDesc& t = a.t[tg_find(a, grow)];
    const int row = grow - a.row0[tg_find(a, grow)];
    const bool second = yy >= TG_SEG;
    const int32_t* ptr = second ? t.ptr2 : t.ptr1;
    if (!ptr) return;
    const int32_t* idx = second ? t.idx2 : t.idx1;
    const float* src = second ? t.src2 : t.src1;
    const int ld = second ? t.ld2 : t.ld1, off = second ? t.off2 : t.off1;
    const int y = second ? yy - TG_SEG : yy;
    const int b = ptr[row], e = ptr[row + 1], per = (e - b + TG_SEG - 1) / TG_SEG;
    const int k0 = b + y * per, k1 = min(e, k0 + per);
    if (k0 >= k1) return;                       // block-uniform: empty segment, nothing written
    const int q = (k1 - k0 + 3) / 4;
    const int ka = min(k1, k0 + ty * q), kb = min(k1, ka + q);
    float* out = a.partial + ((long long)grow * (2 * TG_SEG) + yy) * a.dmax;
    for (int c0 = 0; c0 < t.D; c0 += 64) {
        const int c = c0 + tx;
        float s = 0.f;
        if (c < t.D) {
            // 8 list entries in flight per round trip (slots past the end re-read the last entry and add 0)
            for (int k = ka; k < kb; k += 8) {
                const int last = kb - 1;
                int j[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) j[u] = idx[min(k + u, last)];
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(long long)j[u] * ld + off + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += k + u <= last ? v[u] : 0.f;
            }
        }
        part[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && c < t.D) out[c] = ((part[0][tx] + part[1][tx]) + part[2][tx]) + part[3][tx];
        __syncthreads();
    }
}

__global__ __launch_bounds__(128) void table_grad_final_kernel(const TableArgs a) {
    const int grow = blockIdx.x;
    const int ti = tg_find(a, grow);
    const TableGradDesc& t = a.t[ti];
    const int row = grow - a.row0[ti];
    const int len1 = t.ptr1[row + 1] - t.ptr1[row], per1 = (len1 + TG_SEG - 1) / TG_SEG;
    const int len2 = t.ptr2 ? t.ptr2[row + 1] - t.ptr2[row] : 0, per2 = (len2 + TG_SEG - 1) / TG_SEG;
    const float* in = a.partial + (long long)grow * (2 * TG_SEG) * a.dmax;
    for (int c = threadIdx.x; c < t.D; c += blockDim.x) {
        float v = 0.f;
        for (int y = 0; y < TG_SEG && y * per1 < len1; ++y) v += in[(long long)y * a.dmax + c];
        for (int y = 0; y < TG_SEG && y * per2 < len2; ++y) v += in[(long long)(TG_SEG + y) * a.dmax + c];
        t.dtable[(long long)row * t.D + c] = v;
    }
}

// pooled[b, c] = sum over the graph's rows, in row order.  The input is a sequence of slices [rows, ldx] that lie
// `slice_stride` floats apart; slice k supplies the output columns [k * slice_w, (k+1) * slice_w).
__global__ __launch_bounds__(256) void segment_pool_kernel(const float* __restrict__ x, int ldx, long long slice_stride,
                                                           int slice_w4, const int32_t* __restrict__ graph_off,
                                                           float* __restrict__ pooled, int ldp, int W4) {
    const int b = blockIdx.y;
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= W4) return;
    const int c = c4 * 4;
    const int slice = c4 / slice_w4;
    x += slice * slice_stride + (c4 - slice * slice_w4) * 4 - c;   // so that x[row * ldx + c] is the element
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int i1 = graph_off[b + 1];
    // 8 rows in flight per round trip (rows past the end re-read the last one and add 0); summed in row order
    for (int i = graph_off[b]; i < i1; i += 8) {
        const int last = i1 - 1;
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ld4(x + (long long)min(i + k, last) * ldx + c);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = add4(acc, i + k <= last ? v[k] : zero);
    }
    st4(pooled + (long long)b * ldp + c, acc);
}

// pair_in[s] = [emb[src_s] | emb[dst_s] | z[graph_s]]
__global__ __launch_bounds__(256) void pair_gather_kernel(const float* __restrict__ emb, int lde, const float* __restrict__ z,
                                                          const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                          const int32_t* __restrict__ graph, float* __restrict__ out,
                                                          int S, int H4, int Z4) {
    const int P4 = 2 * H4 + Z4;
    const long long total = (long long)S * P4;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(t / P4), c4 = (int)(t - (long long)s * P4);
        float4 v;
        if (c4 < H4) v = ld4(emb + (long long)src[s] * lde + c4 * 4);
        else if (c4 < 2 * H4) v = ld4(emb + (long long)dst[s] * lde + (c4 - H4) * 4);
        else v = ld4(z + (long long)graph[s] * (Z4 * 4) + (c4 - 2 * H4) * 4);
        st4(out + t * 4, v);
    }
}

// demb[j] = sum_{s: src_s = j} dpin[s, 0:H] + sum_{s: dst_s = j} dpin[s, H:2H]   (blocks 0..node_blocks-1)
// dz[b]  (+)= sum_{s in graph b} dpin[s, 2H:2H+Z]                                (one block per graph after them)
// Index and row loads go out four at a time (slots past the end re-read the last entry and add 0), so a
// node costs two memory round trips per four pairs instead of two per pair.  A graph's latent gradient is
// summed by 16 slots that stride over its pairs and are then combined in slot order through LDS.
__device__ __forceinline__ float4 sum_rows4(const float* __restrict__ base, int P, int off, const int32_t* __restrict__ idx,
                                            int k0, int k1, float4 acc) {
    for (int k = k0; k < k1; k += 4) {
        const int last = k1 - 1;
        const int r0 = idx[k], r1 = idx[min(k + 1, last)], r2 = idx[min(k + 2, last)], r3 = idx[min(k + 3, last)];
        const float4 v0 = ld4(base + (long long)r0 * P + off), v1 = ld4(base + (long long)r1 * P + off);
        const float4 v2 = ld4(base + (long long)r2 * P + off), v3 = ld4(base + (long long)r3 * P + off);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        acc = add4(acc, v0);
        acc = add4(acc, k + 1 <= last ? v1 : zero);
        acc = add4(acc, k + 2 <= last ? v2 : zero);
        acc = add4(acc, k + 3 <= last ? v3 : zero);
    }
    return acc;
}

__global__ __launch_bounds__(256) void pair_scatter_kernel(const float* __restrict__ dpin, int P,
                                                           const int32_t* __restrict__ sp, const int32_t* __restrict__ si,
                                                           const int32_t* __restrict__ dp, const int32_t* __restrict__ di,
                                                           const int32_t* __restrict__ goff, float* __restrict__ demb,
                                                           int ldde, float* __restrict__ dz, int dz_acc, int n, int B,
                                                           int H4, int Z4, int node_blocks) {
    __shared__ float4 part[16][16];
    if ((int)blockIdx.x < node_blocks) {
        const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (t >= (long long)n * H4) return;
        const int j = (int)(t / H4), c = (int)(t - (long long)j * H4) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        acc = sum_rows4(dpin, P, c, si, sp[j], sp[j + 1], acc);
        acc = sum_rows4(dpin, P, H4 * 4 + c, di, dp[j], dp[j + 1], acc);
        st4(demb + (long long)j * ldde + c, acc);
        return;
    }
    const int b = blockIdx.x - node_blocks, slot = threadIdx.x >> 4, cq = threadIdx.x & 15;
    const int s0 = goff[b], s1 = goff[b + 1], off = 2 * H4 * 4;
    for (int c4 = cq; c4 - cq < Z4; c4 += 16) {      // uniform trip count over the block
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < Z4) {
            const float* col = dpin + off + c4 * 4;
            for (int s = s0 + slot; s < s1; s += 64) {
                const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
                const int last = s1 - 1;
                const float4 v0 = ld4(col + (long long)s * P), v1 = ld4(col + (long long)min(s + 16, last) * P);
                const float4 v2 = ld4(col + (long long)min(s + 32, last) * P), v3 = ld4(col + (long long)min(s + 48, last) * P);
                acc = add4(acc, v0);
                acc = add4(acc, s + 16 <= last ? v1 : zero);
                acc = add4(acc, s + 32 <= last ? v2 : zero);
                acc = add4(acc, s + 48 <= last ? v3 : zero);
            }
        }
        part[slot][cq] = acc;
        __syncthreads();
        if (slot == 0 && c4 < Z4) {
            float4 v = part[0][cq];
            for (int q = 1; q < 16; ++q) v = add4(v, part[q][cq]);
            float* d = dz + (long long)b * (Z4 * 4) + c4 * 4;
            if (dz_acc) v = add4(v, ld4(d));
            st4(d, v);
        }
        __syncthreads();
    }
}

__global__ void fill_zero_kernel(float* p, long long count) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (long long)gridDim.x * blockDim.x) p[t] = 0.f;
}

inline int grid_for(long long items, int block, int cap = 4096) {
    long long b = (items + block - 1) / block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

}  // namespace

int gine_agg_fwd(hipStream_t s, const AggGroup* g, int ngroups, int ldx, int ldo, int n, int H) {
    if (n == 0) return 0;
    if (H % 4 || ldx % 4 || ldo % 4) return PSVAE_E_ALIGN;
    if (H > 3072) return PSVAE_E_SHAPE;   // the edge table of a block lives in LDS (20 bytes per channel)
    if (ngroups < 1 || ngroups > 2) return PSVAE_E_SHAPE;
    AggArgs a;
    for (int k = 0; k < ngroups; ++k) {
        a.g[k] = g[k];
        if (!aligned16(g[k].x) || !aligned16(g[k].out) || !aligned16(g[k].edge_w) || !aligned16(g[k].edge_b)) return PSVAE_E_ALIGN;
    }
    if (ngroups == 1) a.g[1] = g[0];
    a.ldx = ldx; a.ldo = ldo; a.n = n; a.H4 = H / 4; a.stream_out = (long long)n * H * 4 > AGG_STREAM_BYTES;
    const long long items = (long long)n * a.H4;
    if (items + 2LL * AGG_FWD_MAX_BLOCKS * 256 >= (1LL << 31)) return PSVAE_E_SHAPE;   // the kernel counts items in 32 bits
    // ... and addresses rows by 32-bit byte offsets built with 24-bit multiplies
    if (n >= (1 << 24) || ldx >= (1 << 22) || ldo >= (1 << 22) || (long long)n * ldx >= (1LL << 30) || (long long)n * ldo >= (1LL << 30)) return PSVAE_E_SHAPE;
    const long long blocks = (items + 255) / 256;
    dim3 grid((unsigned)(blocks <= AGG_FWD_MAX_BLOCKS ? blocks : AGG_FWD_LOOP_BLOCKS), ngroups);
    double bytes = 0;
    for (int k = 0; k < ngroups; ++k) bytes += 2.0 * n * 4 * H + 5.0 * g[k].E + 4.0 * (n + 1);
    ProfScope prof(s, KC_AGG_FWD, bytes);
    if (a.stream_out) hipLaunchKernelGGL(gine_agg_fwd_kernel<true>, grid, dim3(256), (size_t)20 * H, s, a);
    else hipLaunchKernelGGL(gine_agg_fwd_kernel<false>, grid, dim3(256), (size_t)20 * H, s, a);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

// A block of the backward kernel owns nps * passes nodes.  Small inputs use 2 passes (many short blocks);
// larger ones lengthen the walk instead of adding blocks beyond AGG_BWD_MAX_BLOCKS, which also bounds the
// number of edge-table partials the reduction has to read.
constexpr int AGG_BWD_MAX_BLOCKS = 1024;
static int agg_bwd_passes(int n, int H) {
    const int H4 = H / 4, H4b = H4 < 256 ? (H4 < 1 ? 1 : H4) : 256, nps = 256 / H4b;
    const long long per_cap = ((long long)n + (long long)nps * AGG_BWD_MAX_BLOCKS - 1) / ((long long)nps * AGG_BWD_MAX_BLOCKS);
    return (int)(per_cap > 2 ? per_cap : 2);
}
int gine_agg_bwd_blocks(int n, int H) {
    const int H4 = H / 4, H4b = H4 < 256 ? (H4 < 1 ? 1 : H4) : 256;
    const int per = (256 / H4b) * agg_bwd_passes(n, H);
    return (n + per - 1) / per;
}

int gine_agg_bwd(hipStream_t s, const AggBwdGroup* g, int ngroups, int ldda, int ldx, int lddx, int ldp,
                 const int32_t* node_graph, int n, int H) {
    if (n == 0) return 0;
    if (H % 4 || ldda % 4 || ldx % 4 || lddx % 4 || ldp % 4) return PSVAE_E_ALIGN;
    if (ngroups < 1 || ngroups > 2 || H < 4) return PSVAE_E_SHAPE;
    {   // the kernel addresses rows by 32-bit byte offsets built with 24-bit multiplies
        const int ldmax = ldda > ldx ? (ldda > lddx ? ldda : lddx) : (ldx > lddx ? ldx : lddx);
        if (n >= (1 << 24) || ldmax >= (1 << 22) || ldp >= (1 << 22) || (long long)n * ldmax >= (1LL << 30)) return PSVAE_E_SHAPE;
        // the pooled gradient is indexed by graph; there are at most n graphs
        for (int k = 0; k < ngroups; ++k)
            if (g[k].dpool && (long long)n * ldp >= (1LL << 30)) return PSVAE_E_SHAPE;
    }
    bool need_tab = false;
    for (int k = 0; k < ngroups; ++k) need_tab = need_tab || g[k].da != nullptr;
    if (need_tab && H > 2048) return PSVAE_E_SHAPE;   // the edge table sits in LDS (20 bytes per channel) beside the partials
    AggBwdArgs a;
    for (int k = 0; k < ngroups; ++k) {
        a.g[k] = g[k];
        if (!aligned16(g[k].dx) || (g[k].da && (!aligned16(g[k].da) || !aligned16(g[k].x))) ||
            (g[k].dpool && !aligned16(g[k].dpool)) || (g[k].partial && !aligned16(g[k].partial))) return PSVAE_E_ALIGN;
        if (g[k].dpool && !node_graph) return PSVAE_E_NULL;
        if (g[k].partial && !g[k].da) return PSVAE_E_NULL;
    }
    if (ngroups == 1) a.g[1] = g[0];
    a.ldda = ldda; a.ldx = ldx; a.lddx = lddx; a.ldp = ldp; a.n = n; a.H4 = H / 4; a.H4b = a.H4 < 256 ? a.H4 : 256;
    a.nps = 256 / a.H4b; a.passes = agg_bwd_passes(n, H); a.node_graph = node_graph; a.stream_out = (long long)n * H * 4 > AGG_STREAM_BYTES;
    dim3 grid((unsigned)gine_agg_bwd_blocks(n, H), ngroups, (a.H4 + a.H4b - 1) / a.H4b);
    double bytes = 0;
    for (int k = 0; k < ngroups; ++k) bytes += g[k].da ? 3.0 * n * 4 * H + 5.0 * g[k].E + 4.0 * (n + 1) : 2.0 * n * 4 * H;
    ProfScope prof(s, KC_AGG_BWD, bytes);
    if (a.stream_out) hipLaunchKernelGGL(gine_agg_bwd_kernel<true>, grid, dim3(256), need_tab ? (size_t)20 * H : 0, s, a);
    else hipLaunchKernelGGL(gine_agg_bwd_kernel<false>, grid, dim3(256), need_tab ? (size_t)20 * H : 0, s, a);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int gine_etab_final(hipStream_t s, const EtabFinalGroup* g, int ngroups, int np, int H) {
    if (H % 4) return PSVAE_E_ALIGN;
    if (ngroups < 1 || ngroups > 2 || np < 1) return PSVAE_E_SHAPE;
    EtabArgs a;
    for (int k = 0; k < ngroups; ++k) {
        a.g[k] = g[k];
        if (!g[k].partial || !g[k].seg_partial || !g[k].d_edge_w || !g[k].d_edge_b) return PSVAE_E_NULL;
    }
    if (ngroups == 1) a.g[1] = g[0];
    a.H4 = H / 4;
    const int segs = np < ETAB_SEGMENTS ? np : ETAB_SEGMENTS;
    ProfScope prof(s, KC_ETAB, (double)ngroups * (np + 2.0 * segs) * 16.0 * H);
    a.np = np; a.final = 0;
    hipLaunchKernelGGL(gine_etab_reduce_kernel, dim3((a.H4 + 15) / 16, ngroups, segs), dim3(16, 16), 0, s, a);
    PSVAE_LAUNCH_CHECK();
    a.np = segs; a.final = 1;
    hipLaunchKernelGGL(gine_etab_reduce_kernel, dim3((a.H4 + 15) / 16, ngroups, 1), dim3(16, 16), 0, s, a);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int embed3_gather(hipStream_t s, const int32_t* a, const int32_t* p, const int32_t* q, const float* ta,
                  const float* tp, const float* tq, int da, int dp, int dq, float* out, int n) {
    if (n == 0) return 0;
    ProfScope prof(s, KC_EMBED, (double)n * (12 + 4.0 * (da + dp + dq)));
    hipLaunchKernelGGL(embed3_gather_kernel, dim3(grid_for((long long)n * (da + dp + dq), 256)), dim3(256), 0, s,
                       a, p, q, ta, tp, tq, da, dp, dq, out, n);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int rows_gather(hipStream_t s, const int32_t* idx, int /*idx_stride_outer*/, int inner, int idx_ld, const float* table,
                int width, float* out, int rows) {
    if (rows == 0) return 0;
    ProfScope prof(s, KC_EMBED, (double)rows * (4 + 4.0 * width));
    hipLaunchKernelGGL(rows_gather_kernel, dim3(grid_for((long long)rows * width, 256)), dim3(256), 0, s,
                       idx, inner, idx_ld, table, width, out, rows);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

size_t table_grads_scratch(const TableGradDesc* t, int nt) {
    size_t rows = 0; int dmax = 0;
    for (int k = 0; k < nt; ++k) { rows += t[k].rows; if (t[k].D > dmax) dmax = t[k].D; }
    return rows * 2 * TG_SEG * (size_t)dmax;
}

int table_grads(hipStream_t s, const TableGradDesc* t, int nt, float* partial, size_t partial_floats) {
    if (nt < 1 || nt > 3) return PSVAE_E_SHAPE;
    if (!partial || partial_floats < table_grads_scratch(t, nt)) return PSVAE_E_WORKSPACE;
    TableArgs a;
    int rows = 0, dmax = 0, any2 = 0;
    double bytes = 0;
    for (int k = 0; k < nt; ++k) {
        if (!t[k].dtable || !t[k].ptr1 || !t[k].idx1 || !t[k].src1) return PSVAE_E_NULL;
        a.t[k] = t[k]; a.row0[k] = rows; rows += t[k].rows;
        if (t[k].D > dmax) dmax = t[k].D;
        if (t[k].ptr2) any2 = 1;
        bytes += (double)t[k].rows * t[k].D * 4.0;
    }
    a.row0[nt] = rows; a.nt = nt; a.dmax = dmax; a.partial = partial;
    if (rows == 0 || dmax == 0) return 0;
    ProfScope prof(s, KC_TABLE_GRAD, bytes);
    hipLaunchKernelGGL(table_grad_partial_kernel, dim3(rows, any2 ? 2 * TG_SEG : TG_SEG), dim3(64, 4), 0, s, a);
    PSVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(table_grad_final_kernel, dim3(rows), dim3(128), 0, s, a);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int segment_pool_fwd(hipStream_t s, const float* x, int ldx, int64_t slice_stride, int slice_width,
                     const int32_t* graph_off, float* pooled, int ldp, int B, int width) {
    if (B == 0) return 0;
    if (width % 4 || ldx % 4 || ldp % 4 || slice_stride % 4 || slice_width % 4 || !aligned16(x) || !aligned16(pooled)) return PSVAE_E_ALIGN;
    if (slice_width < 4) return PSVAE_E_SHAPE;
    const int W4 = width / 4;
    ProfScope prof(s, KC_POOL, 4.0 * width * (graph_off ? 1 : 1) * (double)B);
    hipLaunchKernelGGL(segment_pool_kernel, dim3((W4 + 255) / 256, B), dim3(256), 0, s, x, ldx, (long long)slice_stride, slice_width / 4,
                       graph_off, pooled, ldp, W4);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int pair_gather(hipStream_t s, const float* emb, int lde, const float* z, const int32_t* src, const int32_t* dst,
                const int32_t* graph, float* out, int S, int H, int Z) {
    if (S == 0) return 0;
    if (H % 4 || Z % 4 || lde % 4 || !aligned16(emb) || !aligned16(z) || !aligned16(out)) return PSVAE_E_ALIGN;
    ProfScope prof(s, KC_PAIR_GATHER, (double)S * 4.0 * (2 * H + Z) * 2);
    hipLaunchKernelGGL(pair_gather_kernel, dim3(grid_for((long long)S * (2 * H + Z) / 4, 256)), dim3(256), 0, s,
                       emb, lde, z, src, dst, graph, out, S, H / 4, Z / 4);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int pair_scatter(hipStream_t s, const float* dpin, int P, const int32_t* by_src_ptr, const int32_t* by_src_idx,
                 const int32_t* by_dst_ptr, const int32_t* by_dst_idx, const int32_t* pair_graph_off, float* demb,
                 int ldde, float* dz, int dz_accumulate, int n, int B, int H, int Z) {
    if (H % 4 || Z % 4 || P % 4 || ldde % 4) return PSVAE_E_ALIGN;
    const long long items = (long long)n * (H / 4) + (long long)B * (Z / 4);
    if (items == 0) return 0;
    ProfScope prof(s, KC_PAIR_SCATTER, (double)items * 16.0 * 2);
    const int node_blocks = (int)(((long long)n * (H / 4) + 255) / 256);
    hipLaunchKernelGGL(pair_scatter_kernel, dim3((unsigned)(node_blocks + B)), dim3(256), 0, s, dpin, P, by_src_ptr,
                       by_src_idx, by_dst_ptr, by_dst_idx, pair_graph_off, demb, ldde, dz, dz_accumulate, n, B, H / 4, Z / 4,
                       node_blocks);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int fill_zero(hipStream_t s, float* p, int64_t count) {
    if (count <= 0) return 0;
    ProfScope prof(s, KC_HEAD, 4.0 * count);
    hipLaunchKernelGGL(fill_zero_kernel, dim3(grid_for(count, 256)), dim3(256), 0, s, p, (long long)count);
    PSVAE_LAUNCH_CHECK();
    return 0;
}
