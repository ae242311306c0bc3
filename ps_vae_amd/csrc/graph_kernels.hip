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

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Edge term of 4 consecutive channels for each of the 4 bond types:
// e[type][k] = edge_w[(c+k)*4 + type] + edge_b[c+k]   (edge_trans applied to a one-hot)
struct EdgeTab { float4 t0, t1, t2, t3; };
__device__ __forceinline__ EdgeTab load_edge_tab(const float* __restrict__ w, const float* __restrict__ b, int c) {
    const float4 w0 = ld4(w + (long long)c * 4), w1 = ld4(w + (long long)(c + 1) * 4);
    const float4 w2 = ld4(w + (long long)(c + 2) * 4), w3 = ld4(w + (long long)(c + 3) * 4);
    const float4 bb = ld4(b + c);
    EdgeTab e;
    e.t0 = make_float4(w0.x + bb.x, w1.x + bb.y, w2.x + bb.z, w3.x + bb.w);
    e.t1 = make_float4(w0.y + bb.x, w1.y + bb.y, w2.y + bb.z, w3.y + bb.w);
    e.t2 = make_float4(w0.z + bb.x, w1.z + bb.y, w2.z + bb.z, w3.z + bb.w);
    e.t3 = make_float4(w0.w + bb.x, w1.w + bb.y, w2.w + bb.z, w3.w + bb.w);
    return e;
}
__device__ __forceinline__ float4 pick(const EdgeTab& e, int ty) {
    return ty == 1 ? e.t1 : (ty == 2 ? e.t2 : (ty == 3 ? e.t3 : e.t0));
}

struct AggArgs { AggGroup g[2]; int ldx, ldo, n, H4; };

__global__ __launch_bounds__(256) void gine_agg_fwd_kernel(const AggArgs a) {
    const AggGroup& g = a.g[blockIdx.y];
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)a.n * a.H4) return;
    const int i = (int)(t / a.H4), c = (int)(t - (long long)i * a.H4) * 4;
    const EdgeTab et = load_edge_tab(g.edge_w, g.edge_b, c);
    const float4 xi = ld4(g.x + (long long)i * a.ldx + c);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int e1 = g.rowptr[i + 1];
    for (int e = g.rowptr[i]; e < e1; ++e) {
        const float4 xj = ld4(g.x + (long long)g.col[e] * a.ldx + c);
        const float4 ee = pick(et, g.etype[e]);
        acc.x += fmaxf(xj.x + ee.x, 0.f);
        acc.y += fmaxf(xj.y + ee.y, 0.f);
        acc.z += fmaxf(xj.z + ee.z, 0.f);
        acc.w += fmaxf(xj.w + ee.w, 0.f);
    }
    st4(g.out + (long long)i * a.ldo + c,
        make_float4(acc.x + g.ope * xi.x, acc.y + g.ope * xi.y, acc.z + g.ope * xi.z, acc.w + g.ope * xi.w));
}

struct AggBwdArgs { AggBwdGroup g[2]; int ldda, ldx, lddx, ldp, n, H4; const int32_t* node_graph; };

__global__ __launch_bounds__(256) void gine_agg_bwd_kernel(const AggBwdArgs a) {
    const AggBwdGroup& g = a.g[blockIdx.y];
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)a.n * a.H4) return;
    const int j = (int)(t / a.H4), c = (int)(t - (long long)j * a.H4) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.da) {
        const EdgeTab et = load_edge_tab(g.edge_w, g.edge_b, c);
        const float4 xj = ld4(g.x + (long long)j * a.ldx + c);
        const int e1 = g.rowptr_t[j + 1];
        for (int e = g.rowptr_t[j]; e < e1; ++e) {
            const float4 d = ld4(g.da + (long long)g.col_t[e] * a.ldda + c);
            const float4 ee = pick(et, g.etype_t[e]);
            acc.x += (xj.x + ee.x > 0.f) ? d.x : 0.f;
            acc.y += (xj.y + ee.y > 0.f) ? d.y : 0.f;
            acc.z += (xj.z + ee.z > 0.f) ? d.z : 0.f;
            acc.w += (xj.w + ee.w > 0.f) ? d.w : 0.f;
        }
        const float4 dj = ld4(g.da + (long long)j * a.ldda + c);
        acc = make_float4(acc.x + g.ope * dj.x, acc.y + g.ope * dj.y, acc.z + g.ope * dj.z, acc.w + g.ope * dj.w);
    }
    if (g.dpool) acc = add4(acc, ld4(g.dpool + (long long)a.node_graph[j] * a.ldp + c));
    st4(g.dx + (long long)j * a.lddx + c, acc);
}

// Edge-table gradient, stage 1: block (chunk, layer, group) sums, for its node
// range, da[dst] * (x[src] + e(type) > 0) per bond type into partial[..][4][H].
struct EtabArgs { EtabGroup g[2]; int t, ldda, ldx, n, H4, chunks; };

__global__ __launch_bounds__(128) void gine_etab_partial_kernel(const EtabArgs a) {
    const EtabGroup& g = a.g[blockIdx.z];
    const int chunk = blockIdx.x, layer = blockIdx.y;
    const int per = (a.n + a.chunks - 1) / a.chunks;
    const int i0 = chunk * per, i1 = min(a.n, i0 + per);
    const float* da = g.da + (long long)layer * g.da_layer;
    const float* x = g.x + (long long)layer * g.x_layer;
    float* out = g.partial + ((long long)layer * a.chunks + chunk) * 4 * (a.H4 * 4);
    for (int c4 = threadIdx.x; c4 < a.H4; c4 += blockDim.x) {
        const int c = c4 * 4;
        const EdgeTab et = load_edge_tab(g.edge_w, g.edge_b, c);
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        for (int i = i0; i < i1; ++i) {
            const int e1 = g.rowptr[i + 1];
            int e = g.rowptr[i];
            if (e == e1) continue;
            const float4 d = ld4(da + (long long)i * a.ldda + c);
            for (; e < e1; ++e) {
                const int ty = g.etype[e];
                const float4 xs = ld4(x + (long long)g.col[e] * a.ldx + c);
                const float4 ee = pick(et, ty);
                const float4 v = make_float4(xs.x + ee.x > 0.f ? d.x : 0.f, xs.y + ee.y > 0.f ? d.y : 0.f,
                                             xs.z + ee.z > 0.f ? d.z : 0.f, xs.w + ee.w > 0.f ? d.w : 0.f);
                if (ty == 1) s1 = add4(s1, v);
                else if (ty == 2) s2 = add4(s2, v);
                else if (ty == 3) s3 = add4(s3, v);
                else s0 = add4(s0, v);
            }
        }
        const int H = a.H4 * 4;
        st4(out + c, s0); st4(out + H + c, s1); st4(out + 2 * H + c, s2); st4(out + 3 * H + c, s3);
    }
}
// stage 2: d_edge_w[c][ty] = sum of partials in (layer, chunk) order; d_edge_b[c] = sum over ty.
__global__ __launch_bounds__(256) void gine_etab_final_kernel(const EtabArgs a) {
    const EtabGroup& g = a.g[blockIdx.y];
    const int H = a.H4 * 4;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const int np = a.t * a.chunks;
    for (int p = 0; p < np; ++p) {
        const float* src = g.partial + (long long)p * 4 * H;
#pragma unroll
        for (int ty = 0; ty < 4; ++ty) s[ty] += src[ty * H + c];
    }
    st4(g.d_edge_w + (long long)c * 4, make_float4(s[0], s[1], s[2], s[3]));
    g.d_edge_b[c] = ((s[0] + s[1]) + s[2]) + s[3];
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

// Embedding-table gradient: block = one table row; threadIdx.x = column lane,
// threadIdx.y = slice of the row's contribution list.  Slices are combined in
// slice order through LDS, so the sum order is fixed.
constexpr int TG_X = 64, TG_Y = 8;
__global__ __launch_bounds__(TG_X * TG_Y) void table_grad_kernel(float* __restrict__ dtable, int D,
                                                                  const int32_t* __restrict__ ptr1, const int32_t* __restrict__ idx1,
                                                                  const float* __restrict__ src1, int ld1, int off1,
                                                                  const int32_t* __restrict__ ptr2, const int32_t* __restrict__ idx2,
                                                                  const float* __restrict__ src2, int ld2, int off2) {
    __shared__ float part[TG_Y][TG_X];
    const int row = blockIdx.x, tx = threadIdx.x, ty = threadIdx.y;
    for (int c0 = 0; c0 < D; c0 += TG_X) {
        const int c = c0 + tx;
        float s = 0.f;
        if (c < D) {
            {
                const int b = ptr1[row], e = ptr1[row + 1], len = e - b, per = (len + TG_Y - 1) / TG_Y;
                const int k0 = b + ty * per, k1 = min(e, k0 + per);
                for (int k = k0; k < k1; ++k) s += src1[(long long)idx1[k] * ld1 + off1 + c];
            }
            if (ptr2) {
                const int b = ptr2[row], e = ptr2[row + 1], len = e - b, per = (len + TG_Y - 1) / TG_Y;
                const int k0 = b + ty * per, k1 = min(e, k0 + per);
                float s2 = 0.f;
                for (int k = k0; k < k1; ++k) s2 += src2[(long long)idx2[k] * ld2 + off2 + c];
                s += s2;
            }
        }
        part[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && c < D) {
            float v = 0.f;
#pragma unroll
            for (int y = 0; y < TG_Y; ++y) v += part[y][tx];
            dtable[(long long)row * D + c] = v;
        }
        __syncthreads();
    }
}

// pooled[b, c] = sum over the graph's rows, in row order.
__global__ __launch_bounds__(256) void segment_pool_kernel(const float* __restrict__ x, int ldx,
                                                           const int32_t* __restrict__ graph_off,
                                                           float* __restrict__ pooled, int ldp, int W4) {
    const int b = blockIdx.y;
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= W4) return;
    const int c = c4 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int i1 = graph_off[b + 1];
    for (int i = graph_off[b]; i < i1; ++i) acc = add4(acc, ld4(x + (long long)i * ldx + c));
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

// demb[j] = sum_{s: src_s = j} dpin[s, 0:H] + sum_{s: dst_s = j} dpin[s, H:2H]   (rows 0..n-1)
// dz[b]  (+)= sum_{s in graph b} dpin[s, 2H:2H+Z]                                (rows n..n+B-1 of the grid)
__global__ __launch_bounds__(256) void pair_scatter_kernel(const float* __restrict__ dpin, int P,
                                                           const int32_t* __restrict__ sp, const int32_t* __restrict__ si,
                                                           const int32_t* __restrict__ dp, const int32_t* __restrict__ di,
                                                           const int32_t* __restrict__ goff, float* __restrict__ demb,
                                                           int ldde, float* __restrict__ dz, int dz_acc, int n, int B,
                                                           int H4, int Z4) {
    const long long nodes_items = (long long)n * H4;
    const long long total = nodes_items + (long long)B * Z4;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    if (t < nodes_items) {
        const int j = (int)(t / H4), c = (int)(t - (long long)j * H4) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = sp[j]; k < sp[j + 1]; ++k) acc = add4(acc, ld4(dpin + (long long)si[k] * P + c));
        const int H = H4 * 4;
        for (int k = dp[j]; k < dp[j + 1]; ++k) acc = add4(acc, ld4(dpin + (long long)di[k] * P + H + c));
        st4(demb + (long long)j * ldde + c, acc);
    } else {
        const long long u = t - nodes_items;
        const int b = (int)(u / Z4), c = (int)(u - (long long)b * Z4) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int off = 2 * H4 * 4;
        for (int s = goff[b]; s < goff[b + 1]; ++s) acc = add4(acc, ld4(dpin + (long long)s * P + off + c));
        float* d = dz + (long long)b * (Z4 * 4) + c;
        if (dz_acc) acc = add4(acc, ld4(d));
        st4(d, acc);
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
    if (ngroups < 1 || ngroups > 2) return PSVAE_E_SHAPE;
    AggArgs a;
    for (int k = 0; k < ngroups; ++k) {
        a.g[k] = g[k];
        if (!aligned16(g[k].x) || !aligned16(g[k].out) || !aligned16(g[k].edge_w) || !aligned16(g[k].edge_b)) return PSVAE_E_ALIGN;
    }
    if (ngroups == 1) a.g[1] = g[0];
    a.ldx = ldx; a.ldo = ldo; a.n = n; a.H4 = H / 4;
    const long long items = (long long)n * a.H4;
    dim3 grid((unsigned)((items + 255) / 256), ngroups);
    hipLaunchKernelGGL(gine_agg_fwd_kernel, grid, dim3(256), 0, s, a);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int gine_agg_bwd(hipStream_t s, const AggBwdGroup* g, int ngroups, int ldda, int ldx, int lddx, int ldp,
                 const int32_t* node_graph, int n, int H) {
    if (n == 0) return 0;
    if (H % 4 || ldda % 4 || ldx % 4 || lddx % 4 || ldp % 4) return PSVAE_E_ALIGN;
    if (ngroups < 1 || ngroups > 2) return PSVAE_E_SHAPE;
    AggBwdArgs a;
    for (int k = 0; k < ngroups; ++k) {
        a.g[k] = g[k];
        if (!aligned16(g[k].dx) || (g[k].da && (!aligned16(g[k].da) || !aligned16(g[k].x))) ||
            (g[k].dpool && !aligned16(g[k].dpool))) return PSVAE_E_ALIGN;
        if (g[k].dpool && !node_graph) return PSVAE_E_NULL;
    }
    if (ngroups == 1) a.g[1] = g[0];
    a.ldda = ldda; a.ldx = ldx; a.lddx = lddx; a.ldp = ldp; a.n = n; a.H4 = H / 4; a.node_graph = node_graph;
    const long long items = (long long)n * a.H4;
    dim3 grid((unsigned)((items + 255) / 256), ngroups);
    hipLaunchKernelGGL(gine_agg_bwd_kernel, grid, dim3(256), 0, s, a);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int gine_etab_grad(hipStream_t s, const EtabGroup* g, int ngroups, int t, int ldda, int ldx, int n, int H, int chunks) {
    if (H % 4 || ldda % 4 || ldx % 4) return PSVAE_E_ALIGN;
    if (ngroups < 1 || ngroups > 2 || t < 1 || chunks < 1) return PSVAE_E_SHAPE;
    EtabArgs a;
    for (int k = 0; k < ngroups; ++k) a.g[k] = g[k];
    if (ngroups == 1) a.g[1] = g[0];
    a.t = t; a.ldda = ldda; a.ldx = ldx; a.n = n; a.H4 = H / 4; a.chunks = chunks;
    hipLaunchKernelGGL(gine_etab_partial_kernel, dim3(chunks, t, ngroups), dim3(128), 0, s, a);
    PSVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(gine_etab_final_kernel, dim3((H + 255) / 256, ngroups), dim3(256), 0, s, a);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int embed3_gather(hipStream_t s, const int32_t* a, const int32_t* p, const int32_t* q, const float* ta,
                  const float* tp, const float* tq, int da, int dp, int dq, float* out, int n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(embed3_gather_kernel, dim3(grid_for((long long)n * (da + dp + dq), 256)), dim3(256), 0, s,
                       a, p, q, ta, tp, tq, da, dp, dq, out, n);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int rows_gather(hipStream_t s, const int32_t* idx, int /*idx_stride_outer*/, int inner, int idx_ld, const float* table,
                int width, float* out, int rows) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(rows_gather_kernel, dim3(grid_for((long long)rows * width, 256)), dim3(256), 0, s,
                       idx, inner, idx_ld, table, width, out, rows);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int table_grad(hipStream_t s, float* dtable, int rows, int D, const int32_t* ptr1, const int32_t* idx1,
               const float* src1, int ld1, int off1, const int32_t* ptr2, const int32_t* idx2, const float* src2,
               int ld2, int off2) {
    if (rows == 0 || D == 0) return 0;
    hipLaunchKernelGGL(table_grad_kernel, dim3(rows), dim3(TG_X, TG_Y), 0, s, dtable, D, ptr1, idx1, src1, ld1, off1,
                       ptr2, idx2, src2, ld2, off2);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int segment_pool_fwd(hipStream_t s, const float* x, int ldx, const int32_t* graph_off, float* pooled, int ldp,
                     int B, int width) {
    if (B == 0) return 0;
    if (width % 4 || ldx % 4 || ldp % 4 || !aligned16(x) || !aligned16(pooled)) return PSVAE_E_ALIGN;
    const int W4 = width / 4;
    hipLaunchKernelGGL(segment_pool_kernel, dim3((W4 + 255) / 256, B), dim3(256), 0, s, x, ldx, graph_off, pooled, ldp, W4);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int pair_gather(hipStream_t s, const float* emb, int lde, const float* z, const int32_t* src, const int32_t* dst,
                const int32_t* graph, float* out, int S, int H, int Z) {
    if (S == 0) return 0;
    if (H % 4 || Z % 4 || lde % 4 || !aligned16(emb) || !aligned16(z) || !aligned16(out)) return PSVAE_E_ALIGN;
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
    hipLaunchKernelGGL(pair_scatter_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, dpin, P, by_src_ptr,
                       by_src_idx, by_dst_ptr, by_dst_idx, pair_graph_off, demb, ldde, dz, dz_accumulate, n, B, H / 4, Z / 4);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int fill_zero(hipStream_t s, float* p, int64_t count) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(fill_zero_kernel, dim3(grid_for(count, 256)), dim3(256), 0, s, p, (long long)count);
    PSVAE_LAUNCH_CHECK();
    return 0;
}
