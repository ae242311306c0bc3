// fp32 GEMM family on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact
// fp32 products, fp32 accumulate).  One kernel template covers the three
// products a Linear layer needs:
//   forward        y  = x  w^T      A row-major (reduction contiguous), B row-major
//   input grad     dx = dy w        A row-major,                        B "transposed" (output contiguous)
//   weight grad    dw = dy^T x      A and B transposed, reduction over the rows,
//                                   split over grid.z into slabs that a second
//                                   kernel sums in a fixed order (bitwise reproducible)
// Workgroup = 256 threads = 4 waves, 64x64 output tile, each wave one 32x32
// accumulator; K-step 16, operands staged through LDS as [k][row] so that the
// MFMA operand read (lane l: row l&31, k l>>5) is one conflict-free ds_read_b32.
#include "psvae_internal.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDS_LD = 64 + 4;
typedef float floatx16 __attribute__((ext_vector_type(16)));

struct GemmP {
    const float* A; const float* B; float* C;
    const float* bias; const float* mask; float* c_extra;
    int M, N, R;
    int lda, ldb, ldc, ldmask;
    long long seg_a, seg_b, seg_c, seg_bias, seg_mask;
    int splits, rchunk;
    int relu, accumulate, ones_col;
    int a_vec, b_vec;
    int to_slab;            // 1: raw partial sums to C + z*slab_stride (ldc = N)
    long long slab_stride;
};

// Load this thread's 4 staged elements of one operand tile.
//  !T: X[o*ld + r]: thread -> row o = o0 + tid/4, r = r0 + 4*(tid%4) .. +3
//   T: X[r*ld + o]: thread -> r = r0 + tid/16, o = o0 + 4*(tid%16) .. +3
template <bool T>
__device__ __forceinline__ float4 load_stage(const float* __restrict__ X, int ld, int o0, int o_lim, int r0, int r_lim,
                                             int tid, int vec, int ones_col) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!T) {
        const int o = o0 + (tid >> 2), r = r0 + ((tid & 3) << 2);
        if (o < o_lim) {
            const float* p = X + (long long)o * ld + r;
            if (vec && r + 3 < r_lim) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                if (r < r_lim) v.x = p[0];
                if (r + 1 < r_lim) v.y = p[1];
                if (r + 2 < r_lim) v.z = p[2];
                if (r + 3 < r_lim) v.w = p[3];
            }
        }
    } else {
        const int r = r0 + (tid >> 4), o = o0 + ((tid & 15) << 2);
        if (r < r_lim) {
            const int real_lim = ones_col >= 0 ? ones_col : o_lim;  // columns that exist in memory
            const float* p = X + (long long)r * ld + o;
            if (vec && o + 3 < real_lim) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                if (o < real_lim) v.x = p[0];
                if (o + 1 < real_lim) v.y = p[1];
                if (o + 2 < real_lim) v.z = p[2];
                if (o + 3 < real_lim) v.w = p[3];
            }
            if (ones_col >= 0) {
                if (o == ones_col) v.x = 1.f;
                if (o + 1 == ones_col) v.y = 1.f;
                if (o + 2 == ones_col) v.z = 1.f;
                if (o + 3 == ones_col) v.w = 1.f;
            }
        }
    }
    return v;
}

template <bool T>
__device__ __forceinline__ void store_stage(float (*s)[LDS_LD], float4 v, int tid) {
    if (!T) {
        const int row = tid >> 2, k = (tid & 3) << 2;
        s[k][row] = v.x; s[k + 1][row] = v.y; s[k + 2][row] = v.z; s[k + 3][row] = v.w;
    } else {
        *reinterpret_cast<float4*>(&s[tid >> 4][(tid & 15) << 2]) = v;
    }
}

template <bool AT, bool BT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmP p) {
    __shared__ __attribute__((aligned(16))) float sA[2][BK][LDS_LD];
    __shared__ __attribute__((aligned(16))) float sB[2][BK][LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z, seg = z / p.splits, sp = z - seg * p.splits;
    const float* A = p.A + seg * p.seg_a;
    const float* B = p.B + seg * p.seg_b;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int rbeg = sp * p.rchunk;
    const int rend = min(p.R, rbeg + p.rchunk);
    const int nk = (rend - rbeg + BK - 1) / BK;

    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    if (nk > 0) {
        float4 ra = load_stage<AT>(A, p.lda, m0, p.M, rbeg, rend, tid, p.a_vec, -1);
        float4 rb = load_stage<BT>(B, p.ldb, n0, p.N, rbeg, rend, tid, p.b_vec, p.ones_col);
        store_stage<AT>(sA[0], ra, tid);
        store_stage<BT>(sB[0], rb, tid);
        __syncthreads();
        const int arow = wm * 32 + (lane & 31), brow = wn * 32 + (lane & 31), kh = lane >> 5;
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) {
                const int r0 = rbeg + (kt + 1) * BK;
                ra = load_stage<AT>(A, p.lda, m0, p.M, r0, rend, tid, p.a_vec, -1);
                rb = load_stage<BT>(B, p.ldb, n0, p.N, r0, rend, tid, p.b_vec, p.ones_col);
            }
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const float a = sA[buf][kk * 2 + kh][arow];
                const float b = sB[buf][kk * 2 + kh][brow];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            if (kt + 1 < nk) {
                store_stage<AT>(sA[buf ^ 1], ra, tid);
                store_stage<BT>(sB[buf ^ 1], rb, tid);
            }
            __syncthreads();
        }
    }

    // epilogue: lane holds column j, rows (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int j = n0 + wn * 32 + (lane & 31);
    if (j >= p.N) return;
    if (p.to_slab) {
        float* C = p.C + (long long)z * p.slab_stride;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (i < p.M) C[(long long)i * p.N + j] = acc[reg];
        }
        return;
    }
    float* C = p.C + seg * p.seg_c;
    const float bj = p.bias ? p.bias[seg * p.seg_bias + j] : 0.f;
    const float* mask = p.mask ? p.mask + seg * p.seg_mask : nullptr;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int i = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (i >= p.M) continue;
        float v = acc[reg] + bj;
        if (p.relu) v = fmaxf(v, 0.f);
        if (j == p.ones_col) {
            if (p.c_extra) p.c_extra[i] = p.accumulate ? p.c_extra[i] + v : v;
            continue;
        }
        if (mask && !(mask[(long long)i * p.ldmask + j] > 0.f)) v = 0.f;
        float* dst = C + (long long)i * p.ldc + j;
        *dst = p.accumulate ? *dst + v : v;
    }
}

// Sum `nz` slabs (fixed order z = 0..nz-1) and apply the epilogue.
struct ReduceP {
    const float* slabs; long long slab_stride; int nz;
    float* C; int ldc; long long seg_c;
    const float* bias; long long seg_bias;
    const float* mask; int ldmask; long long seg_mask;
    float* c_extra;
    int M, N, relu, accumulate, ones_col;
};
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const ReduceP p) {
    const long long total = (long long)p.M * p.N;
    const int seg = blockIdx.y;
    const float* s = p.slabs + (long long)seg * p.nz * p.slab_stride;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e / p.N), j = (int)(e - (long long)i * p.N);
        float v = 0.f;
        for (int z = 0; z < p.nz; ++z) v += s[(long long)z * p.slab_stride + e];
        if (p.bias) v += p.bias[seg * p.seg_bias + j];
        if (p.relu) v = fmaxf(v, 0.f);
        if (j == p.ones_col) {
            if (p.c_extra) p.c_extra[i] = p.accumulate ? p.c_extra[i] + v : v;
            continue;
        }
        if (p.mask && !(p.mask[seg * p.seg_mask + (long long)i * p.ldmask + j] > 0.f)) v = 0.f;
        float* dst = p.C + seg * p.seg_c + (long long)i * p.ldc + j;
        *dst = p.accumulate ? *dst + v : v;
    }
}

struct Plan { int splits, rchunk; };
Plan plan_splits(int M, int N, int R, int nseg) {
    const long long tiles = (long long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * nseg;
    Plan pl{1, (int)round_up(R > 0 ? R : 1, BK)};
    if (tiles >= 160 || R <= 128) return pl;
    int want = (int)((256 + tiles - 1) / tiles);
    int maxs = (R + 63) / 64;
    if (want > maxs) want = maxs;
    if (want <= 1) return pl;
    pl.rchunk = (int)round_up((R + want - 1) / want, BK);
    pl.splits = (R + pl.rchunk - 1) / pl.rchunk;
    return pl;
}

}  // namespace

size_t gemm_scratch_floats(int M, int N, int R, int nseg, bool sum_segments) {
    Plan pl = plan_splits(M, N, R, nseg);
    const long long nz = (long long)nseg * pl.splits;
    if (nz == 1 || (!sum_segments && pl.splits == 1)) return 0;
    return (size_t)(nz * (long long)M * N);
}

int gemm_f32(hipStream_t s, const GemmDesc& d) {
    if (d.M < 0 || d.N < 0 || d.R < 0 || d.nseg < 1) return PSVAE_E_SHAPE;
    if (d.M == 0 || d.N == 0) return 0;
    if (!d.A || !d.B || !d.C) return PSVAE_E_NULL;
    Plan pl = plan_splits(d.M, d.N, d.R, d.nseg);
    long long nz_total = (long long)d.nseg * pl.splits;
    bool slab = d.sum_segments ? (nz_total > 1) : (pl.splits > 1);
    if (slab) {
        const size_t need = (size_t)(nz_total * (long long)d.M * d.N);
        if (!d.scratch || d.scratch_floats < need) {
            if (d.sum_segments && d.nseg > 1) return PSVAE_E_WORKSPACE;
            pl.splits = 1; pl.rchunk = (int)round_up(d.R > 0 ? d.R : 1, BK);
            nz_total = d.nseg; slab = false;
        }
    }
    GemmP p;
    p.A = d.A; p.B = d.B; p.C = slab ? d.scratch : d.C;
    p.bias = d.bias; p.mask = d.mask; p.c_extra = d.c_extra;
    p.M = d.M; p.N = d.N; p.R = d.R;
    p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc; p.ldmask = d.ldmask;
    p.seg_a = d.seg_a; p.seg_b = d.seg_b; p.seg_c = d.seg_c; p.seg_bias = d.seg_bias; p.seg_mask = d.seg_mask;
    p.splits = pl.splits; p.rchunk = pl.rchunk;
    p.relu = d.relu; p.accumulate = d.accumulate; p.ones_col = d.ones_col;
    p.a_vec = aligned16(d.A) && d.lda % 4 == 0 && d.seg_a % 4 == 0;
    p.b_vec = aligned16(d.B) && d.ldb % 4 == 0 && d.seg_b % 4 == 0;
    p.to_slab = slab ? 1 : 0;
    p.slab_stride = (long long)d.M * d.N;
    dim3 grid((d.M + BM - 1) / BM, (d.N + BN - 1) / BN, (unsigned)nz_total);
    if (grid.y > 65535 || grid.z > 65535) return PSVAE_E_SHAPE;
    if (!d.a_t && !d.b_t) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, s, p);
    else if (!d.a_t && d.b_t) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, s, p);
    else if (d.a_t && d.b_t) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, s, p);
    PSVAE_LAUNCH_CHECK();
    if (slab) {
        ReduceP r;
        r.slabs = d.scratch; r.slab_stride = p.slab_stride;
        r.nz = d.sum_segments ? (int)nz_total : pl.splits;
        r.C = d.C; r.ldc = d.ldc; r.seg_c = d.seg_c;
        r.bias = d.bias; r.seg_bias = d.seg_bias;
        r.mask = d.mask; r.ldmask = d.ldmask; r.seg_mask = d.seg_mask;
        r.c_extra = d.c_extra;
        r.M = d.M; r.N = d.N; r.relu = d.relu; r.accumulate = d.accumulate; r.ones_col = d.ones_col;
        const long long total = (long long)d.M * d.N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        dim3 rgrid(blocks, d.sum_segments ? 1 : d.nseg);
        hipLaunchKernelGGL(gemm_reduce_kernel, rgrid, dim3(256), 0, s, r);
        PSVAE_LAUNCH_CHECK();
    }
    return 0;
}

int linear_fwd(hipStream_t s, const float* x, int ldx, const float* w, const float* b, float* y, int ldy,
               int M, int N, int K, int relu, float* scratch, size_t scratch_floats) {
    GemmDesc d;
    d.A = x; d.lda = ldx; d.B = w; d.ldb = K; d.C = y; d.ldc = ldy;
    d.M = M; d.N = N; d.R = K; d.bias = b; d.relu = relu;
    d.scratch = scratch; d.scratch_floats = scratch_floats;
    return gemm_f32(s, d);
}

int linear_bwd_x(hipStream_t s, const float* dy, int lddy, const float* w, float* dx, int lddx,
                 int M, int N, int K, const float* mask, int ldmask, int accumulate) {
    GemmDesc d;
    d.A = dy; d.lda = lddy; d.B = w; d.ldb = K; d.b_t = true; d.C = dx; d.ldc = lddx;
    d.M = M; d.N = K; d.R = N; d.mask = mask; d.ldmask = ldmask; d.accumulate = accumulate;
    return gemm_f32(s, d);
}

int linear_bwd_w(hipStream_t s, const float* dy, int lddy, int64_t seg_dy, const float* x, int ldx, int64_t seg_x,
                 int nseg, float* dw, float* db, int M, int N, int K, float* scratch, size_t scratch_floats) {
    GemmDesc d;
    d.A = dy; d.lda = lddy; d.a_t = true; d.B = x; d.ldb = ldx; d.b_t = true; d.C = dw; d.ldc = K;
    d.M = N; d.N = db ? K + 1 : K; d.R = M;
    d.ones_col = db ? K : -1; d.c_extra = db;
    d.nseg = nseg; d.seg_a = seg_dy; d.seg_b = seg_x; d.sum_segments = true;
    d.scratch = scratch; d.scratch_floats = scratch_floats;
    return gemm_f32(s, d);
}
