// fp32 GEMM family on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact
// fp32 products, fp32 accumulate).  One kernel template covers the three
// products a Linear layer needs:
//   forward        y  = x  w^T      A row-major (reduction contiguous), B row-major
//   input grad     dx = dy w        A row-major,                        B "transposed" (output contiguous)
//   weight grad    dw = dy^T x      A and B transposed, reduction over the rows,
//                                   split over grid.z into slabs that a second
//                                   kernel sums in a fixed order (bitwise reproducible)
// Workgroup = 256 threads = 4 waves, 64x64 output tile, each wave one 32x32
// accumulator; K-step 32.  Operands are staged through LDS as [k][row] so that
// the MFMA operand read (lane l: row l&31, k l>>5) is one conflict-free
// ds_read_b32.  The problems here are small (a few thousand rows), so a
// workgroup's time is set by load latency, not bandwidth: tiles are fetched two
// K-steps ahead into two named register stages (global -> VGPR -> LDS), which
// keeps ~2 x 16 MFMAs of work between a load's issue and its first use.
#include "psvae_internal.h"
#include <stdlib.h>
#include <mutex>

namespace {

constexpr int BM = 64, BN = 64, BK = 32;
// LDS row pitch of a [k][row] operand image.  A tile that arrives with the reduction index contiguous is
// written element by element (lane -> k = 4*(q&7).., row = q>>3): pitch 65 spreads those writes over all
// 32 banks.  A tile that arrives with the output index contiguous is written as float4 rows: pitch 68 keeps
// them 16-byte aligned.  MFMA operand reads (32 consecutive rows of one k) are conflict-free for any pitch.
template <bool T> struct Pitch { static constexpr int value = T ? 68 : 65; };
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
    int to_slab;            // 1: raw partial sums to C + z*slab_stride, rows ld_slab floats apart
    int ld_slab;            // N, or N rounded up to 16 floats (160-tile kernel: 64-byte aligned row segments)
    long long slab_stride;
};

// One operand of the product as this workgroup sees it: X[o][r] (T = false, reduction index contiguous) or
// X[r][o] (T = true, output index contiguous); the tile covers outputs o0 .. o0+63.
struct TileGeo { const float* X; int ld, o0, o_lim, r_lim, ones_col; };

// This thread's 8 elements (two float4 "quads") of one 64 x 32 operand tile, as loaded.
struct Stage { float4 v0, v1; };

//  !T: slot q -> row o = o0 + q/8, r = r0 + 4*(q%8) .. +3    (vector along r)
//   T: slot q -> r = r0 + q/16,    o = o0 + 4*(q%16) .. +3   (vector along o)
// Validity (bits 0-3) and "is the ones column" (bits 8-11) of the 4 elements of quad q.
template <bool T>
__device__ __forceinline__ unsigned quad_flags(const TileGeo& g, int r0, int q) {
    const int o = T ? g.o0 + ((q & 15) << 2) : g.o0 + (q >> 3);
    const int r = T ? r0 + (q >> 4) : r0 + ((q & 7) << 2);
    const int first = T ? o : r;                                              // index along the vector
    const int lim = T ? (g.ones_col >= 0 ? g.ones_col : g.o_lim) : g.r_lim;   // valid extent along the vector
    const bool line_ok = T ? (r < g.r_lim) : (o < g.o_lim);
    const int left = line_ok ? lim - first : 0;                               // valid elements from `first` on
    unsigned flags = left >= 4 ? 0xFu : (left <= 0 ? 0u : (1u << left) - 1u);
    if (T && g.ones_col >= 0 && line_ok) {
        const int d = g.ones_col - o;
        if (d >= 0 && d < 4) flags |= 0x100u << d;
    }
    return flags;
}

// One float4 of an operand tile, without branches (a branch around a load makes the compiler wait at the
// join for loads it has just issued, which would serialise the prefetch): the address is clamped into the
// matrix and the load is unconditional; what lies outside the valid range is zeroed when the stage is
// written to LDS.  VEC: base 16-byte aligned and ld % 4 == 0, so a float4 that starts inside a row stays
// inside that row's ld floats even when it overhangs the valid width.
template <bool T, bool VEC>
__device__ __forceinline__ float4 load_quad(const TileGeo& g, int r0, int q) {
    const int o = T ? g.o0 + ((q & 15) << 2) : g.o0 + (q >> 3);
    const int r = T ? r0 + (q >> 4) : r0 + ((q & 7) << 2);
    const int first = T ? o : r;
    const int lim = T ? (g.ones_col >= 0 ? g.ones_col : g.o_lim) : g.r_lim;
    const bool any = (T ? (r < g.r_lim) : (o < g.o_lim)) && first < lim;
    const float* p = g.X + (any ? (long long)(T ? r : o) * g.ld + first : 0);
    float4 v;
    if (VEC) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        // unaligned operand: four scalar loads, each clamped to the first element when out of range
        const bool ok1 = any && first + 1 < lim, ok2 = any && first + 2 < lim, ok3 = any && first + 3 < lim;
        v.x = p[0]; v.y = p[ok1 ? 1 : 0]; v.z = p[ok2 ? 2 : 0]; v.w = p[ok3 ? 3 : 0];
    }
    return v;
}

template <bool T, bool VEC>
__device__ __forceinline__ Stage load_stage(const TileGeo& g, int r0, int tid) {
    Stage s;
    s.v0 = load_quad<T, VEC>(g, r0, tid);
    s.v1 = load_quad<T, VEC>(g, r0, tid + 256);
    return s;
}

// The same two loads with the address arithmetic taken off the vector ALU (aligned operands only).  What a thread
// adds to the tile's origin - (row, column) of its two quads - never changes, and what changes from K-step to K-step
// is uniform: the loads become "scalar base + 32-bit lane offset", the base advanced by scalar instructions.  A quad
// outside the matrix points at the tile's origin (offset 0).  Only the one tile that overhangs the reduction range
// needs other offsets (quads past the end redirected as well); they are prepared once and chosen by a uniform
// select.  A tile that lies entirely past the range (fetched by the fixed-depth prefetch, never used) is fetched as
// the range's LAST tile once more - the only start index for which these lane offsets are known to stay inside the
// matrix when the range is shorter than a tile.  Inside the range the bytes loaded are exactly those
// load_stage<T, true> loads.
struct FastGeo { const char* tile0; long long step; unsigned off0, off1, offp0, offp1; int r_last, r_part; };

template <bool T>
__device__ __forceinline__ FastGeo fast_geo(const TileGeo& g, int rbeg, int tid) {
    FastGeo f;
    const int lim_o = T ? (g.ones_col >= 0 ? g.ones_col : g.o_lim) : g.o_lim;
    // (a tile that holds nothing but the ones column has no column of its own in the matrix: it reads column 0)
    f.tile0 = reinterpret_cast<const char*>(T ? g.X + (g.o0 < lim_o ? g.o0 : 0) : g.X + (long long)g.o0 * g.ld);
    f.step = T ? (long long)g.ld * 4 : 4;
    const int span = g.r_lim - rbeg, nk = (span + BK - 1) / BK;
    f.r_last = rbeg + (nk - 1) * BK;
    f.r_part = (span % BK) ? f.r_last : -1;
    unsigned off[2], offp[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = tid + 256 * h;
        const int dro = T ? ((q & 15) << 2) : (q >> 3);       // output index inside the tile
        const int drr = T ? (q >> 4) : ((q & 7) << 2);        // reduction index inside the tile
        const bool ok_o = g.o0 + dro < lim_o;
        const unsigned o = (unsigned)(T ? drr * g.ld + dro : dro * g.ld + drr) * 4u;
        off[h] = ok_o ? o : 0u;
        offp[h] = (ok_o && f.r_part >= 0 && f.r_part + drr < g.r_lim) ? o : 0u;
    }
    f.off0 = off[0]; f.off1 = off[1]; f.offp0 = offp[0]; f.offp1 = offp[1];
    return f;
}

__device__ __forceinline__ Stage fast_stage(const FastGeo& f, int r0) {
    const int r = min(r0, f.r_last);
    const char* sb = f.tile0 + (long long)r * f.step;
    const bool part = r == f.r_part;
    const unsigned o0 = part ? f.offp0 : f.off0, o1 = part ? f.offp1 : f.off1;
    Stage s;
    s.v0 = *reinterpret_cast<const float4*>(sb + o0);
    s.v1 = *reinterpret_cast<const float4*>(sb + o1);
    return s;
}

__device__ __forceinline__ float4 apply_flags(float4 v, unsigned flags) {
    v.x = (flags & 1u) ? v.x : 0.f; v.y = (flags & 2u) ? v.y : 0.f;
    v.z = (flags & 4u) ? v.z : 0.f; v.w = (flags & 8u) ? v.w : 0.f;
    v.x = (flags & 0x100u) ? 1.f : v.x; v.y = (flags & 0x200u) ? 1.f : v.y;
    v.z = (flags & 0x400u) ? 1.f : v.z; v.w = (flags & 0x800u) ? 1.f : v.w;
    return v;
}

template <bool T>
__device__ __forceinline__ void store_quad(float (*s)[Pitch<T>::value], float4 v, int q) {
    if (!T) {
        const int row = q >> 3, k = (q & 7) << 2;
        s[k][row] = v.x; s[k + 1][row] = v.y; s[k + 2][row] = v.z; s[k + 3][row] = v.w;
    } else {
        *reinterpret_cast<float4*>(&s[q >> 4][(q & 15) << 2]) = v;
    }
}
// Write a loaded stage into its LDS image.  Only the REDUCTION range needs care: an element past its end would
// add to every output, so a tile that overhangs it has those elements zeroed.  A tile that overhangs the
// OUTPUT range is written as loaded: row o of an operand only ever meets output row (column) o, rows past the
// matrix produce outputs the epilogue never writes, and their loads come from clamped, valid addresses.  The
// one output index with a value of its own is the ones column (bias-gradient trick).  The branch is uniform
// per workgroup and K-step.
// The quad that holds the ones column gets its 1 by four selects (ones_col = -1 matches no quad), so the tile
// with that column needs the flagged path no more often than any other tile.
template <bool T>
__device__ __forceinline__ float4 patch_ones(float4 v, const TileGeo& g, int q) {
    if (T) {
        const int d = g.ones_col - (g.o0 + ((q & 15) << 2));
        v.x = d == 0 ? 1.f : v.x; v.y = d == 1 ? 1.f : v.y; v.z = d == 2 ? 1.f : v.z; v.w = d == 3 ? 1.f : v.w;
    }
    return v;
}
template <bool T>
__device__ __forceinline__ void store_stage(float (*s)[Pitch<T>::value], const Stage& st, const TileGeo& g, int r0, int tid) {
    if (r0 + BK <= g.r_lim) {
        store_quad<T>(s, patch_ones<T>(st.v0, g, tid), tid);
        store_quad<T>(s, patch_ones<T>(st.v1, g, tid + 256), tid + 256);
    } else {
        store_quad<T>(s, apply_flags(st.v0, quad_flags<T>(g, r0, tid)), tid);
        store_quad<T>(s, apply_flags(st.v1, quad_flags<T>(g, r0, tid + 256)), tid + 256);
    }
}

// One K-step of a wave: 16 MFMAs over the [k][row] images.  The operand reads run one group of four MFMAs
// ahead of the matrix pipe (two named register groups), so that the LDS latency of a group is covered by the
// 256 cycles the previous group's MFMAs take instead of being paid between them.
template <int LA, int LB>
__device__ __forceinline__ floatx16 tile_mfma(const float (*sa)[LA], const float (*sb)[LB], int arow, int brow,
                                              int kh, floatx16 acc) {
    float a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a0[u] = sa[u * 2 + kh][arow]; b0[u] = sb[u * 2 + kh][brow]; }
#pragma unroll
    for (int g = 0; g < BK / 8; g += 2) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { a1[u] = sa[(g + 1) * 8 + u * 2 + kh][arow]; b1[u] = sb[(g + 1) * 8 + u * 2 + kh][brow]; }
        __builtin_amdgcn_sched_barrier(0);      // (the scheduler would otherwise sink every read to just before its MFMA)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 < BK / 8) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0[u] = sa[(g + 2) * 8 + u * 2 + kh][arow]; b0[u] = sb[(g + 2) * 8 + u * 2 + kh][brow]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// ONES: the operand may hold the ones column (the B operand of a weight-gradient product); off the edge path it
// is patched in by four selects.
// The first half of a K-step only (k = 0 .. 15): for a last K-step whose valid reduction range ends there.
template <int LA, int LB>
__device__ __forceinline__ floatx16 tile_mfma_half(const float (*sa)[LA], const float (*sb)[LB], int arow, int brow,
                                                   int kh, floatx16 acc) {
    float a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a0[u] = sa[u * 2 + kh][arow]; b0[u] = sb[u * 2 + kh][brow]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { a1[u] = sa[8 + u * 2 + kh][arow]; b1[u] = sb[8 + u * 2 + kh][brow]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc, 0, 0, 0);
    return acc;
}

template <bool T, bool EDGE, bool ONES>
__device__ __forceinline__ void store_half(float (*s)[Pitch<T>::value], float4 v, const TileGeo& g, int r0, int q) {
    store_quad<T>(s, EDGE ? apply_flags(v, quad_flags<T>(g, r0, q)) : (ONES ? patch_ones<T>(v, g, q) : v), q);
}

// One K-step of the main loop as ONE instruction stream per wave.  With a few thousand rows there is at most one
// wave per SIMD, so nothing but the wave's own instruction order can keep the matrix pipe busy: a wave issues in
// order and every MFMA waits for the previous one (same accumulator), which leaves ~60 idle issue cycles after
// each of the 16 MFMAs.  Everything else the K-step has to do is dealt into those gaps - the global loads of the
// tile four steps ahead, the LDS reads of the next group of four MFMAs, the LDS writes of the next tile (other
// buffer) - instead of running before and after the MFMA chain.  The scheduling barriers pin that order.
// EDGE: the tile being stored overhangs the reduction range or holds the ones column (see store_stage).
#define PSVAE_SB() __builtin_amdgcn_sched_barrier(0)
#define PSVAE_MFMA(a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)
template <bool AT, bool BT, bool VEC, bool EDGE>
__device__ __forceinline__ void kstep(const float (*cA)[Pitch<AT>::value], const float (*cB)[Pitch<BT>::value],
                                      float (*nA)[Pitch<AT>::value], float (*nB)[Pitch<BT>::value],
                                      const TileGeo& ga, const TileGeo& gb, const FastGeo& fga, const FastGeo& fgb,
                                      Stage& la, Stage& lb, int r_load,
                                      const Stage& sa, const Stage& sb, int r_store, int arow, int brow, int kh, int tid,
                                      floatx16& acc) {
    float a0[4], b0[4], a1[4], b1[4];
#define PSVAE_READ(A_, B_, G_)                                                                      \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                 \
        A_[u] = cA[(G_) * 8 + u * 2 + kh][arow]; B_[u] = cB[(G_) * 8 + u * 2 + kh][brow];           \
    }
    PSVAE_READ(a0, b0, 0)
    PSVAE_SB();
    PSVAE_MFMA(a0[0], b0[0]);
    la = VEC ? fast_stage(fga, r_load) : load_stage<AT, VEC>(ga, r_load, tid);
    PSVAE_SB();
    PSVAE_MFMA(a0[1], b0[1]);
    lb = VEC ? fast_stage(fgb, r_load) : load_stage<BT, VEC>(gb, r_load, tid);
    PSVAE_SB();
    PSVAE_MFMA(a0[2], b0[2]);
    PSVAE_READ(a1, b1, 1)
    PSVAE_SB();
    PSVAE_MFMA(a0[3], b0[3]);
    store_half<AT, EDGE, false>(nA, sa.v0, ga, r_store, tid);
    PSVAE_SB();
    PSVAE_MFMA(a1[0], b1[0]);
    PSVAE_SB();
    PSVAE_MFMA(a1[1], b1[1]);
    store_half<AT, EDGE, false>(nA, sa.v1, ga, r_store, tid + 256);
    PSVAE_SB();
    PSVAE_MFMA(a1[2], b1[2]);
    PSVAE_READ(a0, b0, 2)
    PSVAE_SB();
    PSVAE_MFMA(a1[3], b1[3]);
    PSVAE_SB();
    PSVAE_MFMA(a0[0], b0[0]);
    store_half<BT, EDGE, true>(nB, sb.v0, gb, r_store, tid);
    PSVAE_SB();
    PSVAE_MFMA(a0[1], b0[1]);
    PSVAE_SB();
    PSVAE_MFMA(a0[2], b0[2]);
    PSVAE_READ(a1, b1, 3)
    PSVAE_SB();
    PSVAE_MFMA(a0[3], b0[3]);
    store_half<BT, EDGE, true>(nB, sb.v1, gb, r_store, tid + 256);
    PSVAE_SB();
    PSVAE_MFMA(a1[0], b1[0]);
    PSVAE_MFMA(a1[1], b1[1]);
    PSVAE_MFMA(a1[2], b1[2]);
    PSVAE_MFMA(a1[3], b1[3]);
    PSVAE_SB();
    __syncthreads();
#undef PSVAE_READ
}

// PIPE selects the main loop: the hand-interleaved K-step (one instruction stream that keeps the matrix pipe fed
// from a single wave per SIMD; ~200 VGPRs, two waves per SIMD at most) for launches of a few workgroups per CU,
// the plain loop (~140 VGPRs, relies on several co-resident workgroups to overlap its phases) for large ones.
template <bool AT, bool BT, bool VEC, bool PIPE>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmP p) {
    __shared__ __attribute__((aligned(16))) float sA[2][BK][Pitch<AT>::value];
    __shared__ __attribute__((aligned(16))) float sB[2][BK][Pitch<BT>::value];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroups are handed to the 8 XCDs round-robin in dispatch order, and each XCD has an L2 of its own.
    // Tiles that read the same operand panel (one row of tiles; all tiles of one reduction slab) are therefore
    // renumbered so that they run on ONE XCD, next to each other in time: XCD k works through the k-th
    // contiguous eighth of the tile list (z slowest, then tile row, tile column fastest), and a panel is
    // fetched into one L2 once instead of into five of them.
    const unsigned gxy = gridDim.x * gridDim.y, nb = gxy * gridDim.z;
    const unsigned bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = bid & 7u, per = nb >> 3, rem = nb & 7u;
    const unsigned tile = xcd * per + min(xcd, rem) + (bid >> 3);
    const int z = tile / gxy, txy = tile - z * gxy;
    const int tx = txy / gridDim.y, ty = txy - tx * gridDim.y;
    const int seg = z / p.splits, sp = z - seg * p.splits;
    const float* A = p.A + seg * p.seg_a;
    const float* B = p.B + seg * p.seg_b;
    const int m0 = tx * BM, n0 = ty * BN;
    const int rbeg = sp * p.rchunk;
    const int rend = min(p.R, rbeg + p.rchunk);
    const int nk = (rend - rbeg + BK - 1) / BK;

    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    if (nk > 0) {
        const int arow = wm * 32 + (lane & 31), brow = wn * 32 + (lane & 31), kh = lane >> 5;
        const TileGeo ga{A, p.lda, m0, p.M, rend, -1}, gb{B, p.ldb, n0, p.N, rend, p.ones_col};
        const FastGeo fga = VEC ? fast_geo<AT>(ga, rbeg, tid) : FastGeo{}, fgb = VEC ? fast_geo<BT>(gb, rbeg, tid) : FastGeo{};
        // Four register stages, named so that every access is static.  Invariant at the top of a quad of K-steps
        // (kt a multiple of 4): LDS buffer 0 holds tile kt, stages 0/1/2 hold tiles kt+1/kt+2/kt+3 (loaded or in
        // flight), stage 3 is free.  Every K-step fetches the tile four steps ahead into the stage that has just
        // been freed, so a load has three K-steps of MFMA work (~1.3 us) to arrive before it is written to LDS:
        // with one or two workgroups on a CU nothing else hides the latency of these loads.  Loads and LDS
        // stores are unconditional (a branch around a load makes the compiler wait at the join for the loads it
        // has just issued): tiles past the end of the reduction range are fetched from a clamped address and,
        // if they are stored at all, stored as zeros.  The loop runs whole quads; the 0-3 K-steps left over
        // only consume stages that are already on their way, so their uniform branches contain no loads.
        Stage a3 = (VEC ? fast_stage(fga, rbeg) : load_stage<AT, VEC>(ga, rbeg, tid));
        Stage b3 = (VEC ? fast_stage(fgb, rbeg) : load_stage<BT, VEC>(gb, rbeg, tid));
        Stage a0 = (VEC ? fast_stage(fga, rbeg + BK) : load_stage<AT, VEC>(ga, rbeg + BK, tid));
        Stage b0 = (VEC ? fast_stage(fgb, rbeg + BK) : load_stage<BT, VEC>(gb, rbeg + BK, tid));
        Stage a1 = (VEC ? fast_stage(fga, rbeg + 2 * BK) : load_stage<AT, VEC>(ga, rbeg + 2 * BK, tid));
        Stage b1 = (VEC ? fast_stage(fgb, rbeg + 2 * BK) : load_stage<BT, VEC>(gb, rbeg + 2 * BK, tid));
        Stage a2 = (VEC ? fast_stage(fga, rbeg + 3 * BK) : load_stage<AT, VEC>(ga, rbeg + 3 * BK, tid));
        Stage b2 = (VEC ? fast_stage(fgb, rbeg + 3 * BK) : load_stage<BT, VEC>(gb, rbeg + 3 * BK, tid));
        store_stage<AT>(sA[0], a3, ga, rbeg, tid);
        store_stage<BT>(sB[0], b3, gb, rbeg, tid);
        __syncthreads();
        const int nq = nk >> 2, rem = nk & 3;
        if constexpr (PIPE) {
            for (int q = 0; q < nq; ++q) {
                const int r1 = rbeg + (4 * q + 1) * BK, r2 = r1 + BK, r3 = r2 + BK, r4 = r3 + BK;
                if (r4 + BK <= rend) {
                    kstep<AT, BT, VEC, false>(sA[0], sB[0], sA[1], sB[1], ga, gb, fga, fgb, a3, b3, r4, a0, b0, r1, arow, brow, kh, tid, acc);
                    kstep<AT, BT, VEC, false>(sA[1], sB[1], sA[0], sB[0], ga, gb, fga, fgb, a0, b0, r4 + BK, a1, b1, r2, arow, brow, kh, tid, acc);
                    kstep<AT, BT, VEC, false>(sA[0], sB[0], sA[1], sB[1], ga, gb, fga, fgb, a1, b1, r4 + 2 * BK, a2, b2, r3, arow, brow, kh, tid, acc);
                    kstep<AT, BT, VEC, false>(sA[1], sB[1], sA[0], sB[0], ga, gb, fga, fgb, a2, b2, r4 + 3 * BK, a3, b3, r4, arow, brow, kh, tid, acc);
                } else {
                    kstep<AT, BT, VEC, true>(sA[0], sB[0], sA[1], sB[1], ga, gb, fga, fgb, a3, b3, r4, a0, b0, r1, arow, brow, kh, tid, acc);
                    kstep<AT, BT, VEC, true>(sA[1], sB[1], sA[0], sB[0], ga, gb, fga, fgb, a0, b0, r4 + BK, a1, b1, r2, arow, brow, kh, tid, acc);
                    kstep<AT, BT, VEC, true>(sA[0], sB[0], sA[1], sB[1], ga, gb, fga, fgb, a1, b1, r4 + 2 * BK, a2, b2, r3, arow, brow, kh, tid, acc);
                    kstep<AT, BT, VEC, true>(sA[1], sB[1], sA[0], sB[0], ga, gb, fga, fgb, a2, b2, r4 + 3 * BK, a3, b3, r4, arow, brow, kh, tid, acc);
                }
            }
        } else {
            for (int q = 0; q < nq; ++q) {
                const int r1 = rbeg + (4 * q + 1) * BK, r2 = r1 + BK, r3 = r2 + BK, r4 = r3 + BK;
                a3 = (VEC ? fast_stage(fga, r4) : load_stage<AT, VEC>(ga, r4, tid));
                b3 = (VEC ? fast_stage(fgb, r4) : load_stage<BT, VEC>(gb, r4, tid));
                acc = tile_mfma(sA[0], sB[0], arow, brow, kh, acc);
                store_stage<AT>(sA[1], a0, ga, r1, tid);
                store_stage<BT>(sB[1], b0, gb, r1, tid);
                __syncthreads();
                a0 = (VEC ? fast_stage(fga, r4 + BK) : load_stage<AT, VEC>(ga, r4 + BK, tid));
                b0 = (VEC ? fast_stage(fgb, r4 + BK) : load_stage<BT, VEC>(gb, r4 + BK, tid));
                acc = tile_mfma(sA[1], sB[1], arow, brow, kh, acc);
                store_stage<AT>(sA[0], a1, ga, r2, tid);
                store_stage<BT>(sB[0], b1, gb, r2, tid);
                __syncthreads();
                a1 = (VEC ? fast_stage(fga, r4 + 2 * BK) : load_stage<AT, VEC>(ga, r4 + 2 * BK, tid));
                b1 = (VEC ? fast_stage(fgb, r4 + 2 * BK) : load_stage<BT, VEC>(gb, r4 + 2 * BK, tid));
                acc = tile_mfma(sA[0], sB[0], arow, brow, kh, acc);
                store_stage<AT>(sA[1], a2, ga, r3, tid);
                store_stage<BT>(sB[1], b2, gb, r3, tid);
                __syncthreads();
                a2 = (VEC ? fast_stage(fga, r4 + 3 * BK) : load_stage<AT, VEC>(ga, r4 + 3 * BK, tid));
                b2 = (VEC ? fast_stage(fgb, r4 + 3 * BK) : load_stage<BT, VEC>(gb, r4 + 3 * BK, tid));
                acc = tile_mfma(sA[1], sB[1], arow, brow, kh, acc);
                store_stage<AT>(sA[0], a3, ga, r4, tid);
                store_stage<BT>(sB[0], b3, gb, r4, tid);
                __syncthreads();
            }
        }
        // buffer 0 holds tile 4 nq; stages 0 and 1 hold the next two
        // (a last K-step that holds at most 16 valid reduction indices runs half the MFMAs; uniform branches)
        const int rt = rbeg + 4 * nq * BK;
        // (only in the interleaved-loop kernel: in the plain-loop kernel the extra branches cost the large
        //  launches 8 % through a worse register assignment)
        const bool short_last = PIPE && rend - (rbeg + (nk - 1) * BK) <= BK / 2;
        if (rem >= 1) acc = (rem == 1 && short_last) ? tile_mfma_half(sA[0], sB[0], arow, brow, kh, acc) : tile_mfma(sA[0], sB[0], arow, brow, kh, acc);
        if (rem >= 2) {
            store_stage<AT>(sA[1], a0, ga, rt + BK, tid);
            store_stage<BT>(sB[1], b0, gb, rt + BK, tid);
            __syncthreads();
            acc = (rem == 2 && short_last) ? tile_mfma_half(sA[1], sB[1], arow, brow, kh, acc) : tile_mfma(sA[1], sB[1], arow, brow, kh, acc);
        }
        if (rem >= 3) {
            store_stage<AT>(sA[0], a1, ga, rt + 2 * BK, tid);
            store_stage<BT>(sB[0], b1, gb, rt + 2 * BK, tid);
            __syncthreads();
            acc = short_last ? tile_mfma_half(sA[0], sB[0], arow, brow, kh, acc) : tile_mfma(sA[0], sB[0], arow, brow, kh, acc);
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
    const int row0 = m0 + wm * 32 + 4 * (lane >> 5);
    // what the epilogue reads (gate values, the old C when accumulating) is fetched for all 16 rows at once,
    // from clamped addresses: one memory round trip instead of one per row
    float gate[16], old[16];
    if (mask) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            gate[reg] = mask[(long long)min(row0 + (reg & 3) + 8 * (reg >> 2), p.M - 1) * p.ldmask + min(j, p.N - 1)];
    }
    if (p.accumulate && j != p.ones_col) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) old[reg] = C[(long long)min(row0 + (reg & 3) + 8 * (reg >> 2), p.M - 1) * p.ldc + j];
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int i = row0 + (reg & 3) + 8 * (reg >> 2);
        if (i >= p.M) continue;
        float v = acc[reg] + bj;
        if (p.relu) v = fmaxf(v, 0.f);
        if (j == p.ones_col) {
            if (p.c_extra) p.c_extra[i] = p.accumulate ? p.c_extra[i] + v : v;
            continue;
        }
        if (mask && !(gate[reg] > 0.f)) v = 0.f;
        C[(long long)i * p.ldc + j] = p.accumulate ? old[reg] + v : v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-gradient product (both operands arrive [reduction][output]) on the bf16 matrix instruction at fp32
// accuracy.  Every fp32 element is cut EXACTLY into three bf16 pieces x = h + m + l (each the round-to-nearest bf16
// of what the pieces before it leave: |m| <= 2^-9 |x|, |l| <= 2^-18 |x|) as it is written to LDS, and a product a.b
// becomes the six piece products down to 2^-18 of it, hh + (hm + mh) + (mm + hl + lh); what is left out is below
// 2^-26 |a||b|, a quarter of one fp32 rounding.  The piece products are exact in fp32; hh goes to the main
// accumulator, the five small ones to a second one that joins it at the end (the instruction rounds its accumulator
// at the accumulator's magnitude, so small terms must not be added to the large sum one by one).  Twelve
// v_mfma_f32_32x32x16_bf16 (32 cycles each) replace sixteen v_mfma_f32_32x32x2_f32 (64 cycles each) per K-step.
// LDS image of an operand tile: three planes [32 reduction rows][64 outputs] of bf16, written as they arrive
// (8 bytes per thread and plane) with a row pitch of 192 bytes; the matrix instruction wants 8 consecutive
// reduction indices of one output per lane, i.e. columns of that image, which ds_read_b64_tr_b16 delivers (per 16
// lanes a block of 4 rows x 16 outputs, transposed; the pitch puts the four rows of a 32-lane half on disjoint
// banks).  Same tiles, grid, slab split and epilogue as gemm_f32_kernel<true, true, ..>.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
constexpr int SP_PITCH = 192, SP_PLANE = BK * SP_PITCH, SP_OPER = 3 * SP_PLANE, SP_BUF = 2 * SP_OPER;

__device__ __forceinline__ unsigned bf_pair(float a, float b) {         // (a, b) -> two bf16, round to nearest even
    return __builtin_bit_cast(unsigned, __builtin_convertvector(floatx2{a, b}, bf16x2));
}
// four consecutive outputs of one reduction row into the three planes
__device__ __forceinline__ void cut_quad(char* plane0, int off, float4 v) {
    const unsigned h0 = bf_pair(v.x, v.y), h1 = bf_pair(v.z, v.w);
    const float rx = v.x - __uint_as_float(h0 << 16), ry = v.y - __uint_as_float(h0 & 0xFFFF0000u);
    const float rz = v.z - __uint_as_float(h1 << 16), rw = v.w - __uint_as_float(h1 & 0xFFFF0000u);
    const unsigned m0 = bf_pair(rx, ry), m1 = bf_pair(rz, rw);
    const float sx = rx - __uint_as_float(m0 << 16), sy = ry - __uint_as_float(m0 & 0xFFFF0000u);
    const float sz = rz - __uint_as_float(m1 << 16), sw = rw - __uint_as_float(m1 & 0xFFFF0000u);
    const unsigned l0 = bf_pair(sx, sy), l1 = bf_pair(sz, sw);
    *reinterpret_cast<uintx2*>(plane0 + off) = uintx2{h0, h1};
    *reinterpret_cast<uintx2*>(plane0 + SP_PLANE + off) = uintx2{m0, m1};
    *reinterpret_cast<uintx2*>(plane0 + 2 * SP_PLANE + off) = uintx2{l0, l1};
}
// (no branch: reduction rows past the range are zeroed by selects, so that the cut can be scheduled between the
//  matrix instructions of the step before; outputs past the matrix keep whatever their clamped loads returned, they
//  only meet outputs the epilogue never writes)
__device__ __forceinline__ void cut_stage(char* oper, const Stage& st, const TileGeo& g, int r0, int tid) {
    // quad q: reduction row q >> 4, outputs 4 (q & 15) .. + 3
    const int off0 = (tid >> 4) * SP_PITCH + (tid & 15) * 8, off1 = off0 + 16 * SP_PITCH;
    const bool ok0 = r0 + (tid >> 4) < g.r_lim, ok1 = r0 + 16 + (tid >> 4) < g.r_lim;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    cut_quad(oper, off0, ok0 ? patch_ones<true>(st.v0, g, tid) : zero);
    cut_quad(oper, off1, ok1 ? patch_ones<true>(st.v1, g, tid + 256) : zero);
}

#define PSVAE_TR_READ(DST_, ADDR_, OFF_) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST_) : "v"(ADDR_), "n"(OFF_))

__global__ __launch_bounds__(256) void gemm_dw_split_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char sp_lds[];      // [2 buffers][A, B][3 planes][32][192 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // (tile order: see gemm_f32_kernel)
    const unsigned gxy = gridDim.x * gridDim.y, nb = gxy * gridDim.z;
    const unsigned bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = bid & 7u, per = nb >> 3, rem = nb & 7u;
    const unsigned tile = xcd * per + min(xcd, rem) + (bid >> 3);
    const int z = tile / gxy, txy = tile - z * gxy;
    const int tx = txy / gridDim.y, ty = txy - tx * gridDim.y;
    const int seg = z / p.splits, sp = z - seg * p.splits;
    const float* A = p.A + seg * p.seg_a;
    const float* B = p.B + seg * p.seg_b;
    const int m0 = tx * BM, n0 = ty * BN;
    const int rbeg = sp * p.rchunk;
    const int rend = min(p.R, rbeg + p.rchunk);
    const int nk = (rend - rbeg + BK - 1) / BK;

    floatx16 acc, lo;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; lo[i] = 0.f; }

    if (nk > 0) {
        const TileGeo ga{A, p.lda, m0, p.M, rend, -1}, gb{B, p.ldb, n0, p.N, rend, p.ones_col};
        // transposed-read addresses: lane 4 q + pp of a 16-lane group supplies row q, outputs 4 pp .. + 3 of the
        // group's 4 x 16 block; group g = lane >> 4 takes outputs 16 (g & 1) .. + 15 of the wave's 32 and the
        // reduction rows 8 (g >> 1) .. + 7 of a 16-row chunk (the instruction's lane map: row l & 31, k = 8 (l >> 5) + j)
        const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
        const unsigned lane_off = (unsigned)((8 * (grp >> 1) + q) * SP_PITCH + (16 * (grp & 1) + 4 * pp) * 2);
        const unsigned a_rd = lane_off + wm * 64, b_rd = SP_OPER + lane_off + wn * 64;     // + buffer offset
        const int rlast = rbeg + (nk - 1) * BK;
        const unsigned lds0 = (unsigned)(size_t)sp_lds;       // (the low half of a shared-memory pointer is its LDS offset)
        // Two register stages: while tile t is multiplied out of one LDS buffer, tile t+1 (loaded a step earlier) is
        // cut into the other buffer BETWEEN the matrix instructions (they hold the vector issue port for a quarter
        // of their time; the cut is ~150 vector instructions per step), and the loads of tile t+2 are in flight.
        Stage sa0 = load_stage<true, true>(ga, rbeg, tid), sb0 = load_stage<true, true>(gb, rbeg, tid);
        Stage sa1 = load_stage<true, true>(ga, min(rbeg + BK, rlast), tid), sb1 = load_stage<true, true>(gb, min(rbeg + BK, rlast), tid);
        cut_stage(sp_lds, sa0, ga, rbeg, tid);
        cut_stage(sp_lds + SP_OPER, sb0, gb, rbeg, tid);
        __syncthreads();
#define PSVAE_DW_STEP(T_, CUR_, NXT_, SA_CUT_, SB_CUT_, SA_LD_, SB_LD_)                                                  \
        {                                                                                                                 \
            SA_LD_ = load_stage<true, true>(ga, min(rbeg + ((T_) + 2) * BK, rlast), tid);   /* (past the end: unused) */  \
            SB_LD_ = load_stage<true, true>(gb, min(rbeg + ((T_) + 2) * BK, rlast), tid);                                 \
            const unsigned aa = lds0 + (CUR_) + a_rd, bb = lds0 + (CUR_) + b_rd;                                          \
            uintx2 fa[2][3][2], fb[2][3][2];                                                                              \
            _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                 \
                _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                                          \
                    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                    \
                        PSVAE_TR_READ(fa[c][pc][jj], aa, pc * SP_PLANE + (16 * c + 4 * jj) * SP_PITCH);                   \
                        PSVAE_TR_READ(fb[c][pc][jj], bb, pc * SP_PLANE + (16 * c + 4 * jj) * SP_PITCH);                   \
                    }                                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                               \
                bf16x8 a3[3], b3[3];                                                                                      \
                _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) {                                                        \
                    a3[pc] = __builtin_bit_cast(bf16x8, uintx4{fa[c][pc][0].x, fa[c][pc][0].y, fa[c][pc][1].x, fa[c][pc][1].y}); \
                    b3[pc] = __builtin_bit_cast(bf16x8, uintx4{fb[c][pc][0].x, fb[c][pc][0].y, fb[c][pc][1].x, fb[c][pc][1].y}); \
                }                                                                                                         \
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[2], b3[0], lo, 0, 0, 0);                                  \
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[2], lo, 0, 0, 0);                                  \
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[1], lo, 0, 0, 0);                                  \
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[0], lo, 0, 0, 0);                                  \
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[1], lo, 0, 0, 0);                                  \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[0], acc, 0, 0, 0);                                \
            }                                                                                                             \
            /* (on the last step this writes an all-zero tile nobody reads) */                                           \
            cut_stage(sp_lds + (NXT_), SA_CUT_, ga, rbeg + ((T_) + 1) * BK, tid);                                         \
            cut_stage(sp_lds + (NXT_) + SP_OPER, SB_CUT_, gb, rbeg + ((T_) + 1) * BK, tid);                               \
            /* one matrix instruction, then a dozen of the cut's vector instructions and one of its LDS stores */        \
            _Pragma("unroll") for (int u = 0; u < 12; ++u) {                                                              \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                        \
                __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);                                                       \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                                        \
            }                                                                                                             \
            __syncthreads();                                                                                              \
        }
        for (int t = 0; t < nk; t += 2) {
            PSVAE_DW_STEP(t, 0, SP_BUF, sa1, sb1, sa0, sb0)
            if (t + 1 >= nk) break;
            PSVAE_DW_STEP(t + 1, SP_BUF, 0, sa0, sb0, sa1, sb1)
        }
#undef PSVAE_DW_STEP
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += lo[i];
    }

    // epilogue: as in gemm_f32_kernel (lane holds column j, rows (reg&3) + 8*(reg>>2) + 4*(lane>>5))
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
    const int row0 = m0 + wm * 32 + 4 * (lane >> 5);
    float gate[16], old[16];
    if (mask) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            gate[reg] = mask[(long long)min(row0 + (reg & 3) + 8 * (reg >> 2), p.M - 1) * p.ldmask + min(j, p.N - 1)];
    }
    if (p.accumulate && j != p.ones_col) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) old[reg] = C[(long long)min(row0 + (reg & 3) + 8 * (reg >> 2), p.M - 1) * p.ldc + j];
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int i = row0 + (reg & 3) + 8 * (reg >> 2);
        if (i >= p.M) continue;
        float v = acc[reg] + bj;
        if (p.relu) v = fmaxf(v, 0.f);
        if (j == p.ones_col) {
            if (p.c_extra) p.c_extra[i] = p.accumulate ? p.c_extra[i] + v : v;
            continue;
        }
        if (mask && !(gate[reg] > 0.f)) v = 0.f;
        C[(long long)i * p.ldc + j] = p.accumulate ? old[reg] + v : v;
    }
}
#undef PSVAE_TR_READ

// ---------------------------------------------------------------------------------------------------------------
// The same product on 160 x 160 output tiles: 6.25 times the matrix instructions per barrier, per operand byte
// fetched and per workgroup prologue / epilogue of the 64 x 64 kernel, whose K-step (12 matrix instructions, 0.16 us of
// matrix-pipe time) takes 0.6 us however its pieces are arranged.  160 divides the widths that matter here (300 + the
// bias column -> 320, 150 -> 160, 600 -> 640) with the padding the 64-tiles have.  Each of the four waves owns 80 x 80
// outputs as 5 x 5 tiles of v_mfma_f32_16x16x32_bf16: per K-step 60 transposed fragment reads feed 150 matrix
// instructions (2400 cycles; the chip holds 1.5 - 1.9 GHz under them).  LDS image per operand: three planes
// [32 reduction rows][160 outputs] of bf16, row pitch 320 bytes (the four rows a 16-lane group reads lie 16 banks
// apart), two buffers: 120 KB, one workgroup per CU, one wave per SIMD.  Accumulators: 2 x 100 registers per lane (main
// and small-term sums), held in the accumulator half of the register file.  Always writes slabs.
// Measured at 95 k x 300 x 300: 105 us (154 TFLOP/s with the slab sum) against 166 us for the fp32 64-tile kernel and
// 180 for the cut-operand one; with the cut left out 80 us, with the matrix instructions left out 55.
constexpr int WD_T = 160, WD_PITCH = 2 * WD_T, WD_PLANE = BK * WD_PITCH, WD_OPER = 3 * WD_PLANE, WD_BUF = 2 * WD_OPER;
constexpr int WD_Q = BK * (WD_T / 4) / 256;          // float4 quads per thread and operand tile: 5
typedef short shortx4 __attribute__((ext_vector_type(4)));
typedef float floatx4w __attribute__((ext_vector_type(4)));
struct WStage { float4 v[WD_Q]; };
// A thread's quads of an operand tile: (row, column) never change, so the loads are "uniform tile base + 32-bit lane
// offset".  Only a tile that overhangs the reduction range needs other offsets (rows past the end read the last
// row, and are zeroed when cut): prepared once, chosen by a uniform select.
struct WGeo { const char* X; long long step; int ones_d, r_part; unsigned off[WD_Q], offp[WD_Q]; };

__device__ __forceinline__ WGeo wd_geo(const float* X, int ld, int o0, int o_lim, int ones_col, int r_lim, int r_last, int tid) {
    WGeo g;
    g.X = reinterpret_cast<const char*>(X); g.step = (long long)ld * 4;
    g.ones_d = ones_col >= 0 ? ones_col - o0 : -(1 << 20);
    g.r_part = r_last + BK > r_lim ? r_last : -1;
    const int lim = ones_col >= 0 ? ones_col : o_lim;
    const int safe = o0 < lim ? o0 : 0;          // (a tile that holds nothing but the ones column reads column 0)
#pragma unroll
    for (int u = 0; u < WD_Q; ++u) {
        const int q = tid + 256 * u, row = q / (WD_T / 4), c = (q - row * (WD_T / 4)) * 4;
        const int col = o0 + c < lim ? o0 + c : safe;             // a quad past the matrix reads the tile's first one (never used)
        g.off[u] = (unsigned)(row * ld + col) * 4u;
        g.offp[u] = (unsigned)(min(row, r_lim - 1 - r_last) * ld + col) * 4u;
    }
    return g;
}
__device__ __forceinline__ float4 wd_load_one(const WGeo& g, int u, int r0) {
    const char* base = g.X + (long long)r0 * g.step;
    return *reinterpret_cast<const float4*>(base + (r0 == g.r_part ? g.offp[u] : g.off[u]));
}
__device__ __forceinline__ WStage wd_load(const WGeo& g, int r0) {
    WStage s;
#pragma unroll
    for (int u = 0; u < WD_Q; ++u) s.v[u] = wd_load_one(g, u, r0);
    return s;
}
__device__ __forceinline__ floatx2 bf_widen(unsigned pair) {              // two bf16 -> two fp32
    return floatx2{__uint_as_float(pair << 16), __uint_as_float(pair & 0xFFFF0000u)};
}
__device__ __forceinline__ void wd_cut_quad(char* plane0, int off, float4 v) {
    const floatx2 a{v.x, v.y}, b{v.z, v.w};                              // (two-wide fp32 subtractions)
    const unsigned h0 = bf_pair(a.x, a.y), h1 = bf_pair(b.x, b.y);
    const floatx2 ra = a - bf_widen(h0), rb = b - bf_widen(h1);
    const unsigned m0 = bf_pair(ra.x, ra.y), m1 = bf_pair(rb.x, rb.y);
    const floatx2 sa = ra - bf_widen(m0), sb = rb - bf_widen(m1);
    const unsigned l0 = bf_pair(sa.x, sa.y), l1 = bf_pair(sb.x, sb.y);
    *reinterpret_cast<uintx2*>(plane0 + off) = uintx2{h0, h1};
    *reinterpret_cast<uintx2*>(plane0 + WD_PLANE + off) = uintx2{m0, m1};
    *reinterpret_cast<uintx2*>(plane0 + 2 * WD_PLANE + off) = uintx2{l0, l1};
}
// Quad u of a tile into the three planes.  Rows 8 .. 15 and 24 .. 31 of an image hold their 16-output tiles swapped
// in pairs (byte offset ^ 32): the two 16-lane groups that a transposed read serves together read rows eight apart,
// which would otherwise lie on the same banks.
__device__ __forceinline__ void wd_cut_one(char* oper, float4 v, const WGeo& g, int u, int r0, int r_lim, int tid) {
    const int q = tid + 256 * u, row = q / (WD_T / 4), c = (q - row * (WD_T / 4)) * 4;
    const int d = g.ones_d - c;
    v.x = d == 0 ? 1.f : v.x; v.y = d == 1 ? 1.f : v.y; v.z = d == 2 ? 1.f : v.z; v.w = d == 3 ? 1.f : v.w;
    const bool ok = r0 + row < r_lim;
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    wd_cut_quad(oper, row * WD_PITCH + ((c * 2) ^ ((row & 8) << 2)), v);
}
__device__ __forceinline__ void wd_cut(char* oper, const WStage& s, const WGeo& g, int r0, int r_lim, int tid) {
#pragma unroll
    for (int u = 0; u < WD_Q; ++u) wd_cut_one(oper, s.v[u], g, u, r0, r_lim, tid);
}
// 8 consecutive reduction indices (8 (lane >> 4) .. + 7) of output (lane & 15) of a 16-output tile: two transposed reads
__device__ __forceinline__ bf16x8 wd_frag(const char* p) {
    typedef __attribute__((address_space(3))) shortx4 lds_shortx4;
    const shortx4 f0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_shortx4*)(p));
    const shortx4 f1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_shortx4*)(p + 4 * WD_PITCH));
    typedef short shortx8 __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(bf16x8, shortx8{f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w});
}

__global__ __launch_bounds__(256) void gemm_dw_wide_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char sp_lds[];      // [2 buffers][A, B][3 planes][32][320 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // (tile order: see gemm_f32_kernel)
    const unsigned gxy = gridDim.x * gridDim.y, nb = gxy * gridDim.z;
    const unsigned bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = bid & 7u, per = nb >> 3, rem = nb & 7u;
    const unsigned tile = xcd * per + min(xcd, rem) + (bid >> 3);
    const int z = tile / gxy, txy = tile - z * gxy;
    const int tx = txy / gridDim.y, ty = txy - tx * gridDim.y;
    const int seg = z / p.splits, sp = z - seg * p.splits;
    const int m0 = tx * WD_T, n0 = ty * WD_T;
    const int rbeg = sp * p.rchunk;
    const int rend = min(p.R, rbeg + p.rchunk);
    const int nk = (rend - rbeg + BK - 1) / BK;

    // (An operand of the accumulator register class anywhere in the kernel makes the compiler keep the matrix
    //  instructions' accumulators in the accumulator half of the register file - 200 of them here - instead of
    //  treating that half as spill space for a 256-register allocation, which costs a copy in and out per instruction.)
    asm volatile("; accumulators live in a[]" ::"a"(0.f));
    floatx4w acc[5][5], lo[5][5];
#pragma unroll
    for (int rt = 0; rt < 5; ++rt)
#pragma unroll
        for (int ct = 0; ct < 5; ++ct) { acc[rt][ct] = floatx4w{0.f, 0.f, 0.f, 0.f}; lo[rt][ct] = floatx4w{0.f, 0.f, 0.f, 0.f}; }

    if (nk > 0) {
        const int rlast = rbeg + (nk - 1) * BK;
        const WGeo ga = wd_geo(p.A + seg * p.seg_a, p.lda, m0, p.M, -1, rend, rlast, tid);
        const WGeo gb = wd_geo(p.B + seg * p.seg_b, p.ldb, n0, p.N, p.ones_col, rend, rlast, tid);
        // lane 4 q + pp of a 16-lane group supplies reduction row q, outputs 4 pp .. + 3 of the group's 4 x 16 block;
        // group g = lane >> 4 reads rows 8 g .. + 7 (the instruction's lane map: output l & 15, k = 8 (l >> 4) + j)
        const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
        const int lane_off = (8 * grp + q) * WD_PITCH + 8 * pp;
        // Tile rt of the wave's five sits at byte y = 160 w + 32 rt of a row; in this group's rows it may be swapped with
        // its neighbour (y ^ 32 = y + 32 or y - 32 by the parity of y / 32): two lane bases, for even and for odd rt.
        const int swz = (grp & 1) << 5;                   // (this group's rows are 8 .. 15 or 24 .. 31 of the image)
        const int a_sw = ((wm * 5) & 1) ? -swz : swz, b_sw = ((wn * 5) & 1) ? -swz : swz;
        const int a_rd0 = lane_off + wm * 160 + a_sw, a_rd1 = lane_off + wm * 160 - a_sw;
        const int b_rd0 = WD_OPER + lane_off + wn * 160 + b_sw, b_rd1 = WD_OPER + lane_off + wn * 160 - b_sw;
        WStage sa = wd_load(ga, rbeg), sb = wd_load(gb, rbeg);
        wd_cut(sp_lds, sa, ga, rbeg, rend, tid);
        wd_cut(sp_lds + WD_OPER, sb, gb, rbeg, rend, tid);
        sa = wd_load(ga, min(rbeg + BK, rlast));
        sb = wd_load(gb, min(rbeg + BK, rlast));
        __syncthreads();
#define PSVAE_WD_TERM(ACC_, PA_, PB_)                                                                                    \
        _Pragma("unroll") for (int rt = 0; rt < 5; ++rt)                                                                 \
            ACC_[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[rt][PA_], fb[PB_], ACC_[rt][ct], 0, 0, 0);
        // One register stage, each quad of it in flight for a whole step: while the 30 matrix instructions of column
        // tile ct are issued, quad ct of either operand's NEXT tile (requested a step ago) is cut into the other
        // buffer and its registers are reloaded with the tile after that.  A wave issues in order: the cut's ~350 vector
        // instructions behind the 150 matrix instructions would add their time to the step; dealt out between them,
        // two or three to a matrix instruction, they run in its shadow.
        for (int t = 0; t < nk; ++t) {
            const int cur = (t & 1) * WD_BUF, nxt = WD_BUF - cur;
            const char* aa0 = sp_lds + cur + a_rd0;
            const char* aa1 = sp_lds + cur + a_rd1;
            const char* bb0 = sp_lds + cur + b_rd0;
            const char* bb1 = sp_lds + cur + b_rd1;
            const int r_cut = rbeg + (t + 1) * BK, r_ld = min(rbeg + (t + 2) * BK, rlast);     // (past the end: unused)
            bf16x8 fa[5][3];
#pragma unroll
            for (int rt = 0; rt < 5; ++rt)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) fa[rt][pc] = wd_frag(((rt & 1) ? aa1 : aa0) + pc * WD_PLANE + rt * 32);
#pragma unroll
            for (int ct = 0; ct < 5; ++ct) {
                bf16x8 fb[3];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) fb[pc] = wd_frag(((ct & 1) ? bb1 : bb0) + pc * WD_PLANE + ct * 32);
                PSVAE_WD_TERM(lo, 2, 0) PSVAE_WD_TERM(lo, 0, 2) PSVAE_WD_TERM(lo, 1, 1)
                PSVAE_WD_TERM(lo, 1, 0) PSVAE_WD_TERM(lo, 0, 1) PSVAE_WD_TERM(acc, 0, 0)
                // (on the last step this writes an all-zero tile nobody reads)
                wd_cut_one(sp_lds + nxt, sa.v[ct], ga, ct, r_cut, rend, tid);
                wd_cut_one(sp_lds + nxt + WD_OPER, sb.v[ct], gb, ct, r_cut, rend, tid);
                sa.v[ct] = wd_load_one(ga, ct, r_ld);
                sb.v[ct] = wd_load_one(gb, ct, r_ld);
#pragma unroll
                for (int u = 0; u < 30; ++u) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    if (u % 5 == 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    if (u == 27 || u == 28) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
#undef PSVAE_WD_TERM
    }
    // slab: this lane holds column n0 + 80 wn + 16 ct + (lane & 15), rows m0 + 80 wm + 16 rt + 4 (lane >> 4) + i
    float* C = p.C + (long long)z * p.slab_stride;
#pragma unroll
    for (int ct = 0; ct < 5; ++ct) {
        const int j = n0 + wn * 80 + ct * 16 + (lane & 15);
        if (j >= p.N) continue;
#pragma unroll
        for (int rt = 0; rt < 5; ++rt) {
            const int i0 = m0 + wm * 80 + rt * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i0 + i < p.M) C[(long long)(i0 + i) * p.ld_slab + j] = acc[rt][ct][i] + lo[rt][ct][i];
        }
    }
}

// Sum `nz` slabs (fixed order z = 0..nz-1) and apply the epilogue.
struct ReduceP {
    const float* slabs; long long slab_stride; int nz, ld_slab;
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
        const long long o = (long long)i * p.ld_slab + j;
        float v = 0.f;
        int z = 0;
        for (; z + 4 <= p.nz; z += 4) {   // 4 independent loads in flight, summed in slab order
            const float v0 = s[(long long)z * p.slab_stride + o], v1 = s[(long long)(z + 1) * p.slab_stride + o];
            const float v2 = s[(long long)(z + 2) * p.slab_stride + o], v3 = s[(long long)(z + 3) * p.slab_stride + o];
            v = (((v + v0) + v1) + v2) + v3;
        }
        for (; z < p.nz; ++z) v += s[(long long)z * p.slab_stride + o];
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

// Reduction split: a launch of few tiles is split along the reduction range into at most 512 workgroups - two per
// CU, all resident at once, each with at least 4 K-steps; these launches run beside other streams' work in the
// training step.  (While the weight gradients ran on the fp32 kernel - 190 registers, one to two workgroups per CU -
// 384 was the better target there: 512 cost the step 1.3 %; with the cut-operand kernel, which shares a CU better,
// 512 gains 0.7 %, 256 and 768 lose as much.)
struct Plan { int splits, rchunk; };
Plan plan_splits(int M, int N, int R, int nseg) {
    const long long tiles = (long long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * nseg;
    Plan pl{1, (int)round_up(R > 0 ? R : 1, BK)};
    if (tiles >= 200 || R <= 4 * BK) return pl;
    const int target = 512;
    int want = (int)(target / tiles);      // rounded down
    int maxs = R / (4 * BK);
    if (want > maxs) want = maxs;
    if (want <= 1) return pl;
    pl.rchunk = (int)round_up((R + want - 1) / want, BK);
    pl.splits = (R + pl.rchunk - 1) / pl.rchunk;
    return pl;
}

// The 160-tile weight-gradient kernel takes a product when its tiles pad the output no more than the 64-tiles do
// (widths near a multiple of 160; more padding is accepted over long reductions, see plan_wide) and the reduction is
// long enough for every workgroup of the split to run WIDE_MIN_STEPS K-steps.  One workgroup per CU (120 KB of LDS);
// the split aims at 128 workgroups: in the training step these products run beside two other streams' kernels, which
// take the other CUs (128 against 256: +2.3 % on the 128-graph step, +1 % at 256 graphs, equal at 1024; alone on the
// chip 256 is 40 % faster).  Option wide_gemm = 0 / 1 forces the kernel off / on (wherever the reduction is long enough).
constexpr int WIDE_MIN_STEPS = 4;
Plan plan_wide(int M, int N, int R, int nseg) {
    Plan none{0, 0};
    const int forced = opt(OPT_WIDE_GEMM);
    constexpr int min_steps = WIDE_MIN_STEPS;
    if (forced == 0) return none;
    const long long tw = (long long)((M + WD_T - 1) / WD_T) * ((N + WD_T - 1) / WD_T), t64 = (long long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    // Padding the 160-tiles may add over the 64-tiles': none for short reductions, 30 % from 12 k rows (node_trans,
    // 600 x 201), and 60 % from 18 432 rows on, where the kernel that would take the product otherwise is the fp32
    // one: the 160-tiles beat it by 1.5 x per padded flop alone and by more beside other kernels (they hold 128 - 192
    // CUs, it floods all of them).  That puts the link scorer's 656 x 329, 328 x 329 and 328 x 657 on them from 768
    // graphs per step (+1 %; +1.7 % at 1024), together with a split for 192 workgroups instead of 128 (+1.5 % more at
    // 1024).  From 10 k rows on the same rule loses 1 % at 512 graphs.  (Option wide_gemm = 1 accepts any padding.)
    constexpr int r160 = 18432;
    const int pad_pct = forced == 1 ? 1000 : (R >= r160 ? 160 : (R >= 12288 ? 130 : 100));
    if (tw * WD_T * WD_T * 100 > t64 * BM * BN * pad_pct) return none;
    const int target = R >= r160 ? 192 : 128;
    const long long tiles = tw * nseg;
    int want = (int)(target / tiles);
    const int maxs = R / (min_steps * BK);
    if (want > maxs) want = maxs;
    if (want < 1 || (long long)want * tiles < 64) return none;       // too little work to fill a quarter of the chip
    if ((long long)want * nseg < 2) return none;                     // (the kernel only writes slabs)
    Plan pl;
    pl.rchunk = (int)round_up((R + want - 1) / want, BK);
    pl.splits = (R + pl.rchunk - 1) / pl.rchunk;
    return pl;
}

}  // namespace

size_t gemm_scratch_floats(int M, int N, int R, int nseg, bool sum_segments) {
    Plan pl = plan_splits(M, N, R, nseg);
    long long nz = (long long)nseg * pl.splits;
    if (nz == 1 || (!sum_segments && pl.splits == 1)) nz = 0;
    if (sum_segments) {                                              // (weight gradients: the 160-tile plan, if it is the larger)
        const Plan pw = plan_wide(M, N, R, nseg);
        return (size_t)std::max(nz * (long long)M * N, (long long)nseg * pw.splits * M * round_up(N, 16));
    }
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
            // a scratch buffer too small for the planned split still gets the largest split that fits
            const size_t per_split = (size_t)d.nseg * d.M * d.N;
            const long long fit = d.scratch ? (long long)(d.scratch_floats / per_split) : 0;
            if (fit >= 2 || (fit >= 1 && d.sum_segments && d.nseg > 1)) {
                pl.rchunk = (int)round_up((d.R + fit - 1) / fit, BK);
                pl.splits = (d.R + pl.rchunk - 1) / pl.rchunk;
            } else if (d.sum_segments && d.nseg > 1) {
                return PSVAE_E_WORKSPACE;       // summed segments cannot do without their slabs
            } else {
                slab = false;
                pl.splits = 1; pl.rchunk = (int)round_up(d.R > 0 ? d.R : 1, BK);
            }
            nz_total = (long long)d.nseg * pl.splits;
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
    p.slab_stride = (long long)d.M * d.N; p.ld_slab = d.N;
    dim3 grid((d.M + BM - 1) / BM, (d.N + BN - 1) / BN, (unsigned)nz_total);
    if (grid.y > 65535 || grid.z > 65535) return PSVAE_E_SHAPE;
    const double flops = 2.0 * d.M * d.N * d.R * d.nseg;
    {
        ProfScope prof(s, d.a_t ? KC_GEMM_DW : (d.b_t ? KC_GEMM_DX : KC_GEMM_FWD), flops);
        const bool vec = p.a_vec && p.b_vec;
#define PSVAE_GEMM_LAUNCH(AT_, BT_)                                                                          \
        do {                                                                                                  \
            if (vec && pipe) hipLaunchKernelGGL((gemm_f32_kernel<AT_, BT_, true, true>), grid, dim3(256), 0, s, p);   \
            else if (vec) hipLaunchKernelGGL((gemm_f32_kernel<AT_, BT_, true, false>), grid, dim3(256), 0, s, p);   \
            else hipLaunchKernelGGL((gemm_f32_kernel<AT_, BT_, false, false>), grid, dim3(256), 0, s, p);           \
        } while (0)
        // the interleaved loop pays where a CU holds few workgroups (split reductions are planned for at most 512)
        const bool pipe = (long long)grid.x * grid.y * grid.z <= 1024;
        // Weight gradients of aligned operands: the cut-operand kernels on the bf16 matrix instruction.  Shapes the
        // 160-tiles fit (plan_wide) go to that kernel.  For the rest, the 64-tile cut-operand kernel runs a K-step in
        // the time the fp32 kernel needs (97 against 105 TFLOP/s at 95 k rows) but with 128 + 32 registers instead of
        // 190 + 32 it shares a CU better with the other streams' kernels: +2 % on the 128-graph step, +1 % at 256 and
        // 512 graphs, -1.5 % at 1024 (reductions over 19 k - 95 k rows there).  Used for reductions under 18 k rows;
        // option split_gemm = 0 / 1 forces fp32 / cut operands.
        const int split_forced = opt(OPT_SPLIT_GEMM);
        const Plan pw = (d.a_t && d.b_t && vec && d.sum_segments) ? plan_wide(d.M, d.N, d.R, d.nseg) : Plan{0, 0};
        if (pw.splits > 0 && d.scratch && d.scratch_floats >= (size_t)d.nseg * pw.splits * d.M * round_up(d.N, 16)) {
            static bool wide_attr_set = false;
            if (!wide_attr_set) {
                if (hipFuncSetAttribute((const void*)gemm_dw_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WD_BUF) != hipSuccess)
                    return (int)hipGetLastError();
                wide_attr_set = true;
            }
            pl = pw;
            nz_total = (long long)d.nseg * pl.splits;
            slab = true;
            p.C = d.scratch; p.to_slab = 1; p.splits = pl.splits; p.rchunk = pl.rchunk;
            p.ld_slab = (int)round_up(d.N, 16); p.slab_stride = (long long)d.M * p.ld_slab;
            const dim3 wgrid((d.M + WD_T - 1) / WD_T, (d.N + WD_T - 1) / WD_T, (unsigned)nz_total);
            hipLaunchKernelGGL(gemm_dw_wide_kernel, wgrid, dim3(256), 2 * WD_BUF, s, p);
        } else
        if (d.a_t && d.b_t && vec && (split_forced >= 0 ? split_forced == 1 : d.R < 18432)) {
            static bool attr_set = false;
            if (!attr_set) {
                if (hipFuncSetAttribute((const void*)gemm_dw_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SP_BUF) != hipSuccess)
                    return (int)hipGetLastError();
                attr_set = true;
            }
            hipLaunchKernelGGL(gemm_dw_split_kernel, grid, dim3(256), 2 * SP_BUF, s, p);
        } else
        if (!d.a_t && !d.b_t) PSVAE_GEMM_LAUNCH(false, false);
        else if (!d.a_t && d.b_t) PSVAE_GEMM_LAUNCH(false, true);
        else if (d.a_t && d.b_t) PSVAE_GEMM_LAUNCH(true, true);
        else PSVAE_GEMM_LAUNCH(true, false);
#undef PSVAE_GEMM_LAUNCH
    }
    PSVAE_LAUNCH_CHECK();
    if (d.slabs_out) {
        *d.slabs_out = GemmSlabs{slab ? d.scratch : nullptr, slab ? pl.splits : 0, p.slab_stride, p.ld_slab};
        if (slab) return 0;      // the caller's next kernel sums the slabs
    }
    if (slab) {
        ReduceP r;
        r.slabs = d.scratch; r.slab_stride = p.slab_stride; r.ld_slab = p.ld_slab;
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
        ProfScope prof(s, KC_GEMM_REDUCE, 4.0 * total * (nz_total + 1));
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

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient without any allocation: the slab partial sums of the split reduction live in a file-scope device
// array instead of a caller-supplied scratch buffer, for callers that cannot allocate (HIP graph capture, serving
// loops, harnesses that forbid hipMalloc).  Every GPU the library runs on holds its own copy of the array (32 MiB:
// the largest split the 160-tile kernel plans, 256 slabs of 160 x 160, always fits)
// until the process ends.  In host code the array's NAME is only a stand-in: the device address is asked for once
// per device with hipGetSymbolAddress and kept; kernels are given that address.
namespace {
constexpr size_t NOALLOC_SCRATCH_FLOATS = (size_t)8 << 20;       // 32 MiB
__device__ __align__(256) float scratch[NOALLOC_SCRATCH_FLOATS];
float* noalloc_ptr[64] = {};
std::mutex noalloc_mu;

int noalloc_scratch(float** out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return PSVAE_E_SHAPE;
    std::lock_guard<std::mutex> lock(noalloc_mu);
    if (!noalloc_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));        // this device's copy (the current device's)
        if (e != hipSuccess) return (int)e;
        if (!p) return PSVAE_E_NULL;
        noalloc_ptr[dev] = (float*)p;
    }
    *out = noalloc_ptr[dev];
    return 0;
}
}  // namespace

extern "C" size_t psvae_noalloc_scratch_bytes(void) { return NOALLOC_SCRATCH_FLOATS * sizeof(float); }

// Looks the current device's copy of the static array up now (psvae_streams_init calls this), so that the first call
// of the no-allocation entry point on this device makes no runtime call but its launches.
int noalloc_scratch_init() {
    float* p = nullptr;
    return noalloc_scratch(&p);
}

extern "C" int psvae_linear_backward_weight_noalloc(const float* dy, int lddy, int64_t seg_stride_dy, const float* x, int ldx,
                                                    int64_t seg_stride_x, int nseg, float* dw, float* db, int M, int N, int K,
                                                    void* stream) {
    if (!dy || !x || !dw) return PSVAE_E_NULL;
    if (M < 0 || N < 1 || K < 1 || nseg < 1 || lddy < N || ldx < K) return PSVAE_E_SHAPE;
    // what the planned split needs (same plan as psvae_linear_backward_weight with a scratch buffer of
    // psvae_linear_scratch_bytes): beyond the static array's size the call is refused, never degraded
    const size_t need = gemm_scratch_floats(N, db ? K + 1 : K, M, nseg, true);
    if (need > NOALLOC_SCRATCH_FLOATS) return PSVAE_E_STATIC_LIMIT;
    float* ptr = nullptr;
    PSVAE_TRY(noalloc_scratch(&ptr));
    // nothing in the array has to start at zero: every slab element the reduction reads is written by this call's
    // product kernel first.  Calls that overlap on one GPU (two streams, two threads) share the array and race.
    return linear_bwd_w((hipStream_t)stream, dy, lddy, seg_stride_dy, x, ldx, seg_stride_x, nseg, dw, db, M, N, K, ptr,
                        NOALLOC_SCRATCH_FLOATS);
}
