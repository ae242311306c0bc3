// Device-side building blocks shared by the fused kernels (fused_layer.hip, fused_stack.hip): products of an LDS
// operand image with packed weights on the bf16 matrix instruction over operands cut into three bf16 pieces, and the
// stores that go with them.  Everything here has internal linkage (one copy per translation unit).
#pragma once
#include "psvae_internal.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int FL_NB = 8, FL_THREADS = 256;

__device__ __forceinline__ float4 fl_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int CT, int RT>
__device__ __forceinline__ void fl_zero(floatx4 (&acc)[RT][CT]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = floatx4{0.f, 0.f, 0.f, 0.f};
}

template <int CT>
__device__ __forceinline__ void fl_bias(const float* __restrict__ b, int N, int wave, int lane, float (&out)[CT]) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) out[ct] = b[min((wave * CT + ct) * 16 + (lane & 15), N - 1)];
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf_bits(float x) {              // round to nearest even (v_cvt_pk_bf16_f32)
    const __bf16 b = (__bf16)x;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ void bf_cut(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = bf_bits(x);
    const float r1 = x - __uint_as_float(h << 16);                 // exact: at most 16 significand bits, |r1| <= 2^-9 |x|
    m = bf_bits(r1);
    const float r2 = r1 - __uint_as_float(m << 16);                // exact: at most 7 significand bits, |r2| <= 2^-18 |x|
    l = bf_bits(r2);                                               // exact
}
// four consecutive k of one row into the three planes (8 bytes each)
__device__ __forceinline__ void bf_put4(char* img, int plane, int off, float4 v) {
    unsigned h[4], m[4], l[4];
    bf_cut(v.x, h[0], m[0], l[0]); bf_cut(v.y, h[1], m[1], l[1]); bf_cut(v.z, h[2], m[2], l[2]); bf_cut(v.w, h[3], m[3], l[3]);
    *reinterpret_cast<uint2*>(img + off) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    *reinterpret_cast<uint2*>(img + plane + off) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
    *reinterpret_cast<uint2*>(img + 2 * plane + off) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}
__device__ __forceinline__ void bf_put1(char* img, int plane, int off, float v) {
    unsigned h, m, l;
    bf_cut(v, h, m, l);
    *reinterpret_cast<unsigned short*>(img + off) = (unsigned short)h;
    *reinterpret_cast<unsigned short*>(img + plane + off) = (unsigned short)m;
    *reinterpret_cast<unsigned short*>(img + 2 * plane + off) = (unsigned short)l;
}

__device__ __forceinline__ float4 fl_ld16(const char* p) { return *reinterpret_cast<const float4*>(p); }

// blocks 0 and 1 of a product's weights, requested ahead of the phase in front of the product
template <int CT>
__device__ __forceinline__ void flb_prefetch(const char* __restrict__ Wp, int KB, int wave, int lane, float4 (&h0)[3][CT], float4 (&h1)[3][CT]) {
    constexpr int Npad = 64 * CT;
    const char* wlane = Wp + ((size_t)(lane >> 4) * Npad + wave * CT * 16 + (lane & 15)) * 16;
    const char* p1 = wlane + (size_t)min(1, KB - 1) * (3 * 4 * Npad * 16);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) h0[p][ct] = fl_ld16(wlane + ((size_t)p * 4 * Npad + ct * 16) * 16);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) h1[p][ct] = fl_ld16(p1 + ((size_t)p * 4 * Npad + ct * 16) * 16);
}

// acc[rt][ct] += image rows of tile rt . W columns of tile ct over KB blocks of 32 reduction indices.  On entry
// b0 / b1 hold (or have requested) the weights of blocks 0 / 1.  Every block's weights are requested a block and a
// half ahead, its image rows half a block ahead.
template <int CT, int RT>
__device__ __forceinline__ void flb_product(const char* img, int PB, int plane, const char* __restrict__ Wp, int KB, int wave, int lane,
                                            floatx4 (&acc)[RT][CT], float4 (&b0)[3][CT], float4 (&b1)[3][CT]) {
    const int nl = lane & 15, kq = lane >> 4;
    constexpr int Npad = 64 * CT;
    const char* wlane = Wp + ((size_t)kq * Npad + wave * CT * 16 + nl) * 16;
    const char* arow = img + nl * PB + kq * 16;
    float4 a0[3][RT], a1[3][RT];
#define FLB_LOADB(B_, KB_)                                                                                  \
    {                                                                                                       \
        const char* p_ = wlane + (size_t)min((KB_), KB - 1) * (3 * 4 * Npad * 16);                          \
        _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                       \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                               \
                B_[p][ct] = fl_ld16(p_ + ((size_t)p * 4 * Npad + ct * 16) * 16);                            \
    }
#define FLB_READA(A_, KB_)                                                                                  \
    {                                                                                                       \
        const int o_ = min((KB_), KB - 1) * 64;                                                             \
        _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                       \
            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                               \
                A_[p][rt] = *reinterpret_cast<const float4*>(arow + p * plane + rt * 16 * PB + o_);         \
    }
#define FLB_TERM(ACC_, A_, PA_, B_, PB_)                                                                    \
    _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                                       \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                   \
            ACC_[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A_[PA_][rt]), \
                                                                   __builtin_bit_cast(bf16x8, B_[PB_][ct]), ACC_[rt][ct], 0, 0, 0);
    // The instruction rounds its accumulator as it adds, at the accumulator's magnitude whatever the size of the
    // terms: the five small products therefore go to an accumulator of their own (2^-9 of the main one, and so are
    // its roundings) that joins the main one once, at the end - the main accumulator is rounded as often as a chain
    // of fp32 multiply-adds rounds, not six times as often.
    floatx4 lo[RT][CT];
    fl_zero<CT, RT>(lo);
    // One block's matrix instructions with the loads of the block two ahead dealt in between them: a wave issues in
    // order, and the packed weights arrive through the CU's vector-memory path at about the rate the matrix pipe
    // consumes them (a 1-KiB wave load per four to six matrix instructions), so requests queued behind a block's
    // instructions leave the pipe idle while they are issued.  The piece products are ordered so that a plane of the
    // weights is free as early as possible: l first (one product), then m (two), then h (three); each plane's
    // registers are reloaded right behind its last product.
#define FLB_LOADP(B_, KB_, P_)                                                                              \
    {                                                                                                       \
        const char* p_ = wlane + (size_t)min((KB_), KB - 1) * (3 * 4 * Npad * 16);                          \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                                   \
            B_[P_][ct] = fl_ld16(p_ + ((size_t)(P_) * 4 * Npad + ct * 16) * 16);                            \
    }
#define FLB_MFMA(A_, B_, KBN_)                                                                              \
    {                                                                                                       \
        FLB_TERM(lo, A_, 0, B_, 2) FLB_LOADP(B_, KBN_, 2)                                                   \
        FLB_TERM(lo, A_, 1, B_, 1) FLB_TERM(lo, A_, 0, B_, 1) FLB_LOADP(B_, KBN_, 1)                        \
        FLB_TERM(lo, A_, 2, B_, 0) FLB_TERM(lo, A_, 1, B_, 0) FLB_TERM(acc, A_, 0, B_, 0) FLB_LOADP(B_, KBN_, 0) \
        __builtin_amdgcn_sched_group_barrier(0x008, RT * CT, 0);                                            \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                                 \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * RT, 0);                                         \
        }                                                                                                   \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                                 \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * RT, 0);                                         \
        }                                                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x020, CT, 0);                                                 \
    }
    // Everything requested so far - the two prefetched blocks and whatever the phase in front of the product left in
    // flight - is waited for HERE, once: a wait in the loop is one instruction for the first and for every later
    // iteration, and with older requests pending on entry the compiler sizes it for those (all but one load complete),
    // which in every later iteration exposes the latency of the block requested half an iteration earlier.
    __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0)
    FLB_READA(a0, 0)
    __builtin_amdgcn_sched_barrier(0);
    for (int kb = 0; kb < KB; kb += 2) {        // (the guards are uniform and hold no loads the next block needs)
        FLB_READA(a1, kb + 1)
        __builtin_amdgcn_sched_barrier(0);
        FLB_MFMA(a0, b0, kb + 2)
        __builtin_amdgcn_sched_barrier(0);
        FLB_READA(a0, kb + 2)
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < KB) FLB_MFMA(a1, b1, kb + 3)
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] += lo[rt][ct];
#undef FLB_MFMA
#undef FLB_LOADP
#undef FLB_TERM
#undef FLB_READA
#undef FLB_LOADB
}

// fl_store for a split image: the same values, cut into the three planes (2-byte stores)
template <int CT, int RT, bool RELU, bool ZERO_PAD>
__device__ __forceinline__ void flb_store(const floatx4 (&acc)[RT][CT], const float (&bias)[CT], int N, char* img, int PB, int plane,
                                          int kpad, float* gout, size_t ldg, int row0, int nrows, int wave, int lane) {
    const int nl = lane & 15, kq = lane >> 4;
    const bool full = row0 + 16 * RT <= nrows;
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
        if (img && (cok || (ZERO_PAD && c < kpad))) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) bf_put1(img, plane, (rt * 16 + 4 * kq + i) * PB + 2 * c, v[rt][i]);
        }
        if (gout && cok) {
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

template <int CT, int RT, bool SYNC>
__device__ __forceinline__ void flb_store_gated(const floatx4 (&acc)[RT][CT], const float* __restrict__ act, size_t ldact, int N,
                                                char* img, int PB, int plane, int kpad, float* gout, size_t ldg, int row0, int nrows,
                                                int wave, int lane) {
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
    const bool full = row0 + 16 * RT <= nrows;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = (wave * CT + ct) * 16 + nl;
        const bool cok = c < N;
        float v[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[rt][i] = (cok && gate[rt][ct][i] > 0.f) ? acc[rt][ct][i] : 0.f;
        if (c < kpad) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) bf_put1(img, plane, (rt * 16 + 4 * kq + i) * PB + 2 * c, v[rt][i]);
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

int flb_pitch_bytes(int K) { return (K + 31) / 32 * 64 + 16; }     // an odd number of 16-byte units

}  // namespace
