// Single-layer GRU over the piece sequence (torch.nn.GRU semantics, gates r,z,n;
// the reference builds it at src/modules/vae_piece_decoder.py:29 and runs it at :89).
//
// The input products gi = x W_ih^T + b_ih for all steps are one large GEMM done
// elsewhere; what remains is the recurrence.  One workgroup keeps a tile of 16
// sequences resident for all T steps: the hidden state lives in LDS, every wave
// owns 16 hidden units and computes their three gate pre-activations with
// v_mfma_f32_16x16x4_f32 (A = hidden tile from LDS, B = W_hh rows streamed from
// L2), so r, z and n of one (sequence, unit) land in the same lane and the gate
// math needs no exchange.  One barrier per time step.
//
// LDS image of a [16 x K] tile: element (row, k) at ((k>>2)*16 + row)*4 + (k&3),
// i.e. one ds_read_b128 gives a lane 4 consecutive k of its row, conflict-free.
#include "psvae_internal.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 16;

__device__ __forceinline__ int tile_idx(int k, int row) { return (((k >> 2) * ROWS + row) << 2) + (k & 3); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(1024) void gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ h0,
                                                       const float* __restrict__ whh, const float* __restrict__ bhh,
                                                       float* __restrict__ hseq, float* __restrict__ hprev,
                                                       float* __restrict__ gates, int B, int T, int R) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int R16 = (R + 15) & ~15;
    float* buf0 = smem;
    float* buf1 = smem + R16 * ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int b0 = blockIdx.x * ROWS;
    const int ntiles = R16 / 16;
    for (int e = tid; e < R16 * ROWS; e += blockDim.x) {
        const int row = e / R16, k = e - row * R16;
        const int b = b0 + row;
        buf0[tile_idx(k, row)] = (k < R && b < B) ? h0[(long long)b * R + k] : 0.f;
        buf1[tile_idx(k, row)] = 0.f;
    }
    __syncthreads();
    const int j = lane & 15, kq = lane >> 4;
    for (int t = 0; t < T; ++t) {
        const float* cur = (t & 1) ? buf1 : buf0;
        float* nxt = (t & 1) ? buf0 : buf1;
        for (int tile = wave; tile < ntiles; tile += nwaves) {
            const int unit = tile * 16 + j;
            const bool uok = unit < R;
            const float* wr = whh + (long long)(uok ? unit : 0) * R;
            const float* wu = wr + (long long)R * R;
            const float* wn = wu + (long long)R * R;
            floatx4 ar = {0.f, 0.f, 0.f, 0.f}, au = ar, an = ar;
            // W_hh fragments are fetched two K-groups ahead of the MFMAs that consume them
            // (unconditional loads at a clamped address: a branch around a load would drain the
            // load queue; columns k >= R meet zeros of the hidden tile, rows of units >= R are discarded)
            auto wload = [&](const float* w, int m) -> float4 {
                const int k = min(16 * m + 4 * kq, R - 4);
                return *reinterpret_cast<const float4*>(w + k);
            };
            float4 br = wload(wr, 0), bu = wload(wu, 0), bn = wload(wn, 0);
            float4 br1 = wload(wr, 1), bu1 = wload(wu, 1), bn1 = wload(wn, 1);
            // this lane's input-gate terms for the step (consumed after the MFMA loop)
            float gir[4], giu[4], gin[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int b = b0 + kq * 4 + reg;
                const float* g = gi + ((long long)min(b, B - 1) * T + t) * 3 * R + (uok ? unit : 0);
                gir[reg] = g[0]; giu[reg] = g[R]; gin[reg] = g[2 * R];   // unused when the row or unit is out of range
            }
            for (int m = 0; m < ntiles; ++m) {
                const float4 br2 = wload(wr, m + 2), bu2 = wload(wu, m + 2), bn2 = wload(wn, m + 2);
                const float4 a = *reinterpret_cast<const float4*>(&cur[((4 * m + kq) * ROWS + j) << 2]);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, br.x, ar, 0, 0, 0);
                au = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bu.x, au, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bn.x, an, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, br.y, ar, 0, 0, 0);
                au = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bu.y, au, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bn.y, an, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, br.z, ar, 0, 0, 0);
                au = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bu.z, au, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bn.z, an, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, br.w, ar, 0, 0, 0);
                au = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bu.w, au, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bn.w, an, 0, 0, 0);
                br = br1; bu = bu1; bn = bn1;
                br1 = br2; bu1 = bu2; bn1 = bn2;
            }
            if (uok) {
                const float cr = bhh[unit], cu = bhh[R + unit], cn = bhh[2 * R + unit];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = kq * 4 + reg, b = b0 + row;
                    float hn = 0.f;
                    if (b < B) {
                        const long long base = (long long)b * T + t;
                        const float ghn = an[reg] + cn;
                        const float r = sigmoidf_(gir[reg] + ar[reg] + cr);
                        const float u = sigmoidf_(giu[reg] + au[reg] + cu);
                        const float c = tanhf(gin[reg] + r * ghn);
                        const float hp = cur[tile_idx(unit, row)];
                        hn = (1.0f - u) * c + u * hp;
                        hseq[base * R + unit] = hn;
                        hprev[base * R + unit] = hp;
                        float* gs = gates + base * 4 * R;
                        gs[unit] = r; gs[R + unit] = u; gs[2 * R + unit] = c; gs[3 * R + unit] = ghn;
                    }
                    nxt[tile_idx(unit, row)] = hn;
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void gru_bwd_kernel(const float* __restrict__ dhseq, const float* __restrict__ hprev,
                                                       const float* __restrict__ gates, const float* __restrict__ whh,
                                                       float* __restrict__ dgi, float* __restrict__ dgh,
                                                       float* __restrict__ dh0, int B, int T, int R) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int R16 = (R + 15) & ~15, G = 3 * R, G16 = (G + 15) & ~15;
    float* sDH = smem;                 // [16][R16] running dL/dh
    float* sG = smem + ROWS * R16;     // tile image of d(gh) [16 x G16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int b0 = blockIdx.x * ROWS;
    for (int e = tid; e < ROWS * R16 + ROWS * G16; e += blockDim.x) smem[e] = 0.f;
    __syncthreads();
    const int j = lane & 15, kq = lane >> 4;
    const int ntiles = R16 / 16, msteps = G16 / 16;
    for (int t = T - 1; t >= 0; --t) {
        for (int e = tid; e < ROWS * R; e += blockDim.x) {
            const int row = e / R, u = e - row * R;
            const int b = b0 + row;
            if (b >= B) continue;
            const long long base = (long long)b * T + t;
            const float d = sDH[row * R16 + u] + dhseq[base * R + u];
            const float* gs = gates + base * 4 * R;
            const float r = gs[u], z = gs[R + u], c = gs[2 * R + u], ghn = gs[3 * R + u];
            const float hp = hprev[base * R + u];
            const float dpc = d * (1.0f - z) * (1.0f - c * c);
            const float dpu = d * (hp - c) * z * (1.0f - z);
            const float dghn = dpc * r;
            const float dpr = dpc * ghn * r * (1.0f - r);
            float* o1 = dgi + base * G;
            float* o2 = dgh + base * G;
            o1[u] = dpr; o1[R + u] = dpu; o1[2 * R + u] = dpc;
            o2[u] = dpr; o2[R + u] = dpu; o2[2 * R + u] = dghn;
            sG[tile_idx(u, row)] = dpr;
            sG[tile_idx(R + u, row)] = dpu;
            sG[tile_idx(2 * R + u, row)] = dghn;
            sDH[row * R16 + u] = d * z;
        }
        __syncthreads();
        for (int tile = wave; tile < ntiles; tile += nwaves) {
            const int kcol = tile * 16 + j;
            const bool kok = kcol < R;
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            // W_hh rows are fetched two K-groups ahead of the MFMAs that consume them
            auto wload = [&](int m) -> float4 {
                // unconditional loads at clamped addresses (rows >= G meet zeros of the d(gh) tile,
                // columns >= R are discarded): no branch may sit between a load and its use
                const int jb = min(16 * m + 4 * kq, G - 4);
                const float* w = whh + (long long)jb * R + min(kcol, R - 1);
                return make_float4(w[0], w[R], w[2 * R], w[3 * R]);
            };
            float4 w0 = wload(0), w1 = wload(1);
            for (int m = 0; m < msteps; ++m) {
                const float4 w2 = wload(m + 2);
                const float4 a = *reinterpret_cast<const float4*>(&sG[((4 * m + kq) * ROWS + j) << 2]);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w0.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w0.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w0.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w0.w, acc, 0, 0, 0);
                w0 = w1; w1 = w2;
            }
            if (kok) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) sDH[(kq * 4 + reg) * R16 + kcol] += acc[reg];
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < ROWS * R; e += blockDim.x) {
        const int row = e / R, u = e - row * R;
        if (b0 + row < B) dh0[(long long)(b0 + row) * R + u] = sDH[row * R16 + u];
    }
}

}  // namespace

int gru_fwd(hipStream_t s, const float* gi, const float* h0, const float* whh, const float* bhh, float* hseq,
            float* hprev, float* gates, int B, int T, int R) {
    if (B == 0 || T == 0) return 0;
    if (R % 4 || !aligned16(whh)) return PSVAE_E_ALIGN;
    const int R16 = (R + 15) & ~15, ntiles = R16 / 16;
    const size_t lds = (size_t)2 * R16 * ROWS * sizeof(float);
    if (lds > 64 * 1024) return PSVAE_E_SHAPE;
    const int nwaves = ntiles < 16 ? ntiles : 16;
    ProfScope prof(s, KC_GRU_FWD, 2.0 * B * T * 3 * R * R);
    hipLaunchKernelGGL(gru_fwd_kernel, dim3((B + ROWS - 1) / ROWS), dim3(64 * nwaves), lds, s, gi, h0, whh, bhh, hseq,
                       hprev, gates, B, T, R);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int gru_bwd(hipStream_t s, const float* dhseq, const float* hprev, const float* gates, const float* whh,
            float* dgi, float* dgh, float* dh0, int B, int T, int R) {
    if (B == 0) return 0;
    if (R % 4) return PSVAE_E_ALIGN;
    const int R16 = (R + 15) & ~15, G16 = (3 * R + 15) & ~15, ntiles = R16 / 16;
    const size_t lds = (size_t)(ROWS * R16 + ROWS * G16) * sizeof(float);
    if (lds > 64 * 1024) return PSVAE_E_SHAPE;
    const int nwaves = ntiles < 16 ? ntiles : 16;
    ProfScope prof(s, KC_GRU_BWD, 2.0 * B * T * 3 * R * R);
    hipLaunchKernelGGL(gru_bwd_kernel, dim3((B + ROWS - 1) / ROWS), dim3(64 * nwaves), lds, s, dhseq, hprev, gates, whh,
                       dgi, dgh, dh0, B, T, R);
    PSVAE_LAUNCH_CHECK();
    return 0;
}
