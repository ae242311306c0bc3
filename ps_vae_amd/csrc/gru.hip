// Single-layer GRU over the piece sequence (torch.nn.GRU semantics, gates r,z,n;
// the reference builds it at src/modules/vae_piece_decoder.py:29 and runs it at :89).
//
// The input products gi = x W_ih^T + b_ih for all steps are one large GEMM done
// elsewhere; what remains is the recurrence.  Two implementations:
//
//  * multi-CU ("coop"): the recurrence of 16 sequences is spread over R/16
//    workgroups, one per 16 hidden units.  Each keeps its slice of W_hh in
//    registers for the whole sequence, computes its units' gate pre-activations
//    with v_mfma_f32_16x16x4_f32 (K split over the 4 waves, summed through LDS)
//    and hands its 16x16 piece of the new hidden state to the other workgroups of
//    the group through global memory as 8-byte {step tag, value} words written
//    with agent-scope atomic stores; consumers re-read their MFMA fragments with
//    agent-scope atomic loads until every tag carries the step they wait for, so
//    no flag, fence or grid barrier is needed and the step costs one cross-CU
//    hop.  Needs a caller-provided exchange buffer (zeroed here before each
//    launch) and all workgroups of a group resident at once, which the launch
//    guarantees by capping the grid at the CU count; every wait is bounded.
//  * single-CU: one workgroup per 16 sequences with the hidden state in LDS,
//    W_hh streamed from L2 every step.  Used when no exchange buffer is given or
//    R is outside the coop kernels' register budget.
//
// LDS image of a [16 x K] tile (single-CU kernels): element (row, k) at
// ((k>>2)*16 + row)*4 + (k&3), i.e. one ds_read_b128 gives a lane 4 consecutive k
// of its row, conflict-free.
#include "psvae_internal.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 16;

__device__ __forceinline__ int tile_idx(int k, int row) { return (((k >> 2) * ROWS + row) << 2) + (k & 3); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(1024) void gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ h0,
                                                       const float* __restrict__ whh, const float* __restrict__ bhh,
                                                       float* __restrict__ hseq, float* __restrict__ hprev,
                                                       float* __restrict__ gates, int B, int T, int R,
                                                       const unsigned int* run_if, unsigned int* redo_count) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // as the redo of a multi-CU recurrence that gave up (run_if = its give-up word): nothing to do unless it did
    if (run_if && __hip_atomic_load(run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    if (redo_count && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(redo_count, 1u);
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
                                                       float* __restrict__ dh0, int B, int T, int R,
                                                       const unsigned int* run_if, unsigned int* redo_count) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (run_if && __hip_atomic_load(run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    if (redo_count && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(redo_count, 1u);
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


// ---------------------------------------------------------------------------
// multi-CU recurrence
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr unsigned long long SPIN_LIMIT_TICKS = 200000000ull;   // 2 s of the 100 MHz wall clock (option gru_spin_ticks)
constexpr int EXCH_HEADER_WORDS = 32;                            // 256 B: word 0 = give-up flag

__device__ __forceinline__ unsigned long long granule(unsigned tag, float v) {
    return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}

// One wave re-reads its exchange words until every one carries `tag`: `count` (<= N, wave-uniform, >= 1) words,
// `stride` words apart, for each of `ntl` (<= RB, wave-uniform, >= 1) sequence tiles that lie `tile_stride` words
// apart.  All RB * N loads of a round are in flight together (surplus slots repeat the last word / the last tile:
// no branch around a load).  Gives up (returns false) when another wave has given up or after `limit` ticks of
// the 100 MHz wall clock (0: at once); the launcher queues the single-CU kernel behind this one, which redoes the
// whole recurrence when the give-up word is set.
template <int N, int RB>
__device__ __forceinline__ bool sweep(gu64* first, long long stride, int count, long long tile_stride, int ntl, unsigned tag,
                                      float (&v)[RB][N], gu32* giveup, unsigned long long limit, int lane) {
    const unsigned long long t0 = wall_clock64();
    if (limit == 0ull) {
        if (lane == 0) __hip_atomic_store(giveup, 1u, RLX_AGENT);
        return false;
    }
    for (unsigned spins = 1;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            gu64* base = first + (rb < ntl ? rb : ntl - 1) * tile_stride;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int ii = i < count ? i : count - 1;
                const unsigned long long x = __hip_atomic_load(base + ii * stride, RLX_AGENT);
                v[rb][i] = __uint_as_float((unsigned)x);
                ok &= (unsigned)(x >> 32) == tag;
            }
        }
        if (__all(ok)) return true;
        if ((spins & 63u) == 0u) {
            if (__hip_atomic_load(giveup, RLX_AGENT) != 0u) return false;
            if (wall_clock64() - t0 > limit) {
                if (lane == 0) __hip_atomic_store(giveup, 1u, RLX_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// grid (hidden tiles, sequence-group slots); block 256.  A workgroup steps RB sequence tiles of 16 sequences
// each through time together (the W_hh slice in its registers serves all of them, and the cross-CU hop of a
// time step is paid once for the RB tiles).  Thread (wave w, lane l) owns element (row 4*(l>>4)+w, unit
// j0+(l&15)) of every tile for the gate arithmetic = register w of lane l in the MFMA output layout, so the
// cross-wave K reduction reads LDS without bank conflicts.
// Exchange buffer: [parity][sequence tile][k][16 rows] words; h_t carries tag t.
// STAMP: thread 0 of workgroup (0, 0) writes the 100 MHz wall clock to stamps[t][0..3] at four points of every time
// step (top of the step, operands of h_t arrived, K-split partial sums exchanged through LDS, gates done and h_{t+1}
// published) - the instrumented variant behind psvae_gru_forward_stamped.
template <int KSW, int RB, bool STAMP = false>
__global__ __launch_bounds__(256) void gru_fwd_coop_kernel(const float* __restrict__ gi, const float* __restrict__ h0,
                                                           const float* __restrict__ whh, const float* __restrict__ bhh,
                                                           float* __restrict__ hseq, float* __restrict__ hprev,
                                                           float* __restrict__ gates, unsigned long long* exch_,
                                                           int B, int T, int R, int n_st, unsigned long long spin_limit,
                                                           unsigned long long* stamps = nullptr) {
    const bool stamper = STAMP && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0;
    __shared__ float red[2][RB][4][3][4][64];
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int j0 = blockIdx.x * 16;
    const int nks = R >> 2;
    const int cnt = nks > wave ? (nks - wave + 3) >> 2 : 0;        // this wave's K-steps: ks = wave + 4 i
    gu32* giveup = (gu32*)exch_;
    gu64* exch = (gu64*)exch_ + EXCH_HEADER_WORDS;
    const int ucol = min(j0 + j, R - 1);                           // B-operand column (columns >= R are discarded)
    float wr[KSW], wz[KSW], wn[KSW];
#pragma unroll
    for (int i = 0; i < KSW; ++i) {
        const int k = min(4 * (wave + 4 * i) + kq, R - 1);
        const float* w = whh + (long long)ucol * R + k;
        const bool ok = i < cnt;
        const float a = w[0], b = w[(long long)R * R], c = w[2ll * R * R];
        wr[i] = ok ? a : 0.f; wz[i] = ok ? b : 0.f; wn[i] = ok ? c : 0.f;
    }
    const int erow = 4 * kq + wave, eunit = j0 + j;
    const bool uok = eunit < R;
    const int euc = min(eunit, R - 1);
    const float cr = bhh[euc], cz = bhh[R + euc], cn = bhh[2 * R + euc];
    if (tid == 0) s_fail = 0;
    __syncthreads();
    bool dead = false;
    for (int st0 = blockIdx.y * RB; st0 < n_st && !dead; st0 += gridDim.y * RB) {
        const int ntl = min(RB, n_st - st0);                       // live tiles of this group (uniform)
        int ebc[RB], arow[RB];
        bool eok[RB], aok[RB];
        float hp[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int b0 = (st0 + rb) * 16, eb = b0 + erow;
            ebc[rb] = min(eb, B - 1);
            eok[rb] = uok && eb < B;
            arow[rb] = min(b0 + j, B - 1);
            aok[rb] = b0 + j < B;
            const float v = h0[(long long)ebc[rb] * R + euc];
            hp[rb] = eok[rb] ? v : 0.f;
        }
        for (int t = 0; t < T; ++t) {
            if (stamper) stamps[4 * t] = wall_clock64();
            float gir[RB], giz[RB], gin[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float* g = gi + ((long long)ebc[rb] * T + t) * 3 * R + euc;
                gir[rb] = g[0]; giz[rb] = g[R]; gin[rb] = g[2 * R];
            }
            float a[RB][KSW];
            if (t == 0) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int i = 0; i < KSW; ++i) {
                        const int k = min(4 * (wave + 4 * i) + kq, R - 1);
                        const float v = h0[(long long)arow[rb] * R + k];
                        a[rb][i] = aok[rb] ? v : 0.f;
                    }
            } else if (cnt > 0) {
                gu64* first = exch + ((long long)((t & 1) * n_st + st0) * R + 4 * wave + kq) * 16 + j;
                if (!sweep<KSW, RB>(first, 16 * 16, cnt, (long long)R * 16, ntl, (unsigned)t, a, giveup, spin_limit, lane)) s_fail = 1;
            }
            if (stamper) stamps[4 * t + 1] = wall_clock64();
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                floatx4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar;
#pragma unroll
                for (int i = 0; i < KSW; ++i) {
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][i], wr[i], ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][i], wz[i], az, 0, 0, 0);
                    an = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][i], wn[i], an, 0, 0, 0);
                }
                float(*rd)[3][4][64] = red[t & 1][rb];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    rd[wave][0][reg][lane] = ar[reg];
                    rd[wave][1][reg][lane] = az[reg];
                    rd[wave][2][reg][lane] = an[reg];
                }
            }
            __syncthreads();
            if (s_fail) { dead = true; break; }
            if (stamper) stamps[4 * t + 2] = wall_clock64();
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float(*rd)[3][4][64] = red[t & 1][rb];
                const float ghr = rd[0][0][wave][lane] + rd[1][0][wave][lane] + rd[2][0][wave][lane] + rd[3][0][wave][lane];
                const float ghz = rd[0][1][wave][lane] + rd[1][1][wave][lane] + rd[2][1][wave][lane] + rd[3][1][wave][lane];
                const float ghn = rd[0][2][wave][lane] + rd[1][2][wave][lane] + rd[2][2][wave][lane] + rd[3][2][wave][lane] + cn;
                const float r = sigmoidf_(gir[rb] + ghr + cr);
                const float u = sigmoidf_(giz[rb] + ghz + cz);
                const float c = tanhf(gin[rb] + r * ghn);
                const float hn = eok[rb] ? (1.0f - u) * c + u * hp[rb] : 0.f;
                if (uok && rb < ntl && t + 1 < T)   // rows past the batch publish zeros: consumers sweep all 16 rows
                    __hip_atomic_store(exch + ((long long)(((t + 1) & 1) * n_st + st0 + rb) * R + eunit) * 16 + erow,
                                       granule((unsigned)(t + 1), hn), RLX_AGENT);
                if (eok[rb]) {
                    const long long base = ((long long)(st0 + rb) * 16 + erow) * T + t;
                    hseq[base * R + eunit] = hn;
                    hprev[base * R + eunit] = hp[rb];
                    float* gs = gates + base * 4 * R;
                    gs[eunit] = r; gs[R + eunit] = u; gs[2 * R + eunit] = c; gs[3 * R + eunit] = ghn;
                }
                hp[rb] = hn;
            }
            if (stamper) stamps[4 * t + 3] = wall_clock64();
        }
        __syncthreads();   // the next sequence group starts over on red[0]
    }
}

// Backward recurrence, same partition: a workgroup owns dL/dh of its 16 units for RB sequence tiles.  Per step
// it turns that into the three gate gradients, publishes them (tag = step count from the end), gathers the full
// d(gh) row block [16 x 3R] of every tile and multiplies it with its 16 columns of W_hh.
template <int KSW, int RB>
__global__ __launch_bounds__(256) void gru_bwd_coop_kernel(const float* __restrict__ dhseq, const float* __restrict__ hprev,
                                                           const float* __restrict__ gates, const float* __restrict__ whh,
                                                           float* __restrict__ dgi, float* __restrict__ dgh,
                                                           float* __restrict__ dh0, unsigned long long* exch_,
                                                           int B, int T, int R, int n_st, unsigned long long spin_limit) {
    __shared__ float red[2][RB][4][4][64];
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int j0 = blockIdx.x * 16;
    const int G = 3 * R, nks = G >> 2;
    const int cnt = nks > wave ? (nks - wave + 3) >> 2 : 0;
    gu32* giveup = (gu32*)exch_;
    gu64* exch = (gu64*)exch_ + EXCH_HEADER_WORDS;
    const int ucol = min(j0 + j, R - 1);
    float wb[KSW];
#pragma unroll
    for (int i = 0; i < KSW; ++i) {
        const int k = min(4 * (wave + 4 * i) + kq, G - 1);
        const float v = whh[(long long)k * R + ucol];
        wb[i] = i < cnt ? v : 0.f;
    }
    const int erow = 4 * kq + wave, eunit = j0 + j;
    const bool uok = eunit < R;
    const int euc = min(eunit, R - 1);
    if (tid == 0) s_fail = 0;
    __syncthreads();
    bool dead = false;
    for (int st0 = blockIdx.y * RB; st0 < n_st && !dead; st0 += gridDim.y * RB) {
        const int ntl = min(RB, n_st - st0);
        int eb[RB], ebc[RB];
        bool eok[RB];
        float dh[RB], n_d[RB], n_hp[RB], n_r[RB], n_z[RB], n_c[RB], n_g[RB];
        // operands of a step are fetched one step ahead (plain loads of the forward pass's outputs)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            eb[rb] = (st0 + rb) * 16 + erow; ebc[rb] = min(eb[rb], B - 1);
            eok[rb] = uok && eb[rb] < B;
            dh[rb] = 0.f;
            const long long base = (long long)ebc[rb] * T + (T - 1);
            n_d[rb] = dhseq[base * R + euc]; n_hp[rb] = hprev[base * R + euc];
            n_r[rb] = gates[base * 4 * R + euc]; n_z[rb] = gates[base * 4 * R + R + euc];
            n_c[rb] = gates[base * 4 * R + 2 * R + euc]; n_g[rb] = gates[base * 4 * R + 3 * R + euc];
        }
        for (int t = T - 1, sidx = 0; t >= 0; --t, ++sidx) {
            float dz_[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float d = eok[rb] ? dh[rb] + n_d[rb] : 0.f;
                const float r = n_r[rb], z = n_z[rb], c = n_c[rb], ghn = n_g[rb], hp = n_hp[rb];
                const float dpc = d * (1.0f - z) * (1.0f - c * c);
                const float dpu = d * (hp - c) * z * (1.0f - z);
                const float dghn = dpc * r;
                const float dpr = dpc * ghn * r * (1.0f - r);
                dz_[rb] = d * z;
                gu64* slab = exch + (long long)((sidx & 1) * n_st + st0 + rb) * G * 16;
                if (uok && rb < ntl) {
                    const unsigned tag = (unsigned)(sidx + 1);
                    __hip_atomic_store(slab + (long long)eunit * 16 + erow, granule(tag, dpr), RLX_AGENT);
                    __hip_atomic_store(slab + (long long)(R + eunit) * 16 + erow, granule(tag, dpu), RLX_AGENT);
                    __hip_atomic_store(slab + (long long)(2 * R + eunit) * 16 + erow, granule(tag, dghn), RLX_AGENT);
                }
                if (eok[rb]) {
                    const long long bs = (long long)eb[rb] * T + t;
                    float* o1 = dgi + bs * G;
                    float* o2 = dgh + bs * G;
                    o1[eunit] = dpr; o1[R + eunit] = dpu; o1[2 * R + eunit] = dpc;
                    o2[eunit] = dpr; o2[R + eunit] = dpu; o2[2 * R + eunit] = dghn;
                }
                if (t > 0) {
                    const long long base = (long long)ebc[rb] * T + (t - 1);
                    n_d[rb] = dhseq[base * R + euc]; n_hp[rb] = hprev[base * R + euc];
                    n_r[rb] = gates[base * 4 * R + euc]; n_z[rb] = gates[base * 4 * R + R + euc];
                    n_c[rb] = gates[base * 4 * R + 2 * R + euc]; n_g[rb] = gates[base * 4 * R + 3 * R + euc];
                }
            }
            float a[RB][KSW];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int i = 0; i < KSW; ++i) a[rb][i] = 0.f;
            if (cnt > 0) {
                gu64* first = exch + (long long)((sidx & 1) * n_st + st0) * G * 16 + (long long)(4 * wave + kq) * 16 + j;
                if (!sweep<KSW, RB>(first, 16 * 16, cnt, (long long)G * 16, ntl, (unsigned)(sidx + 1), a, giveup, spin_limit, lane)) s_fail = 1;
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < KSW; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][i], wb[i], acc, 0, 0, 0);
                float(*rd)[4][64] = red[sidx & 1][rb];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) rd[wave][reg][lane] = acc[reg];
            }
            __syncthreads();
            if (s_fail) { dead = true; break; }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float(*rd)[4][64] = red[sidx & 1][rb];
                dh[rb] = rd[0][wave][lane] + rd[1][wave][lane] + rd[2][wave][lane] + rd[3][wave][lane] + dz_[rb];
            }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            if (eok[rb] && !dead) dh0[(long long)eb[rb] * R + eunit] = dh[rb];
        __syncthreads();
    }
}

// Sequence-group slots of a launch: one workgroup per CU with headroom, so that every group stays resident.
int coop_slots(int ntile, int n_groups) {
    int slots = 240 / ntile;
    if (slots < 1) slots = 1;
    return slots < n_groups ? slots : n_groups;
}
// Sequence tiles a workgroup steps together: the smallest of 1, 2, max_rb that brings all tiles of the batch
// into one round of resident workgroups (a round costs T cross-CU hops whatever a workgroup carries, and the
// spinning workgroups of every further round hold their CUs against the kernels of the other streams).
// Option gru_tiles = 1 / 2 / 4 forces.
int coop_tiles_per_group(int ntile, int n_st, int max_rb) {
    const int forced = opt(OPT_GRU_TILES);
    if (forced == 1 || forced == 2 || forced == 4) return forced < max_rb ? forced : max_rb;
    int slots = 240 / ntile;
    if (slots < 1) slots = 1;
    int rb = 1;
    while (rb < max_rb && (n_st + rb - 1) / rb > slots) rb *= 2;
    return rb;
}

}  // namespace

size_t gru_exchange_bytes(int B, int T, int R) {
    (void)T;
    const size_t n_st = (size_t)(B + ROWS - 1) / ROWS;
    return (size_t)EXCH_HEADER_WORDS * 8 + 2 * n_st * 3 * (size_t)R * ROWS * 8;
}

// Sequence tiles up to which the multi-CU kernels are the default.  Their workgroups each hold a compute unit's
// registers for the whole recurrence (most of the time waiting for a hand-off), so beyond a few rounds of
// resident groups they cost the kernels of the other streams more than the shorter recurrence gains: measured on
// the training step, the multi-CU kernels win up to 640 sequences, tie at 768 and lose 5 % at 1024, 6 % at 2048.
constexpr int COOP_MAX_TILES = 48;

static unsigned long long spin_limit() {
    const int v = opt(OPT_GRU_SPIN_TICKS);
    return v < 0 ? SPIN_LIMIT_TICKS : (unsigned long long)v;
}

// Would a recurrence of B sequences and R units run on the multi-CU kernels (given an exchange buffer)?
bool gru_multi_cu_selected(int B, int R) {
    const int forced = opt(OPT_GRU_MULTI_CU);
    const bool wanted = forced >= 0 ? forced == 1 : (B + ROWS - 1) / ROWS <= COOP_MAX_TILES;
    return wanted && R % 4 == 0 && R >= 16 && R <= 256;
}

static bool coop_ok(int R, const void* exch, size_t exch_bytes, int B, int T) {
    // option gru_multi_cu = 0 forces the single-CU kernels (e.g. on a GPU that other processes keep busy: the
    // multi-CU kernels need all workgroups of a sequence tile resident at the same time), 1 the multi-CU ones
    const int forced = opt(OPT_GRU_MULTI_CU);
    const bool wanted = forced >= 0 ? forced == 1 : (B + ROWS - 1) / ROWS <= COOP_MAX_TILES;
    return wanted && exch && R % 4 == 0 && R >= 16 && R <= 256 && (reinterpret_cast<uintptr_t>(exch) & 15) == 0 &&
           exch_bytes >= gru_exchange_bytes(B, T, R);
}

// Zero the exchange buffer of the multi-CU kernels ahead of time (gru_fwd / gru_bwd then take exch_clean = true
// and skip their own memset, which otherwise sits on the dependent chain right in front of the recurrence).
int gru_exchange_clear(hipStream_t s, void* exch, size_t exch_bytes, int B, int T, int R) {
    if (!exch || B == 0 || T == 0 || !coop_ok(R, exch, exch_bytes, B, T)) return 0;
    hipError_t e = hipMemsetAsync(exch, 0, gru_exchange_bytes(B, T, R), s);
    return e == hipSuccess ? 0 : (int)e;
}

int gru_fwd(hipStream_t s, const float* gi, const float* h0, const float* whh, const float* bhh, float* hseq,
            float* hprev, float* gates, int B, int T, int R, void* exch, size_t exch_bytes, bool exch_clean, int32_t* status) {
    if (B == 0 || T == 0) return 0;
    if (R % 4 || !aligned16(whh)) return PSVAE_E_ALIGN;
    const int R16 = (R + 15) & ~15, ntiles = R16 / 16;
    const bool coop = coop_ok(R, exch, exch_bytes, B, T);
    if (coop) {
        const int n_st = (B + ROWS - 1) / ROWS;
        const size_t used = (size_t)EXCH_HEADER_WORDS * 8 + 2 * (size_t)n_st * R * ROWS * 8;
        ProfScope prof(s, KC_GRU_FWD, 2.0 * B * T * 3 * R * R);
        if (!exch_clean) {
            hipError_t e = hipMemsetAsync(exch, 0, used, s);
            if (e != hipSuccess) return (int)e;
        }
        const bool narrow = (R >> 2) <= 52;     // wider rows keep more of W_hh per lane: registers for two tiles only
        const int rb = coop_tiles_per_group(ntiles, n_st, narrow ? 4 : 2);
        const dim3 grid(ntiles, coop_slots(ntiles, (n_st + rb - 1) / rb));
#define PSVAE_GRU_FWD(KSW, RB) hipLaunchKernelGGL((gru_fwd_coop_kernel<KSW, RB>), grid, dim3(256), 0, s, gi, h0, whh, bhh, hseq, hprev, \
                                                  gates, (unsigned long long*)exch, B, T, R, n_st, spin_limit())
        if (narrow) { if (rb == 4) PSVAE_GRU_FWD(13, 4); else if (rb == 2) PSVAE_GRU_FWD(13, 2); else PSVAE_GRU_FWD(13, 1); }
        else { if (rb == 2) PSVAE_GRU_FWD(16, 2); else PSVAE_GRU_FWD(16, 1); }
#undef PSVAE_GRU_FWD
        PSVAE_LAUNCH_CHECK();
    }
    const size_t lds = (size_t)2 * R16 * ROWS * sizeof(float);
    if (lds > 64 * 1024) return PSVAE_E_SHAPE;
    const int nwaves = ntiles < 16 ? ntiles : 16;
    // after a multi-CU launch: the same recurrence on the single-CU kernel, whose workgroups leave at once unless a
    // workgroup of the multi-CU kernel gave up waiting (the give-up word heads the exchange buffer); word 1 of
    // `status` counts the recurrences redone this way
    const unsigned int* run_if = coop ? (const unsigned int*)exch : nullptr;
    unsigned int* redone = coop && status ? (unsigned int*)status + 1 : nullptr;
    ProfScope prof(s, KC_GRU_FWD, coop ? 0.0 : 2.0 * B * T * 3 * R * R);
    hipLaunchKernelGGL(gru_fwd_kernel, dim3((B + ROWS - 1) / ROWS), dim3(64 * nwaves), lds, s, gi, h0, whh, bhh, hseq,
                       hprev, gates, B, T, R, run_if, redone);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

// The multi-CU forward recurrence with time stamps (see gru_fwd_coop_kernel<.., STAMP>): only the form the library
// takes for rows of up to 208 units and batches that step one sequence tile per workgroup.
int gru_fwd_stamped(hipStream_t s, const float* gi, const float* h0, const float* whh, const float* bhh, float* hseq,
                    float* hprev, float* gates, int B, int T, int R, void* exch, size_t exch_bytes, unsigned long long* stamps) {
    if (B <= 0 || T <= 0 || !stamps) return PSVAE_E_SHAPE;
    if (R % 4 || !aligned16(whh)) return PSVAE_E_ALIGN;
    const int ntiles = ((R + 15) & ~15) / 16, n_st = (B + ROWS - 1) / ROWS;
    if (!coop_ok(R, exch, exch_bytes, B, T) || (R >> 2) > 52 || coop_tiles_per_group(ntiles, n_st, 4) != 1) return PSVAE_E_SHAPE;
    hipError_t e = hipMemsetAsync(exch, 0, gru_exchange_bytes(B, T, R), s);
    if (e != hipSuccess) return (int)e;
    const dim3 grid(ntiles, coop_slots(ntiles, n_st));
    hipLaunchKernelGGL((gru_fwd_coop_kernel<13, 1, true>), grid, dim3(256), 0, s, gi, h0, whh, bhh, hseq, hprev, gates,
                       (unsigned long long*)exch, B, T, R, n_st, spin_limit(), stamps);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int gru_bwd(hipStream_t s, const float* dhseq, const float* hprev, const float* gates, const float* whh,
            float* dgi, float* dgh, float* dh0, int B, int T, int R, void* exch, size_t exch_bytes, bool exch_clean,
            int32_t* status) {
    if (B == 0) return 0;
    if (R % 4) return PSVAE_E_ALIGN;
    const int R16 = (R + 15) & ~15, G16 = (3 * R + 15) & ~15, ntiles = R16 / 16;
    const bool coop = T > 0 && coop_ok(R, exch, exch_bytes, B, T);
    if (coop) {
        const int n_st = (B + ROWS - 1) / ROWS;
        ProfScope prof(s, KC_GRU_BWD, 2.0 * B * T * 3 * R * R);
        if (!exch_clean) {
            hipError_t e = hipMemsetAsync(exch, 0, gru_exchange_bytes(B, T, R), s);
            if (e != hipSuccess) return (int)e;
        }
        const bool narrow = (3 * R >> 2) <= 152;
        const int rb = coop_tiles_per_group(ntiles, n_st, narrow ? 2 : 1);
        const dim3 grid(ntiles, coop_slots(ntiles, (n_st + rb - 1) / rb));
#define PSVAE_GRU_BWD(KSW, RB) hipLaunchKernelGGL((gru_bwd_coop_kernel<KSW, RB>), grid, dim3(256), 0, s, dhseq, hprev, gates, whh, dgi, \
                                                  dgh, dh0, (unsigned long long*)exch, B, T, R, n_st, spin_limit())
        if (narrow) { if (rb == 2) PSVAE_GRU_BWD(38, 2); else PSVAE_GRU_BWD(38, 1); }
        else PSVAE_GRU_BWD(48, 1);
#undef PSVAE_GRU_BWD
        PSVAE_LAUNCH_CHECK();
    }
    const size_t lds = (size_t)(ROWS * R16 + ROWS * G16) * sizeof(float);
    if (lds > 64 * 1024) return PSVAE_E_SHAPE;
    const int nwaves = ntiles < 16 ? ntiles : 16;
    const unsigned int* run_if = coop ? (const unsigned int*)exch : nullptr;      // the redo, as in gru_fwd
    unsigned int* redone = coop && status ? (unsigned int*)status + 1 : nullptr;
    ProfScope prof(s, KC_GRU_BWD, coop ? 0.0 : 2.0 * B * T * 3 * R * R);
    hipLaunchKernelGGL(gru_bwd_kernel, dim3((B + ROWS - 1) / ROWS), dim3(64 * nwaves), lds, s, dhseq, hprev, gates, whh,
                       dgi, dgh, dh0, B, T, R, run_if, redone);
    PSVAE_LAUNCH_CHECK();
    return 0;
}
