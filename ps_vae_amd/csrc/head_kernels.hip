// Small kernels around the latent: reparameterisation + KL, cross-entropy rows,
// MSE, loss finalisation, and the fused global-norm clip + Adam update.
// Loss denominators and upstream loss gradients are read from DEVICE memory
// (`denoms`, `gscale`) so that no launch depends on a host round trip.
#include "psvae_internal.h"

namespace {

// rsample (src/modules/vae_piece_decoder.py:56-63): one wave per graph, lanes over the latent
// dimensions; the KL terms are reduced across the wave with a fixed shuffle tree.
__global__ __launch_bounds__(256) void vae_head_fwd_kernel(const float* __restrict__ mulv, const float* __restrict__ eps,
                                                           float* __restrict__ z, float* __restrict__ klrow, int B, int Z) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const float* mu = mulv + (long long)b * 2 * Z;
    const float* raw = mu + Z;
    float s = 0.f;
    for (int k = lane; k < Z; k += 64) {
        const float m = mu[k], lv = -fabsf(raw[k]);
        s += 1.0f + lv - m * m - expf(lv);
        z[(long long)b * Z + k] = m + expf(lv * 0.5f) * eps[(long long)b * Z + k];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) klrow[b] = s;
}

// d(mean | raw log-var) from dz (the sum of up to three partial gradients, added in a fixed
// order) and the KL term; gscale[3] = dL/dkl, denoms[2] = B.
__global__ void vae_head_bwd_kernel(const float* __restrict__ mulv, const float* __restrict__ eps, const float* __restrict__ dz0,
                                    const float* __restrict__ dz1, const float* __restrict__ dz2,
                                    const GScale gscale, const float* __restrict__ denoms,
                                    float* __restrict__ dmulv, int B, int Z) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * Z) return;
    const int b = t / Z, k = t - b * Z;
    const float gk = gscale_at(gscale, 3) / denoms[2];
    const float m = mulv[(long long)b * 2 * Z + k], raw = mulv[(long long)b * 2 * Z + Z + k];
    float d = dz0[t];
    if (dz1) d += dz1[t];
    if (dz2) d += dz2[t];
    const float lv = -fabsf(raw);
    const float dmu = d + gk * m;
    const float dlv = d * eps[t] * 0.5f * expf(lv * 0.5f) + gk * (-0.5f) * (1.0f - expf(lv));
    const float sgn = raw > 0.f ? 1.f : (raw < 0.f ? -1.f : 0.f);
    dmulv[(long long)b * 2 * Z + k] = dmu;
    dmulv[(long long)b * 2 * Z + Z + k] = -sgn * dlv;
}

__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int ce_target(const int32_t* target, int tstride, int toffset, int rpg, int r) {
    return target[(long long)(r / rpg) * tstride + toffset + (r % rpg)];
}

// One wave per row: lse and the row's loss (0 for ignored rows).
__global__ __launch_bounds__(256) void softmax_ce_fwd_kernel(const float* __restrict__ logits, int ld,
                                                             const int32_t* __restrict__ target, int tstride, int toffset,
                                                             int rpg, int rows, int classes, int ignore_index,
                                                             float* __restrict__ row_loss, float* __restrict__ lse_out) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* x = logits + (long long)r * ld;
    float mx = -INFINITY;
    for (int c = lane; c < classes; c += 64) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < classes; c += 64) s += expf(x[c] - mx);
    s = wave_sum(s);
    const float lse = mx + logf(s);
    if (lane == 0) {
        const int tg = ce_target(target, tstride, toffset, rpg, r);
        lse_out[r] = lse;
        row_loss[r] = (tg == ignore_index) ? 0.f : lse - x[tg];
    }
}

// In place: logits <- (softmax - onehot) * gscale[gidx] / denoms[didx]  (0 for ignored rows)
__global__ __launch_bounds__(256) void softmax_ce_bwd_kernel(float* __restrict__ logits, int ld,
                                                             const int32_t* __restrict__ target, int tstride, int toffset,
                                                             int rpg, int rows, int classes, int ignore_index,
                                                             const float* __restrict__ lse, const GScale gscale,
                                                             int gidx, const float* __restrict__ denoms, int didx) {
    const long long total = (long long)rows * classes;
    const float scale = gscale_at(gscale, gidx) / denoms[didx];
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(t / classes), c = (int)(t - (long long)r * classes);
        const int tg = ce_target(target, tstride, toffset, rpg, r);
        float* p = logits + (long long)r * ld + c;
        *p = (tg == ignore_index) ? 0.f : (expf(*p - lse[r]) - (c == tg ? 1.f : 0.f)) * scale;
    }
}

// Both of the above in one launch, for a caller that knows the upstream gradient when the losses are formed (the
// training step as one call): a wave keeps its row in registers (NC values per lane: up to 64 * NC classes), does
// the forward kernel's arithmetic in the forward kernel's order, then writes the backward kernel's values over the
// row.  Same bits as the two launches.
template <int NC>
__global__ __launch_bounds__(256) void softmax_ce_fwd_bwd_kernel(float* __restrict__ logits, int ld,
                                                                 const int32_t* __restrict__ target, int tstride, int toffset,
                                                                 int rpg, int rows, int classes, int ignore_index,
                                                                 float* __restrict__ row_loss, float* __restrict__ lse_out,
                                                                 const GScale gscale, int gidx, const float* __restrict__ denoms, int didx) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    float* x = logits + (long long)r * ld;
    float v[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) v[j] = lane + 64 * j < classes ? x[lane + 64 * j] : -INFINITY;
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NC; ++j) if (lane + 64 * j < classes) mx = fmaxf(mx, v[j]);
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) if (lane + 64 * j < classes) s += expf(v[j] - mx);
    s = wave_sum(s);
    const float lse = mx + logf(s);
    const int tg = ce_target(target, tstride, toffset, rpg, r);
    float xt = 0.f;       // the row's value at the target class, from the lane that holds it
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const float t = __shfl(v[j], tg & 63, 64);
        if ((tg >> 6) == j) xt = t;
    }
    if (lane == 0) {
        lse_out[r] = lse;
        row_loss[r] = (tg == ignore_index) ? 0.f : lse - xt;
    }
    const float scale = gscale_at(gscale, gidx) / denoms[didx];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = lane + 64 * j;
        if (c < classes) x[c] = (tg == ignore_index) ? 0.f : (expf(v[j] - lse) - (c == tg ? 1.f : 0.f)) * scale;
    }
}

// Last layer of the link scorer (Linear(P -> 4), src/modules/vae_piece_decoder.py:42-46,110) together with its
// cross-entropy (:115): with four outputs per row the product is a handful of dot products, not a tile problem.
// A wave keeps its share of the four weight rows in registers (NCH float4 per row and lane) and walks over rows
// of x: forward = 4 dot products, bias, log-sum-exp and the row's loss; backward = softmax gradient (left in
// `logits` for the weight gradient) and dx = dlogits W.  One launch each instead of product + slab sum + loss
// (forward) and loss gradient + product (backward).
__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }

template <int NCH>
__device__ __forceinline__ void load_w4(const float* __restrict__ w, int P4, int lane, float4 (&wr)[4][NCH]) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int idx = lane + 64 * j;
            float4 v = reinterpret_cast<const float4*>(w + (long long)c * 4 * P4)[min(idx, P4 - 1)];   // clamped, unconditional
            if (idx >= P4) v = make_float4(0.f, 0.f, 0.f, 0.f);
            wr[c][j] = v;
        }
}

template <int NCH>
__global__ __launch_bounds__(256) void pair_head_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const int32_t* __restrict__ gold,
                                                            float* __restrict__ logits, float* __restrict__ row_loss,
                                                            float* __restrict__ lse_out, int S, int P4) {
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    float4 wr[4][NCH];
    load_w4<NCH>(w, P4, lane, wr);
    const float b0 = bias[0], b1 = bias[1], b2 = bias[2], b3 = bias[3];
    // rows in groups of RB: all their loads go out before the first reduction (one memory round trip per group)
    constexpr int RB = 4;
    for (int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RB; r0 < S; r0 += nw * RB) {
        float4 xv[RB][NCH];
        int tgs[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int r = min(r0 + i, S - 1);
            const float4* xr = reinterpret_cast<const float4*>(x + (long long)r * ldx);
#pragma unroll
            for (int j = 0; j < NCH; ++j) xv[i][j] = xr[min(lane + 64 * j, P4 - 1)];      // surplus chunks meet zero weights
            tgs[i] = gold[r];
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                a0 += dot4(xv[i][j], wr[0][j]); a1 += dot4(xv[i][j], wr[1][j]); a2 += dot4(xv[i][j], wr[2][j]); a3 += dot4(xv[i][j], wr[3][j]);
            }
            a0 = wave_sum(a0) + b0; a1 = wave_sum(a1) + b1; a2 = wave_sum(a2) + b2; a3 = wave_sum(a3) + b3;
            const float mx = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
            const float lse = mx + logf(((expf(a0 - mx) + expf(a1 - mx)) + expf(a2 - mx)) + expf(a3 - mx));
            const int r = r0 + i, tg = tgs[i];
            if (lane == 0 && r < S) {
                reinterpret_cast<float4*>(logits)[r] = make_float4(a0, a1, a2, a3);
                lse_out[r] = lse;
                row_loss[r] = lse - (tg == 0 ? a0 : (tg == 1 ? a1 : (tg == 2 ? a2 : a3)));
            }
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void pair_head_bwd_kernel(float* __restrict__ logits, const float* __restrict__ lse,
                                                            const int32_t* __restrict__ gold, const float* __restrict__ w,
                                                            const GScale gscale, const float* __restrict__ denoms,
                                                            float* __restrict__ dx, int lddx, int S, int P4) {
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    float4 wr[4][NCH];
    load_w4<NCH>(w, P4, lane, wr);
    const float scale = gscale_at(gscale, 2) / denoms[1];
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < S; r += nw) {
        const float4 lg = reinterpret_cast<const float4*>(logits)[r];
        const float l = lse[r];
        const int tg = gold[r];
        const float d0 = (expf(lg.x - l) - (tg == 0 ? 1.f : 0.f)) * scale, d1 = (expf(lg.y - l) - (tg == 1 ? 1.f : 0.f)) * scale;
        const float d2 = (expf(lg.z - l) - (tg == 2 ? 1.f : 0.f)) * scale, d3 = (expf(lg.w - l) - (tg == 3 ? 1.f : 0.f)) * scale;
        if (lane == 0) reinterpret_cast<float4*>(logits)[r] = make_float4(d0, d1, d2, d3);
        float4* out = reinterpret_cast<float4*>(dx + (long long)r * lddx);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int idx = lane + 64 * j;
            if (idx < P4) {
                float4 v;
                v.x = fmaf(d3, wr[3][j].x, fmaf(d2, wr[2][j].x, fmaf(d1, wr[1][j].x, d0 * wr[0][j].x)));
                v.y = fmaf(d3, wr[3][j].y, fmaf(d2, wr[2][j].y, fmaf(d1, wr[1][j].y, d0 * wr[0][j].y)));
                v.z = fmaf(d3, wr[3][j].z, fmaf(d2, wr[2][j].z, fmaf(d1, wr[1][j].z, d0 * wr[0][j].z)));
                v.w = fmaf(d3, wr[3][j].w, fmaf(d2, wr[2][j].w, fmaf(d1, wr[1][j].w, d0 * wr[0][j].w)));
                out[idx] = v;
            }
        }
    }
}

__global__ void mse_fwd_kernel(const float* __restrict__ pred, int ldp, const float* __restrict__ gold,
                               float* __restrict__ row, int B, int P) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int k = 0; k < P; ++k) {
        const float d = pred[(long long)b * ldp + k] - gold[(long long)b * P + k];
        s += d * d;
    }
    row[b] = s;
}
__global__ void mse_bwd_kernel(const float* __restrict__ pred, int ldp, const float* __restrict__ gold,
                               const GScale gscale, const float* __restrict__ denoms,
                               float* __restrict__ dpred, int B, int P) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * P) return;
    const int b = t / P, k = t - b * P;
    dpred[(long long)b * ldp + k] = 2.0f * (pred[(long long)b * ldp + k] - gold[t]) * gscale_at(gscale, 0) / denoms[3];
}

// losses = [pred, rec, piece, edge, kl]
__global__ __launch_bounds__(256) void losses_finalize_kernel(const float* mse_row, int B, const float* piece_row, int rows_p,
                                                              const float* edge_row, int S, const float* klrow,
                                                              const float* denoms, float w_pred, float w_rec, float w_kl,
                                                              float* losses) {
    // four block-wide sums in a fixed order (thread-strided partials, then an LDS tree), with the loads of
    // all four in flight together and one tree for the four of them: this kernel sits between forward and
    // backward on the caller's stream, so its latency is paid in full on every step
    __shared__ float sh[4][256];
    const int tid = threadIdx.x;
    const int nmax = max(max(B, rows_p), S);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    // eight strides per round, from clamped addresses: 32 loads in flight instead of one round trip per stride
    // (a thread still adds its elements in index order, so the sums are what the one-by-one loop gave)
    for (int i0 = tid; i0 < nmax; i0 += 8 * 256) {
        float a0[8], a1[8], a2[8], a3[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            a0[u] = mse_row[max(0, min(i, B - 1))]; a1[u] = piece_row[max(0, min(i, rows_p - 1))];
            a2[u] = edge_row[max(0, min(i, S - 1))]; a3[u] = klrow[max(0, min(i, B - 1))];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            s0 += i < B ? a0[u] : 0.f; s1 += i < rows_p ? a1[u] : 0.f; s2 += i < S ? a2[u] : 0.f; s3 += i < B ? a3[u] : 0.f;
        }
    }
    sh[0][tid] = s0; sh[1][tid] = s1; sh[2][tid] = s2; sh[3][tid] = s3;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; sh[2][tid] += sh[2][tid + o]; sh[3][tid] += sh[3][tid + o]; }
        __syncthreads();
    }
    const float mse = sh[0][0], piece = sh[1][0], edge = sh[2][0], kl = sh[3][0];
    if (threadIdx.x == 0) {
        const float pl = piece / denoms[0], el = edge / denoms[1];
        losses[0] = mse / denoms[3];
        losses[1] = pl + el;
        losses[2] = pl;
        losses[3] = el;
        losses[4] = -0.5f * kl / denoms[2];
        losses[5] = w_rec * losses[1] + w_pred * losses[0] + w_kl * losses[4];
    }
}

// ---- clip_grad_norm_ + Adam ------------------------------------------------
constexpr int NORM_BLOCKS = 512;
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long long count, float* __restrict__ partials) {
    __shared__ float sh[256];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) s += g[i] * g[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}
// Every block re-derives the clip coefficient from the same partials in the same
// order, so all blocks agree bit for bit.  torch: coef = min(1, max_norm / (norm + 1e-6)).
// The constants arrive as torch forms them: 1 - beta, lr / bias1 and sqrt(bias2) are worked out in double on the host
// and rounded once (1.f - 0.999f is 1.3e-5 off 0.001, which would scale every update by 6e-6).
__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long long count, const float* __restrict__ partials,
                                                        float max_norm, float omb1, float b2, float omb2, float eps,
                                                        float step, float sqrt_b2, float* __restrict__ norm_out) {
    __shared__ float sh[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < NORM_BLOCKS; i += 256) s += partials[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    const float norm = sqrtf(sh[0]);
    if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = norm;
    float coef = max_norm / (norm + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * coef, m0 = m[i];
        const float mi = m0 + omb1 * (gi - m0);                     // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + (omb2 * gi) * gi;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        m[i] = mi; v[i] = vi; g[i] = gi;
        p[i] += (-step * mi) / (sqrtf(vi) / sqrt_b2 + eps);         // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
}

// ---- Linear layers over a handful of rows (one row per graph) ------------------------------------------------
// y[m][n] = b[n] + sum_k x[m][k] w[n][k] with M*N of a few ten thousand and K of 56..400: as a tiled MFMA product
// this is 8-12 workgroups (or a split reduction plus its slab sum) whose cost is launch, prologue and epilogue,
// 10-16 us on the dependent chain of the step.  Here one wave takes one output: lanes split k (16-byte coalesced
// loads of both rows), partial sums meet in a fixed butterfly order, so results are reproducible.
__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                               const float* __restrict__ b, float* __restrict__ y, int ldy,
                                                               int M, int N, int K4) {
    const int lane = threadIdx.x & 63;
    const long long out = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (out >= (long long)M * N) return;       // wave-uniform
    const int m = (int)(out / N), n = (int)(out - (long long)m * N);
    const float4* xr = reinterpret_cast<const float4*>(x + (long long)m * ldx);
    const float4* wr = reinterpret_cast<const float4*>(w + (long long)n * (K4 * 4));
    float acc = 0.f;
    for (int k = lane; k < K4; k += 64) {
        const float4 a = xr[k], c = wr[k];
        acc = fmaf(a.x, c.x, acc); acc = fmaf(a.y, c.y, acc); acc = fmaf(a.z, c.z, acc); acc = fmaf(a.w, c.w, acc);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) y[(long long)m * ldy + n] = acc + (b ? b[n] : 0.f);
}

// The same for a short reduction (K <= 64, e.g. latent_to_rnn_hidden with K = 56): a wave per output would leave
// most lanes idle, so one THREAD takes one output and walks its K in order (x[m] is a broadcast read, the rows
// of w are 16-byte loads from a matrix that sits in L1/L2).
template <int K4MAX>
__global__ __launch_bounds__(256) void small_linear_fwd_shortk_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                                      const float* __restrict__ b, float* __restrict__ y, int ldy,
                                                                      int M, int N, int K4) {
    const long long out = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool on = out < (long long)M * N;
    const long long o = on ? out : 0;
    const int m = (int)(o / N), n = (int)(o - (long long)m * N);
    const float4* xr = reinterpret_cast<const float4*>(x + (long long)m * ldx);
    const float4* wr = reinterpret_cast<const float4*>(w + (long long)n * (K4 * 4));
    float4 c[K4MAX], a[K4MAX];
#pragma unroll
    for (int k = 0; k < K4MAX; ++k) { c[k] = wr[min(k, K4 - 1)]; a[k] = xr[min(k, K4 - 1)]; }     // all in flight at once
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < K4MAX; ++k) {
        if (k < K4) {
            acc = fmaf(a[k].x, c[k].x, acc); acc = fmaf(a[k].y, c[k].y, acc); acc = fmaf(a[k].z, c[k].z, acc); acc = fmaf(a[k].w, c[k].w, acc);
        }
    }
    if (on) y[(long long)m * ldy + n] = acc + (b ? b[n] : 0.f);
}

// dx[m][k] = sum_n dy[m][n] w[n][k]: one block per row m; thread (slice q, column k) sums a quarter of the n range
// in order with 32 rows of w (and the matching dy values) in flight, the four slices are added in slice order through LDS.
__global__ __launch_bounds__(256) void small_linear_dx_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ w,
                                                              float* __restrict__ dx, int lddx, int M, int N, int K) {
    __shared__ float part[4][64];
    const int m = blockIdx.x, q = threadIdx.x >> 6, kk = threadIdx.x & 63, k = blockIdx.y * 64 + kk;
    const int per = (N + 3) / 4, n0 = min(N, q * per), n1 = min(N, n0 + per);
    const float* dr = dy + (long long)m * lddy;
    float acc = 0.f;
    if (k < K) {
        for (int n = n0; n < n1; n += 32) {
            float wv[32], dv[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) { wv[u] = w[(long long)min(n + u, n1 - 1) * K + k]; dv[u] = dr[min(n + u, n1 - 1)]; }
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = n + u < n1 ? fmaf(dv[u], wv[u], acc) : acc;
        }
    }
    part[q][kk] = acc;
    __syncthreads();
    if (q == 0 && k < K) dx[(long long)m * lddx + k] = ((part[0][kk] + part[1][kk]) + part[2][kk]) + part[3][kk];
}

// W_mean / W_log_var (one [2Z, G] matrix) and rsample in one launch: a block of 16 waves per graph.  Each wave
// takes every sixteenth output of the 2Z, eight of them per round with all their weight loads in flight at once
// (lanes split the G inputs, same order of additions as small_linear_fwd_kernel); the results meet in LDS, then
// the first wave does what vae_head_fwd_kernel does.  Saves a dependent launch.
// `sl.nz > 0`: the graph's row of cond does not exist yet - the product that makes it (encoder.linear) was split along
// its reduction and left its slabs behind; the block first adds them up in slab order, adds cond_bias, writes the row
// to cond (backward reads it) and goes on from its copy in LDS: the slab-sum launch between the two falls away.
template <int NCH>
__global__ __launch_bounds__(1024) void latent_head_fwd_kernel(const float* __restrict__ cond, int ldc, const float* __restrict__ w,
                                                               const float* __restrict__ bias, const float* __restrict__ eps,
                                                               float* __restrict__ mulv, float* __restrict__ z,
                                                               float* __restrict__ klrow, int Z, int G4, const GemmSlabs sl,
                                                               const float* __restrict__ cond_bias, float* __restrict__ cond_out) {
    __shared__ float sm[1024];
    __shared__ __attribute__((aligned(16))) float srow[1024];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, N = 2 * Z;
    if (sl.nz > 0) {       // uniform over the launch
        const int k = threadIdx.x;
        if (k < G4 * 4) {
            const float* s = sl.slabs + (long long)b * sl.ld + k;
            float v = 0.f;
            int zz = 0;
            for (; zz + 4 <= sl.nz; zz += 4) {      // the order of gemm_reduce_kernel
                const float v0 = s[(long long)zz * sl.stride], v1 = s[(long long)(zz + 1) * sl.stride];
                const float v2 = s[(long long)(zz + 2) * sl.stride], v3 = s[(long long)(zz + 3) * sl.stride];
                v = (((v + v0) + v1) + v2) + v3;
            }
            for (; zz < sl.nz; ++zz) v += s[(long long)zz * sl.stride];
            v += cond_bias[k];
            srow[k] = v;
            cond_out[(long long)b * ldc + k] = v;
        }
        __syncthreads();
    }
    const float4* xr = sl.nz > 0 ? reinterpret_cast<const float4*>(srow) : reinterpret_cast<const float4*>(cond + (long long)b * ldc);
    float4 xv[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        xv[j] = xr[min(lane + 64 * j, G4 - 1)];
        if (lane + 64 * j >= G4) xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int o0 = wave; o0 < N; o0 += 16 * 8) {
        float4 wv[8][NCH];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NCH; ++j)
                wv[u][j] = reinterpret_cast<const float4*>(w + (long long)min(o0 + 16 * u, N - 1) * (G4 * 4))[min(lane + 64 * j, G4 - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                acc = fmaf(xv[j].x, wv[u][j].x, acc); acc = fmaf(xv[j].y, wv[u][j].y, acc);
                acc = fmaf(xv[j].z, wv[u][j].z, acc); acc = fmaf(xv[j].w, wv[u][j].w, acc);
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
            const int o = o0 + 16 * u;
            if (lane == 0 && o < N) sm[o] = acc + bias[o];
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < N; o += 1024) mulv[(long long)b * N + o] = sm[o];
    if (wave == 0) {
        float sum = 0.f;
        for (int k = lane; k < Z; k += 64) {
            const float m = sm[k], lv = -fabsf(sm[Z + k]);
            sum += 1.0f + lv - m * m - expf(lv);
            z[(long long)b * Z + k] = m + expf(lv * 0.5f) * eps[(long long)b * Z + k];
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0) klrow[b] = sum;
    }
}

// vae_head_bwd_kernel and the input gradient of [W_mean; W_log_var] in one launch: block (graph, 64 columns of
// d_cond).  Every block forms the graph's 2Z values of d_mulv in LDS (block column 0 also stores them for the
// weight gradient), then sums its columns as small_linear_dx_kernel does.
__global__ __launch_bounds__(256) void latent_head_bwd_kernel(const float* __restrict__ mulv, const float* __restrict__ eps,
                                                              const float* __restrict__ dz0, const float* __restrict__ dz1,
                                                              const float* __restrict__ dz2, const GScale gscale,
                                                              const float* __restrict__ denoms, const float* __restrict__ w,
                                                              float* __restrict__ dmulv, float* __restrict__ dcond, int ldd,
                                                              int Z, int G) {
    __shared__ float sd[1024];
    __shared__ float part[4][64];
    const int b = blockIdx.x, N = 2 * Z;
    // this thread's column of the weights (first 32 rows of its slice) is requested before anything else: it
    // does not depend on d_mulv, so its latency overlaps the loads and the arithmetic of the first phase
    const int q = threadIdx.x >> 6, kk = threadIdx.x & 63, k = blockIdx.y * 64 + kk, kc = min(k, G - 1);
    const int per = (N + 3) / 4, n0 = min(N, q * per), n1 = min(N, n0 + per);
    float wpre[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) wpre[u] = w[(long long)min(n0 + u, N - 1) * G + kc];
    for (int k = threadIdx.x; k < Z; k += 256) {
        const long long t = (long long)b * Z + k;
        const float gk = gscale_at(gscale, 3) / denoms[2];
        const float m = mulv[(long long)b * N + k], raw = mulv[(long long)b * N + Z + k];
        float d = dz0[t];
        if (dz1) d += dz1[t];
        if (dz2) d += dz2[t];
        const float lv = -fabsf(raw);
        const float dmu = d + gk * m;
        const float dlv = d * eps[t] * 0.5f * expf(lv * 0.5f) + gk * (-0.5f) * (1.0f - expf(lv));
        const float sgn = raw > 0.f ? 1.f : (raw < 0.f ? -1.f : 0.f);
        sd[k] = dmu; sd[Z + k] = -sgn * dlv;
        if (blockIdx.y == 0) { dmulv[(long long)b * N + k] = dmu; dmulv[(long long)b * N + Z + k] = -sgn * dlv; }
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 32; ++u) acc = n0 + u < n1 ? fmaf(sd[min(n0 + u, N - 1)], wpre[u], acc) : acc;
    for (int n = n0 + 32; n < n1; n += 8) {          // slices longer than 32 rows (2Z > 128)
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = w[(long long)min(n + u, n1 - 1) * G + kc];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = n + u < n1 ? fmaf(sd[n + u], wv[u], acc) : acc;
    }
    part[q][kk] = acc;
    __syncthreads();
    if (q == 0 && k < G) dcond[(long long)b * ldd + k] = ((part[0][kk] + part[1][kk]) + part[2][kk]) + part[3][kk];
}

}  // namespace

int small_linear_fwd(hipStream_t s, const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int N, int K) {
    if (M <= 0 || N <= 0) return 0;
    if (K % 4 || ldx % 4 || !aligned16(x) || !aligned16(w)) return PSVAE_E_ALIGN;
    ProfScope prof(s, KC_HEAD, 2.0 * M * N * K);
    const long long outs = (long long)M * N;
    if (K <= 64)
        hipLaunchKernelGGL(small_linear_fwd_shortk_kernel<16>, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, s, x, ldx, w, b, y, ldy, M, N, K / 4);
    else
        hipLaunchKernelGGL(small_linear_fwd_kernel, dim3((unsigned)((outs + 3) / 4)), dim3(256), 0, s, x, ldx, w, b, y, ldy, M, N, K / 4);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

int small_linear_bwd_x(hipStream_t s, const float* dy, int lddy, const float* w, float* dx, int lddx, int M, int N, int K) {
    if (M <= 0 || K <= 0) return 0;
    ProfScope prof(s, KC_HEAD, 2.0 * M * N * K);
    hipLaunchKernelGGL(small_linear_dx_kernel, dim3(M, (K + 63) / 64), dim3(256), 0, s, dy, lddy, w, dx, lddx, M, N, K);
    PSVAE_LAUNCH_CHECK();
    return 0;
}

bool latent_head_supported(const float* cond, int ldc, const float* w, int Z, int G) {
    return Z > 0 && 2 * Z <= 1024 && G > 0 && G % 4 == 0 && G <= 1024 && ldc % 4 == 0 && aligned16(cond) && aligned16(w);
}
int latent_head_fwd(hipStream_t s, const float* cond, int ldc, const float* w, const float* bias, const float* eps,
                    float* mulv, float* z, float* klrow, int B, int Z, int G) {
    if (B == 0) return 0;
    if (!latent_head_supported(cond, ldc, w, Z, G)) return PSVAE_E_ALIGN;
    ProfScope prof(s, KC_HEAD, 2.0 * B * 2 * Z * G);
    const int G4 = G / 4;
    const GemmSlabs none{nullptr, 0, 0, 0};
    if (G4 <= 128) hipLaunchKernelGGL(latent_head_fwd_kernel<2>, dim3(B), dim3(1024), 0, s, cond, ldc, w, bias, eps, mulv, z, klrow, Z, G4, none, nullptr, nullptr);
    else hipLaunchKernelGGL(latent_head_fwd_kernel<4>, dim3(B), dim3(1024), 0, s, cond, ldc, w, bias, eps, mulv, z, klrow, Z, G4, none, nullptr, nullptr);
    PSVAE_LAUNCH_CHECK();
    return 0;
}
// The same with the rows of cond still lying in the slabs of the split product that makes them (see the kernel).
int latent_head_fwd_slabs(hipStream_t s, const GemmSlabs& sl, const float* cond_bias, float* cond, int ldc, const float* w,
                          const float* bias, const float* eps, float* mulv, float* z, float* klrow, int B, int Z, int G) {
    if (B == 0) return 0;
    if (!latent_head_supported(cond, ldc, w, Z, G) || sl.nz < 1 || !sl.slabs || !cond_bias) return PSVAE_E_ALIGN;
    ProfScope prof(s, KC_HEAD, 2.0 * B * 2 * Z * G);
    const int G4 = G / 4;
    if (G4 <= 128) hipLaunchKernelGGL(latent_head_fwd_kernel<2>, dim3(B), dim3(1024), 0, s, cond, ldc, w, bias, eps, mulv, z, klrow, Z, G4, sl, cond_bias, cond);
    else hipLaunchKernelGGL(latent_head_fwd_kernel<4>, dim3(B), dim3(1024), 0, s, cond, ldc, w, bias, eps, mulv, z, klrow, Z, G4, sl, cond_bias, cond);
    PSVAE_LAUNCH_CHECK();
    return 0;
}
int latent_head_bwd(hipStream_t s, const float* mulv, const float* eps, const float* dz0, const float* dz1, const float* dz2,
                    GScale gscale, const float* denoms, const float* w, float* dmulv, float* dcond, int ldd, int B, int Z, int G) {
    if (B == 0) return 0;
    if (2 * Z > 1024) return PSVAE_E_SHAPE;
    ProfScope prof(s, KC_HEAD, 2.0 * B * 2 * Z * G);
    hipLaunchKernelGGL(latent_head_bwd_kernel, dim3(B, (G + 63) / 64), dim3(256), 0, s, mulv, eps, dz0, dz1, dz2, gscale, denoms, w,
                       dmulv, dcond, ldd, Z, G);
    PSVAE_LAUNCH_CHECK();
    return 0;
}
int vae_head_fwd(hipStream_t s, const float* mulv, const float* eps, float* z, float* klrow, int B, int Z) {
    if (B == 0) return 0;
    { ProfScope prof(s, KC_HEAD, 0);
    hipLaunchKernelGGL(vae_head_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, s, mulv, eps, z, klrow, B, Z);
    }
    PSVAE_LAUNCH_CHECK();
    return 0;
}
int vae_head_bwd(hipStream_t s, const float* mulv, const float* eps, const float* dz0, const float* dz1, const float* dz2,
                 GScale gscale,
                 const float* denoms, float* dmulv, int B, int Z) {
    if (B == 0) return 0;
    { ProfScope prof(s, KC_HEAD, 0);
    hipLaunchKernelGGL(vae_head_bwd_kernel, dim3((B * Z + 255) / 256), dim3(256), 0, s, mulv, eps, dz0, dz1, dz2, gscale, denoms, dmulv, B, Z);
    }
    PSVAE_LAUNCH_CHECK();
    return 0;
}
int softmax_ce_fwd(hipStream_t s, const float* logits, int ld, const int32_t* target, int tstride, int toffset,
                   int rows_per_group, int rows, int classes, int ignore_index, float* row_loss, float* lse) {
    if (rows == 0) return 0;
    if (rows_per_group < 1) return PSVAE_E_SHAPE;
    { ProfScope prof(s, KC_CE, 4.0 * rows * classes);
    hipLaunchKernelGGL(softmax_ce_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, logits, ld, target, tstride, toffset,
                       rows_per_group, rows, classes, ignore_index, row_loss, lse);
    }
    PSVAE_LAUNCH_CHECK();
    return 0;
}
bool softmax_ce_fused_supported(int classes) { return classes >= 1 && classes <= 512; }
int softmax_ce_fwd_bwd(hipStream_t s, float* logits_inout, int ld, const int32_t* target, int tstride, int toffset,
                       int rows_per_group, int rows, int classes, int ignore_index, float* row_loss, float* lse,
                       GScale gscale, int gidx, const float* denoms, int didx) {
    if (rows == 0) return 0;
    if (!softmax_ce_fused_supported(classes)) return PSVAE_E_SHAPE;
    { ProfScope prof(s, KC_CE, 12.0 * rows * classes);
    if (classes <= 320)
        hipLaunchKernelGGL(softmax_ce_fwd_bwd_kernel<5>, dim3((rows + 3) / 4), dim3(256), 0, s, logits_inout, ld, target, tstride, toffset,
                           rows_per_group, rows, classes, ignore_index, row_loss, lse, gscale, gidx, denoms, didx);
    else
        hipLaunchKernelGGL(softmax_ce_fwd_bwd_kernel<8>, dim3((rows + 3) / 4), dim3(256), 0, s, logits_inout, ld, target, tstride, toffset,
                           rows_per_group, rows, classes, ignore_index, row_loss, lse, gscale, gidx, denoms, didx);
    }
    PSVAE_LAUNCH_CHECK();
    return 0;
}
int softmax_ce_bwd(hipStream_t s, float* logits_inout, int ld, const int32_t* target, int tstride, int toffset,
                   int rows_per_group, int rows, int classes, int ignore_index, const float* lse,
                   GScale gscale, int gidx, const float* denoms, int didx) {
    if (rows == 0) return 0;
    long long total = (long long)rows * classes;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    { ProfScope prof(s, KC_CE, 8.0 * rows * classes);
    hipLaunchKernelGGL(softmax_ce_bwd_kernel, dim3(blocks), dim3(256), 0, s, logits_inout, ld, target, tstride, toffset,
                       rows_per_group, rows, classes, ignore_index, lse, gscale, gidx, denoms, didx);
    }
    PSVAE_LAUNCH_CHECK();
    return 0;
}
bool pair_head_supported(const float* x, int ldx, const float* w, const float* dx, int lddx, int P) {
    return P > 0 && P % 4 == 0 && P <= 5 * 256 && ldx % 4 == 0 && lddx % 4 == 0 && aligned16(x) && aligned16(w) && (!dx || aligned16(dx));
}
int pair_head_fwd(hipStream_t s, const float* x, int ldx, const float* w, const float* bias, const int32_t* gold,
                  float* logits, float* row_loss, float* lse, int S, int P) {
    if (S == 0) return 0;
    if (!pair_head_supported(x, ldx, w, nullptr, 4, P) || !aligned16(logits)) return PSVAE_E_ALIGN;
    const int P4 = P / 4, nch = (P4 + 63) / 64;
    const int blocks = (S + 15) / 16 < 1024 ? (S + 15) / 16 : 1024;      // a wave takes 4 rows per round
    { ProfScope prof(s, KC_CE, 4.0 * S * (P + 4));
    if (nch <= 3) hipLaunchKernelGGL(pair_head_fwd_kernel<3>, dim3(blocks), dim3(256), 0, s, x, ldx, w, bias, gold, logits, row_loss, lse, S, P4);
    else hipLaunchKernelGGL(pair_head_fwd_kernel<5>, dim3(blocks), dim3(256), 0, s, x, ldx, w, bias, gold, logits, row_loss, lse, S, P4);
    }
    PSVAE_LAUNCH_CHECK();
    return 0;
}
int pair_head_bwd(hipStream_t s, float* logits_inout, const float* lse, const int32_t* gold, const float* w, GScale gscale,
                  const float* denoms, float* dx, int lddx, int S, int P) {
    if (S == 0) return 0;
    if (!pair_head_supported(dx, lddx, w, dx, lddx, P) || !aligned16(logits_inout)) return PSVAE_E_ALIGN;
    const int P4 = P / 4, nch = (P4 + 63) / 64;
    const int blocks = (S + 3) / 4 < 1024 ? (S + 3) / 4 : 1024;
    { ProfScope prof(s, KC_CE, 4.0 * S * (P + 8));
    if (nch <= 3) hipLaunchKernelGGL(pair_head_bwd_kernel<3>, dim3(blocks), dim3(256), 0, s, logits_inout, lse, gold, w, gscale, denoms, dx, lddx, S, P4);
    else hipLaunchKernelGGL(pair_head_bwd_kernel<5>, dim3(blocks), dim3(256), 0, s, logits_inout, lse, gold, w, gscale, denoms, dx, lddx, S, P4);
    }
    PSVAE_LAUNCH_CHECK();
    return 0;
}
int mse_fwd(hipStream_t s, const float* pred, int ldp, const float* gold, float* row, int B, int P) {
    if (B == 0) return 0;
    { ProfScope prof(s, KC_HEAD, 0);
    hipLaunchKernelGGL(mse_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, s, pred, ldp, gold, row, B, P);
    }
    PSVAE_LAUNCH_CHECK();
    return 0;
}
int mse_bwd(hipStream_t s, const float* pred, int ldp, const float* gold, GScale gscale, const float* denoms,
            float* dpred, int B, int P) {
    if (B == 0) return 0;
    { ProfScope prof(s, KC_HEAD, 0);
    hipLaunchKernelGGL(mse_bwd_kernel, dim3((B * P + 255) / 256), dim3(256), 0, s, pred, ldp, gold, gscale, denoms, dpred, B, P);
    }
    PSVAE_LAUNCH_CHECK();
    return 0;
}
int losses_finalize(hipStream_t s, const float* mse_row, int B, const float* piece_row, int rows_p,
                    const float* edge_row, int S, const float* klrow, const float* denoms, const float* w3, float* losses) {
    { ProfScope prof(s, KC_HEAD, 0);
    hipLaunchKernelGGL(losses_finalize_kernel, dim3(1), dim3(256), 0, s, mse_row, B, piece_row, rows_p, edge_row, S, klrow, denoms,
                       w3 ? w3[0] : 0.f, w3 ? w3[1] : 0.f, w3 ? w3[2] : 0.f, losses);
    }
    PSVAE_LAUNCH_CHECK();
    return 0;
}

size_t psvae_adam_partials(int64_t) { return NORM_BLOCKS; }

int clip_adam(hipStream_t s, float* p, float* g, float* m, float* v, int64_t count, float* partials,
              float max_norm, double lr, double b1, double b2, double eps, double bias1, double bias2, float* norm_out) {
    if (count <= 0) return 0;
    if (!p || !g || !m || !v || !partials) return PSVAE_E_NULL;
    { ProfScope prof(s, KC_ADAM, 4.0 * count);
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, s, g, (long long)count, partials);
    }
    PSVAE_LAUNCH_CHECK();
    int blocks = (int)((count + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    { ProfScope prof(s, KC_ADAM, 28.0 * count);
    hipLaunchKernelGGL(clip_adam_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, (long long)count, partials, max_norm,
                       (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps, (float)(lr / bias1), (float)sqrt(bias2), norm_out);
    }
    PSVAE_LAUNCH_CHECK();
    return 0;
}
