// Batched front half of molecule generation (src/modules/vae_piece_decoder.py:128-241): the two
// GPU-shaped stages of VAEPieceDecoder.inference, for many latents at once.
//   1. psvae_sample_pieces: the piece sequence of every latent, sampled token by token from the GRU
//      (:134-158).  One step = W_hh product (MFMA GEMM over all sequences), gate kernel, vocabulary
//      product, one sampling wave per sequence.  Sequences that have ended are frozen.
//   2. psvae_score_links: the decoder-side GINE stack over the fragments' atoms and in-piece bonds,
//      then the link scorer and a softmax over the 4 bond classes for every candidate pair (:203-238).
// What lies between and after them (piece SMILES -> atoms, valence and cycle checks, molecule
// assembly) needs RDKit and stays on the CPU.
#include "psvae_internal.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct SampleState { int32_t *cur, *atoms, *len, *active, *n_active; };

__global__ void sample_init_kernel(SampleState st, int32_t* __restrict__ lengths, int Bz, int start_token) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) *st.n_active = Bz;
    if (b >= Bz) return;
    st.cur[b] = start_token; st.atoms[b] = 0; st.len[b] = 0; st.active[b] = 1; lengths[b] = 0;
}

// h <- GRU cell (gate order r, z, n; n = tanh(gi_n + r * gh_n); h' = (1 - z) n + z h) for the sequences
// still running.  gi is a row of the per-token table  piece_embedding W_ih^T + b_ih.
__global__ void gru_cell_kernel(const float* __restrict__ xproj, const float* __restrict__ gh, float* __restrict__ h,
                                const int32_t* __restrict__ cur, const int32_t* __restrict__ active, int Bz, int R) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Bz * R) return;
    const int b = t / R, j = t - b * R;
    if (!active[b]) return;
    const float* gi = xproj + (long long)cur[b] * 3 * R;
    const float* g = gh + (long long)b * 3 * R;
    const float r = sigmoidf_(gi[j] + g[j]);
    const float u = sigmoidf_(gi[R + j] + g[R + j]);
    const float c = tanhf(gi[2 * R + j] + r * g[2 * R + j]);
    h[t] = (1.0f - u) * c + u * h[t];
}

// One wave per sequence: softmax(logits / temperature) with <pad> (and, before the first piece, the end
// token) masked out, then the class whose cumulative-probability interval holds u (classes in index
// order; lane l owns a contiguous run of classes).  Lane 0 then applies the stop rules of the loop.
__global__ __launch_bounds__(256) void sample_step_kernel(const float* __restrict__ logits, int ld, int V2, int pad_idx,
                                                          int end_idx, float temperature, const float* __restrict__ uniforms,
                                                          int step, int max_steps, const int32_t* __restrict__ piece_atoms,
                                                          int max_atom_num, SampleState st, int32_t* __restrict__ piece_ids,
                                                          int32_t* __restrict__ lengths, int Bz) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= Bz || !st.active[b]) return;
    const float* x = logits + (long long)b * ld;
    const int len = st.len[b];
    const int chunk = (V2 + 63) / 64, c0 = lane * chunk, c1 = min(V2, c0 + chunk);
    float mx = -INFINITY;
    for (int c = c0; c < c1; ++c)
        if (c != pad_idx && !(len == 0 && c == end_idx)) mx = fmaxf(mx, x[c] / temperature);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float mine = 0.f;
    for (int c = c0; c < c1; ++c)
        if (c != pad_idx && !(len == 0 && c == end_idx)) mine += expf(x[c] / temperature - mx);
    // inclusive scan of the lane sums
    float incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const float up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    const float total = __shfl(incl, 63, 64);
    const float target = uniforms[(long long)b * max_steps + step] * total;
    const float before = incl - mine;
    // the first lane whose inclusive sum exceeds the target owns the sample
    const unsigned long long owners = __ballot(incl > target && mine > 0.f);
    int pick = -1;
    if (owners) {
        const int owner = __ffsll((long long)owners) - 1;
        if (lane == owner) {
            float run = before;
            for (int c = c0; c < c1; ++c) {
                if (c == pad_idx || (len == 0 && c == end_idx)) continue;
                run += expf(x[c] / temperature - mx);
                pick = c;                       // falls through to the lane's last class on rounding
                if (run > target) break;
            }
        }
        pick = __shfl(pick, owner, 64);
    } else {
        // u so close to 1 that rounding left no interval: the last admissible class
        for (int c = V2 - 1; c >= 0; --c)
            if (c != pad_idx && !(len == 0 && c == end_idx)) { pick = c; break; }
    }
    if (lane == 0) {
        const int atoms = st.atoms[b] + piece_atoms[pick];
        piece_ids[(long long)b * max_steps + len] = pick;
        st.cur[b] = pick; st.atoms[b] = atoms; st.len[b] = len + 1;
        // the loop of the reference ends on the end token, on too many atoms, or at max_pos pieces,
        // and then drops the last token whichever it was
        if (pick == end_idx || atoms > max_atom_num || len + 1 == max_steps) {
            st.active[b] = 0;
            lengths[b] = len;
            atomicSub(st.n_active, 1);
        }
    }
}

// softmax over the 4 bond classes of every pair + the winning class and its probability
__global__ void link_softmax_kernel(const float* __restrict__ logits, float* __restrict__ probs, float* __restrict__ conf,
                                    int32_t* __restrict__ type, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const float4 v = *(const float4*)(logits + (long long)i * 4);
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    float4 e = make_float4(expf(v.x - mx), expf(v.y - mx), expf(v.z - mx), expf(v.w - mx));
    const float s = e.x + e.y + e.z + e.w;
    e.x /= s; e.y /= s; e.z /= s; e.w /= s;
    if (probs) *(float4*)(probs + (long long)i * 4) = e;
    float best = e.x; int arg = 0;
    if (e.y > best) { best = e.y; arg = 1; }
    if (e.z > best) { best = e.z; arg = 2; }
    if (e.w > best) { best = e.w; arg = 3; }
    if (conf) conf[i] = best;
    if (type) type[i] = arg;
}

struct SampleWS { float *xproj, *h, *gh, *logits; SampleState st; size_t floats; };
size_t carve_sample(const psvae_dims& d, int Bz, float* base, SampleWS* w) {
    const size_t V2 = d.num_piece, R = d.rnn, ldl = (size_t)round_up(d.num_piece, 4);
    size_t off = 0;
    auto take = [&](size_t count) { size_t o = off; off += (size_t)round_up((int64_t)count, 64); return o; };
    const size_t o_xp = take(V2 * 3 * R), o_h = take((size_t)Bz * R), o_gh = take((size_t)Bz * 3 * R), o_lg = take((size_t)Bz * ldl);
    const size_t o_cur = take(Bz), o_at = take(Bz), o_len = take(Bz), o_act = take(Bz), o_n = take(1);
    if (w) {
        w->xproj = base + o_xp; w->h = base + o_h; w->gh = base + o_gh; w->logits = base + o_lg;
        w->st.cur = (int32_t*)(base + o_cur); w->st.atoms = (int32_t*)(base + o_at); w->st.len = (int32_t*)(base + o_len);
        w->st.active = (int32_t*)(base + o_act); w->st.n_active = (int32_t*)(base + o_n);
        w->floats = off;
    }
    return off;
}

int check_sample_dims(const psvae_dims* d, int Bz) {
    if (!d) return PSVAE_E_NULL;
    if (Bz < 1 || d->num_piece < 3 || d->max_pos < 1 || d->d_piece < 1) return PSVAE_E_SHAPE;
    if (d->rnn <= 0 || d->rnn % 4 || d->latent <= 0 || d->latent % 4) return PSVAE_E_ALIGN;
    return 0;
}

}  // namespace

extern "C" size_t psvae_sample_pieces_workspace_bytes(const psvae_dims* dims, int Bz) {
    if (check_sample_dims(dims, Bz) != 0) return 0;
    return carve_sample(*dims, Bz, nullptr, nullptr) * sizeof(float);
}

extern "C" int psvae_sample_pieces(const psvae_dims* dims, const psvae_params* params, const float* z, int Bz,
                                   const int32_t* piece_atoms, int max_atom_num, float temperature,
                                   const float* uniforms, int32_t* piece_ids, int32_t* lengths, int check_every,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    PSVAE_TRY(check_sample_dims(dims, Bz));
    if (!params || !z || !piece_atoms || !uniforms || !piece_ids || !lengths || !workspace) return PSVAE_E_NULL;
    if (!aligned16(workspace) || !aligned16(z)) return PSVAE_E_ALIGN;
    if (!(temperature > 0.f)) return PSVAE_E_SHAPE;
    const psvae_dims& d = *dims; const psvae_params& p = *params;
    SampleWS w;
    if (carve_sample(d, Bz, (float*)workspace, &w) * sizeof(float) > workspace_bytes) return PSVAE_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int R = d.rnn, V2 = d.num_piece, ldl = (int)round_up(V2, 4), steps = d.max_pos;
    const int pad_idx = d.pad_idx, end_idx = d.pad_idx + 1;
    // per-token input products, the initial hidden state, the start token (:134-135)
    PSVAE_TRY(linear_fwd(st, p.piece_emb, d.d_piece, p.gru_wih, p.gru_bih, w.xproj, 3 * R, V2, 3 * R, d.d_piece, 0, nullptr, 0));
    PSVAE_TRY(linear_fwd(st, z, d.latent, p.l2h_w, p.l2h_b, w.h, R, Bz, R, d.latent, 0, nullptr, 0));
    hipLaunchKernelGGL(sample_init_kernel, dim3((Bz + 255) / 256), dim3(256), 0, st, w.st, lengths, Bz, end_idx);
    PSVAE_LAUNCH_CHECK();
    for (int step = 0; step < steps; ++step) {
        PSVAE_TRY(linear_fwd(st, w.h, R, p.gru_whh, p.gru_bhh, w.gh, 3 * R, Bz, 3 * R, R, 0, nullptr, 0));
        hipLaunchKernelGGL(gru_cell_kernel, dim3((Bz * R + 255) / 256), dim3(256), 0, st, w.xproj, w.gh, w.h, w.st.cur,
                           w.st.active, Bz, R);
        PSVAE_LAUNCH_CHECK();
        PSVAE_TRY(linear_fwd(st, w.h, R, p.vocab_w, p.vocab_b, w.logits, ldl, Bz, V2, R, 0, nullptr, 0));
        hipLaunchKernelGGL(sample_step_kernel, dim3((Bz + 3) / 4), dim3(256), 0, st, w.logits, ldl, V2, pad_idx, end_idx,
                           temperature, uniforms, step, steps, piece_atoms, max_atom_num, w.st, piece_ids, lengths, Bz);
        PSVAE_LAUNCH_CHECK();
        if (check_every > 0 && (step + 1) % check_every == 0 && step + 1 < steps) {
            int32_t running = 0;    // the one host round trip of this entry point, and only on request
            if (hipMemcpyAsync(&running, w.st.n_active, sizeof(running), hipMemcpyDeviceToHost, st) != hipSuccess) return (int)hipGetLastError();
            if (hipStreamSynchronize(st) != hipSuccess) return (int)hipGetLastError();
            if (running == 0) break;
        }
    }
    return 0;
}

int link_softmax(hipStream_t s, const float* logits, float* probs, float* conf, int32_t* type, int S) {
    if (S <= 0) return 0;
    hipLaunchKernelGGL(link_softmax_kernel, dim3((S + 255) / 256), dim3(256), 0, s, logits, probs, conf, type, S);
    PSVAE_LAUNCH_CHECK();
    return 0;
}
