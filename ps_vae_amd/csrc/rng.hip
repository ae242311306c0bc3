// Counter-based N(0,1) rows for rsample's epsilon (src/modules/vae_piece_decoder.py:61 draws it with
// torch.randn_like, whose stream depends on how graphs are dealt to processes).  Here the draw of a graph is a
// pure function of (seed, draw index, GLOBAL row of the graph, column): a data-parallel run over any number of
// GPUs samples exactly what one process samples for the undivided batch.
//
// Generator: Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11).
//   counter = (column group c/4, row low word, row high word, draw low word)
//   key     = (seed low word, seed high word XOR draw high word)
// One call yields the four columns 4g .. 4g+3 of a row by two Box-Muller transforms over 24-bit uniforms:
//   u = ((x >> 8) + 0.5) * 2^-24,   n0 = sqrt(-2 ln u0) cos(2 pi u1),   n1 = sqrt(-2 ln u0) sin(2 pi u1).
#include "psvae_internal.h"

namespace {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ __launch_bounds__(256) void normal_rows_kernel(uint64_t seed, uint64_t draw, int64_t first_row, int rows,
                                                          int width, float* __restrict__ out) {
    const int groups = (width + 3) >> 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)rows * groups) return;
    const int r = (int)(t / groups), g = (int)(t - (long long)r * groups);
    const uint64_t row = (uint64_t)(first_row + r);
    uint32_t c[4] = {(uint32_t)g, (uint32_t)row, (uint32_t)(row >> 32), (uint32_t)draw};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(draw >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    float v[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u0 = ((float)(c[2 * h] >> 8) + 0.5f) * 5.9604644775390625e-8f;      // 2^-24
        const float u1 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * 5.9604644775390625e-8f;
        const float rad = sqrtf(-2.0f * logf(u0));
        float sn, cs;
        sincosf(6.283185307179586f * u1, &sn, &cs);
        v[2 * h] = rad * cs; v[2 * h + 1] = rad * sn;
    }
    float* dst = out + (size_t)r * width + 4 * g;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (4 * g + i < width) dst[i] = v[i];
}

}  // namespace

extern "C" int psvae_normal_rows(uint64_t seed, uint64_t draw, int64_t first_row, int rows, int width, float* out,
                                 void* stream) {
    if (rows < 0 || width < 1 || first_row < 0) return PSVAE_E_SHAPE;
    if (rows == 0) return 0;
    if (!out) return PSVAE_E_NULL;
    const long long items = (long long)rows * ((width + 3) / 4);
    ProfScope prof((hipStream_t)stream, KC_HEAD, 4.0 * rows * width);
    hipLaunchKernelGGL(normal_rows_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, draw,
                       first_row, rows, width, out);
    PSVAE_LAUNCH_CHECK();
    return 0;
}
