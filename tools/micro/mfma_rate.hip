// Issue rate of the fp32-input MFMA forms on one wave per SIMD: cycles per instruction (shader clock, s_memtime)
// and the shader clock itself against the 100 MHz constant counter (s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, long long* t, int iters) {
    floatx4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    const long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, long long* t, int iters) {
    floatx16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    const long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}

template <typename F>
void run(const char* name, F launch, int blocks, int nacc, double flop_per) {
    float* out; long long* t;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&t, 16);
    const int iters = 2000;
    launch(out, t, 10, blocks);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch(out, t, iters, blocks);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * 4 * nacc;
    printf("%-28s blocks %4d: %.1f cycles/MFMA (s_memtime), %.1f ns/MFMA (100 MHz counter) -> clock %.2f GHz; whole GPU %.1f TFLOP/s\n", name, blocks,
           h[0] / n, h[1] * 10.0 / n, (double)h[0] / (h[1] * 10.0), n * 4 * blocks * flop_per / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(t);
}

int main() {
    for (int blocks : {1, 104, 256, 512}) {
        run("16x16x4 f32, 6 accumulators", [](float* o, long long* t, int it, int b) { hipLaunchKernelGGL(k16<6>, dim3(b), dim3(256), 0, 0, o, t, it); }, blocks, 6, 2048.0);
        run("16x16x4 f32, 10 accumulators", [](float* o, long long* t, int it, int b) { hipLaunchKernelGGL(k16<10>, dim3(b), dim3(256), 0, 0, o, t, it); }, blocks, 10, 2048.0);
        run("32x32x2 f32, 2 accumulators", [](float* o, long long* t, int it, int b) { hipLaunchKernelGGL(k32<2>, dim3(b), dim3(256), 0, 0, o, t, it); }, blocks, 2, 4096.0);
    }
    return 0;
}
