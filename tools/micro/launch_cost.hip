// Host cost of the HIP calls a step is made of: an empty kernel launched N times on one stream, round-robin on three
// streams, in runs of 8 per stream; an event record + stream wait pair between two streams.  Prints microseconds of
// host time per call (the queue is drained every 64 calls so that the host never blocks on a full queue).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 999) *p = 1; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s[3];
    for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    hipEvent_t ev[8];
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const int N = 6400;
    auto run = [&](const char* what, auto body) {
        for (int rep = 0; rep < 2; ++rep) {
            double host = 0;
            for (int i = 0; i < N; i += 64) {
                const double t0 = now();
                for (int j = 0; j < 64; ++j) body(i + j);
                host += now() - t0;
                hipDeviceSynchronize();
            }
            if (rep) printf("%-60s %.2f us per call\n", what, host / N);
        }
    };
    run("launch, one stream", [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[0], nullptr); });
    run("launch, three streams round-robin", [&](int i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[i % 3], nullptr); });
    run("launch, three streams in runs of 8", [&](int i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[(i / 8) % 3], nullptr); });
    run("event record (after a launch on that stream) + launch", [&](int i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[0], nullptr); hipEventRecord(ev[i % 8], s[0]); });
    run("launch on A, record on A, wait on B, launch on B", [&](int i) {
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[0], nullptr); hipEventRecord(ev[i % 8], s[0]);
        hipStreamWaitEvent(s[1], ev[i % 8], 0); hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[1], nullptr); });
    run("two launches (A, B) without the hand-over", [&](int) {
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[0], nullptr); hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[1], nullptr); });
    {   // the same hand-over through a 64-bit word in signal memory: stream A writes a sequence number, stream B waits for it
        uint64_t* sig = nullptr;
        hipError_t e = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
        if (e != hipSuccess) printf("hipMallocSignalMemory: %s\n", hipGetErrorString(e));
        else {
            hipMemset(sig, 0, 8);
            uint64_t seq = 0;
            bool ok = true;
            run("launch on A, write value on A, wait value on B, launch on B", [&](int) {
                ++seq;
                hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[0], nullptr);
                if (hipStreamWriteValue64(s[0], sig, seq, 0) != hipSuccess) ok = false;
                if (hipStreamWaitValue64(s[1], sig, seq, hipStreamWaitValueGte, 0xffffffffffffffffull) != hipSuccess) ok = false;
                hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[1], nullptr); });
            uint64_t last = 0;
            hipDeviceSynchronize();
            hipMemcpy(&last, sig, 8, hipMemcpyDeviceToHost);
            printf("  (calls %s, last value %llu of %llu)\n", ok ? "succeeded" : "FAILED", (unsigned long long)last, (unsigned long long)seq);
        }
    }
    run("hipMemsetAsync 64 KB", [&](int) { static void* p = nullptr; if (!p) hipMalloc(&p, 65536); hipMemsetAsync(p, 0, 65536, s[0]); });
    run("hipGetLastError", [&](int) { (void)hipGetLastError(); });
    return 0;
}
