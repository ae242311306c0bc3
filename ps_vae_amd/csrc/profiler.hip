// Opt-in per-launch timer: when enabled, every kernel launch of the library is
// bracketed by hipEvents recorded on the launch stream, tagged with a kernel
// class and the algorithmic work (FLOPs or bytes) of the launch.  Meant for the
// benchmark's roofline pass only: off by default (one branch per launch), not
// thread-safe, and the only global state the library has.
#include "psvae_internal.h"
#include <vector>
#include <algorithm>

namespace {
struct Rec { int kc; double work; hipEvent_t a, b; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

ProfScope::ProfScope(hipStream_t s, int kc, double work) : s_(s), idx_(-1) {
    if (!g_on) return;
    Rec r{kc, work, get_event(), get_event()};
    if (!r.a || !r.b) return;
    (void)hipEventRecord(r.a, s);
    idx_ = (int)g_recs.size();
    g_recs.push_back(r);
}
ProfScope::~ProfScope() {
    if (idx_ >= 0) (void)hipEventRecord(g_recs[idx_].b, s_);
}

extern "C" int psvae_profile_enable(int on) {
    g_on = on != 0;
    return 0;
}

extern "C" int psvae_profile_num_classes(void) { return KC_COUNT; }

extern "C" const char* psvae_profile_class_name(int kc) {
    static const char* names[KC_COUNT] = {
        "gemm_forward", "gemm_input_grad", "gemm_weight_grad", "gemm_slab_reduce", "gine_aggregate_fwd",
        "gine_aggregate_bwd", "gine_edge_grad", "embed_gather", "table_grad", "segment_pool", "pair_gather",
        "pair_scatter", "gru_forward", "gru_backward", "softmax_ce", "head_misc", "clip_adam", "gine_layer_fused", "scorer_fused", "gine_layer_fused_bwd"};
    return (kc >= 0 && kc < KC_COUNT) ? names[kc] : "?";
}

// Waits for the recorded launches, adds their durations (ms), work and counts per
// class into the three arrays of psvae_profile_num_classes() entries, and clears the log.
extern "C" int psvae_profile_collect(double* ms, double* work, int64_t* launches) {
    if (!ms || !work || !launches) return PSVAE_E_NULL;
    for (const Rec& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return (int)hipGetLastError();
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return (int)hipGetLastError();
        ms[r.kc] += t; work[r.kc] += r.work; launches[r.kc] += 1;
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    g_recs.clear();
    return 0;
}

namespace {
__global__ void profile_empty_kernel() {}
}

// What a hipEvent pair adds around a launch: the median time between the two events around an empty
// kernel, in ms (64 samples on `stream`, which is synchronised).  Subtract it per launch from
// psvae_profile_collect's durations to compare with a profiler's kernel-only durations.
extern "C" double psvae_profile_event_overhead_ms(void* stream) {
    hipStream_t s = (hipStream_t)stream;
    std::vector<float> t;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1.0;
    for (int i = 0; i < 72; ++i) {
        (void)hipEventRecord(a, s);
        hipLaunchKernelGGL(profile_empty_kernel, dim3(256), dim3(256), 0, s);
        (void)hipEventRecord(b, s);
        if (hipEventSynchronize(b) != hipSuccess) return -1.0;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return -1.0;
        if (i >= 8) t.push_back(ms);
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}
