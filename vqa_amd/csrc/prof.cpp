// Per-kernel timing: a pair of HIP events around every launch of the MFMA kernels while
// profiling is enabled, resolved at vqa_prof_collect.  Used by bench.py for the roofline figure.
#include "prof.h"
#include "plan.h"

#include <mutex>
#include <utility>
#include <vector>

namespace {
struct Rec { int kind; double flops, bytes; hipEvent_t a, b; int launches; };
std::mutex g_mu;
unsigned g_mask = 0;   // bit k: launches of kind k are timed
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
Rec g_open;
Rec g_run;               // the open run (kind < 0: none)
bool g_run_open = false;
bool g_nested = false;   // the launch being recorded belongs to the open run

hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

bool vqa_prof_on(int kind) { return (g_mask >> kind) & 1u; }

void vqa_prof_begin(int kind, double flops, double bytes, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_nested = g_run_open && g_run.kind == kind;
    if (g_nested) {
        g_run.flops += flops; g_run.bytes += bytes; g_run.launches++;
        return;
    }
    g_open = Rec{kind, flops, bytes, get_event(), get_event(), 1};
    (void)hipEventRecord(g_open.a, s);
}

void vqa_prof_end(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_nested) { g_nested = false; return; }
    (void)hipEventRecord(g_open.b, s);
    g_recs.push_back(g_open);
}

void vqa_prof_run_begin(int kind, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_run_open) return;          // one run at a time: an inner one is part of the outer
    g_run = Rec{kind, 0.0, 0.0, get_event(), get_event(), 0};
    g_run_open = true;
    (void)hipEventRecord(g_run.a, s);
}

void vqa_prof_run_end(int kind, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_run_open || g_run.kind != kind) return;
    (void)hipEventRecord(g_run.b, s);
    g_run_open = false;
    if (g_run.launches > 0) g_recs.push_back(g_run);
    else { g_pool.push_back(g_run.a); g_pool.push_back(g_run.b); }
}

extern "C" int vqa_prof_enable(int32_t kind_mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_mask = (unsigned)kind_mask;
    return VQA_OK;
}

// Waits for the recorded launches, then adds per kind: out[kind*4 + 0] launches, +1 total ms,
// +2 algorithmic FLOPs, +3 algorithmic bytes.  Clears the records.
extern "C" int vqa_prof_collect(double* out, int32_t kinds) {
    if (!out || kinds < VQA_K_COUNT) { vqa_set_error("vqa_prof_collect: need room for 6 kinds"); return VQA_EINVAL; }
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < kinds * 4; i++) out[i] = 0.0;
    for (Rec& r : g_recs) {
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            out[r.kind * 4 + 0] += (double)r.launches;
            out[r.kind * 4 + 1] += ms;
            out[r.kind * 4 + 2] += r.flops;
            out[r.kind * 4 + 3] += r.bytes;
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    return VQA_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize), once per (kernel, device).  A kernel's code object
// is loaded per device, so the limit is raised on each device the library is used on.
// Compute units of the current device (asked once per device; 0 if the runtime will not say).
int vqa_cu_count() {
    static std::mutex mu;
    static std::vector<std::pair<int, int>> known;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(mu);
    for (const auto& k : known)
        if (k.first == dev) return k.second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    known.emplace_back(dev, n);
    return n;
}

bool vqa_allow_lds(const void* kernel, int bytes, const char* who) {
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(mu);
        for (const auto& d : done)
            if (d.first == kernel && d.second == dev) return true;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) { done.emplace_back(kernel, dev); return true; }
    }
    vqa_set_error(std::string(who) + ": raising the dynamic LDS limit: " + hipGetErrorString(e));
    return false;
}
