// Optional per-kernel timing with HIP events recorded on the launch stream (off by default).
#pragma once
#include <hip/hip_runtime.h>

enum { VQA_K_CONV9 = 0, VQA_K_CONV1, VQA_K_WGRAD9, VQA_K_WGRAD1, VQA_K_GEMM, VQA_K_RESBLOCK, VQA_K_COUNT };

bool vqa_prof_on(int kind);
// call right before / right after enqueueing a kernel of `kind`; flops / bytes are the launch's
// algorithmic work
void vqa_prof_begin(int kind, double flops, double bytes, hipStream_t s);
void vqa_prof_end(hipStream_t s);

// A run of back-to-back launches of one kind (the levels of a tree pass) under ONE pair of events: while a run
// is open, vqa_prof_begin / vqa_prof_end of that kind only add their launch's work to it.  An event pair per
// launch keeps a launch from overlapping its neighbours and costs about 5 us; a pair per run costs that once,
// and the run's time per launch includes what a kernel boundary costs.
void vqa_prof_run_begin(int kind, hipStream_t s);
void vqa_prof_run_end(int kind, hipStream_t s);

struct VqaProfRun {
    hipStream_t s;
    int kind;
    bool on;
    VqaProfRun(int k, hipStream_t st) : s(st), kind(k), on(vqa_prof_on(k)) { if (on) vqa_prof_run_begin(kind, s); }
    void end() { if (on) { vqa_prof_run_end(kind, s); on = false; } }
    ~VqaProfRun() { end(); }
};

struct VqaProfScope {
    hipStream_t s;
    bool on;
    VqaProfScope(int kind, double flops, double bytes, hipStream_t st) : s(st), on(vqa_prof_on(kind)) {
        if (on) vqa_prof_begin(kind, flops, bytes, s);
    }
    ~VqaProfScope() { if (on) vqa_prof_end(s); }
};
