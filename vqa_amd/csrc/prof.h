// Optional per-kernel timing with HIP events recorded on the launch stream (off by default).
#pragma once
#include <hip/hip_runtime.h>

enum { VQA_K_CONV9 = 0, VQA_K_CONV1, VQA_K_WGRAD9, VQA_K_WGRAD1, VQA_K_GEMM, VQA_K_RESBLOCK, VQA_K_COUNT };

bool vqa_prof_on(int kind);
// call right before / right after enqueueing a kernel of `kind`; flops / bytes are the launch's
// algorithmic work
void vqa_prof_begin(int kind, double flops, double bytes, hipStream_t s);
void vqa_prof_end(hipStream_t s);

struct VqaProfScope {
    hipStream_t s;
    bool on;
    VqaProfScope(int kind, double flops, double bytes, hipStream_t st) : s(st), on(vqa_prof_on(kind)) {
        if (on) vqa_prof_begin(kind, flops, bytes, s);
    }
    ~VqaProfScope() { if (on) vqa_prof_end(s); }
};
