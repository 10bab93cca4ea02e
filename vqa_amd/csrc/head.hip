// The classifier's last linear layer (F -> A answers, A <= 64) forward and backward, and the
// cross-entropy loss behind it (gfx950).  33 MFLOP at the default sizes: what these passes cost is
// launches, not arithmetic - as GEMM + bias, cast + column sum + two GEMMs + column sum, and torch's
// softmax / nll_loss kernels they were sixteen launches per training step; here they are six.
// Operands are rounded to bf16 where the MFMA path rounds them (activations, weights, the gradient of
// the logits), products are summed in fp32 in a fixed order: every result is the same on every run.
//
// Replaces nn.Linear(1024, num_answers) of Module_Model.py:92, autograd's gradients of it, and
// torch.nn.CrossEntropyLoss of Train_model.py:185,312.
#include "vqa_dev.h"
#include "plan.h"
#include <string>

#define HEAD_ROWS 4          // rows of the batch per workgroup, forward
#define HEAD_MAX_SLICE 64    // rows of the batch per workgroup, backward

#define HEAD_CHECK_LAUNCH(name)                                                               \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) {                                                               \
            vqa_set_error(std::string(name ": ") + hipGetErrorString(e_));                    \
            return VQA_ELAUNCH;                                                               \
        }                                                                                     \
    } while (0)

// logits[n][o] = sum_i act[n][i] * w[o][i] + bias[o].  Workgroup = HEAD_ROWS rows; thread = (output o,
// one of 256 / AT ranges of i); the rows sit in LDS as floats (every lane of a range reads the same
// address: a broadcast), a thread streams its range of w's row o in 16-byte pieces; the ranges are
// added in a fixed order.
template <int AT>
__global__ __launch_bounds__(256) void head_forward_kernel(const bf16_t* __restrict__ act, int64_t ld_act,
                                                           const bf16_t* __restrict__ w, int64_t ld_w,
                                                           const float* __restrict__ bias, float* __restrict__ logits,
                                                           int N, int F, int A) {
    extern __shared__ __attribute__((aligned(16))) float hs[];     // [HEAD_ROWS][F], then [PARTS][HEAD_ROWS][AT]
    constexpr int PARTS = 256 / AT;
    float* rows = hs;
    float* red = hs + HEAD_ROWS * F;
    const int tid = threadIdx.x, n0 = blockIdx.x * HEAD_ROWS;
    for (int idx = tid; idx < HEAD_ROWS * F / 8; idx += 256) {
        const int r = idx / (F / 8), j = idx - r * (F / 8);
        u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (n0 + r < N) v = *(const u16x8*)(act + (int64_t)(n0 + r) * ld_act + j * 8);
#pragma unroll
        for (int k = 0; k < 8; k++) rows[r * F + j * 8 + k] = bf2f(v[k]);
    }
    __syncthreads();
    const int o = tid % AT, part = tid / AT;
    const int per = F / PARTS;                                     // F % (8 * PARTS) == 0
    float acc[HEAD_ROWS];
#pragma unroll
    for (int r = 0; r < HEAD_ROWS; r++) acc[r] = 0.f;
    if (o < A) {
        const bf16_t* wr = w + (int64_t)o * ld_w + part * per;
        const float* xr = rows + part * per;
        for (int j = 0; j < per; j += 8) {
            const u16x8 wv = *(const u16x8*)(wr + j);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float wf = bf2f(wv[k]);
#pragma unroll
                for (int r = 0; r < HEAD_ROWS; r++) acc[r] = fmaf(xr[r * F + j + k], wf, acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < HEAD_ROWS; r++) red[(part * HEAD_ROWS + r) * AT + o] = acc[r];
    __syncthreads();
    if (tid < HEAD_ROWS * AT) {
        const int r = tid / AT, oo = tid % AT;
        if (oo < A && n0 + r < N) {
            float t = bias ? bias[oo] : 0.f;
#pragma unroll
            for (int p = 0; p < PARTS; p++) t += red[(p * HEAD_ROWS + r) * AT + oo];
            logits[(int64_t)(n0 + r) * A + oo] = t;
        }
    }
}

extern "C" int vqa_head_forward(const void* act, int64_t ld_act, const void* w, int64_t ld_w, const float* bias,
                                float* logits, int32_t N, int32_t F, int32_t A, void* stream) {
    if (!act || !w || !logits || N <= 0 || F <= 0 || F % 64 != 0 || F > 8192 || A <= 0 || A > 64 || ld_act < F ||
        ld_w < F || ld_act % 8 != 0 || ld_w % 8 != 0) {
        vqa_set_error("vqa_head_forward: bad argument");
        return VQA_EINVAL;
    }
    const int at = A <= 32 ? 32 : 64;
    const size_t lds = (size_t)(HEAD_ROWS * F + (256 / at) * HEAD_ROWS * at) * sizeof(float);
    const void* fn = at == 32 ? (const void*)head_forward_kernel<32> : (const void*)head_forward_kernel<64>;
    if (lds > 48 * 1024 && !vqa_allow_lds(fn, (int)lds, "vqa_head_forward")) return VQA_ELAUNCH;
    dim3 grid((N + HEAD_ROWS - 1) / HEAD_ROWS), block(256);
    if (at == 32)
        hipLaunchKernelGGL(head_forward_kernel<32>, grid, block, lds, (hipStream_t)stream, (const bf16_t*)act, ld_act,
                           (const bf16_t*)w, ld_w, bias, logits, N, F, A);
    else
        hipLaunchKernelGGL(head_forward_kernel<64>, grid, block, lds, (hipStream_t)stream, (const bf16_t*)act, ld_act,
                           (const bf16_t*)w, ld_w, bias, logits, N, F, A);
    HEAD_CHECK_LAUNCH("vqa_head_forward");
    return VQA_OK;
}

// Backward of the same layer over one slice of the batch and 256 input columns; thread = column i.
//   g = bf16(dlogits)                       (the linear layers' operands are bf16)
//   gin[n][i]  = bf16(sum_o g[n][o] w[o][i]) * (act[n][i] > 0)     the gradient into the layer's input,
//                                                                  times the ReLU that produced it
//   dW[o][i]  += g[n][o] act[n][i],   db_in[i] += gin[n][i]        (partial sums of the slice)
//   db[o]     += dlogits[n][o]                                     (unrounded, like the fp32 column sum)
// The slices' partial sums are added by head_backward_reduce_kernel in slice order.
template <int AT>
__global__ __launch_bounds__(256) void head_backward_kernel(const float* __restrict__ dl, const bf16_t* __restrict__ act,
                                                            int64_t ld_act, const bf16_t* __restrict__ w, int64_t ld_w,
                                                            bf16_t* __restrict__ gin, int64_t ld_gin,
                                                            float* __restrict__ part, float* __restrict__ part_b,
                                                            int N, int F, int A, int slice) {
    __shared__ __attribute__((aligned(16))) float g[HEAD_MAX_SLICE * AT];
    const int tid = threadIdx.x, s = blockIdx.y, i = blockIdx.x * 256 + tid;
    const int n0 = s * slice, rows = min(slice, N - n0);
    for (int idx = tid; idx < slice * AT; idx += 256) {
        const int r = idx / AT, o = idx - r * AT;
        g[idx] = (r < rows && o < A) ? bf2f(f2bf(dl[(int64_t)(n0 + r) * A + o])) : 0.f;
    }
    __syncthreads();
    if (i < F) {
        float wv[AT], dw[AT];
#pragma unroll
        for (int o = 0; o < AT; o++) {
            wv[o] = o < A ? bf2f(w[(int64_t)o * ld_w + i]) : 0.f;
            dw[o] = 0.f;
        }
        float dbi = 0.f;
        for (int r = 0; r < rows; r++) {
            const float a = bf2f(act[(int64_t)(n0 + r) * ld_act + i]);
            float gi = 0.f;
#pragma unroll
            for (int o = 0; o < AT; o++) {
                const float gv = g[r * AT + o];
                gi = fmaf(gv, wv[o], gi);
                dw[o] = fmaf(gv, a, dw[o]);
            }
            const bf16_t q = a > 0.f ? f2bf(gi) : (bf16_t)0;
            gin[(int64_t)(n0 + r) * ld_gin + i] = q;
            dbi += bf2f(q);
        }
        float* P = part + (int64_t)s * (AT + 1) * F;
#pragma unroll
        for (int o = 0; o < AT; o++) P[(int64_t)o * F + i] = dw[o];
        P[(int64_t)AT * F + i] = dbi;
    }
    if (blockIdx.x == 0 && tid < AT) {
        float t = 0.f;
        if (tid < A)
            for (int r = 0; r < rows; r++) t += dl[(int64_t)(n0 + r) * A + tid];
        part_b[s * AT + tid] = t;
    }
}

__global__ __launch_bounds__(256) void head_backward_reduce_kernel(const float* __restrict__ part,
                                                                   const float* __restrict__ part_b,
                                                                   float* __restrict__ dw, float* __restrict__ db_in,
                                                                   float* __restrict__ db, int S, int AT, int F, int A) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = (AT + 1) * F;
    if (idx < total) {
        const int o = idx / F, i = idx - o * F;
        if (o < A || o == AT) {
            float t = 0.f;
#pragma unroll 8
            for (int s = 0; s < S; s++) t += part[(int64_t)s * total + idx];      // (eight loads in flight, added in slice order)
            if (o == AT) db_in[i] = t;
            else dw[(int64_t)o * F + i] = t;
        }
    } else if (idx - total < A) {
        const int o = idx - total;
        float t = 0.f;
        for (int s = 0; s < S; s++) t += part_b[s * AT + o];
        db[o] = t;
    }
}

extern "C" int64_t vqa_head_backward_workspace(int32_t N, int32_t F, int32_t A) {
    if (N <= 0 || F <= 0 || A <= 0 || A > 64) return 0;
    const int at = A <= 32 ? 32 : 64;
    const int slice = N >= 64 * 16 ? 16 : (N >= 256 ? 8 : 4);
    const int64_t S = (N + slice - 1) / slice;
    return S * ((int64_t)(at + 1) * F + at);
}

extern "C" int vqa_head_backward(const float* dlogits, const void* act, int64_t ld_act, const void* w, int64_t ld_w,
                                 void* gin, int64_t ld_gin, float* dw, float* db_in, float* db, float* workspace,
                                 int32_t N, int32_t F, int32_t A, void* stream) {
    if (!dlogits || !act || !w || !gin || !dw || !db_in || !db || !workspace || N <= 0 || F <= 0 || A <= 0 || A > 64 ||
        ld_act < F || ld_w < F || ld_gin < F) {
        vqa_set_error("vqa_head_backward: bad argument");
        return VQA_EINVAL;
    }
    const int at = A <= 32 ? 32 : 64;
    const int slice = N >= 64 * 16 ? 16 : (N >= 256 ? 8 : 4);
    const int S = (N + slice - 1) / slice;
    float* part = workspace;
    float* part_b = workspace + (int64_t)S * (at + 1) * F;
    dim3 grid((F + 255) / 256, S), block(256);
    if (at == 32)
        hipLaunchKernelGGL(head_backward_kernel<32>, grid, block, 0, (hipStream_t)stream, dlogits, (const bf16_t*)act, ld_act,
                           (const bf16_t*)w, ld_w, (bf16_t*)gin, ld_gin, part, part_b, N, F, A, slice);
    else
        hipLaunchKernelGGL(head_backward_kernel<64>, grid, block, 0, (hipStream_t)stream, dlogits, (const bf16_t*)act, ld_act,
                           (const bf16_t*)w, ld_w, (bf16_t*)gin, ld_gin, part, part_b, N, F, A, slice);
    HEAD_CHECK_LAUNCH("vqa_head_backward");
    const int total = (at + 1) * F + A;
    hipLaunchKernelGGL(head_backward_reduce_kernel, dim3((total + 255) / 256), block, 0, (hipStream_t)stream, part, part_b,
                       dw, db_in, db, S, at, F, A);
    HEAD_CHECK_LAUNCH("vqa_head_backward");
    return VQA_OK;
}

// loss = mean_n( logsumexp(logits[n]) - logits[n][target[n]] ),  dlogits = (softmax - onehot) / N:
// torch.nn.CrossEntropyLoss (mean reduction) forward and backward.  A row is spread over 32 lanes (coalesced
// accesses, the row's maximum and sum by lane exchanges), a workgroup of 256 threads takes 8 rows at a time and a
// contiguous range of rows in all; it leaves the sum of its rows' losses, added in row order, in `partial`, and
// a one-workgroup launch adds the partial sums in workgroup order: the same bits on every run.  (As ONE
// workgroup with a row per thread the pass took 68 us on the critical path of a step.)
#define XENT_BLOCKS 128
__global__ __launch_bounds__(256) void xent_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                        float* __restrict__ partial, float* __restrict__ dlogits, int N, int A,
                                                        int rows_per_block) {
    __shared__ float rowloss[8];
    const int tid = threadIdx.x, sub = tid >> 5, l = tid & 31;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
    const float inv_n = 1.f / (float)N;
    float total = 0.f;                                   // (thread 0 only)
    for (int base = r0; base < r1; base += 8) {
        const int n = base + sub;
        const bool live = n < r1;
        const float* x = logits + (int64_t)(live ? n : r0) * A;
        float m = -INFINITY;
        for (int o = l; o < A; o += 32) m = fmaxf(m, x[o]);
#pragma unroll
        for (int d = 16; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 32));
        float sum = 0.f;
        for (int o = l; o < A; o += 32) sum += expf(x[o] - m);
#pragma unroll
        for (int d = 16; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 32);
        const float lse = m + logf(sum);
        const int64_t t = target[live ? n : r0];
        const bool valid = t >= 0 && t < A;              // (out of range: the row adds nothing, like an ignored index)
        if (live) {
            float* dl = dlogits + (int64_t)n * A;
            for (int o = l; o < A; o += 32) dl[o] = valid ? (expf(x[o] - lse) - (o == t ? 1.f : 0.f)) * inv_n : 0.f;
        }
        if (l == 0) rowloss[sub] = (live && valid) ? lse - x[t] : 0.f;
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) total += rowloss[k];
        }
        __syncthreads();
    }
    if (tid == 0) partial[blockIdx.x] = total;
}

__global__ __launch_bounds__(64) void xent_sum_kernel(const float* __restrict__ partial, float* __restrict__ loss, int blocks, int N) {
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int b = 0; b < blocks; b++) t += partial[b];
        loss[0] = t / (float)N;
    }
}

extern "C" int vqa_xent(const float* logits, const int64_t* target, float* loss, float* dlogits, float* workspace, int32_t N,
                        int32_t A, void* stream) {
    if (!logits || !target || !loss || !dlogits || !workspace || N <= 0 || A <= 0) {
        vqa_set_error("vqa_xent: bad argument");
        return VQA_EINVAL;
    }
    const int rows_per_block = ((N + XENT_BLOCKS - 1) / XENT_BLOCKS + 7) / 8 * 8;
    const int blocks = (N + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(xent_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, target, workspace, dlogits, N, A,
                       rows_per_block);
    HEAD_CHECK_LAUNCH("vqa_xent");
    hipLaunchKernelGGL(xent_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, loss, blocks, N);
    HEAD_CHECK_LAUNCH("vqa_xent");
    return VQA_OK;
}
