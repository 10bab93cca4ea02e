// Layout changes and small elementwise / reduction kernels around the MFMA kernels (gfx950).
// All of them are HBM-bound; loads and stores are arranged so that a wave touches whole lines.
#include "vqa_dev.h"
#include "plan.h"

typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2n __attribute__((ext_vector_type(2)));

#define CHECK_LAUNCH(name)                                                                    \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) {                                                               \
            vqa_set_error(std::string(name ": ") + hipGetErrorString(e_));                    \
            return VQA_ELAUNCH;                                                               \
        }                                                                                     \
    } while (0)

// x [n][Cin][196] (fp32) -> out [n][196][Cpad] (bf16); channels Cin..Cpad-1 are written as zero.
// Block = one sample x 64 channels.  Reads are 16 bytes per lane with every load of a thread in
// flight at once; the tile sits in LDS as bf16 [channel][198] (odd dword stride: the column reads
// of the second phase are conflict-free, 25 KB per block so six blocks share a CU); writes are
// 16 bytes per lane, 128 contiguous bytes per pixel.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x,
                                                           bf16_t* __restrict__ out, int Cin, int Cpad) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 198];
    const int n = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
    const float4* xs = (const float4*)(x + ((int64_t)n * Cin + c0) * VQA_HW);
    const int cvalid = min(64, Cin - c0);
    float4 v[13];
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const int f = tid + 256 * i;            // float4 index: channel f / 49, pixels 4 * (f % 49)..+3
        const int c = f / 49;
        // streaming read: the fp32 input is touched once
        if (f < 64 * 49 && c < cvalid) {
            const f32x4n t = __builtin_nontemporal_load((const f32x4n*)(xs + f));
            v[i] = make_float4(t[0], t[1], t[2], t[3]);
        } else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const int f = tid + 256 * i;
        if (f < 64 * 49) {
            const int c = f / 49, pix = (f - c * 49) * 4;
            uint32_t* t = (uint32_t*)(tile + c * 198 + pix);
            t[0] = (uint32_t)f2bf(v[i].x) | ((uint32_t)f2bf(v[i].y) << 16);
            t[1] = (uint32_t)f2bf(v[i].z) | ((uint32_t)f2bf(v[i].w) << 16);
        }
    }
    __syncthreads();
    bf16_t* os = out + (int64_t)n * VQA_HW * Cpad + c0;
    for (int idx = tid; idx < VQA_HW * 8; idx += 256) {
        // lanes run along the pixels (bank-conflict-free column reads); a pixel's eight 16-byte
        // pieces are written by eight different waves' iterations, each a full 16 B
        const int cg = (idx / VQA_HW) * 8, pix = idx - (idx / VQA_HW) * VQA_HW;
        const bf16_t* t = tile + cg * 198 + pix;
        uint4 o;
        o.x = (uint32_t)t[0] | ((uint32_t)t[198] << 16);
        o.y = (uint32_t)t[2 * 198] | ((uint32_t)t[3 * 198] << 16);
        o.z = (uint32_t)t[4 * 198] | ((uint32_t)t[5 * 198] << 16);
        o.w = (uint32_t)t[6 * 198] | ((uint32_t)t[7 * 198] << 16);
        *(uint4*)(os + (int64_t)pix * Cpad + cg) = o;
    }
}

extern "C" int vqa_nchw_to_nhwc(const float* x, void* out, int32_t N, int32_t Cin, int32_t Cpad, void* stream) {
    if (!x || !out || N <= 0 || Cin <= 0 || Cpad < Cin || Cpad % 64 != 0) {
        vqa_set_error("vqa_nchw_to_nhwc: bad argument");
        return VQA_EINVAL;
    }
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(Cpad / 64, N), dim3(256), 0, (hipStream_t)stream,
                       x, (bf16_t*)out, Cin, Cpad);
    CHECK_LAUNCH("vqa_nchw_to_nhwc");
    return VQA_OK;
}

extern "C" int vqa_fc_weight_to_nhwc(const float* w, void* out, int32_t F, int32_t Fpad, int32_t P, void* stream) {
    if (Fpad < F || !w || !out || P % 64 != 0) { vqa_set_error("vqa_fc_weight_to_nhwc: bad argument"); return VQA_EINVAL; }
    return vqa_nchw_to_nhwc(w, out, F, P, P, stream);
}

// in [n][196][C] (bf16 or fp32) -> out [n][C][196] fp32
template <typename Tin>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const Tin* __restrict__ in, float* __restrict__ out, int C) {
    __shared__ float tile[64 * 197];
    const int n = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
    const Tin* is = in + (int64_t)n * VQA_HW * C + c0;
    for (int idx = tid; idx < VQA_HW * 64; idx += 256) {
        const int pix = idx >> 6, c = idx & 63;
        float v;
        if constexpr (sizeof(Tin) == 2) v = bf2f(is[(int64_t)pix * C + c]);
        else v = is[(int64_t)pix * C + c];
        tile[c * 197 + pix] = v;
    }
    __syncthreads();
    float* os = out + ((int64_t)n * C + c0) * VQA_HW;
    for (int idx = tid; idx < 64 * VQA_HW; idx += 256) {
        const int c = idx / VQA_HW, pix = idx - c * VQA_HW;
        os[idx] = tile[c * 197 + pix];
    }
}

extern "C" int vqa_nhwc_to_nchw(const void* x, float* out, int32_t N, int32_t C, void* stream) {
    if (!x || !out || N <= 0 || C <= 0 || C % 64 != 0) { vqa_set_error("vqa_nhwc_to_nchw: bad argument"); return VQA_EINVAL; }
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(C / 64, N), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, out, C);
    CHECK_LAUNCH("vqa_nhwc_to_nchw");
    return VQA_OK;
}

// in [n][196][C] bf16 -> out [n][C][196] bf16: the per-sample transposition that turns a flattened
// channels-last activation into the reference's flatten order (Layers.py:42-44: c * 196 + hw).
// Block = one sample x 64 channels; 16-byte loads (8 channels of a pixel), the tile in LDS as
// bf16 [channel][198], 8-byte stores along the 25 KB contiguous output range of the block.
__global__ __launch_bounds__(256) void transpose_hw_c_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                             int C) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 198];
    const int n = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
    const bf16_t* is = in + (int64_t)n * VQA_HW * C + c0;
    uint4 v[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const int idx = min(tid + 256 * i, VQA_HW * 8 - 1);
        v[i] = *(const uint4*)(is + (int64_t)(idx >> 3) * C + (idx & 7) * 8);
    }
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const int idx = tid + 256 * i;
        if (idx < VQA_HW * 8) {
            bf16_t* t = tile + (idx & 7) * 8 * 198 + (idx >> 3);
            const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                t[(2 * j) * 198] = (bf16_t)(w[j] & 0xffffu);
                t[(2 * j + 1) * 198] = (bf16_t)(w[j] >> 16);
            }
        }
    }
    __syncthreads();
    uint2* os = (uint2*)(out + ((int64_t)n * C + c0) * VQA_HW);
    const uint32_t* tw = (const uint32_t*)tile;
    for (int q = tid; q < 64 * 49; q += 256) {
        const int c = q / 49, k = q - c * 49;
        __builtin_nontemporal_store((u32x2n){tw[c * 99 + 2 * k], tw[c * 99 + 2 * k + 1]}, (u32x2n*)(os + q));   // read once, much later
    }
}

extern "C" int vqa_transpose_hw_c(const void* x, void* out, int32_t N, int32_t C, void* stream) {
    if (!x || !out || N <= 0 || C <= 0 || C % 64 != 0) { vqa_set_error("vqa_transpose_hw_c: bad argument"); return VQA_EINVAL; }
    hipLaunchKernelGGL(transpose_hw_c_kernel, dim3(C / 64, N), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)out, C);
    CHECK_LAUNCH("vqa_transpose_hw_c");
    return VQA_OK;
}

extern "C" int vqa_fc_grad_from_nhwc(const float* g, float* out, int32_t F, int32_t P, void* stream) {
    if (!g || !out || F <= 0 || P <= 0 || P % 64 != 0) { vqa_set_error("vqa_fc_grad_from_nhwc: bad argument"); return VQA_EINVAL; }
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(P / 64, F), dim3(256), 0, (hipStream_t)stream, g, out, P);
    CHECK_LAUNCH("vqa_fc_grad_from_nhwc");
    return VQA_OK;
}

__global__ void cast_pad_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int rows, int cols,
                                int rows_pad, int cols_pad) {
    const int64_t total = (int64_t)rows_pad * cols_pad;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = idx / cols_pad, c = idx - (int64_t)r * cols_pad;
        out[idx] = (r < rows && c < cols) ? f2bf(w[(int64_t)r * cols + c]) : (bf16_t)0;
    }
}

extern "C" int vqa_cast_pad(const float* w, void* out, int32_t rows, int32_t cols, int32_t rows_pad,
                            int32_t cols_pad, void* stream) {
    if (!w || !out || rows <= 0 || cols <= 0 || rows_pad < rows || cols_pad < cols) {
        vqa_set_error("vqa_cast_pad: bad argument");
        return VQA_EINVAL;
    }
    const int64_t total = (int64_t)rows_pad * cols_pad;
    hipLaunchKernelGGL(cast_pad_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0,
                       (hipStream_t)stream, w, (bf16_t*)out, rows, cols, rows_pad, cols_pad);
    CHECK_LAUNCH("vqa_cast_pad");
    return VQA_OK;
}

// out[row][col] = act(sum_s acc[s][row][col] + bias[col]); the sum over s folds split-K slabs
__global__ void bias_act_kernel(const float* __restrict__ acc, const float* __restrict__ bias,
                                bf16_t* __restrict__ out, int rows, int cols, int64_t ld_acc, int64_t ld_out,
                                int relu, int slabs, int64_t slab_stride) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = idx / cols, c = idx - (int64_t)r * cols;
        float v = bias ? bias[c] : 0.f;
        for (int s = 0; s < slabs; s++) v += acc[s * slab_stride + r * ld_acc + c];
        if (relu) v = fmaxf(v, 0.f);
        out[r * ld_out + c] = f2bf(v);
    }
}
__global__ void bias_act_f32_kernel(const float* __restrict__ acc, const float* __restrict__ bias,
                                    float* __restrict__ out, int rows, int cols, int64_t ld_acc, int64_t ld_out,
                                    int relu, int slabs, int64_t slab_stride) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = idx / cols, c = idx - (int64_t)r * cols;
        float v = bias ? bias[c] : 0.f;
        for (int s = 0; s < slabs; s++) v += acc[s * slab_stride + r * ld_acc + c];
        if (relu) v = fmaxf(v, 0.f);
        out[r * ld_out + c] = v;
    }
}

extern "C" int vqa_bias_act(const float* acc, const float* bias, void* out, int32_t rows, int32_t cols,
                            int64_t ld_acc, int64_t ld_out, int32_t relu, int32_t slabs,
                            int64_t slab_stride, int32_t out_f32, void* stream) {
    if (!acc || !out || rows <= 0 || cols <= 0 || slabs < 1) { vqa_set_error("vqa_bias_act: bad argument"); return VQA_EINVAL; }
    const int64_t total = (int64_t)rows * cols;
    dim3 grid((int)std::min<int64_t>((total + 255) / 256, 4096));
    if (out_f32)
        hipLaunchKernelGGL(bias_act_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, acc, bias, (float*)out,
                           rows, cols, ld_acc, ld_out, relu, slabs, slab_stride);
    else
        hipLaunchKernelGGL(bias_act_kernel, grid, dim3(256), 0, (hipStream_t)stream, acc, bias, (bf16_t*)out,
                           rows, cols, ld_acc, ld_out, relu, slabs, slab_stride);
    CHECK_LAUNCH("vqa_bias_act");
    return VQA_OK;
}

// out[col] += sum_rows x[row][col]; block = 32 columns x 32 row lanes, one block per column range:
// the sum of a column is formed in one fixed order (no atomics), the same bits on every run
__global__ __launch_bounds__(1024) void col_sum_kernel(const void* __restrict__ x, float* __restrict__ out, int rows,
                                                       int cols, int64_t ld, int is_bf16) {
    __shared__ float red[1024];
    const int col = blockIdx.x * 32 + (threadIdx.x & 31), part = threadIdx.x >> 5;
    float s = 0.f;
    if (col < cols)
        for (int r = part; r < rows; r += 32)
            s += is_bf16 ? bf2f(((const bf16_t*)x)[r * ld + col]) : ((const float*)x)[r * ld + col];
    red[threadIdx.x] = s;
    __syncthreads();
    if (part == 0 && col < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; k++) t += red[threadIdx.x + 32 * k];
        out[col] += t;
    }
}

extern "C" int vqa_col_sum(const void* x, float* out, int32_t rows, int32_t cols, int64_t ld, int32_t is_bf16,
                           void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0) { vqa_set_error("vqa_col_sum: bad argument"); return VQA_EINVAL; }
    hipLaunchKernelGGL(col_sum_kernel, dim3((cols + 31) / 32), dim3(1024), 0, (hipStream_t)stream, x, out, rows, cols, ld,
                       is_bf16);
    CHECK_LAUNCH("vqa_col_sum");
    return VQA_OK;
}

// dst slot d <- (sum over src slots offsets[d]..offsets[d+1]) * (mask slot > 0), 8 bf16 per lane
__global__ __launch_bounds__(256) void slot_sum_mask_kernel(bf16_t* __restrict__ arena, const int32_t* __restrict__ offsets,
                                                            const int32_t* __restrict__ src_slots,
                                                            const int32_t* __restrict__ dst_slots,
                                                            const int32_t* __restrict__ mask_slots, int64_t slot_elems) {
    const int d = blockIdx.y;
    const int beg = offsets[d], end = offsets[d + 1];
    bf16_t* dst = arena + (int64_t)dst_slots[d] * slot_elems;
    const bf16_t* mk = arena + (int64_t)mask_slots[d] * slot_elems;
    typedef __attribute__((ext_vector_type(8))) uint16_t u16x8;
    for (int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; e < slot_elems;
         e += (int64_t)gridDim.x * blockDim.x * 8) {
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = beg; k < end; k++) {
            const u16x8 v = *(const u16x8*)(arena + (int64_t)src_slots[k] * slot_elems + e);
#pragma unroll
            for (int j = 0; j < 8; j++) s[j] += bf2f(v[j]);
        }
        const u16x8 m = *(const u16x8*)(mk + e);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = ((int16_t)m[j] > 0) ? f2bf(s[j]) : (bf16_t)0;
        *(u16x8*)(dst + e) = o;
    }
}

extern "C" int vqa_slot_sum_mask(void* arena, const int32_t* offsets, const int32_t* src_slots,
                                 const int32_t* dst_slots, const int32_t* mask_slots, int32_t n_dst,
                                 int64_t slot_elems, void* stream) {
    if (!arena || !offsets || !src_slots || !dst_slots || !mask_slots || n_dst <= 0 || slot_elems % 8 != 0) {
        vqa_set_error("vqa_slot_sum_mask: bad argument");
        return VQA_EINVAL;
    }
    const int bx = (int)std::min<int64_t>((slot_elems / 8 + 255) / 256, 64);
    hipLaunchKernelGGL(slot_sum_mask_kernel, dim3(bx, n_dst), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)arena, offsets, src_slots, dst_slots, mask_slots, slot_elems);
    CHECK_LAUNCH("vqa_slot_sum_mask");
    return VQA_OK;
}
