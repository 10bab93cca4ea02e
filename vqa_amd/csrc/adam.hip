// Adam update of the executor's parameters (Train_model.py:165-167, 315: torch.optim.Adam with L2
// weight decay), HBM-bound: 28 bytes of traffic per parameter.
//
//   g  = grad + weight_decay * p
//   m  = m + (1 - beta1) * (g - m)
//   v  = beta2 * v + (1 - beta2) * g * g
//   p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
//
// Two kernels.  vqa_adam_tensors updates any number of tensors in one launch from a device table
// (tensors whose token was not used in the batch have no gradient and are simply not in the table:
// the reference's `grad is None` skip; each tensor carries its own step count).  vqa_adam_fc1
// updates the first linear layer's weight - 88 % of the network's parameters - and in the same pass
// writes the bf16, channels-last copy the GEMM reads (Layers.py:42-44 flattens c*196+hw, the
// executor's maps are hw*P+c), which otherwise costs a second trip over the 411 MB tensor every step.
#include "vqa_dev.h"
#include "plan.h"

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float wd, float b1c, float b2,
                                      float b2c, float step_size, float inv_bc2_sqrt, float eps) {
    g += wd * p;
    m += b1c * (g - m);
    v = b2 * v + b2c * g * g;
    p -= step_size * m / (sqrtf(v) * inv_bc2_sqrt + eps);
}

// streaming accesses: every element is touched once per step, so it should not displace the weights
// and tile lists the next kernels find in L2
typedef float f32x4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntload4(const float* a) {
    const f32x4n v = __builtin_nontemporal_load((const f32x4n*)a);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void ntstore4(float* a, float4 x) {
    __builtin_nontemporal_store((f32x4n){x.x, x.y, x.z, x.w}, (f32x4n*)a);
}

// chunk c: elements [offset, offset + 4096) of tensor chunks[c].tensor
__global__ __launch_bounds__(256) void adam_tensors_kernel(const vqa_adam_tensor* __restrict__ tensors,
                                                           const vqa_adam_chunk* __restrict__ chunks, float wd,
                                                           float beta1, float beta2, float eps) {
    const vqa_adam_chunk ck = chunks[blockIdx.x];
    const vqa_adam_tensor t = tensors[ck.tensor];
    const float b1c = 1.f - beta1, b2c = 1.f - beta2;
    const int64_t end = min(ck.offset + (int64_t)4096, t.numel);
    if ((t.numel & 3) == 0) {       // 16-byte accesses
        for (int64_t e = ck.offset + threadIdx.x * 4; e < end; e += 1024) {
            float4 p = ntload4(t.p + e), m = ntload4(t.m + e), v = ntload4(t.v + e);
            const float4 g = ntload4(t.g + e);
            adam1(p.x, g.x, m.x, v.x, wd, b1c, beta2, b2c, t.step_size, t.inv_bc2_sqrt, eps);
            adam1(p.y, g.y, m.y, v.y, wd, b1c, beta2, b2c, t.step_size, t.inv_bc2_sqrt, eps);
            adam1(p.z, g.z, m.z, v.z, wd, b1c, beta2, b2c, t.step_size, t.inv_bc2_sqrt, eps);
            adam1(p.w, g.w, m.w, v.w, wd, b1c, beta2, b2c, t.step_size, t.inv_bc2_sqrt, eps);
            ntstore4(t.p + e, p); ntstore4(t.m + e, m); ntstore4(t.v + e, v);
        }
    } else {
        for (int64_t e = ck.offset + threadIdx.x; e < end; e += 256) {
            float p = t.p[e], m = t.m[e], v = t.v[e];
            adam1(p, t.g[e], m, v, wd, b1c, beta2, b2c, t.step_size, t.inv_bc2_sqrt, eps);
            t.p[e] = p; t.m[e] = m; t.v[e] = v;
        }
    }
}

extern "C" int vqa_adam_tensors(const vqa_adam_tensor* tensors, const vqa_adam_chunk* chunks, int32_t num_chunks,
                                float weight_decay, float beta1, float beta2, float eps, void* stream) {
    if (!tensors || !chunks || num_chunks < 0) { vqa_set_error("vqa_adam_tensors: bad argument"); return VQA_EINVAL; }
    if (num_chunks == 0) return VQA_OK;
    hipLaunchKernelGGL(adam_tensors_kernel, dim3(num_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunks,
                       weight_decay, beta1, beta2, eps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_adam_tensors: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    return VQA_OK;
}

// First linear layer: p, g, m, v are [F][P][196] fp32 (the reference's layout); packed is bf16
// [Fpad][196][P].  Block = one output row f x 64 channels: a contiguous [64][196] tile of each array
// (16-byte accesses), the updated weights also rounded to bf16 into an LDS tile [channel][198] and
// written out pixel-major, 16 bytes per lane - the layout pass of vqa_nchw_to_nhwc riding along.
__global__ __launch_bounds__(256) void adam_fc1_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       bf16_t* __restrict__ packed, int P, float wd, float beta1,
                                                       float beta2, float eps, float step_size, float inv_bc2_sqrt) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 198];
    const int f = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
    const int64_t base = ((int64_t)f * P + c0) * VQA_HW;
    const float b1c = 1.f - beta1, b2c = 1.f - beta2;
#pragma unroll 1
    for (int i0 = 0; i0 < 13; i0 += 4) {
        float4 pv[4], gv[4], mv[4], vv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int q = min(tid + 256 * (i0 + i), 64 * 49 - 1);       // float4 index in the tile
            pv[i] = ntload4(p + base + q * 4); gv[i] = ntload4(g + base + q * 4);
            mv[i] = ntload4(m + base + q * 4); vv[i] = ntload4(v + base + q * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int q = tid + 256 * (i0 + i);
            if (i0 + i < 13 && q < 64 * 49) {
                adam1(pv[i].x, gv[i].x, mv[i].x, vv[i].x, wd, b1c, beta2, b2c, step_size, inv_bc2_sqrt, eps);
                adam1(pv[i].y, gv[i].y, mv[i].y, vv[i].y, wd, b1c, beta2, b2c, step_size, inv_bc2_sqrt, eps);
                adam1(pv[i].z, gv[i].z, mv[i].z, vv[i].z, wd, b1c, beta2, b2c, step_size, inv_bc2_sqrt, eps);
                adam1(pv[i].w, gv[i].w, mv[i].w, vv[i].w, wd, b1c, beta2, b2c, step_size, inv_bc2_sqrt, eps);
                ntstore4(p + base + q * 4, pv[i]); ntstore4(m + base + q * 4, mv[i]);
                ntstore4(v + base + q * 4, vv[i]);
                const int c = q / 49, pix = (q - c * 49) * 4;
                uint32_t* t = (uint32_t*)(tile + c * 198 + pix);
                t[0] = cvt_pk_bf16((f32x2){pv[i].x, pv[i].y});
                t[1] = cvt_pk_bf16((f32x2){pv[i].z, pv[i].w});
            }
        }
    }
    __syncthreads();
    bf16_t* os = packed + (int64_t)f * VQA_HW * P + c0;
    for (int idx = tid; idx < VQA_HW * 8; idx += 256) {
        const int cg = (idx / VQA_HW) * 8, pix = idx - (idx / VQA_HW) * VQA_HW;
        const bf16_t* t = tile + cg * 198 + pix;
        uint4 o;
        o.x = (uint32_t)t[0] | ((uint32_t)t[198] << 16);
        o.y = (uint32_t)t[2 * 198] | ((uint32_t)t[3 * 198] << 16);
        o.z = (uint32_t)t[4 * 198] | ((uint32_t)t[5 * 198] << 16);
        o.w = (uint32_t)t[6 * 198] | ((uint32_t)t[7 * 198] << 16);
        *(uint4*)(os + (int64_t)pix * P + cg) = o;
    }
}

extern "C" int vqa_adam_fc1(float* p, const float* g, float* m, float* v, void* packed, int32_t F, int32_t P,
                            float weight_decay, float beta1, float beta2, float eps, float step_size,
                            float inv_bc2_sqrt, void* stream) {
    if (!p || !g || !m || !v || !packed || F <= 0 || P <= 0 || P % 64 != 0) {
        vqa_set_error("vqa_adam_fc1: bad argument");
        return VQA_EINVAL;
    }
    hipLaunchKernelGGL(adam_fc1_kernel, dim3(P / 64, F), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (bf16_t*)packed, P, weight_decay, beta1, beta2, eps, step_size, inv_bc2_sqrt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_adam_fc1: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    return VQA_OK;
}

// Convolution weights [cout][cin][taps] fp32 (torch's OIHW with the taps flattened): the Adam update
// and, in the same pass, the two bf16 fragment-major copies the convolution kernels stream
// (vqa_pack_conv_weight's layout): the forward one (n = output channel, k = input channel) and the
// input-gradient one (n = input channel, k = output channel, taps flipped).  Re-packing every
// weight after every optimizer step otherwise costs a dozen launches and a second trip over them.
// Block = one tensor x 8 output channels: a contiguous [8][cin * taps] slice of p, g, m, v
// (16-byte streaming accesses); the updated slice also goes to LDS as bf16, from where each thread
// assembles 16-byte fragment pieces: 8 consecutive input channels of one (output channel, tap) for
// the forward copy, the 8 output channels of one (input channel, tap) for the other.
__global__ __launch_bounds__(256) void adam_conv_pack_kernel(const vqa_adam_conv_tensor* __restrict__ tensors, int cout,
                                                             int cin, int taps, float wd, float beta1, float beta2,
                                                             float eps) {
    extern __shared__ __attribute__((aligned(16))) bf16_t ctile[];     // [8][cin * taps + 8]
    const vqa_adam_conv_tensor t = tensors[blockIdx.y];
    const int o0 = blockIdx.x * 8, tid = threadIdx.x;
    const int row = cin * taps, ld = row + 8;
    const int64_t base = (int64_t)o0 * row;
    const int nq = 2 * row;                       // float4 pieces of the slice (8 * row / 4)
    const float b1c = 1.f - beta1, b2c = 1.f - beta2;
#pragma unroll 1
    for (int q0 = 0; q0 < nq; q0 += 3 * 256) {
        float4 pv[3], gv[3], mv[3], vv[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int q = min(q0 + 256 * i + tid, nq - 1);
            pv[i] = ntload4(t.p + base + q * 4); gv[i] = ntload4(t.g + base + q * 4);
            mv[i] = ntload4(t.m + base + q * 4); vv[i] = ntload4(t.v + base + q * 4);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int q = q0 + 256 * i + tid;
            if (q < nq) {
                adam1(pv[i].x, gv[i].x, mv[i].x, vv[i].x, wd, b1c, beta2, b2c, t.step_size, t.inv_bc2_sqrt, eps);
                adam1(pv[i].y, gv[i].y, mv[i].y, vv[i].y, wd, b1c, beta2, b2c, t.step_size, t.inv_bc2_sqrt, eps);
                adam1(pv[i].z, gv[i].z, mv[i].z, vv[i].z, wd, b1c, beta2, b2c, t.step_size, t.inv_bc2_sqrt, eps);
                adam1(pv[i].w, gv[i].w, mv[i].w, vv[i].w, wd, b1c, beta2, b2c, t.step_size, t.inv_bc2_sqrt, eps);
                ntstore4(t.p + base + q * 4, pv[i]); ntstore4(t.m + base + q * 4, mv[i]);
                ntstore4(t.v + base + q * 4, vv[i]);
                const int e = q * 4, ol = e / row, r = e - ol * row;     // row % 4 == 0: the four stay in one row
                uint32_t* d = (uint32_t*)(ctile + ol * ld + r);
                d[0] = cvt_pk_bf16((f32x2){pv[i].x, pv[i].y});
                d[1] = cvt_pk_bf16((f32x2){pv[i].z, pv[i].w});
            }
        }
    }
    __syncthreads();
    // forward copy: element (n = o, k = ci, tap) sits at
    // [ci / 128][tap][(ci % 128) / 32][o / 16][lane = o % 16 + 16 * ((ci % 32) / 8)][ci % 8]
    if (t.pk_fwd) {
        const int nb16 = cout >> 4;
        for (int u = tid; u < row; u += 256) {
            const int ol = u & 7, w = u >> 3;                 // w enumerates (tap, group of 8 input channels)
            const int cig = w % (cin >> 3), tap = w / (cin >> 3);
            const int o = o0 + ol;
            const bf16_t* s = ctile + ol * ld + cig * 8 * taps + tap;
            uint4 v;
            v.x = (uint32_t)s[0] | ((uint32_t)s[taps] << 16);
            v.y = (uint32_t)s[2 * taps] | ((uint32_t)s[3 * taps] << 16);
            v.z = (uint32_t)s[4 * taps] | ((uint32_t)s[5 * taps] << 16);
            v.w = (uint32_t)s[6 * taps] | ((uint32_t)s[7 * taps] << 16);
            const int chunk = cig >> 4, ks = (cig >> 2) & 3, kq = cig & 3;
            const int64_t idx = ((((int64_t)(chunk * taps + tap) * 4 + ks) * nb16 + (o >> 4)) * 64 + (o & 15) + 16 * kq) * 8;
            *(uint4*)((bf16_t*)t.pk_fwd + idx) = v;
        }
    }
    // input-gradient copy: element (n = ci, k = o, flipped tap) sits at
    // [o / 128][taps - 1 - tap][(o % 128) / 32][ci / 16][lane = ci % 16 + 16 * ((o % 32) / 8)][o % 8]
    if (t.pk_bwd) {
        const int nb16 = cin >> 4;
        const int chunk = o0 >> 7, ks = (o0 >> 5) & 3, kq = (o0 >> 3) & 3;
        for (int u = tid; u < row; u += 256) {
            const int ci = u % cin, tap = u / cin;
            const bf16_t* s = ctile + ci * taps + tap;
            uint4 v;
            v.x = (uint32_t)s[0] | ((uint32_t)s[ld] << 16);
            v.y = (uint32_t)s[2 * ld] | ((uint32_t)s[3 * ld] << 16);
            v.z = (uint32_t)s[4 * ld] | ((uint32_t)s[5 * ld] << 16);
            v.w = (uint32_t)s[6 * ld] | ((uint32_t)s[7 * ld] << 16);
            const int64_t idx = ((((int64_t)(chunk * taps + (taps - 1 - tap)) * 4 + ks) * nb16 + (ci >> 4)) * 64 + (ci & 15) + 16 * kq) * 8;
            *(uint4*)((bf16_t*)t.pk_bwd + idx) = v;
        }
    }
}

extern "C" int vqa_adam_conv_pack(const vqa_adam_conv_tensor* tensors, int32_t num_tensors, int32_t cout, int32_t cin,
                                  int32_t taps, float weight_decay, float beta1, float beta2, float eps, void* stream) {
    if (!tensors || num_tensors < 0 || cout <= 0 || cout % 128 != 0 || cin <= 0 || cin % 128 != 0 || taps <= 0 ||
        (int64_t)(cin * taps + 8) * 16 > 160 * 1024) {
        vqa_set_error("vqa_adam_conv_pack: bad argument (channel counts must be multiples of 128; 8 x cin x taps bf16 must fit in LDS)");
        return VQA_EINVAL;
    }
    if (num_tensors == 0) return VQA_OK;
    const int lds = (cin * taps + 8) * 8 * 2;
    if (lds > 64 * 1024 && !vqa_allow_lds((const void*)adam_conv_pack_kernel, 160 * 1024, "vqa_adam_conv_pack")) return VQA_ELAUNCH;
    hipLaunchKernelGGL(adam_conv_pack_kernel, dim3(cout / 8, num_tensors), dim3(256), lds, (hipStream_t)stream, tensors,
                       cout, cin, taps, weight_decay, beta1, beta2, eps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_adam_conv_pack: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    return VQA_OK;
}
