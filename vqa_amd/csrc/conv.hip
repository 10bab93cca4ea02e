// Grouped implicit-GEMM convolution over 14x14 feature-map tiles (gfx950, bf16 MFMA).
//
// One workgroup (4 waves) = one tile = one feature map (14 image rows, each one MFMA column tile of
// 16 lanes of which 14 are pixels: 16 consecutive pixels would straddle image rows and make every
// LDS fragment read a 2-way bank conflict) x 128 output channels; wave w owns output channels
// [32w, 32w+32).
//   * pixels: for each 128-channel source chunk the map is staged once into an LDS halo image
//     (16 x 16 positions, zero border, 288-byte position stride); the 9 taps are 9 address offsets
//     into that image, so the im2col matrix never exists
//   * weights: every wave streams its own fragments straight from L2 into registers, one
//     coalesced 1 KiB load per 16 x 32 fragment, 4 K-steps ahead; the packed layout is
//     fragment-major so no LDS, no barrier and no bank conflict is spent on them
//   * the K loop therefore has no workgroup barrier at all; LDS holds only the halo (74 KiB), two
//     workgroups share a CU, and one's prologue/epilogue memory latency hides under the other's MFMAs
// Weights are the MFMA A operand (rows = output channels), pixels the B operand, so each lane ends
// with 4 consecutive output channels of one pixel and stores 8 bytes.
//
// Replaces the reference's batch-1 nn.Conv2d calls (Layers.py:33-36, Module_Model.py:38-40,47-50),
// its stem (Module_Model.py:59-62) and classifier 1x1 conv (:72-75).
#include "vqa_dev.h"
#include "plan.h"
#include "prof.h"

#define HALO_STRIDE 288
#define HALO_POS 264
#define HALO_BYTES (HALO_POS * HALO_STRIDE)   // 76032
#define CONV_LDS HALO_BYTES
#define MT 14                                 // pixel tiles = image rows

template <int TAPS>
__global__ __launch_bounds__(256, 2) void conv_tiles_kernel(vqa_conv_params p) {
    extern __shared__ __attribute__((aligned(16))) char halo[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const vqa_conv_tile t = p.tiles[blockIdx.x];
    const int ntile = blockIdx.y;

    // this wave's weight stream: K-step s (32 channels of one tap) is two adjacent 1 KiB blocks
    const int64_t wstep = (int64_t)(p.n_out / 16) * 1024;
    const char* wbase = (const char*)p.w + ((int64_t)t.wgroup * p.w_group_stride) * 2 +
                        (int64_t)(ntile * 8 + wave * 2) * 1024 + lane * 16;
    const int nsteps = p.k_chunks * TAPS * 4;
    bf16x8 wr[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        wr[i][0] = *(const bf16x8*)(wbase + i * wstep);
        wr[i][1] = *(const bf16x8*)(wbase + i * wstep + 1024);
    }

    // zero the border of the halo image (interior positions are overwritten by every chunk)
    for (int idx = tid; idx < HALO_POS * 16; idx += 256) {
        const int pos = idx >> 4, row = pos >> 4, col = pos & 15;
        if (row == 0 || row >= 15 || col == 0 || col == 15)
            *(uint4*)(halo + pos * HALO_STRIDE + (idx & 15) * 16) = make_uint4(0, 0, 0, 0);
    }

    f32x4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // LDS byte offset of this lane's pixel in every pixel tile, at tap (-1,-1), K-chunk lq
    int xaddr[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        xaddr[m] = (m * 16 + l15) * HALO_STRIDE + lq * 16;   // lanes 14, 15 read halo padding; discarded
    }

    int s = 0;
    for (int c = 0; c < p.k_chunks; c++) {
        {
            int slot, choff;
            if (p.pair_chunks > 0) { slot = (c / p.pair_chunks) ? t.src1 : t.src0; choff = (c % p.pair_chunks) * 128; }
            else { slot = t.src0; choff = c * 128; }
            const char* sp = (const char*)p.src + ((int64_t)slot * p.src_slot_stride + choff) * 2;
            __syncthreads();   // every wave is done reading the previous chunk's image
            // 196 pixels x 16 pieces of 16 B, staged through registers 7 pieces per thread at a time
            // (more in flight would not fit beside the accumulators at two workgroups per CU)
#pragma unroll 1
            for (int i0 = 0; i0 < 14; i0 += 7) {
                uint4 v[7];
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    const int e = tid + 256 * (i0 + i), pix = min(e >> 4, VQA_HW - 1), pc = e & 15;
                    v[i] = *(const uint4*)(sp + (int64_t)pix * p.src_pix_stride * 2 + pc * 16);
                }
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    const int e = tid + 256 * (i0 + i), pix = e >> 4, pc = e & 15;
                    if (pix < VQA_HW) {
                        const int y = pix / 14, x = pix - y * 14;
                        *(uint4*)(halo + ((y + 1) * 16 + x + 1) * HALO_STRIDE + pc * 16) = v[i];
                    }
                }
            }
            __syncthreads();
        }
        // K loop of this chunk: taps x 4 K-steps x 14 pixel tiles.  The pixel fragments run through
        // an 8-deep register ring, read 7 tiles ahead of the MFMAs that use them, so that LDS latency
        // is covered even when this wave is alone on its SIMD.
        bf16x8 xq[8];
        {
            const char* x0 = halo + ((TAPS == 9) ? 0 : 17) * HALO_STRIDE;
#pragma unroll
            for (int q = 0; q < 7; q++) xq[q] = *(const bf16x8*)(x0 + xaddr[q]);
        }
#pragma unroll 1
        for (int tap = 0; tap < TAPS; tap++) {
            const int tapoff = (TAPS == 9) ? ((tap / 3) * 16 + (tap % 3)) : 17;
            const int tn = (tap + 1 < TAPS) ? tap + 1 : tap;      // last tap: harmless re-read
            const int tapoff_n = (TAPS == 9) ? ((tn / 3) * 16 + (tn % 3)) : 17;
            const char* xb = halo + tapoff * HALO_STRIDE;
            const char* xn = halo + tapoff_n * HALO_STRIDE;
#pragma unroll
            for (int ks = 0; ks < 4; ks++, s++) {
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    const int q = ks * MT + m;            // 0..55 within the tap
                    const int qa = q + 7;                 // fragment to fetch now
                    const int ka = (qa / MT) & 3, ma = qa % MT;
                    xq[qa & 7] = *(const bf16x8*)((qa < 4 * MT ? xb : xn) + xaddr[ma] + ka * 64);
                    acc[m][0] = mma16(wr[ks][0], xq[q & 7], acc[m][0]);
                    acc[m][1] = mma16(wr[ks][1], xq[q & 7], acc[m][1]);
                }
                // pin the issue order to one LDS read, two MFMAs, ... : left alone the scheduler
                // collapses the ring and waits on every read right after issuing it
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
                // refill this ring slot for K-step s+4 (three K-steps of MFMAs hide the L2 latency)
                const int sn = min(s + 4, nsteps - 1);
                wr[ks][0] = *(const bf16x8*)(wbase + sn * wstep);
                wr[ks][1] = *(const bf16x8*)(wbase + sn * wstep + 1024);
                __builtin_amdgcn_sched_barrier(0);   // keep the refill here: sunk to the loop end it is waited on at once
            }
        }
    }

    // epilogue: lane holds pixel (m, l15), channels ntile*128 + wave*32 + n*16 + lq*4 + 0..3
    if (l15 >= 14) return;
    int dslot, mslot, choff;
    if (p.pair_out > 0) {
        const int which = ntile / p.pair_out;
        dslot = which ? t.dst1 : t.dst0;
        mslot = which ? t.mask1 : t.mask0;
        choff = (ntile % p.pair_out) * 128;
    } else { dslot = t.dst0; mslot = t.mask0; choff = ntile * 128; }
    if (dslot < 0) return;
    const int64_t dbase = (int64_t)dslot * p.dst_slot_stride + choff;
    const int64_t rbase = (int64_t)t.res * p.dst_slot_stride + choff;
    const int64_t mbase = (int64_t)mslot * p.dst_slot_stride + choff;
    const bool has_res = t.res >= 0, has_mask = mslot >= 0;
    const bool has_bias = (p.flags & VQA_CONV_BIAS) != 0, relu = (p.flags & VQA_CONV_RELU) != 0;
    const float* bias = p.bias + (int64_t)t.wgroup * p.n_out + ntile * 128;
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int ch = wave * 32 + n * 16 + lq * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_bias) bv = *(const float4*)(bias + ch);
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const int pix = m * 14 + l15;
            const int64_t off = (int64_t)pix * p.dst_pix_stride + ch;
            float o0 = acc[m][n][0] + bv.x, o1 = acc[m][n][1] + bv.y;
            float o2 = acc[m][n][2] + bv.z, o3 = acc[m][n][3] + bv.w;
            if (has_res) {
                const u16x4 r = *(const u16x4*)((const bf16_t*)p.res + rbase + off);
                o0 += bf2f(r[0]); o1 += bf2f(r[1]); o2 += bf2f(r[2]); o3 += bf2f(r[3]);
            }
            if (relu) { o0 = fmaxf(o0, 0.f); o1 = fmaxf(o1, 0.f); o2 = fmaxf(o2, 0.f); o3 = fmaxf(o3, 0.f); }
            if (has_mask) {
                const u16x4 k = *(const u16x4*)((const bf16_t*)p.mask + mbase + off);
                // bf16 bits > 0 as a signed value <=> positive finite/inf (zero and negatives fail)
                o0 = ((int16_t)k[0] > 0) ? o0 : 0.f; o1 = ((int16_t)k[1] > 0) ? o1 : 0.f;
                o2 = ((int16_t)k[2] > 0) ? o2 : 0.f; o3 = ((int16_t)k[3] > 0) ? o3 : 0.f;
            }
            u16x4 o = {f2bf(o0), f2bf(o1), f2bf(o2), f2bf(o3)};
            *(u16x4*)((bf16_t*)p.dst + dbase + off) = o;
        }
    }
}

extern "C" int vqa_conv_tiles(const vqa_conv_params* p, int32_t taps, void* stream) {
    if (!p || !p->src || !p->dst || !p->w || !p->tiles || (taps != 9 && taps != 1) ||
        p->k_chunks <= 0 || p->n_out <= 0 || p->n_out % 128 != 0 || p->num_tiles < 0 ||
        p->src_pix_stride % 8 != 0 || p->dst_pix_stride % 4 != 0 ||
        ((p->flags & VQA_CONV_BIAS) && !p->bias)) {
        vqa_set_error("vqa_conv_tiles: bad argument");
        return VQA_EINVAL;
    }
    if (p->num_tiles == 0) return VQA_OK;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_tiles_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS);
        (void)hipFuncSetAttribute((const void*)conv_tiles_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS);
        attr_set = true;
    }
    dim3 grid(p->num_tiles, p->n_out / 128), block(256);
    const double kdim = (double)p->k_chunks * 128 * taps;
    VqaProfScope prof(taps == 9 ? VQA_K_CONV9 : VQA_K_CONV1, 2.0 * VQA_HW * kdim * p->n_out * p->num_tiles,
                      (double)p->num_tiles * VQA_HW * 2.0 * (p->k_chunks * 128 + p->n_out), (hipStream_t)stream);
    if (taps == 9) hipLaunchKernelGGL(conv_tiles_kernel<9>, grid, block, CONV_LDS, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL(conv_tiles_kernel<1>, grid, block, CONV_LDS, (hipStream_t)stream, *p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_conv_tiles: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    return VQA_OK;
}

// ---------------------------------------------------------------------------------------------
// Weight packing: fp32 strided tensor -> bf16 MFMA fragments in the order the kernel streams
// them: [group][chunk][tap][K-step of 32][n / 16][lane 0..63][8], where lane l of a fragment holds
// W[n = 16*(n/16) + (l & 15)][k = 128*chunk + 32*step + 8*(l >> 4) + 0..7].
// ---------------------------------------------------------------------------------------------
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ out,
                                        int n_out, int k_in, int kpad, int taps, int64_t g_stride,
                                        int64_t n_stride, int64_t k_stride, int flip,
                                        int64_t per_group) {
    const int g = blockIdx.y;
    const int nb16 = n_out / 16;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per_group;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int j = idx & 7, l = (idx >> 3) & 63;
        int64_t r = idx >> 9;
        const int nt = r % nb16; r /= nb16;
        const int ks = r & 3; r >>= 2;
        const int t = r % taps;
        const int c = r / taps;
        const int n = nt * 16 + (l & 15);
        const int k = c * 128 + ks * 32 + 8 * (l >> 4) + j;
        float v = 0.f;
        if (k < k_in) v = w[g * g_stride + n * n_stride + k * k_stride + (flip ? taps - 1 - t : t)];
        out[g * per_group + idx] = f2bf(v);
    }
    (void)kpad;
}

extern "C" int vqa_pack_conv_weight(const float* w, void* packed, int32_t groups, int32_t n_out,
                                    int32_t k_in, int32_t taps, int64_t g_stride, int64_t n_stride,
                                    int64_t k_stride, int32_t flip, void* stream) {
    if (!w || !packed || groups <= 0 || n_out <= 0 || n_out % 128 != 0 || k_in <= 0 || taps <= 0) {
        vqa_set_error("vqa_pack_conv_weight: bad argument");
        return VQA_EINVAL;
    }
    const int kpad = (k_in + 127) / 128 * 128;
    const int64_t per_group = (int64_t)kpad * taps * n_out;
    int blocks = (int)std::min<int64_t>((per_group + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks, groups), dim3(256), 0, (hipStream_t)stream,
                       w, (bf16_t*)packed, n_out, k_in, kpad, taps, g_stride, n_stride, k_stride, flip, per_group);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_pack_conv_weight: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    return VQA_OK;
}

// Resident workgroups per CU the runtime reports for the 3x3 kernel (diagnostics; advisory).
extern "C" int vqa_debug_conv_occupancy(void) {
    int nb = -1;
    (void)hipFuncSetAttribute((const void*)conv_tiles_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)conv_tiles_kernel<9>, 256, CONV_LDS) != hipSuccess) return -1;
    return nb;
}
