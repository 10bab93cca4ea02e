// Grouped implicit-GEMM convolution over 14x14 feature-map tiles (gfx950, bf16 MFMA).
//
// One workgroup (4 waves) = one tile = one feature map (196 pixels, computed as 14 image rows of
// 16 lanes) x 128 output channels.  K loop: for each 128-channel source chunk the map is staged
// once into an LDS halo image (16x16 positions, zero border, 288-B position stride so that both
// the row reads below are bank-conflict free); for each tap and each 64-channel half of the chunk
// a 128 x 64 weight slab (16 KiB) is DMA-copied global->LDS, double buffered, one barrier per
// slab.  Weights are the MFMA A operand (rows = output channels), pixels the B operand, so each
// lane ends with 4 consecutive output channels of one pixel and stores 8 bytes.
//
// Replaces the reference's batch-1 nn.Conv2d calls (Layers.py:33-36, Module_Model.py:38-40,47-50),
// its stem (Module_Model.py:59-62) and classifier 1x1 conv (:72-75).
#include "vqa_dev.h"
#include "plan.h"
#include "prof.h"

#define HALO_STRIDE 288
#define HALO_POS 264
#define HALO_BYTES (HALO_POS * HALO_STRIDE)   // 76032
#define WSLAB_BYTES 16384
#define CONV_LDS (HALO_BYTES + 2 * WSLAB_BYTES)

template <int TAPS>
__global__ __launch_bounds__(256) void conv_tiles_kernel(vqa_conv_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* wbuf = smem + HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const vqa_conv_tile t = p.tiles[blockIdx.x];
    const int ntile = blockIdx.y;

    // weight slab s of this tile's group: slabs are [chunk][tap][half] major, each n_out x 64 bf16
    const char* wsrc = (const char*)p.w + ((int64_t)t.wgroup * p.w_group_stride) * 2 +
                       (int64_t)ntile * 128 * 128;
    const int64_t slab_stride = (int64_t)p.n_out * 128;
    const int nslabs = p.k_chunks * TAPS * 2;
    auto issue_slab = [&](int s) {
        const char* g = wsrc + (int64_t)s * slab_stride + lane * 16;
        char* l = wbuf + (s & 1) * WSLAB_BYTES;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int piece = wave * 4 + i;
            dma16(g + piece * 1024, l + piece * 1024);
        }
    };
    issue_slab(0);

    for (int i = tid; i < HALO_BYTES / 16; i += 256) ((uint4*)halo)[i] = make_uint4(0, 0, 0, 0);

    f32x4 acc[7][4];
#pragma unroll
    for (int m = 0; m < 7; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane LDS byte offsets: pixel column l15 of image row (7*wm + m) at tap (-1,-1)
    const int xbase = ((wm * 7) * 16 + l15) * HALO_STRIDE + lq * 16;
    const int wrow = wn * 64 + l15;   // + 16*nt

    int slab = 0;
    for (int c = 0; c < p.k_chunks; c++) {
        __syncthreads();   // halo zeroed (c == 0) / all reads of the previous chunk done
        {
            int slot, choff;
            if (p.pair_chunks > 0) { slot = (c / p.pair_chunks) ? t.src1 : t.src0; choff = (c % p.pair_chunks) * 128; }
            else { slot = t.src0; choff = c * 128; }
            const char* sp = (const char*)p.src + ((int64_t)slot * p.src_slot_stride + choff) * 2;
            uint4 v[13];
#pragma unroll
            for (int i = 0; i < 13; i++) {
                const int e = tid + 256 * i, pix = min(e >> 4, VQA_HW - 1), pc = e & 15;
                v[i] = *(const uint4*)(sp + (int64_t)pix * p.src_pix_stride * 2 + pc * 16);
            }
#pragma unroll
            for (int i = 0; i < 13; i++) {
                const int e = tid + 256 * i, pix = e >> 4, pc = e & 15;
                if (pix < VQA_HW) {
                    const int y = pix / 14, x = pix - y * 14;
                    *(uint4*)(halo + ((y + 1) * 16 + x + 1) * HALO_STRIDE + pc * 16) = v[i];
                }
            }
        }
#pragma unroll 1
        for (int tap = 0; tap < TAPS; tap++) {
            const int tapoff = (TAPS == 9) ? ((tap / 3) * 16 + (tap % 3)) : 17;
#pragma unroll
            for (int half = 0; half < 2; half++, slab++) {
                wait_vm0();
                __syncthreads();   // slab landed for every wave; previous slab's reads done
                if (slab + 1 < nslabs) issue_slab(slab + 1);
                const char* wb = wbuf + (slab & 1) * WSLAB_BYTES;
                const char* xb = halo + xbase + tapoff * HALO_STRIDE + half * 128;
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    bf16x8 wf[4];
#pragma unroll
                    for (int n = 0; n < 4; n++) {
                        const int row = wrow + n * 16;
                        const int kc = ks * 4 + lq;
                        wf[n] = *(const bf16x8*)(wb + row * 128 + ((kc ^ (row & 7)) << 4));
                    }
#pragma unroll
                    for (int m = 0; m < 7; m++) {
                        const bf16x8 xf = *(const bf16x8*)(xb + m * 16 * HALO_STRIDE + ks * 64);
#pragma unroll
                        for (int n = 0; n < 4; n++) acc[m][n] = mma16(wf[n], xf, acc[m][n]);
                    }
                }
            }
        }
    }

    // epilogue: lane holds pixel (7*wm+m, l15), channels ntile*128 + wn*64 + n*16 + lq*4 + 0..3
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
    for (int n = 0; n < 4; n++) {
        const int ch = wn * 64 + n * 16 + lq * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_bias) bv = *(const float4*)(bias + ch);
#pragma unroll
        for (int m = 0; m < 7; m++) {
            const int pix = (wm * 7 + m) * 14 + l15;
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
// Weight packing: fp32 strided tensor -> bf16 slabs [group][chunk][tap][half][n][64] with the
// 16-byte k-chunks of each 128-byte row XOR-swizzled by (n & 7), i.e. exactly the LDS image the
// kernel above reads.
// ---------------------------------------------------------------------------------------------
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ out,
                                        int n_out, int k_in, int kpad, int taps, int64_t g_stride,
                                        int64_t n_stride, int64_t k_stride, int flip,
                                        int64_t per_group) {
    const int g = blockIdx.y;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per_group;
         idx += (int64_t)gridDim.x * blockDim.x) {
        // idx = (((c*taps + t)*2 + h)*n_out + n)*64 + pos
        const int pos = idx & 63;
        int64_t r = idx >> 6;
        const int n = r % n_out; r /= n_out;
        const int h = r & 1; r >>= 1;
        const int t = r % taps;
        const int c = r / taps;
        const int j = (pos >> 3) ^ (n & 7);
        const int k = c * 128 + h * 64 + j * 8 + (pos & 7);
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
