// Grouped implicit-GEMM convolution over 14x14 feature-map tiles (gfx950, bf16 MFMA).
//
// One workgroup (4 waves) = one tile = one feature map x 128 output channels; wave w owns output
// channels [32w, 32w+32).  The map's 196 pixels are 13 MFMA column tiles of 16 consecutive pixels
// (the last one has 4): a tile per image row, 14 pixels in 16 lanes, would spend a fourteenth tile's
// MFMAs and LDS reads on lanes that hold nothing.
//   * pixels: for each 128-channel source chunk the map is staged once into an LDS halo image
//     (16 x 16 positions, zero border, 288-byte position stride); the 9 taps are 9 address offsets
//     into that image, so the im2col matrix never exists.  Image rows are 16 positions + 192 bytes
//     apart: from the last pixel of an image row to the first of the next the address then moves by
//     one position stride modulo the 256 bytes of a bank row, so the 16 pixels of a tile, which
//     straddle image rows, fall on the banks like 16 consecutive positions: no conflict
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

#define HALO_STRIDE 288                       // bytes per position
#define HALO_ROW (16 * HALO_STRIDE + 192)     // bytes per image row: 4800
#define HALO_BYTES (16 * HALO_ROW - 192)      // 76608: 16 rows of 16 positions
#define HALO_CENTER (HALO_ROW + HALO_STRIDE)  // from a pixel's (-1,-1) neighbour to the pixel itself
#define CONV_LDS HALO_BYTES
#define MT 13                                 // pixel tiles: 16 consecutive pixels each

// ---------------------------------------------------------------------------------------------
// Building blocks shared by the single-conv kernel and the fused residual-body kernel
// ---------------------------------------------------------------------------------------------

// Workgroup -> tile index.  Consecutive workgroup ids go round-robin to the 8 XCDs, each with its
// own 4 MiB L2, and the tile lists are sorted by token: taken in order, every XCD would stream the
// weights of every token of the level (0.6 MB per token and direction, tens of tokens) through its
// L2 and thrash it.  Instead XCD x takes the x-th contiguous eighth of the list, so it sees an
// eighth of the tokens.  (bijection for any n: XCD x owns n/8 tiles, one more if x < n % 8)
__device__ __forceinline__ int xcd_tile_index(int block, int n) {
    const int x = block & 7, j = block >> 3, q = n >> 3, r = n & 7;
    return x * q + min(x, r) + j;
}

// byte offset of border position j of the halo image: the 60 positions no staging pass ever writes
// (row 0, columns 0 and 15 of rows 1..14, row 15)
#define HALO_BORDER 60
__device__ __forceinline__ int halo_border_at(int j) {
    if (j < 16) return j * HALO_STRIDE;
    if (j < 44) return (((j - 16) >> 1) + 1) * HALO_ROW + ((j & 1) ? 15 * HALO_STRIDE : 0);
    return 15 * HALO_ROW + (j - 44) * HALO_STRIDE;
}

__device__ __forceinline__ void halo_zero_border(char* halo, int tid) {
    for (int idx = tid; idx < HALO_BORDER * 16; idx += 256)
        *(uint4*)(halo + halo_border_at(idx >> 4) + (idx & 15) * 16) = make_uint4(0, 0, 0, 0);
}

// This lane's pixel of every pixel tile: tile m, lane column l15 -> pixel 16 m + l15 (tile 12: lanes 0..3).
// xaddr[m] = byte offset of the pixel's (-1,-1) neighbour in the halo image, K-chunk lq: tap (dy, dx) is
// dy * HALO_ROW + dx * HALO_STRIDE further, the pixel's own cell HALO_CENTER further.  The 12 lanes
// without a pixel read pixel 4's cells (one address: a broadcast, on banks no lane with a pixel is
// on) and their results are never stored.
__device__ __forceinline__ void lane_pixels(int (&xaddr)[MT], int l15, int lq) {
#pragma unroll
    for (int m = 0; m < MT; m++) {
        int px = m * 16 + l15;
        if (m == MT - 1 && l15 >= 4) px = 4;
        const int y = px / 14, x = px - y * 14;
        xaddr[m] = y * HALO_ROW + x * HALO_STRIDE + lq * 16;
    }
}
// whether this lane's column holds a pixel in tile m
__device__ __forceinline__ bool lane_has_pixel(int m, int l15) { return m < MT - 1 || l15 < 4; }

// copy one 128-channel chunk of a source map (196 pixels x 256 B) into the halo image interior,
// through registers, 7 image rows at a time; thread = (image column x = tid >> 4, 16-byte piece
// pc = tid & 15), so addresses step by constants from row to row
__device__ __forceinline__ void halo_stage(char* halo, const char* sp, int pix_stride_bytes, int tid, bool diag_noload) {
    const int x = tid >> 4, pc = tid & 15;
    const bool active = x < 14;
    const char* g = sp + (int64_t)(active ? x : 0) * pix_stride_bytes + pc * 16;
    char* cell = halo + HALO_ROW + (x + 1) * HALO_STRIDE + pc * 16;
#pragma unroll 1
    for (int y0 = 0; y0 < 14; y0 += 7) {
        uint4 v[7];
#pragma unroll
        for (int i = 0; i < 7; i++) {
            if (diag_noload) v[i] = make_uint4(x, pc, i, 0);
            else v[i] = *(const uint4*)(g + (int64_t)(y0 + i) * 14 * pix_stride_bytes);
        }
        if (active) {
#pragma unroll
            for (int i = 0; i < 7; i++) *(uint4*)(cell + (y0 + i) * HALO_ROW) = v[i];
        }
    }
}

// K loop over one 128-channel chunk already staged in the halo image: TAPS x 4 K-steps x 13 pixel
// tiles.  Weights come through the 4-slot register ring `wr` (refilled for K-step s+4 after use,
// so three K-steps of MFMAs cover the L2 latency); pixel fragments through an 8-deep ring read 7
// tiles ahead; the issue order is pinned to 1 LDS read : 2 MFMA (left alone, the scheduler collapses
// the ring and waits on every read right after issuing it).  A tap is 52 fragments, half a turn of
// the ring more than a whole number: taps alternate between two phases of the ring.  With ZERO the accumulators start from the first K-step's products plus `c0`
// (the C operand of those MFMAs: this lane's bias values, or zeros) instead of from registers
// somebody cleared - the bias costs no instruction of its own anywhere.
template <int TAPS, bool ZERO>
__device__ __forceinline__ void conv_kloop(const char* halo, const int (&xaddr)[MT], const char* wbase, int64_t wstep,
                                           int nsteps, int& s, bf16x8 (&wr)[4][2], f32x4 (&acc)[MT][2],
                                           const f32x4 (&c0)[2], bool diag_now) {
    bf16x8 xq[8];
    {
        const char* x0 = halo + ((TAPS == 9) ? 0 : HALO_CENTER);
#pragma unroll
        for (int q = 0; q < 7; q++) xq[q] = *(const bf16x8*)(x0 + xaddr[q]);
    }
    auto tap_body = [&](auto tap_tag) {
        constexpr int tap = decltype(tap_tag)::value;
        constexpr bool FIRST = ZERO && tap == 0;
        constexpr int PH = (tap & 1) * 4;                      // ring slot of the tap's first fragment
        constexpr int tapoff = (TAPS == 9) ? ((tap / 3) * HALO_ROW + (tap % 3) * HALO_STRIDE) : HALO_CENTER;
        constexpr int tn = (tap + 1 < TAPS) ? tap + 1 : tap;  // last tap: harmless re-read
        constexpr int tapoff_n = (TAPS == 9) ? ((tn / 3) * HALO_ROW + (tn % 3) * HALO_STRIDE) : HALO_CENTER;
        const char* xb = halo + tapoff;
        const char* xn = halo + tapoff_n;
#pragma unroll
        for (int ks = 0; ks < 4; ks++, s++) {
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const int q = ks * MT + m;            // 0..51 within the tap
                const int qa = q + 7;                 // fragment to fetch now
                const int ka = (qa / MT) & 3, ma = qa % MT;
                xq[(qa + PH) & 7] = *(const bf16x8*)((qa < 4 * MT ? xb : xn) + xaddr[ma] + ka * 64);
                if (FIRST && ks == 0) {
                    acc[m][0] = mma16(wr[ks][0], xq[(q + PH) & 7], c0[0]);
                    acc[m][1] = mma16(wr[ks][1], xq[(q + PH) & 7], c0[1]);
                } else {
                    acc[m][0] = mma16(wr[ks][0], xq[(q + PH) & 7], acc[m][0]);
                    acc[m][1] = mma16(wr[ks][1], xq[(q + PH) & 7], acc[m][1]);
                }
            }
#pragma unroll
            for (int m = 0; m < MT; m++) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            const int sn = diag_now ? 0 : min(s + 4, nsteps - 1);
            wr[ks][0] = *(const bf16x8*)(wbase + sn * wstep);
            wr[ks][1] = *(const bf16x8*)(wbase + sn * wstep + 1024);
            __builtin_amdgcn_sched_barrier(0);   // keep the refill here: sunk to the loop end it is waited on at once
        }
    };
    static_assert((4 * MT) % 8 == 4, "the two ring phases assume a tap of 8 n + 4 fragments");
    // (taps unrolled: a tap's offset is then part of every read's immediate offset; added at run time it
    // would need a register per pixel tile and tap in flight)
    tap_body(std::integral_constant<int, 0>{});
    if constexpr (TAPS == 9) {
        tap_body(std::integral_constant<int, 1>{}); tap_body(std::integral_constant<int, 2>{});
        tap_body(std::integral_constant<int, 3>{}); tap_body(std::integral_constant<int, 4>{});
        tap_body(std::integral_constant<int, 5>{}); tap_body(std::integral_constant<int, 6>{});
        tap_body(std::integral_constant<int, 7>{}); tap_body(std::integral_constant<int, 8>{});
    }
}

// Final epilogue.  A lane holds 4 channels of one pixel per tile, i.e. 8-byte pieces scattered over
// 16 cache lines per instruction; written, or worse read (residual, mask), that way the epilogue costs
// more than the K loop.  So the raw accumulators go through LDS (fp32, tiles 0..6 = pixels 0..111,
// then tiles 7..12 = pixels 112..195; the halo image is dead by now) and come back pixel-major: every
// global access is 16 bytes per lane, 1 KiB contiguous per wave, and residual, rounding, ReLU and mask
// are applied there, two channels per instruction (the bias is in the accumulators since the first K-step).
//   out = [relu]( acc + res ) * (mask > 0)
template <bool RES, bool MASK>
__device__ __forceinline__ void conv_epilogue_t(char* halo, const f32x4 (&acc)[MT][2], bool relu,
                                                const char* rsrc, const char* msrc, char* dptr, int pix_stride,
                                                int tid) {
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int pc = tid & 15;            // this thread's 16-byte column of every row: channels 8*pc..8*pc+7
    float* stg = (float*)halo;          // [112 pixels][132 floats]: 528-byte rows keep the writes conflict-free
    // ReLU without a select per value: a signed 16-bit maximum with 0, or with the most negative value (no change)
    const uint32_t floor_pk = relu ? 0u : 0x80008000u;
    // (a generic lambda called with a constant: a run-time `half` would select between accumulator
    // registers with one v_cndmask per value)
    auto do_half = [&](auto half_tag) {
        constexpr int half = decltype(half_tag)::value;
        constexpr int tiles = half ? MT - 7 : 7;
        // this thread's 16-byte pieces of the half: piece i -> pixel 112*half + (tid >> 4) + 16*i
        const int p0 = 112 * half + (tid >> 4);
        const int64_t g0 = ((int64_t)p0 * pix_stride + pc * 8) * 2;
        uint4 rv[tiles], mv[tiles];
#pragma unroll
        for (int i = 0; i < tiles; i++) {
            const int64_t goff = g0 + (p0 + 16 * i < VQA_HW ? (int64_t)i * 16 * pix_stride * 2 : 0);
            if (RES) rv[i] = *(const uint4*)(rsrc + goff);
            if (MASK) mv[i] = *(const uint4*)(msrc + goff);
        }
        lds_barrier();       // K loop reads (half 0) / previous half's staging reads are done
        {
            float* row = stg + l15 * 132 + wave * 32 + lq * 4;
#pragma unroll
            for (int mm = 0; mm < tiles; mm++) {
                *(f32x4*)(row + mm * 16 * 132) = acc[7 * half + mm][0];
                *(f32x4*)(row + mm * 16 * 132 + 16) = acc[7 * half + mm][1];
            }
        }
        lds_barrier();
        const float* src = stg + (tid >> 4) * 132 + pc * 8;
#pragma unroll
        for (int i = 0; i < tiles; i++) {
            if (p0 + 16 * i < VQA_HW) {
                const f32x4 f0 = *(const f32x4*)(src + i * 16 * 132);
                const f32x4 f1 = *(const f32x4*)(src + i * 16 * 132 + 4);
                f32x2 o[4] = {{f0[0], f0[1]}, {f0[2], f0[3]}, {f1[0], f1[1]}, {f1[2], f1[3]}};
                uint32_t r[4] = {0, 0, 0, 0}, k[4] = {0, 0, 0, 0}, w[4];
                if (RES) { r[0] = rv[i].x; r[1] = rv[i].y; r[2] = rv[i].z; r[3] = rv[i].w; }
                if (MASK) { k[0] = mv[i].x; k[1] = mv[i].y; k[2] = mv[i].z; k[3] = mv[i].w; }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (RES) o[j] += unpack_pk_bf16(r[j]);
                    w[j] = max_pk_i16(cvt_pk_bf16(o[j]), floor_pk);
                    if (MASK) w[j] = mask_pk_bf16(w[j], k[j]);
                }
                *(uint4*)(dptr + g0 + (int64_t)i * 16 * pix_stride * 2) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    };
    do_half(std::integral_constant<int, 0>{});
    do_half(std::integral_constant<int, 1>{});
}

// one specialisation per combination: a residual or mask array that only some paths fill makes the
// compiler carry placeholders for the others through the whole kernel (scalar-register spills)
__device__ __forceinline__ void conv_epilogue(char* halo, const f32x4 (&acc)[MT][2], bool relu,
                                              const char* rsrc, const char* msrc, char* dptr, int pix_stride,
                                              int tid) {
    if (rsrc) {
        if (msrc) conv_epilogue_t<true, true>(halo, acc, relu, rsrc, msrc, dptr, pix_stride, tid);
        else conv_epilogue_t<true, false>(halo, acc, relu, rsrc, msrc, dptr, pix_stride, tid);
    } else {
        if (msrc) conv_epilogue_t<false, true>(halo, acc, relu, rsrc, msrc, dptr, pix_stride, tid);
        else conv_epilogue_t<false, false>(halo, acc, relu, rsrc, msrc, dptr, pix_stride, tid);
    }
}

// this lane's bias values: 4 channels of each of the wave's two 16-channel blocks (zeros without a bias),
// in the accumulator layout: they are the C operand of the first K-step's MFMAs
__device__ __forceinline__ void lane_bias(f32x4 (&b)[2], const float* bias, int wave, int lq) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    b[0] = b[1] = z;
    if (bias) { b[0] = *(const f32x4*)(bias + wave * 32 + lq * 4); b[1] = *(const f32x4*)(bias + wave * 32 + lq * 4 + 16); }
}

template <int TAPS>
__global__ __launch_bounds__(256, 2) void conv_tiles_kernel(vqa_conv_params p) {
    extern __shared__ __attribute__((aligned(16))) char halo[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const vqa_conv_tile t = p.tiles[xcd_tile_index(blockIdx.x, p.num_tiles)];
    const int ntile = blockIdx.y;
    // an output half nobody wants (pair outputs with one destination missing): uniform over the workgroup
    if (p.pair_out > 0 && ((ntile / p.pair_out) ? t.dst1 : t.dst0) < 0) return;

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
    halo_zero_border(halo, tid);
    f32x4 binit[2];
    lane_bias(binit, (p.flags & VQA_CONV_BIAS) ? p.bias + (int64_t)t.wgroup * p.n_out + ntile * 128 : nullptr, wave, lq);

    f32x4 acc[MT][2];
    // LDS byte offset of this lane's pixel in every pixel tile, at tap (-1,-1), K-chunk lq
    int xaddr[MT];
    lane_pixels(xaddr, l15, lq);

    // (the accumulators start from the bias here, not in the first K-step's MFMAs: two copies of the K loop
    // in the chunk loop, one per form, begin alike, and the compiler then moves their common reads in
    // front of the branch - twenty fragments in flight and their registers spilled)
#pragma unroll
    for (int m = 0; m < MT; m++) { acc[m][0] = binit[0]; acc[m][1] = binit[1]; }
    int s = 0;
    for (int c = 0; c < p.k_chunks; c++) {
        int slot, choff;
        if (p.pair_chunks > 0) { slot = (c / p.pair_chunks) ? t.src1 : t.src0; choff = (c % p.pair_chunks) * 128; }
        else { slot = t.src0; choff = c * 128; }
        const char* sp = (const char*)p.src + ((int64_t)slot * p.src_slot_stride + choff) * 2;
        __syncthreads();   // every wave is done reading the previous chunk's image
        halo_stage(halo, sp, p.src_pix_stride * 2, tid, (p.flags & 0x400) != 0);
        __syncthreads();
        conv_kloop<TAPS, false>(halo, xaddr, wbase, wstep, nsteps, s, wr, acc, binit, (p.flags & 0x200) != 0);
    }

    if (p.flags & 0x100) {   // diagnostic: skip the epilogue
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < MT; m++) sum += acc[m][0][0] + acc[m][1][1];
        if (sum == 12345.678f) ((float*)p.dst)[tid] = sum;
        return;
    }
    int dslot, mslot, choff;
    if (p.pair_out > 0) {
        const int which = ntile / p.pair_out;
        dslot = which ? t.dst1 : t.dst0;
        mslot = which ? t.mask1 : t.mask0;
        choff = (ntile % p.pair_out) * 128;
    } else { dslot = t.dst0; mslot = t.mask0; choff = ntile * 128; }
    if (dslot < 0) return;     // uniform over the workgroup
    const char* rsrc = t.res >= 0 ? (const char*)p.res + ((int64_t)t.res * p.dst_slot_stride + choff) * 2 : nullptr;
    const char* msrc = mslot >= 0 ? (const char*)p.mask + ((int64_t)mslot * p.dst_slot_stride + choff) * 2 : nullptr;
    char* dptr = (char*)p.dst + ((int64_t)dslot * p.dst_slot_stride + choff) * 2;
    conv_epilogue(halo, acc, (p.flags & VQA_CONV_RELU) != 0, rsrc, msrc, dptr, p.dst_pix_stride, tid);
}

// ---------------------------------------------------------------------------------------------
// The stem's first convolution straight from the reference's input: x fp32 [N][Cin][14][14]
// (Train_model.py:311 hands the loader's NCHW batch to the executor; Module_Model.py:59-60,239 convolves it),
// 3x3, Cin -> 128 channels, bias, ReLU, one sample per workgroup.  The layout change (fp32 planes -> bf16
// channels-last) is not a pass of its own (1.2 GB of traffic per 1024 samples in front of an idle matrix
// pipe): four PRODUCER waves turn chunk c + 1 of the sample into the halo image the K loop reads while four
// CONSUMER waves run the K loop of vqa_conv_tiles over the image of chunk c.
//   producer wave, 8 channels at a time: 16-byte loads along the planes (a chunk's 128 planes are 100 KB of
//   contiguous memory), v_cvt_pk_bf16_f32, 8-byte stores into a private channel-major strip [8][128 pixels]
//   in LDS; two transposing reads (ds_read_b64_tr_b16: 4 channels x 16 pixels each) hand every lane the 8
//   channels of one pixel, i.e. one 16-byte cell of the image - and of the bf16 channels-last copy that the
//   weight-gradient pass reads later, written from the same registers.  Strip rows are 320 bytes apart: the
//   4 rows x 2 pixel blocks x 4 pieces of a 32-lane read land on 64 different banks.
//   one workgroup barrier per chunk: the producers hand over image c & 1 and then fill the other one, which
//   the consumers left before they came to that barrier.
// LDS: two images + four strips = 163 456 bytes, one workgroup of 8 waves per CU.  Same fragments, same K
// order, same rounding as vqa_nchw_to_nhwc + vqa_conv_tiles: the same bits.
// ---------------------------------------------------------------------------------------------
#define STRIP_STRIDE 320
#define STRIP_BYTES (8 * STRIP_STRIDE)
#define STEM_LDS (2 * HALO_BYTES + 4 * STRIP_BYTES)

struct vqa_stem_params {
    const float* x;          // fp32 [N][Cin][196]
    void* x_nhwc;            // bf16 [N][196][Cin] or null
    void* dst;               // bf16 base of destination slots
    const void* w;           // packed weights of the convolution (one group)
    const float* bias;       // fp32 [128] or null
    int64_t dst_slot_stride; // elements
    int32_t dst_slot_first;  // sample n goes to slot dst_slot_first + n
    int32_t dst_pix_stride;
    int32_t N, Cin;          // Cin % 128 == 0
    int32_t relu;            // bit 0: ReLU; bits 1-4: diagnostics (producers idle / load only / do not load / read eight samples)
};

__global__ __launch_bounds__(512, 1) void stem_nchw_kernel(vqa_stem_params p) {
    extern __shared__ __attribute__((aligned(16))) char halo[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x;
    const int k_chunks = p.Cin >> 7;
    char* const img2 = halo + HALO_BYTES;
    // borders of both images, by everybody
    for (int idx = tid; idx < HALO_BORDER * 16; idx += 512) {
        const int at = halo_border_at(idx >> 4) + (idx & 15) * 16;
        *(uint4*)(halo + at) = make_uint4(0, 0, 0, 0);
        *(uint4*)(img2 + at) = make_uint4(0, 0, 0, 0);
    }
    if (wave >= 4) {
        // ---------------- producers ----------------
        const int pw = wave - 4;
        char* strip = halo + 2 * HALO_BYTES + pw * STRIP_BYTES;
        typedef float f32x4n __attribute__((ext_vector_type(4)));
        const float* xs = p.x + (int64_t)((p.relu & 16) ? (n & 7) : n) * p.Cin * VQA_HW;     // (bit 16: diagnostics, eight samples for all)
        // a lane's part of a group of 8 planes: pixels 0..127 of the planes are float4 f = lane + 64 i (i < 4:
        // plane f >> 5, pixels 4 (f & 31)..), pixels 128..195 float4 f = lane + 64 i < 136 (plane f / 17)
        int offa[4], offb[3], stra[4], strb[3];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int f = lane + 64 * i;
            offa[i] = (f >> 5) * VQA_HW + (f & 31) * 4;
            stra[i] = (f >> 5) * STRIP_STRIDE + (f & 31) * 8;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int f = min(lane + 64 * i, 135), r = f / 17, k = f - r * 17;
            offb[i] = r * VQA_HW + 128 + 4 * k;
            strb[i] = r * STRIP_STRIDE + k * 8;
        }
        const bool lastb = lane + 128 < 136;
        // transposing reads: lane 4 q + pp of a 16-lane group reads pixels 4 pp..4 pp + 3 of strip row q (and of row
        // 4 + q); group g of pass it covers pixels 16 (4 it + g)..+15 of the strip
        const int l15 = lane & 15, grp = lane >> 4;
        const int troff = (l15 >> 2) * STRIP_STRIDE + (lane & 3) * 8 + grp * 32;
        // image cell and channels-last offset of this lane's pixel in each of the four passes
        int cell[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int px = (k >> 1) * 128 + (k & 1) * 64 + grp * 16 + l15;
            ok[k] = px < VQA_HW;
            const int pc = ok[k] ? px : 0, y = pc / 14, x = pc - y * 14;
            cell[k] = (y + 1) * HALO_ROW + (x + 1) * HALO_STRIDE;
        }
        bf16_t* xn = p.x_nhwc ? (bf16_t*)p.x_nhwc + (int64_t)n * VQA_HW * p.Cin : nullptr;
        // Group gq of chunk c: 8 of the wave's 32 planes.  One group is in flight while the one before it is
        // processed: 7 to 14 loads per wave.  (With a half chunk - 28 loads a wave, 900 cache lines a CU - in flight
        // the consumers' weight fragments, which come from L2 through the same L1, arrived late: the K loops of a
        // sample took a third longer.)
        f32x4n ra[7], rb[7];
        const bool diag_noload = (p.relu & 8) != 0;
        auto ask = [&](f32x4n (&r)[7], int c, int gq) {
            if (diag_noload) return;
            const float* g = xs + (int64_t)(c * 128 + pw * 32 + gq * 8) * VQA_HW;
#pragma unroll
            for (int i = 0; i < 4; i++) r[i] = __builtin_nontemporal_load((const f32x4n*)(g + offa[i]));
#pragma unroll
            for (int i = 0; i < 3; i++) r[4 + i] = __builtin_nontemporal_load((const f32x4n*)(g + offb[i]));
        };
        auto put = [&](const f32x4n (&r)[7], int c, int gq) {
            char* img = (c & 1) ? img2 : halo;
            constexpr int gi = 0;
            {
                const int chl = pw * 32 + gq * 8;                 // first channel of the group within the chunk
#pragma unroll
                for (int part = 0; part < 2; part++) {
                    if (part == 0) {
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const f32x4n v = r[7 * gi + i];
                            *(uint2*)(strip + stra[i]) = make_uint2(cvt_pk_bf16((f32x2){v[0], v[1]}), cvt_pk_bf16((f32x2){v[2], v[3]}));
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 3; i++) {
                            const f32x4n v = r[7 * gi + 4 + i];
                            if (i < 2 || lastb)
                                *(uint2*)(strip + strb[i]) = make_uint2(cvt_pk_bf16((f32x2){v[0], v[1]}), cvt_pk_bf16((f32x2){v[2], v[3]}));
                        }
                    }
                    // (a lane reads what other lanes stored: nothing may move across these two lines; the LDS itself
                    // runs a wave's instructions in order)
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int it = 0; it < 2; it++) {
                        const s16x4 lo = lds_tr4(strip + troff + it * 128);
                        const s16x4 hi = lds_tr4(strip + troff + it * 128 + 4 * STRIP_STRIDE);
                        const uint4 v = __builtin_bit_cast(uint4, frag_from_tr(lo, hi));
                        const int k = part * 2 + it;
                        if (ok[k]) *(uint4*)(img + cell[k] + chl * 2) = v;
                    }
                    asm volatile("" ::: "memory");
                }
            }
        };
        const bool diag_idle = (p.relu & 2) != 0, diag_noput = (p.relu & 4) != 0;
        if (!diag_idle) ask(ra, 0, 0);
#pragma unroll 1
        for (int c = 0; c < k_chunks; c++) {
            if (!diag_idle) {
            ask(rb, c, 1);
            if (!diag_noput) put(ra, c, 0);
            ask(ra, c, 2);
            if (!diag_noput) put(rb, c, 1);
            ask(rb, c, 3);
            if (!diag_noput) put(ra, c, 2);
            if (c + 1 < k_chunks) ask(ra, c + 1, 0);
            if (!diag_noput) put(rb, c, 3);
            if (diag_noput) {
#pragma unroll
                for (int i = 0; i < 7; i++) asm volatile("" :: "v"(ra[i]), "v"(rb[i]));
            }
            }
            lds_barrier();          // image c & 1 is the consumers'; they are out of the other one
            if (xn) {
                // the finished image (the consumers only read it) leaves as the bf16 channels-last copy in whole
                // 256-byte rows: thread = (image column, 16-byte piece), 14 image rows
                const int ptid = tid - 256, cx = ptid >> 4, pcs = ptid & 15;
                if (cx < 14) {
                    const char* im = ((c & 1) ? img2 : halo) + HALO_ROW + (cx + 1) * HALO_STRIDE + pcs * 16;
                    bf16_t* o = xn + (int64_t)cx * p.Cin + c * 128 + pcs * 8;
#pragma unroll
                    for (int y = 0; y < 14; y++)
                        *(uint4*)(o + (int64_t)y * 14 * p.Cin) = *(const uint4*)(im + y * HALO_ROW);
                }
            }
        }
        // (the consumers' epilogue has four workgroup barriers)
        lds_barrier(); lds_barrier(); lds_barrier(); lds_barrier();
        return;
    }
    // ---------------- consumers: the K loop and epilogue of conv_tiles_kernel<9>, images alternating ----------------
    const int l15 = lane & 15, lq = lane >> 4;
    const int64_t wstep = 8 * 1024;
    const char* wbase = (const char*)p.w + (int64_t)(wave * 2) * 1024 + lane * 16;
    const int nsteps = k_chunks * 36;
    bf16x8 wr[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        wr[i][0] = *(const bf16x8*)(wbase + i * wstep);
        wr[i][1] = *(const bf16x8*)(wbase + i * wstep + 1024);
    }
    f32x4 binit[2];
    lane_bias(binit, p.bias, wave, lq);
    int xaddr[MT];
    lane_pixels(xaddr, l15, lq);
    f32x4 acc[MT][2];
    int s = 0;
    lds_barrier();
    conv_kloop<9, true>(halo, xaddr, wbase, wstep, nsteps, s, wr, acc, binit, false);
#pragma unroll 1
    for (int c = 1; c < k_chunks; c++) {
        lds_barrier();
        // (the image's base goes into the lane's address registers, made opaque: every read of the K loop is then
        // a register plus a constant that fits the instruction's offset field)
        int ib = (c & 1) * HALO_BYTES;
        asm volatile("" : "+v"(ib));
        int xa[MT];
#pragma unroll
        for (int m = 0; m < MT; m++) xa[m] = xaddr[m] + ib;
        conv_kloop<9, false>(halo, xa, wbase, wstep, nsteps, s, wr, acc, binit, false);
    }
    char* dptr = (char*)p.dst + (int64_t)(p.dst_slot_first + n) * p.dst_slot_stride * 2;
    conv_epilogue(halo, acc, (p.relu & 1) != 0, nullptr, nullptr, dptr, p.dst_pix_stride, tid);
}

extern "C" int vqa_stem_nchw(const float* x, void* x_nhwc, void* dst, const void* w, const float* bias,
                             int64_t dst_slot_stride, int32_t dst_slot_first, int32_t dst_pix_stride, int32_t N,
                             int32_t Cin, int32_t flags, void* stream) {
    if (!x || !dst || !w || N < 0 || Cin <= 0 || Cin % 128 != 0 || dst_pix_stride < 128 || dst_pix_stride % 8 != 0 ||
        dst_slot_stride < (int64_t)VQA_HW * dst_pix_stride || dst_slot_first < 0) {
        vqa_set_error("vqa_stem_nchw: bad argument");
        return VQA_EINVAL;
    }
    if (N == 0) return VQA_OK;
    if (!vqa_allow_lds((const void*)stem_nchw_kernel, STEM_LDS, "vqa_stem_nchw")) return VQA_ELAUNCH;
    vqa_stem_params p;
    p.x = x; p.x_nhwc = x_nhwc; p.dst = dst; p.w = w; p.bias = bias;
    p.dst_slot_stride = dst_slot_stride; p.dst_slot_first = dst_slot_first; p.dst_pix_stride = dst_pix_stride;
    p.N = N; p.Cin = Cin; p.relu = flags;
    VqaProfScope prof(VQA_K_CONV9, 2.0 * VQA_HW * 9.0 * Cin * 128.0 * N, (double)N * VQA_HW * (4.0 * Cin + 2.0 * 128), (hipStream_t)stream);
    hipLaunchKernelGGL(stem_nchw_kernel, dim3(N), dim3(512), STEM_LDS, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_stem_nchw: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    return VQA_OK;
}

// ---------------------------------------------------------------------------------------------
// Fused module body (module width 128 only): every convolution of one tree node in one workgroup.
//   forward :  [P = relu(proj([x0; x1]) + bp)]   H = relu(conv1(x) + b1);   Y = relu(x + conv2(H) + b2)
//              (Layers.py:33-36, Module_Model.py:38-40,47-50; x = P for a binary node)
//   backward:  [child of a binary node: dZ = dproj_half(dP of the parent) * (Y > 0)]
//              dH = dgrad2(dZ) * (H > 0);   dX = (dgrad1(dH) + dZ) * (X > 0)
//              (a binary node's dX is its dP; each child takes its half of the transposed projection
//              in front of its own body, as forward takes the projection in front of the parent's:
//              no tile is two passes longer than the others of its launch)
// No intermediate map makes a round trip: accumulators are rounded to bf16 straight into the halo
// image that the next convolution reads, and leave for HBM (they are needed again by the backward
// pass / the weight gradients) as whole 256-byte rows from that image.  One launch per tree level,
// one source load and one cold start per node.
// ---------------------------------------------------------------------------------------------

// v_permlane16_swap: the odd 16-lane rows of `a` change places with the even rows of `b`
// (a' = a.row0 b.row0 a.row2 b.row2, b' = a.row1 b.row1 a.row3 b.row3); applied twice it is the identity.
__device__ __forceinline__ void swap16(uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}

// accumulators (the bias is in them) [ReLU] -> bf16 -> interior of the halo image; then a row-wise pass
// applies an optional mask, rewrites the image and stores the map to `dst` in 16-byte pieces.
template <bool MASK>
__device__ __forceinline__ void acc_to_image_t(char* halo, const int (&xaddr)[MT], const f32x4 (&acc)[MT][2], bool relu,
                                               const char* mask, char* dst, int tid) {
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    // row pass mapping: thread = (image column x = tid >> 4, 16-byte piece pc = tid & 15), looping over
    // the 14 image rows: every address is a constant step per row (no per-piece index arithmetic),
    // a wave still covers 4 pixels x 256 B = 1 KiB contiguous per access; columns 14, 15 idle
    const int x = tid >> 4, pc = tid & 15;
    const bool active = x < 14;
    const int goff = (active ? x : 0) * 256 + pc * 16;
    uint4 mv[14];
    if (MASK) {
#pragma unroll
        for (int y = 0; y < 14; y++) mv[y] = *(const uint4*)(mask + goff + y * 14 * 256);
    }
    lds_barrier();          // every wave has finished reading the image
    {
        // 16-byte conflict-free LDS writes after a swap between the lane rows: see solo_epilogue_t
        const uint32_t floor_pk = relu ? 0u : 0x80008000u;
        // this lane's pixel: its own cell, the 8 channels the swap leaves the lane with
        char* cell = halo + HALO_CENTER - lq * 16 + (wave * 32 + 16 * (lq & 1) + 8 * (lq >> 1)) * 2;
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const f32x4 o0 = acc[m][0], o1 = acc[m][1];
            uint32_t a0 = max_pk_i16(cvt_pk_bf16((f32x2){o0[0], o0[1]}), floor_pk);
            uint32_t a1 = max_pk_i16(cvt_pk_bf16((f32x2){o0[2], o0[3]}), floor_pk);
            uint32_t c0 = max_pk_i16(cvt_pk_bf16((f32x2){o1[0], o1[1]}), floor_pk);
            uint32_t c1 = max_pk_i16(cvt_pk_bf16((f32x2){o1[2], o1[3]}), floor_pk);
            swap16(a0, c0);
            swap16(a1, c1);
            if (lane_has_pixel(m, l15)) *(uint4*)(cell + xaddr[m]) = make_uint4(a0, a1, c0, c1);
        }
    }
    lds_barrier();
    if (active) {
        char* cell = halo + HALO_ROW + (x + 1) * HALO_STRIDE + pc * 16;
#pragma unroll
        for (int y = 0; y < 14; y++) {
            uint4 v = *(uint4*)(cell + y * HALO_ROW);
            if (MASK) {
                v = make_uint4(mask_pk_bf16(v.x, mv[y].x), mask_pk_bf16(v.y, mv[y].y), mask_pk_bf16(v.z, mv[y].z),
                               mask_pk_bf16(v.w, mv[y].w));
                *(uint4*)(cell + y * HALO_ROW) = v;
            }
            *(uint4*)(dst + goff + y * 14 * 256) = v;
        }
    }
    lds_barrier();
}

__device__ __forceinline__ void acc_to_image(char* halo, const int (&xaddr)[MT], const f32x4 (&acc)[MT][2], bool relu,
                                             const char* mask, char* dst, int tid) {
    if (mask) acc_to_image_t<true>(halo, xaddr, acc, relu, mask, dst, tid);
    else acc_to_image_t<false>(halo, xaddr, acc, relu, mask, dst, tid);
}

__device__ __forceinline__ void load_wring(bf16x8 (&wr)[4][2], const char* wbase, int64_t wstep) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        wr[i][0] = *(const bf16x8*)(wbase + i * wstep);
        wr[i][1] = *(const bf16x8*)(wbase + i * wstep + 1024);
    }
}

// ---------------------------------------------------------------------------------------------
// The two-image form of the fused kernel (SOLO, see resblock_tiles_kernel): its own staging and epilogues.
// Same arithmetic, same order of the fp32 additions and same rounding points as the one-image form and
// as vqa_conv_tiles: the two forms and the separate launches agree bit for bit.
// ---------------------------------------------------------------------------------------------

// accumulators (the bias is in them) [+ the map `xres` in LDS: the residual connection] -> bf16 -> [ReLU] -> interior of the
// image `img`; then, after a barrier, a row-wise pass applies an optional mask (KEEP: and rewrites the image,
// a convolution reads it next) and stores the map to `dst` in 16-byte pieces.
// A lane's accumulators are 4 channels of each of the wave's two 16-channel blocks, 8 bytes each, and
// 8-byte LDS accesses at this position stride are 4-way bank conflicts.  One register swap per packed
// pair between the lane rows (swap16) leaves every lane with 8 consecutive channels of its pixel -
// channels 32 wave + 16 (row & 1) + 8 (row >> 1) .. + 7 - i.e. one conflict-free 16-byte access; the
// residual map is read the same way and swapped back into the accumulator layout.
// No barrier in front of the writes: `img` is an image nobody reads at this point (the caller's rule).
template <bool MASK, bool RES, bool KEEP>
__device__ __forceinline__ void solo_epilogue_t(char* img, const char* xres, const int (&xaddr)[MT], const f32x4 (&acc)[MT][2],
                                                bool relu, const char* mask, char* dst, int tid) {
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int x = tid >> 4, pc = tid & 15;
    const bool active = x < 14;
    const int goff = (active ? x : 0) * 256 + pc * 16;
    // (asked for here, not in front of the K loop - measured: with 14 more loads in flight and 56 more
    // registers live the K loops of a tile lose 7 k cycles, the epilogues gain 2 k)
    uint4 mv[14];
    if (MASK) {
#pragma unroll
        for (int y = 0; y < 14; y++) mv[y] = *(const uint4*)(mask + goff + y * 14 * 256);
    }
    {
        // ReLU without a branch per row: a signed 16-bit maximum with 0, or with the most negative value (no change)
        const uint32_t floor_pk = relu ? 0u : 0x80008000u;
        // this lane's pixel: its own cell (xaddr[m] further), the 8 channels the swap leaves the lane with;
        // lanes without a pixel read pixel 4's cell and never write
        const int celloff = HALO_CENTER - lq * 16 + (wave * 32 + 16 * (lq & 1) + 8 * (lq >> 1)) * 2;
        char* cell = img + celloff;
        // the residual cells are read three image rows ahead (all 14 at once: 56 registers too many next to
        // the mask rows); the order is pinned row by row, or the scheduler hoists every read to the top
        uint4 xr[MT];
        if (RES) {
#pragma unroll
            for (int m = 0; m < 3; m++) xr[m] = *(const uint4*)(xres + celloff + xaddr[m]);
        }
#pragma unroll
        for (int m = 0; m < MT; m++) {
            if (RES && m + 3 < MT) xr[m + 3] = *(const uint4*)(xres + celloff + xaddr[m + 3]);
            f32x4 o0 = acc[m][0], o1 = acc[m][1];
            if (RES) {
                swap16(xr[m].x, xr[m].z);
                swap16(xr[m].y, xr[m].w);
                const f32x2 r0 = unpack_pk_bf16(xr[m].x), r1 = unpack_pk_bf16(xr[m].y);
                const f32x2 r2 = unpack_pk_bf16(xr[m].z), r3 = unpack_pk_bf16(xr[m].w);
                o0 += (f32x4){r0[0], r0[1], r1[0], r1[1]};
                o1 += (f32x4){r2[0], r2[1], r3[0], r3[1]};
            }
            uint32_t a0 = max_pk_i16(cvt_pk_bf16((f32x2){o0[0], o0[1]}), floor_pk);
            uint32_t a1 = max_pk_i16(cvt_pk_bf16((f32x2){o0[2], o0[3]}), floor_pk);
            uint32_t c0 = max_pk_i16(cvt_pk_bf16((f32x2){o1[0], o1[1]}), floor_pk);
            uint32_t c1 = max_pk_i16(cvt_pk_bf16((f32x2){o1[2], o1[3]}), floor_pk);
            swap16(a0, c0);
            swap16(a1, c1);
            if (lane_has_pixel(m, l15)) *(uint4*)(cell + xaddr[m]) = make_uint4(a0, a1, c0, c1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    lds_barrier();
    if (active) {
        char* cell = img + HALO_ROW + (x + 1) * HALO_STRIDE + pc * 16;
#pragma unroll
        for (int y = 0; y < 14; y++) {
            uint4 v = *(uint4*)(cell + y * HALO_ROW);
            if (MASK) {
                v = make_uint4(mask_pk_bf16(v.x, mv[y].x), mask_pk_bf16(v.y, mv[y].y), mask_pk_bf16(v.z, mv[y].z),
                               mask_pk_bf16(v.w, mv[y].w));
                if (KEEP) *(uint4*)(cell + y * HALO_ROW) = v;
            }
            *(uint4*)(dst + goff + y * 14 * 256) = v;
        }
    }
    if (MASK && KEEP) lds_barrier();      // the rewritten rows are other threads' fragment reads
}

template <bool RES, bool KEEP>
__device__ __forceinline__ void solo_epilogue(char* img, const char* xres, const int (&xaddr)[MT], const f32x4 (&acc)[MT][2],
                                              bool relu, const char* mask, char* dst, int tid) {
    if (mask) solo_epilogue_t<true, RES, KEEP>(img, xres, xaddr, acc, relu, mask, dst, tid);
    else solo_epilogue_t<false, RES, KEEP>(img, xres, xaddr, acc, relu, mask, dst, tid);
}

// RANGE false: one tile per workgroup, entry blockIdx.x of the list in XCD order.  RANGE true: the
// tile range is in device memory (`level_info`: first tile, tiles) and the workgroups of the launch,
// however many, share it; every workgroup leaves its loop, with or without work.
// SOLO: for launches of at most one workgroup per CU.  A workgroup then takes the LDS and the registers of
// two: a second halo image, and 512 registers per lane (the accumulators move to the accumulation
// registers).  The body's input stays in the first image, the intermediate map goes to the second, the
// result to the first again: every map is written into an image that nobody is reading (no barrier in
// front of the writes), the residual connection is read from LDS (not from memory), the three biases and
// all 14 rows of the source map are asked for at entry, and the epilogues write LDS without bank conflicts
// (solo_epilogue_t).  Fixed image addresses: chosen at run time they cost a vector add per LDS read in the
// K loops, which is why the sources of a 1x1 pass in front are staged into the second image.
template <bool RANGE, bool SOLO>
__global__ __launch_bounds__(256, SOLO ? 1 : 2) void resblock_tiles_kernel(vqa_resblock_params p) {
    extern __shared__ __attribute__((aligned(16))) char halo[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    if constexpr (SOLO) {
        static_assert(!RANGE, "the two-image form takes one tile per workgroup");
        const vqa_resblock_tile t = p.tiles[xcd_tile_index(blockIdx.x, p.num_tiles)];
        unsigned long long* stamps = p.phase_stamps ? p.phase_stamps + (int64_t)blockIdx.x * 8 : nullptr;
#define STAMP(i) if (stamps && tid == 0) stamps[i] = __builtin_readcyclecounter()
        STAMP(0);
        const int64_t slot_b = p.slot_stride * 2;
        char* arena = (char*)p.arena;
        const bool relu = (p.flags & VQA_CONV_RELU) != 0, has_bias = (p.flags & VQA_CONV_BIAS) != 0;
        const int64_t wstep = 8 * 1024;
        const int64_t wlane = (int64_t)(wave * 2) * 1024 + lane * 16;
        const int64_t w33 = (int64_t)t.wgroup * p.w_group_stride * 2 + wlane;
        const bool pre2 = t.src1 >= 0, pre1 = t.pre_group >= 0;
        int64_t wstep_pre = wstep;
        const char* wbase = (const char*)p.w_first + w33;
        if (pre2) wbase = (const char*)p.w_pre + (int64_t)t.wgroup * p.w_pre_group_stride * 2 + wlane;
        else if (pre1) {
            wbase = (const char*)p.w_pre + (int64_t)(t.pre_group >> 1) * p.w_pre_group_stride * 2 + (t.pre_group & 1) * wstep + wlane;
            wstep_pre = 2 * wstep;
        }
        // (the second image is beyond the reach of the LDS instructions' 16-bit offsets from the first image's
        // lane address: its own lane address, opaque to constant folding, keeps the K loops' reads free of adds.
        // Made here: a volatile statement between the two halves of the staging below sends its 14 rows through scratch)
        int base2 = HALO_BYTES;
        asm volatile("" : "+v"(base2));
        bf16x8 wr[4][2];
        load_wring(wr, wbase, wstep_pre);
        f32x4 b_pre[2], b_first[2], b_second[2];
        lane_bias(b_pre, (pre2 && has_bias) ? p.bias_pre + (int64_t)t.wgroup * 128 : nullptr, wave, lq);
        lane_bias(b_first, has_bias ? p.bias_first + (int64_t)t.wgroup * 128 : nullptr, wave, lq);
        lane_bias(b_second, has_bias ? p.bias_second + (int64_t)t.wgroup * 128 : nullptr, wave, lq);
        char* const img2 = halo + HALO_BYTES;
        f32x4 acc[MT][2];
        int xaddr[MT], xaddr2[MT];
        lane_pixels(xaddr, l15, lq);
#pragma unroll
        for (int m = 0; m < MT; m++) xaddr2[m] = xaddr[m] + base2;
        // halo_stage with all 14 rows of the source map in flight at once (one memory latency instead of two).
        // Named values, not an array (any statement the compiler cannot see through between an array's
        // writes and reads sends it through scratch), and a scheduler barrier that keeps the loads together
        // (left alone they are paired off with their writes, one exposed latency each).  The threads of the
        // two idle image columns write into the 32 unused bytes behind a position's 256: no branch.
        {
            const int sx = tid >> 4, spc = tid & 15;
            const char* g = arena + (int64_t)t.src0 * slot_b + (sx < 14 ? sx : 0) * 256 + spc * 16;
            char* cell = ((pre2 || pre1) ? img2 : halo) +
                         (sx < 14 ? HALO_ROW + (sx + 1) * HALO_STRIDE + spc * 16 : HALO_CENTER + 256 + (spc & 1) * 16);
#define SOLO_ROWS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#define SOLO_ASK(i) const uint4 st##i = *(const uint4*)(g + i * 14 * 256);
#define SOLO_PUT(i) *(uint4*)(cell + i * HALO_ROW) = st##i;
            SOLO_ROWS(SOLO_ASK)
            __builtin_amdgcn_sched_barrier(0);
            // the borders of both images, cleared under that latency (straight-line code: the last round repeats
            // writes of zeros instead of branching - across a branch the compiler sinks the loads to their uses)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int idx = min(tid + 256 * k, HALO_BORDER * 16 - 1);
            const int at = halo_border_at(idx >> 4) + (idx & 15) * 16;
            *(uint4*)(halo + at) = make_uint4(0, 0, 0, 0);
            *(uint4*)(img2 + at) = make_uint4(0, 0, 0, 0);
        }
            __builtin_amdgcn_sched_barrier(0);
            SOLO_ROWS(SOLO_PUT)
#undef SOLO_ROWS
#undef SOLO_ASK
#undef SOLO_PUT
        }
        __syncthreads();
        STAMP(1);
        int s = 0;
        if (pre2 || pre1) {
#pragma unroll
            for (int m = 0; m < MT; m++) { acc[m][0] = b_pre[0]; acc[m][1] = b_pre[1]; }      // (zeros without a bias)
            const int nsteps_pre = pre2 ? 8 : 4;
#pragma unroll 1
            for (int c = 0; c < (pre2 ? 2 : 1); c++) {
                if (c) {
                    __syncthreads();
                    halo_stage(img2, arena + (int64_t)t.src1 * slot_b, 256, tid, false);
                    __syncthreads();
                }
                conv_kloop<1, false>(halo, xaddr2, wbase, wstep_pre, nsteps_pre, s, wr, acc, b_pre, false);
            }
            wbase = (const char*)p.w_first + w33;
            load_wring(wr, wbase, wstep);
            solo_epilogue<false, true>(halo, nullptr, xaddr, acc, pre2 && relu, t.pre_mask >= 0 ? arena + (int64_t)t.pre_mask * slot_b : nullptr,
                                       arena + (int64_t)t.res * slot_b, tid);
            s = 0;
        }
        STAMP(2);
        conv_kloop<9, true>(halo, xaddr, wbase, wstep, 36, s, wr, acc, b_first, false);
        STAMP(3);
        wbase = (const char*)p.w_second + w33;
        load_wring(wr, wbase, wstep);
        // (every wave is past the 1x1 pass's reads of the second image: the barrier of that pass's epilogue)
        solo_epilogue<false, true>(img2, nullptr, xaddr, acc, relu, t.mask1 >= 0 ? arena + (int64_t)t.mask1 * slot_b : nullptr, arena + (int64_t)t.dst1 * slot_b, tid);
        s = 0;
        STAMP(4);
        conv_kloop<9, true>(halo, xaddr2, wbase, wstep, 36, s, wr, acc, b_second, false);
        STAMP(5);
        // the result goes where the input is: each wave reads its own cells of it first, nobody else reads it
        if (t.res >= 0) solo_epilogue<true, false>(halo, halo, xaddr, acc, relu, t.mask0 >= 0 ? arena + (int64_t)t.mask0 * slot_b : nullptr, arena + (int64_t)t.dst0 * slot_b, tid);
        else solo_epilogue<false, false>(halo, nullptr, xaddr, acc, relu, t.mask0 >= 0 ? arena + (int64_t)t.mask0 * slot_b : nullptr, arena + (int64_t)t.dst0 * slot_b, tid);
        STAMP(6);
#undef STAMP
        return;
    }
    int tile_i = RANGE ? p.level_info[0] + (int)blockIdx.x : xcd_tile_index(blockIdx.x, p.num_tiles);
    const int tile_end = RANGE ? p.level_info[0] + p.level_info[1] : 0;
    if (RANGE && tile_i >= tile_end) return;
    do {
    const vqa_resblock_tile t = p.tiles[tile_i];
    // (with several tiles per workgroup everything derived from the thread's index would be computed once, in front
    // of the loop, and held in registers across the whole tile body - 43 spilled registers; made opaque per
    // tile it is recomputed, a few instructions)
    int tid = threadIdx.x;
    if (RANGE) asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    // diagnostics: shader-clock timestamps of the phase boundaries of every tile
    unsigned long long* stamps = p.phase_stamps ? p.phase_stamps + (int64_t)blockIdx.x * 8 : nullptr;
#define STAMP(i) if (stamps && tid == 0) stamps[i] = __builtin_readcyclecounter()
    STAMP(0);
    const int64_t slot_b = p.slot_stride * 2;       // bytes per slot; pixel stride is 256 B (C = 128)
    char* arena = (char*)p.arena;
    const bool relu = (p.flags & VQA_CONV_RELU) != 0, has_bias = (p.flags & VQA_CONV_BIAS) != 0;

    const int64_t wstep = 8 * 1024;                 // 128 output channels: 8 fragments of 1 KiB per K-step
    const int64_t wlane = (int64_t)(wave * 2) * 1024 + lane * 16;
    const int64_t w33 = (int64_t)t.wgroup * p.w_group_stride * 2 + wlane;
    // A 1x1 pass in front of the body.  Forward (src1 >= 0): the node's own projection of its two
    // inputs, K = 256, 8 fragments per K-step.  Backward (pre_group >= 0): this node's half of its
    // PARENT's transposed projection, K = 128 over the parent's dP map; the pack holds both halves
    // (16 fragments per K-step), half h starting 8 fragments in.
    const bool pre2 = t.src1 >= 0, pre1 = t.pre_group >= 0;
    int64_t wstep_pre = wstep;
    const char* wbase = (const char*)p.w_first + w33;
    if (pre2) wbase = (const char*)p.w_pre + (int64_t)t.wgroup * p.w_pre_group_stride * 2 + wlane;
    else if (pre1) {
        wbase = (const char*)p.w_pre + (int64_t)(t.pre_group >> 1) * p.w_pre_group_stride * 2 + (t.pre_group & 1) * wstep + wlane;
        wstep_pre = 2 * wstep;
    }
    bf16x8 wr[4][2];
    load_wring(wr, wbase, wstep_pre);
    f32x4 bfirst[2];
    lane_bias(bfirst, has_bias ? p.bias_first + (int64_t)t.wgroup * 128 : nullptr, wave, lq);
    halo_zero_border(halo, tid);
    f32x4 acc[MT][2];
    int xaddr[MT];
    lane_pixels(xaddr, l15, lq);

    halo_stage(halo, arena + (int64_t)t.src0 * slot_b, 256, tid, false);
    __syncthreads();
    STAMP(1);
    int s = 0;
    if (pre2 || pre1) {   // x = [relu](pre(sources) + bias_pre) * (pre_mask > 0), kept in slot `res`
        // (registers set to the bias - zeros without one - here, not the first-K-step form: with it the
        // register allocator spills)
        f32x4 bp[2];
        lane_bias(bp, (pre2 && has_bias) ? p.bias_pre + (int64_t)t.wgroup * 128 : nullptr, wave, lq);
#pragma unroll
        for (int m = 0; m < MT; m++) { acc[m][0] = bp[0]; acc[m][1] = bp[1]; }
        const int nsteps_pre = pre2 ? 8 : 4;
#pragma unroll 1
        for (int c = 0; c < (pre2 ? 2 : 1); c++) {
            if (c) {
                __syncthreads();
                halo_stage(halo, arena + (int64_t)t.src1 * slot_b, 256, tid, false);
                __syncthreads();
            }
            conv_kloop<1, false>(halo, xaddr, wbase, wstep_pre, nsteps_pre, s, wr, acc, bp, false);
        }
        wbase = (const char*)p.w_first + w33;
        load_wring(wr, wbase, wstep);
        acc_to_image(halo, xaddr, acc, pre2 && relu,
                     t.pre_mask >= 0 ? arena + (int64_t)t.pre_mask * slot_b : nullptr,
                     arena + (int64_t)t.res * slot_b, tid);
        s = 0;
    }
    STAMP(2);
    conv_kloop<9, true>(halo, xaddr, wbase, wstep, 36, s, wr, acc, bfirst, false);
    STAMP(3);

    // intermediate map -> the image the second convolution reads; its weight stream and its bias start
    // now and land under the epilogue
    wbase = (const char*)p.w_second + w33;
    load_wring(wr, wbase, wstep);
    f32x4 bsecond[2];
    lane_bias(bsecond, has_bias ? p.bias_second + (int64_t)t.wgroup * 128 : nullptr, wave, lq);
    acc_to_image(halo, xaddr, acc, relu,
                 t.mask1 >= 0 ? arena + (int64_t)t.mask1 * slot_b : nullptr, arena + (int64_t)t.dst1 * slot_b, tid);
    s = 0;
    STAMP(4);
    conv_kloop<9, true>(halo, xaddr, wbase, wstep, 36, s, wr, acc, bsecond, false);
    STAMP(5);

    conv_epilogue(halo, acc, relu,
                  t.res >= 0 ? arena + (int64_t)t.res * slot_b : nullptr,
                  t.mask0 >= 0 ? arena + (int64_t)t.mask0 * slot_b : nullptr, arena + (int64_t)t.dst0 * slot_b, 128, tid);
    STAMP(6);
    if (RANGE) __syncthreads();     // the next tile's staging overwrites the image this one's epilogue reads
    } while (RANGE && (tile_i += (int)gridDim.x) < tile_end);
#undef STAMP
}

extern "C" int vqa_conv_tiles(const vqa_conv_params* p, int32_t taps, void* stream) {
    if (!p || !p->src || !p->dst || !p->w || !p->tiles || (taps != 9 && taps != 1) ||
        p->k_chunks <= 0 || p->n_out <= 0 || p->n_out % 128 != 0 || p->num_tiles < 0 ||
        p->src_pix_stride % 8 != 0 || p->dst_pix_stride % 8 != 0 ||
        ((p->flags & VQA_CONV_BIAS) && !p->bias)) {
        vqa_set_error("vqa_conv_tiles: bad argument");
        return VQA_EINVAL;
    }
    if (p->num_tiles == 0) return VQA_OK;
    if (!vqa_allow_lds((const void*)conv_tiles_kernel<9>, CONV_LDS, "vqa_conv_tiles") ||
        !vqa_allow_lds((const void*)conv_tiles_kernel<1>, CONV_LDS, "vqa_conv_tiles")) return VQA_ELAUNCH;
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

extern "C" int vqa_resblock_tiles(const vqa_resblock_params* p, void* stream) {
    if (!p || !p->arena || !p->w_first || !p->w_second || !p->tiles || p->num_tiles < 0 ||
        ((p->flags & VQA_CONV_BIAS) && (!p->bias_first || !p->bias_second)) ||
        (p->num_pre > 0 && (!p->w_pre || ((p->flags & VQA_CONV_BIAS) && !p->bias_pre)))) {
        vqa_set_error("vqa_resblock_tiles: bad argument");
        return VQA_EINVAL;
    }
    if (p->num_tiles == 0) return VQA_OK;
    if (!vqa_allow_lds((const void*)resblock_tiles_kernel<false, false>, CONV_LDS, "vqa_resblock_tiles") ||
        !vqa_allow_lds((const void*)resblock_tiles_kernel<true, false>, CONV_LDS, "vqa_resblock_tiles") ||
        !vqa_allow_lds((const void*)resblock_tiles_kernel<false, true>, 2 * CONV_LDS, "vqa_resblock_tiles")) return VQA_ELAUNCH;
    // a launch that cannot put two workgroups on any CU gives each workgroup the LDS and registers of two
    const bool solo = !p->level_info && p->num_tiles <= vqa_cu_count();
    // two 3x3 convolutions of 128 -> 128 channels per tile, plus the 1x1 passes in front: K = 256 per
    // forward binary node, K = 128 per backward child of one (num_pre counts K = 256 equivalents)
    VqaProfScope prof(VQA_K_RESBLOCK, 2.0 * 2.0 * VQA_HW * 1152.0 * 128.0 * p->num_tiles +
                                          2.0 * VQA_HW * 256.0 * 128.0 * p->num_pre,
                      (double)p->num_tiles * VQA_HW * 2.0 * 128 * 3, (hipStream_t)stream);
    if (p->level_info) hipLaunchKernelGGL((resblock_tiles_kernel<true, false>), dim3(p->num_tiles), dim3(256), CONV_LDS, (hipStream_t)stream, *p);
    else if (solo) hipLaunchKernelGGL((resblock_tiles_kernel<false, true>), dim3(p->num_tiles), dim3(256), 2 * CONV_LDS, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL((resblock_tiles_kernel<false, false>), dim3(p->num_tiles), dim3(256), CONV_LDS, (hipStream_t)stream, *p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_resblock_tiles: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
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
