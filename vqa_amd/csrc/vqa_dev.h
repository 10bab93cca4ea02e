// Device-side building blocks shared by the gfx950 kernels: bf16 helpers, MFMA wrappers, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA 16x16x32 A/B fragment
typedef __attribute__((ext_vector_type(4))) float f32x4;     // one MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint16_t u16x4;
typedef __attribute__((ext_vector_type(8))) uint16_t u16x8;

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

// Packed bf16-pair helpers.  Next to a workgroup that is busy in its MFMA loop a wave gets roughly
// one vector instruction per MFMA slot, so epilogues are priced in instructions: these work on two
// channels per instruction.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2n __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// (a, b) -> packed bf16 pair, round to nearest even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t cvt_pk_bf16(f32x2 v) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2n));
}
// ReLU of a packed pair: as signed 16-bit integers a bf16 is negative iff its sign bit is set,
// and max(., 0) also turns -0 into +0
__device__ __forceinline__ uint32_t relu_pk_bf16(uint32_t v) {
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
}
// signed 16-bit maximum of each half with the matching half of `lim`
__device__ __forceinline__ uint32_t max_pk_i16(uint32_t v, uint32_t lim) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), __builtin_bit_cast(s16x2, lim)));
}
// v * (m > 0) per bf16 of a packed pair, as a bit mask: max(m, 0) clears negatives and -0, then
// (0 - m) >> 15 (arithmetic) is all ones exactly where m was positive.  Inline asm because the
// compiler rewrites any C form of this into per-half compares, selects and a permute.
__device__ __forceinline__ uint32_t mask_pk_bf16(uint32_t v, uint32_t m) {
    uint32_t k;
    asm("v_pk_max_i16 %0, %1, 0\n\t"
        "v_pk_sub_i16 %0, 0, %0\n\t"
        "v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]"
        : "=&v"(k) : "v"(m));
    return v & k;
}
// the two bf16 of a dword as floats
__device__ __forceinline__ f32x2 unpack_pk_bf16(uint32_t v) {
    return (f32x2){__uint_as_float(v << 16), __uint_as_float(v & 0xFFFF0000u)};
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter, which parks every wave until its global loads and stores have completed; where the
// barrier separates LDS phases and the global accesses in flight are prefetches or final stores,
// that wait is pure loss (a microsecond or more per barrier with a busy neighbour on the CU).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// D = A(16 x 32) * B(32 x 16) + C.  Lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15] (j<8)
// and D[4*(l>>4)+r][l&15] (r<4).
__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// One 1 KiB global->LDS DMA per wave: lane l copies 16 B from its own `g` to `lds_wave_base + 16*l`.
__device__ __forceinline__ void dma16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const GLB_AS void*)g, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Transposed LDS read: each lane supplies the address of 4 consecutive 16-bit elements of one row
// (lane 4q+p of a 16-lane group: row q, elements 4p..4p+3 of a 4-row x 16-column block) and gets
// back column (lane&15) of the 4 rows.  Two of these make the 8 k-values of an MFMA fragment whose
// k index runs along LDS rows.
__device__ __forceinline__ s16x4 lds_tr4(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)p);
}
__device__ __forceinline__ bf16x8 frag_from_tr(s16x4 lo, s16x4 hi) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
