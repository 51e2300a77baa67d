// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cutie_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// float -> bfloat16, round-to-nearest-even (the rounding of torch's .to(bfloat16)): the cast lowers to gfx950's
// v_cvt_pk_bf16_f32 -- one instruction per PAIR, where the bit-twiddled form cost ~6 VALU per element in every epilogue
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

__device__ __forceinline__ uint32_t pack_bf2(float a, float b) {
    const bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}

// relu on two packed bf16: zero each half whose sign bit is set
__device__ __forceinline__ uint32_t relu_bf2(uint32_t w) {
    uint32_t neg = (w >> 15) & 0x00010001u;
    return w & ~(neg * 0xffffu);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// max / sum over the 4 lanes {l, l^16, l^32, l^48} (the 4 row groups of an MFMA accumulator column) without touching
// LDS: v_permlane16_swap / v_permlane32_swap (gfx950) instead of ds_bpermute shuffles.
__device__ __forceinline__ float rows_max(float v) {
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    unsigned a = __float_as_uint(fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])));
    auto r2 = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    return fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
}
__device__ __forceinline__ float rows_sum(float v) {
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    unsigned a = __float_as_uint(__uint_as_float(r[0]) + __uint_as_float(r[1]));
    auto r2 = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}

// integer sum over the 64 lanes, result in every lane: butterfly inside each 16-lane row on the DPP path (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror), then the 4 rows through v_permlane16_swap / v_permlane32_swap -- no LDS round trips
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);     // row_mirror
    auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    const unsigned a = r[0] + r[1];
    auto r2 = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    return (int)(r2[0] + r2[1]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// the same sum on the DPP / permlane path (as wave_sum_i32): six dependent ds_bpermute round trips become six VALU instructions.
// The order of the additions differs from wave_sum's butterfly, so the two are not bit-interchangeable.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xf, 0xf, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4E, 0xf, 0xf, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x141, 0xf, 0xf, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x140, 0xf, 0xf, false));
    return rows_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// launchers implemented in the .hip files; all return hipError_t as int
int launch_conv(const cutie_op* op, hipStream_t s);
int launch_elementwise(const cutie_op* op, hipStream_t s);   // everything in elementwise.hip
int launch_attention(const cutie_op* op, hipStream_t s);     // attention.hip
int launch_stem(const cutie_op* op, hipStream_t s);          // stem.hip: IMG_PREP + 7x7 conv + max pool
int launch_qchain(const cutie_op* op, hipStream_t s);        // qchain.hip: attention ops with flags 4 / 8, QFFN
int launch_affinity(const cutie_op* op, hipStream_t s);      // affinity.hip
int launch_bank(const cutie_op* op, hipStream_t s);          // bank.hip
void cutie_set_error(const char* fmt, ...);
