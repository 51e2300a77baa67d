// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cutie_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved (same rounding as torch's float->bfloat16)
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

__device__ __forceinline__ uint32_t pack_bf2(float a, float b) {
    return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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
int launch_affinity(const cutie_op* op, hipStream_t s);      // affinity.hip
int launch_bank(const cutie_op* op, hipStream_t s);          // bank.hip
void cutie_set_error(const char* fmt, ...);
