// Helpers shared by attention.hip (the attention ops) and qchain.hip (the query side of a transformer block in four launches).
#pragma once
#include "common.h"
#include <math.h>

__device__ __forceinline__ float clamp_logit_(float p) {
    p = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
    return logf(p / (1.f - p));
}

// fg[k,p] = (L_k >= max(L_bg, max_j L_j)) with L = logit of the clamped probabilities (object_transformer.py:179-205)
__device__ __forceinline__ bool aux_fg_(const float* __restrict__ lg, int K, int HW, int k, int p) {
    float bg = 1.f, mx = -INFINITY, mine = 0.f;
    for (int j = 0; j < K; ++j) {
        const float pr = 1.f / (1.f + expf(-lg[(long)j * HW + p]));
        const float l = clamp_logit_(pr);
        bg *= (1.f - pr);
        mx = fmaxf(mx, l);
        mine = j == k ? l : mine;
    }
    return mine >= fmaxf(mx, clamp_logit_(bg));
}


typedef __attribute__((ext_vector_type(4))) uint32_t q2p_u32x4;
union q2p_frag { q2p_u32x4 u; bf16x8 b; };

__device__ __forceinline__ void split_bf2(float a, float b, uint32_t& hi, uint32_t& lo) {
    bf16_t ha = f2bf(a), hb = f2bf(b);
    hi = (uint32_t)ha | ((uint32_t)hb << 16);
    lo = (uint32_t)f2bf(a - bf2f(ha)) | ((uint32_t)f2bf(b - bf2f(hb)) << 16);
}


// ---- projections computed inside the attention kernels (round 2: one launch less per attention) ---------------------------------
// The 16 query rows of one object are staged in LDS (LayerNorm'd and / or with the query embedding added, exactly the inputs the
// LINEAR op would have read), and a 16 x 16 output tile of x . W^T runs on MFMA with x split into bf16 hi + lo (the arithmetic of
// linear_mfma_kernel): lane (c, g) of the result holds rows 4g..4g+3 of column c.
#define PROJ_XLD 260                                     // fp32 row pitch of the staged rows (256 + 4: staggers the banks)
struct ProjIn {                                          // what a fused projection needs besides the attention operands
    const float* x;                                      // [K*16, 256] fp32 rows (row stride ldx)
    const float* add;                                    // query embedding [K*16, 256] or null
    const float* ln_g; const float* ln_b;                // LayerNorm in front (null: none)
    float* ln_out;                                       // [K*16, 256]: the normalised rows, written once per object (null: not kept)
    const bf16_t* W; const float* bias;                  // packed linear [N][256] bf16, bias [N]
    int ldx;
};

// rows of object k -> LDS.  xs_add: LN(x) + add (or x + add); xs_plain (nullable): LN(x) (or x).  Wave w takes rows w, w + NW, ...
template <int NW>
__device__ __forceinline__ void stage_rows16(const ProjIn& pi, int k, float* xs_add, float* xs_plain, bool write_ln_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int RPW = 16 / NW;                         // rows per wave; all their loads go out first
    float4 xv[RPW], av[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const long row = (long)k * 16 + wave + j * NW;
        xv[j] = *reinterpret_cast<const float4*>(pi.x + row * pi.ldx + lane * 4);
        av[j] = pi.add ? *reinterpret_cast<const float4*>(pi.add + row * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 gg = make_float4(1.f, 1.f, 1.f, 1.f), bb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pi.ln_g) { gg = *reinterpret_cast<const float4*>(pi.ln_g + lane * 4); bb = *reinterpret_cast<const float4*>(pi.ln_b + lane * 4); }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int r = wave + j * NW;
        float4 v = xv[j];
        if (pi.ln_g) {
            const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.f / 256.f);
            const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
            const float rstd = rsqrtf(wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.f / 256.f) + 1e-5f);
            v.x = dx * rstd * gg.x + bb.x; v.y = dy * rstd * gg.y + bb.y; v.z = dz * rstd * gg.z + bb.z; v.w = dw * rstd * gg.w + bb.w;
            if (pi.ln_out && write_ln_out) *reinterpret_cast<float4*>(pi.ln_out + ((long)k * 16 + r) * 256 + lane * 4) = v;
        }
        if (xs_plain) *reinterpret_cast<float4*>(xs_plain + r * PROJ_XLD + lane * 4) = v;
        v.x += av[j].x; v.y += av[j].y; v.z += av[j].z; v.w += av[j].w;
        *reinterpret_cast<float4*>(xs_add + r * PROJ_XLD + lane * 4) = v;
    }
}

// partial 16 x 16 tile: weight rows n0..n0+15, STEPS 32-wide k steps from ks0.  The weight fragments are requested first, all of
// them (an L2 round trip each: issued one per step in front of its MFMAs they cost 12 dependent round trips in ATTN_SELF).
typedef __attribute__((ext_vector_type(4))) unsigned int proj_u4;
template <int STEPS>
__device__ __forceinline__ void proj16_load(const bf16_t* __restrict__ W, int n0, int ks0, proj_u4* wv, int ldw = 256) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s_ = 0; s_ < STEPS; ++s_) wv[s_] = *reinterpret_cast<const proj_u4*>(W + (long)(n0 + c) * ldw + (ks0 + s_) * 32 + 8 * g);
}
template <int STEPS>
__device__ __forceinline__ f32x4 proj16_mma(const float* xs, int ks0, const proj_u4* wv, int xld = PROJ_XLD) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s_ = 0; s_ < STEPS; ++s_) {
        const int kw = (ks0 + s_) * 32 + 8 * g;
        const float4 a = *reinterpret_cast<const float4*>(xs + c * xld + kw), b = *reinterpret_cast<const float4*>(xs + c * xld + kw + 4);
        const float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        proj_u4 hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) { uint32_t h_, l_; split_bf2(xv[2 * i], xv[2 * i + 1], h_, l_); hi[i] = h_; lo[i] = l_; }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, lo), __builtin_bit_cast(bf16x8, wv[s_]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hi), __builtin_bit_cast(bf16x8, wv[s_]), acc, 0, 0, 0);
    }
    return acc;
}

