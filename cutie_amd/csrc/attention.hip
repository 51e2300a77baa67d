// Object-transformer attention kernels (QueryTransformerBlock, reference object_transformer.py:12-73).
// 16 object queries per object, 8 heads x 32: these are HBM-/latency-bound (arithmetic intensity < 50 flop/B),
// so they are written as streaming kernels with online softmax, not MFMA tiles.  The pixel-side linear
// projections that feed them are MFMA convs (conv_igemm.hip).
#include "common.h"
#include <math.h>

__device__ __forceinline__ float clamp_logit_(float p) {
    p = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
    return logf(p / (1.f - p));
}

// AUX_MASK: fg[k,p] = (L_k >= max(L_bg, max_j L_j));  nfg[k] += count
__global__ void aux_mask_kernel(const float* __restrict__ lg, uint8_t* __restrict__ fg, int* __restrict__ nfg, int K, int HW) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = p < HW;
    float bg = 1.f, mx = -INFINITY;
    if (valid) {
        for (int k = 0; k < K; ++k) {
            float pr = 1.f / (1.f + expf(-lg[(long)k * HW + p]));
            bg *= (1.f - pr);
            mx = fmaxf(mx, clamp_logit_(pr));
        }
        mx = fmaxf(mx, clamp_logit_(bg));
    }
    for (int k = 0; k < K; ++k) {
        bool f = false;
        if (valid) {
            float pr = 1.f / (1.f + expf(-lg[(long)k * HW + p]));
            f = clamp_logit_(pr) >= mx;
            fg[(long)k * HW + p] = f ? 1 : 0;
        }
        unsigned long long b = __ballot(f);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(&nfg[k], __popcll(b));
    }
}

// ATTN_Q2P: grid (heads, K, Q/4); the block owns 4 queries; its 4 waves split the pixels (wave w takes pixels
// w*64 + lane + 256 i) with a per-lane online softmax; lanes are merged by shuffles, waves through LDS.
__global__ __launch_bounds__(256) void attn_q2p_kernel(const float* __restrict__ q, const bf16_t* __restrict__ kv,
                                                       const uint8_t* __restrict__ fg, const int* __restrict__ nfg,
                                                       float* __restrict__ y, int Q, int HW, int C, int ldkv, int voff) {
    __shared__ float sM[4][4], sL[4][4], sA[4][4][32];     // [wave][query]([dim])
    const int hh = blockIdx.x, k = blockIdx.y, q0 = blockIdx.z * 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float scale = rsqrtf(32.f);
    float qv[4][32];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int d = 0; d < 32; ++d) qv[i][d] = q[((long)k * Q + q0 + i) * C + hh * 32 + d] * scale;
    const int n_fg = nfg[k];
    const bool is_fg_query = q0 < Q / 2;                   // queries 0..7 attend foreground only
    // row fully blocked -> unblocked (object_transformer.py:203)
    const bool masked = is_fg_query ? (n_fg != 0) : (n_fg != HW);
    float m[4], l[4], acc[4][32];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m[i] = -INFINITY; l[i] = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) acc[i][d] = 0.f;
    }
    for (int p = wave * 64 + lane; p < HW; p += 256) {
        if (masked) {
            bool f = fg[(long)k * HW + p] != 0;
            if (f != is_fg_query) continue;                // blocked
        }
        const bf16_t* kr = kv + ((long)k * HW + p) * ldkv + hh * 32;
        float kf[32], vf[32];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint4 a = *reinterpret_cast<const uint4*>(kr + c * 8);
            uint4 b = *reinterpret_cast<const uint4*>(kr + voff + c * 8);
            const uint32_t* au = &a.x; const uint32_t* bu = &b.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                kf[c * 8 + 2 * j] = __uint_as_float(au[j] << 16); kf[c * 8 + 2 * j + 1] = __uint_as_float(au[j] & 0xffff0000u);
                vf[c * 8 + 2 * j] = __uint_as_float(bu[j] << 16); vf[c * 8 + 2 * j + 1] = __uint_as_float(bu[j] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) s += qv[i][d] * kf[d];
            float mn = fmaxf(m[i], s);
            float alpha = __expf(m[i] - mn), pe = __expf(s - mn);
            l[i] = l[i] * alpha + pe;
#pragma unroll
            for (int d = 0; d < 32; ++d) acc[i][d] = acc[i][d] * alpha + pe * vf[d];
            m[i] = mn;
        }
    }
    // merge the 64 lanes of each wave (un-normalised, relative to the wave maximum)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float M = wave_max(m[i]);
        float f = (m[i] == -INFINITY) ? 0.f : __expf(m[i] - M);
        float L = wave_sum(l[i] * f);
        if (lane == 0) { sM[wave][i] = M; sL[wave][i] = L; }
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            float v = wave_sum(acc[i][d] * f);
            if (lane == d) sA[wave][i][d] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < 128) {                               // (query i, dim d) merge over the 4 waves
        int i = threadIdx.x >> 5, d = threadIdx.x & 31;
        float Mg = fmaxf(fmaxf(sM[0][i], sM[1][i]), fmaxf(sM[2][i], sM[3][i]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float f = (sM[w][i] == -INFINITY) ? 0.f : __expf(sM[w][i] - Mg);
            num += sA[w][i][d] * f;
            den += sL[w][i] * f;
        }
        y[((long)k * Q + q0 + i) * C + hh * 32 + d] = num / den;
    }
}

// ATTN_SELF: grid (heads, K), block 64: lane = query*4 + part (8 dims each)
__global__ void attn_self_kernel(const float* __restrict__ qk, const float* __restrict__ v, float* __restrict__ y, int Q, int C) {
    const int hh = blockIdx.x, k = blockIdx.y, lane = threadIdx.x, qi = lane >> 2, part = lane & 3;
    const float scale = rsqrtf(32.f);
    float qf[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) qf[d] = qk[((long)k * Q + qi) * 2 * C + hh * 32 + part * 8 + d] * scale;
    float s[16], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float* kr = qk + ((long)k * Q + j) * 2 * C + C + hh * 32 + part * 8;
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) a += qf[d] * kr[d];
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        s[j] = a;
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { s[j] = __expf(s[j] - mx); sum += s[j]; }
    float inv = 1.f / sum;
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float* vr = v + ((long)k * Q + j) * C + hh * 32 + part * 8;
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] += s[j] * vr[d];
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) y[((long)k * Q + qi) * C + hh * 32 + part * 8 + d] = o[d] * inv;
}

// ATTN_P2Q: grid (ceil(HW/256), heads, K): one thread per (pixel, head)
__global__ __launch_bounds__(256) void attn_p2q_kernel(const bf16_t* __restrict__ q, const float* __restrict__ kq,
                                                       const float* __restrict__ vq, bf16_t* __restrict__ y, int Q, int HW,
                                                       int C, int ldq) {
    __shared__ float ks[16][32], vs[16][32];
    const int hh = blockIdx.y, k = blockIdx.z;
    for (int t = threadIdx.x; t < 512; t += 256) {
        int j = t >> 5, d = t & 31;
        ks[j][d] = kq[((long)k * Q + j) * C + hh * 32 + d];
        vs[j][d] = vq[((long)k * Q + j) * C + hh * 32 + d];
    }
    __syncthreads();
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float scale = rsqrtf(32.f);
    const bf16_t* qr = q + ((long)k * HW + p) * ldq + hh * 32;
    float qf[32];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint4 a = *reinterpret_cast<const uint4*>(qr + c * 8);
        const uint32_t* au = &a.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qf[c * 8 + 2 * j] = __uint_as_float(au[j] << 16) * scale;
            qf[c * 8 + 2 * j + 1] = __uint_as_float(au[j] & 0xffff0000u) * scale;
        }
    }
    float s[16], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) a += qf[d] * ks[j][d];
        s[j] = a; mx = fmaxf(mx, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { s[j] = __expf(s[j] - mx); sum += s[j]; }
    float inv = 1.f / sum;
    bf16_t* yr = y + ((long)k * HW + p) * C + hh * 32;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int d = 0; d < 8; ++d) o[d] += s[j] * vs[j][c * 8 + d];
        *reinterpret_cast<uint4*>(yr + c * 8) = make_uint4(pack_bf2(o[0] * inv, o[1] * inv), pack_bf2(o[2] * inv, o[3] * inv),
                                                           pack_bf2(o[4] * inv, o[5] * inv), pack_bf2(o[6] * inv, o[7] * inv));
    }
}

int launch_attention(const cutie_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const uint64_t* p = op->p;
    switch (op->kind) {
        case CUTIE_OP_AUX_MASK:
            hipLaunchKernelGGL(aux_mask_kernel, dim3((i[1] + 255) / 256), dim3(256), 0, s, (const float*)p[0], (uint8_t*)p[1], (int*)p[2], i[0], i[1]);
            break;
        case CUTIE_OP_ATTN_Q2P:
            if (i[1] != 16 || i[3] != i[4] * 32) { cutie_set_error("attn_q2p: Q=16, head dim 32 only"); return -2; }
            hipLaunchKernelGGL(attn_q2p_kernel, dim3(i[4], i[0], i[1] / 4), dim3(256), 0, s, (const float*)p[0], (const bf16_t*)p[1], (const uint8_t*)p[2],
                               (const int*)p[3], (float*)p[4], i[1], i[2], i[3], i[5], i[6]);
            break;
        case CUTIE_OP_ATTN_SELF:
            if (i[1] != 16 || i[2] != i[3] * 32) { cutie_set_error("attn_self: Q=16, head dim 32 only"); return -2; }
            hipLaunchKernelGGL(attn_self_kernel, dim3(i[3], i[0]), dim3(64), 0, s, (const float*)p[0], (const float*)p[1], (float*)p[2], i[1], i[2]);
            break;
        case CUTIE_OP_ATTN_P2Q:
            if (i[1] != 16 || i[3] != i[4] * 32) { cutie_set_error("attn_p2q: Q=16, head dim 32 only"); return -2; }
            hipLaunchKernelGGL(attn_p2q_kernel, dim3((i[2] + 255) / 256, i[4], i[0]), dim3(256), 0, s, (const bf16_t*)p[0], (const float*)p[1],
                               (const float*)p[2], (bf16_t*)p[3], i[1], i[2], i[3], i[5]);
            break;
        default:
            cutie_set_error("attention: unknown op kind %d", op->kind);
            return -3;
    }
    return (int)hipGetLastError();
}
