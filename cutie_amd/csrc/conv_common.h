// Pieces shared by the convolution kernels (conv_igemm.hip, conv_dma.hip, conv_pc.hip): launch parameters, 16-B global load,
// LDS chunk swizzle and the epilogue tail.
#pragma once
#include "common.h"

struct ConvParams {
    const bf16_t* x1; const bf16_t* x2; const bf16_t* w; const float* bias; const bf16_t* res; void* y;
    int B, H, W, C1, C2, ldx1, ldx2, OH, OW, Cout, ldy, KH, KW, stride, pad, ldr, Kpad;
    int flags, M, Cin, OHW;
    float* part; int splitk, ldp, Kslice;                // split-K: fp32 partial tiles [splitk][M][ldp]
    // conv_dma only: per-(object, channel) sums of the stored output as fixed point (x 2^24, int64: integer atomics are order-
    // independent, so the GAP that ECA consumes stays bit-reproducible) accumulated into gap[B][Cout]; zero[0..nzero) is cleared by
    // block 0 (the buffer the NEXT conv of the block accumulates into)
    long long* gap; unsigned long long* zero; int nzero;
    // conv_pc only: bytes [pf, pf + pf_bytes) = the weights of the NEXT conv of the plan, touched by the producer waves when their last
    // DMA piece is out (every XCD covers the whole range: inside a frame every launch starts with its weights cold, r03_conv_ablation.md)
    const unsigned char* pf; int pf_bytes;
    const unsigned char* pf2; int pf2_bytes;             // a second range (the conv after next, when the next one has no producer waves)
    // broadcast residual of a launch that holds several CLIPS (CUTIE_F_RES_BCAST with f0 > 0): the objects come in groups of f0 (one clip
    // each), every group has its own residual map; res_grp_rows = f0 * OHW output rows per group, res_grp_stride = rows between the maps
    int res_grp_rows, res_grp_stride;
};
#define GAP_FIXED_SCALE 16777216.f
// GAP side job of the LDS-DMA / producer-consumer convs: every stored VALUE goes to fixed point (2^-20: exact for a bf16 of magnitude 2^-13 .. 2^11,
// saturating beyond) BEFORE anything is summed -- integer sums depend neither on how a tile groups the rows into fragments and waves nor on the
// order in which the blocks' atomics arrive.  (Rounds 3-6 summed a wave's values in fp32 and converted the partial sums: exact almost always -- but
// one bf16 ulp of a fusion output differed between the 128 x 64 tile of clips in lock step and the 96 x 64 tile of one clip about once per 35 M
// elements, and the transformer carried it into every pixel of the clip: tools/lockstep_soak.py, tools/lockstep_diverge.py.)  One multiply and one
// conversion per value (the two-part 2^-24 form cost the pooled convs 6.6 us of 33: profiles/r06_lockstep.md); the accumulator keeps its 2^-24 unit.
#define GAP_ELEM_SHIFT 4                                  // accumulator unit 2^-24 = element unit 2^-20 >> 4
__device__ __forceinline__ long long conv_gapfx(float v) { return (long long)__float2int_rn(v * 1048576.f); }

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// 16-B load that is a global_load for sure.  Pointers that went through a select, an array of pointers or pointer
// increments lose their address space and hipcc falls back to flat_load; a flat load counts in lgkmcnt as well as vmcnt, so
// every wait for an LDS read would also wait for the whole weight / activation prefetch (no overlap of L2 latency with MFMA).
typedef const __attribute__((address_space(1))) u32x4* gptr16;
#define GLOAD16(ptr) (*(gptr16)(ptr))

template <int CPR>
__device__ __forceinline__ int swz(int row) {
    return CPR == 4 ? ((row >> 3) & 1) * 3 : (row & (CPR - 1));
}

// Row of the residual that output row m adds: itself, or -- broadcast -- the pixel's row inside the (group's) one map.
__device__ __forceinline__ int conv_res_row(const ConvParams& p, int m) {
    if (!(p.flags & CUTIE_F_RES_BCAST)) return m;
    int r = m % p.OHW;
    if (p.res_grp_rows > 0) r += (m / p.res_grp_rows) * p.res_grp_stride;
    return r;
}

// Epilogue tail for 8 consecutive output channels of one pixel: bias, residual, activation, 16-B stores.
__device__ __forceinline__ void conv_finish(const ConvParams& p, float* v, int m, int ch0) {
    const int act = (p.flags >> CUTIE_ACT_SHIFT) & 7;
    const bool out_f32 = p.flags & CUTIE_F_OUT_F32;
    const bool vec_y = out_f32 ? ((p.ldy & 3) == 0) : ((p.ldy & 7) == 0);
    const bool vec_r = (p.ldr & 7) == 0;
    const bool full = ch0 + 7 < p.Cout;
    if (p.bias) {
        if (full && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) {      // two 16-byte loads (the per-channel form waits for each of its eight loads)
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + ch0), b1 = *reinterpret_cast<const float4*>(p.bias + ch0 + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (ch0 + r < p.Cout) v[r] += p.bias[ch0 + r];
        }
    }
    if (p.res) {
        const int mres = conv_res_row(p, m);
        const bf16_t* rp = p.res + (long)mres * p.ldr + ch0;
        if (full && vec_r) {
            const uint4 rr = *reinterpret_cast<const uint4*>(rp);
            v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
            v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
            v[4] += __uint_as_float(rr.z << 16); v[5] += __uint_as_float(rr.z & 0xffff0000u);
            v[6] += __uint_as_float(rr.w << 16); v[7] += __uint_as_float(rr.w & 0xffff0000u);
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (ch0 + r < p.Cout) v[r] += bf2f(rp[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (act == CUTIE_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
        else if (act == CUTIE_ACT_SIGMOID) v[r] = sigmoidf_(v[r]);
        else if (act == CUTIE_ACT_SQ1) v[r] = v[r] * v[r] + 1.f;
    }
    if (out_f32) {
        float* yp = reinterpret_cast<float*>(p.y) + (long)m * p.ldy + ch0;
        if (full && vec_y) {
            *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (ch0 + r < p.Cout) yp[r] = v[r];
        }
    } else {
        bf16_t* yp = reinterpret_cast<bf16_t*>(p.y) + (long)m * p.ldy + ch0;
        if (full && vec_y) {
            *reinterpret_cast<uint4*>(yp) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (ch0 + r < p.Cout) yp[r] = f2bf(v[r]);
        }
    }
}

// dynamic LDS of a conv tile: WK double-buffered operand pipelines, or the fp32 epilogue tile if that is larger
template <int BM, int BN, int BK, int WK>
constexpr int conv_lds_bytes() {
    constexpr int pipe = WK * 2 * (BM + BN) * (BK / 8) * 16, epi = BM * (BN + 4) * 4;
    return pipe > epi ? pipe : epi;
}

// Every kernel argument this block will need, requested in ONE batch at kernel entry.  hipcc otherwise loads them where they are first used,
// behind branches: three to four DEPENDENT scalar-memory round trips (cold scalar cache at kernel start: 400-800 cycles each) in the
// prologue of every launch (profiles/r03_conv_timeline.md: 1.1-1.8 us from kernel entry to the first DMA).
__device__ __forceinline__ void conv_preload_args(const ConvParams& p) {
#ifndef CONV_NO_PRELOAD
    // (one statement = one wait; 30 operands is the limit of an asm statement: x2 / gap / Kslice are loaded where the rarer paths use them)
    asm volatile("" :: "s"(p.x1), "s"(p.w), "s"(p.bias), "s"(p.res), "s"(p.y), "s"(p.zero),
                 "s"(p.B), "s"(p.H), "s"(p.W), "s"(p.C1), "s"(p.C2), "s"(p.ldx1), "s"(p.ldx2), "s"(p.OH), "s"(p.OW), "s"(p.Cout), "s"(p.ldy),
                 "s"(p.KH), "s"(p.KW), "s"(p.stride), "s"(p.pad), "s"(p.ldr), "s"(p.Kpad), "s"(p.flags), "s"(p.M), "s"(p.Cin), "s"(p.OHW),
                 "s"(p.nzero), "s"(gridDim.x), "s"(gridDim.y));
#endif
}

// ---- optional timeline (diagnostic libraries only: make timeline, -DCONV_TIMELINE; tools/conv_timeline.py) ----
// s_memtime stamps of every wave of a few blocks (logical tile 0, 1, nb/2, nb-1), parked in TL_BYTES of extra dynamic LDS behind the
// kernel's own (one ds_write per stamp: no VMEM traffic that would disturb the counted vmcnt waits) and copied to the split-K
// scratch (ConvParams::part, unused by the LDS-DMA kernels) when the wave ends.  Record: [0] = 0x544c0000 | count, [1] = logical << 32 |
// waves, then stamps (cycles << 8 | id).  Loop stamps stop at TL_MAX - 6 so that the closing stamps always fit.
#ifdef CONV_TIMELINE
#define TL_MAX 96
#define TL_BYTES (16 * TL_MAX * 8)
#define TL_DECL(...) unsigned long long* const tlp_ = reinterpret_cast<unsigned long long*>(__VA_ARGS__) + wave * TL_MAX; int tln_ = 0;
#define TL_PUT_(ID, CAP) { if (tln_ < (CAP)) { const unsigned long long t_ = (__builtin_readcyclecounter() << 8) | (unsigned)(ID); if ((threadIdx.x & 63) == 0) tlp_[tln_] = t_; ++tln_; } }
#define TL(ID) TL_PUT_(ID, TL_MAX - 6)
#define TLE(ID) TL_PUT_(ID, TL_MAX)
#define TL_DUMP(LOGICAL, NB, NWAVES)                                                                                        \
    {                                                                                                                       \
        const int slot_ = (LOGICAL) == 0 ? 0 : (LOGICAL) == 1 ? 1 : (LOGICAL) == (NB) / 2 ? 2 : (LOGICAL) == (NB) - 1 ? 3 : -1; \
        if (slot_ >= 0 && p.part) {                                                                                         \
            __builtin_amdgcn_s_waitcnt(0);                                                                                  \
            unsigned long long* o_ = reinterpret_cast<unsigned long long*>(p.part) + ((long)slot_ * 16 + wave) * (TL_MAX + 2); \
            if ((threadIdx.x & 63) == 0) { o_[0] = 0x544c0000ull | (unsigned)tln_; o_[1] = ((unsigned long long)(LOGICAL) << 32) | (unsigned)(NWAVES); } \
            for (int q_ = threadIdx.x & 63; q_ < tln_; q_ += 64) o_[2 + q_] = tlp_[q_];                                     \
        }                                                                                                                   \
    }
#else
#define TL_BYTES 0
#define TL_DECL(...)
#define TL(ID)
#define TLE(ID)
#define TL_DUMP(LOGICAL, NB, NWAVES)
#endif

int launch_conv_dma(const ConvParams& p, int tile, hipStream_t s);
int launch_conv_pc(const ConvParams& p, int tile, hipStream_t s);         // conv_pc.hip: producer / consumer tiles 100..
