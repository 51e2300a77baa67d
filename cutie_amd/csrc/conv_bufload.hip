// Implicit-GEMM convolution, buffer-load variant -- EXPERIMENTAL tiles 50.. (opt-in: CUTIE_AMD_EXPERIMENTAL_TILES=1, see
// cutie_amd/ops.py; never chosen otherwise).  Written at the end of round 1 from the static instruction mix of the default
// kernel (tools/isa_mix.py, profiles/r01_isa_mix.txt): conv_igemm_kernel issues 10-27 non-MFMA instructions per MFMA in its
// steady-state loop, ~300-500 of them VALU (per-chunk im2col address arithmetic in 64 bit, halo predicates, the two-source
// select, zero masking, re-derived LDS indices).  At 2 cycles per wave64 VALU instruction that is more vector-pipe time than
// the 48 MFMAs of the same loop need matrix-pipe time (~16 cycles each), and far more issue slots than one wave can spare
// between MFMAs.  This variant moves that work off the VALU:
//   * operands are fetched with `buffer_load_dwordx4 v, voffset, srsrc, soffset offen`: the per-thread byte offset of a
//     chunk (pixel base + 16-B chunk) is a loop-invariant VGPR, everything that changes per K tile -- filter tap, channel
//     offset, source tensor of a virtual concat, weight column -- is wave-uniform and goes into the SGPR soffset / srsrc;
//   * the halo needs no predicated pointer and no masking: a per-row bit mask over the filter taps is computed once, and an
//     invalid chunk is loaded at voffset 0x80000000 >= num_records, for which the buffer unit returns zeros;
//   * the resource base is shifted by -(pad rows, pad columns), so that every voffset / soffset is non-negative.
// LDS layout, swizzle, MFMA fragment layout, prefetch ring and epilogue are those of conv_igemm_kernel (conv_igemm.hip).
// Requirements (checked at launch): Cin % BK == 0 (and C1 % BK == 0 for a two-source input) so that a K tile never straddles
// a tap or a source; KH*KW <= 16; no split-K; every operand smaller than 2 GiB (32-bit buffer offsets).
#include "conv_common.h"

typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define BUF_WORD3 0x00020000          // gfx9 / CDNA raw buffer descriptor, dword 3: DATA_FORMAT = 32 bit, no swizzle, no tid
#define BUF_RECORDS 0x7fffffff        // bytes addressable through one descriptor
#define BUF_OOB 0x80000000u           // any voffset >= BUF_RECORDS reads as zero

__device__ __forceinline__ rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, BUF_RECORDS, BUF_WORD3);
}
__device__ __forceinline__ u32x4 bload16(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}

// ReLU on two packed bf16 in ONE VALU op: a bf16 is negative iff its bit pattern is a negative int16, and non-negative
// patterns are left alone by max(x, 0) -- v_pk_max_i16.  (Same result as relu_bf2 of common.h for every input incl. -0 / NaN.)
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu_bf2_pk(unsigned w) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), (s16x2){0, 0}));
}

// RELU: the fused input ReLU (CUTIE_F_RELU_IN); TWO: two-source (virtually concatenated) input.  Template flags, so that the
// common single-source / no-ReLU loop carries neither the select nor the branch.
template <int BM, int BN, int WM, int WN, int BK, int S, bool RELU, bool TWO>
__global__ __launch_bounds__(WM * WN * 64) void conv_bufload_kernel(ConvParams p) {
    constexpr int NT = WM * WN * 64;                    // 4 or 8 waves
    constexpr int CPR = BK / 8;                         // 16-B chunks per LDS row
    constexpr int RPT = NT / CPR;                       // rows covered by one pass of the threads
    constexpr int NX = (BM * CPR) / NT;                 // X chunks per thread per tile
    constexpr int NWC = (BN * CPR) / NT;                // W chunks per thread per tile
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16, KSUB = BK / 32;
    constexpr int LDC = BN + 4;
    static_assert((NT == 256 || NT == 512) && NX >= 1 && NWC >= 1 && TM >= 1 && TN >= 1 && S >= 2 &&
                  (BM * CPR) % NT == 0 && (BN * CPR) % NT == 0, "bad tile");
    extern __shared__ __attribute__((aligned(16))) u32x4 smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int m0, n0;                                          // XCD-aware tile mapping, as in conv_igemm_kernel
    {
        const int nb = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = id & 7, kq = id >> 3, q = nb >> 3, r = nb & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kq;
        const int mt = logical / (int)gridDim.y;
        m0 = mt * BM;
        n0 = (logical - mt * (int)gridDim.y) * BN;
    }
    const int kc = tid % CPR, trow = tid / CPR;

    // ---- loop-invariant per-thread state: byte offset of each chunk's centre pixel, and which taps fall inside the image ----
    unsigned xoff1[NX], xoff2[NX], vmask[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int m = m0 + trow + i * RPT;
        const bool valid = m < p.M;
        const int mm = valid ? m : 0;
        const int b = mm / p.OHW;
        const int rem = mm - b * p.OHW;
        const int oh = rem / p.OW;
        const int ow = rem - oh * p.OW;
        const int ih0 = oh * p.stride, iw0 = ow * p.stride;           // the descriptors start at (-pad, -pad)
        const unsigned pix = (unsigned)((b * p.H + ih0) * p.W + iw0);
        xoff1[i] = pix * (unsigned)(p.ldx1 * 2) + (unsigned)kc * 16u;
        xoff2[i] = TWO ? pix * (unsigned)(p.ldx2 * 2) + (unsigned)kc * 16u : 0u;
        unsigned mk = 0;
        int t = 0;
        for (int kh = 0; kh < p.KH; ++kh)
            for (int kw = 0; kw < p.KW; ++kw, ++t) {
                const int ih = ih0 - p.pad + kh, iw = iw0 - p.pad + kw;
                mk |= (valid && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? (1u << t) : 0u;
            }
        vmask[i] = mk;
    }
    unsigned woff[NWC];
#pragma unroll
    for (int i = 0; i < NWC; ++i) woff[i] = (unsigned)((n0 + trow + i * RPT) * p.Kpad * 2) + (unsigned)kc * 16u;
    const long shift = (long)p.pad * p.W + p.pad;                    // pixels
    const bf16_t* xb1 = p.x1 - shift * p.ldx1;
    const bf16_t* xb2 = TWO ? p.x2 - shift * p.ldx2 : xb1;
    const rsrc_t rw = make_rsrc(p.w), rx1 = make_rsrc(xb1);
    // LDS indices (16-B units) are loop invariant: rows handled by one thread differ by multiples of 16, which the chunk
    // swizzle ignores, so every access is <per-thread base> + <compile-time offset>
    static_assert(RPT % 16 == 0 && (BM / WM) % 16 == 0 && (BN / WN) % 16 == 0, "swizzle must not depend on the unrolled row offset");
    const int wrx = trow * CPR + (kc ^ swz<CPR>(trow));              // staging stores: X row trow (+ i*RPT), W row BM + trow (+ i*RPT)

    // ---- wave-uniform K-tile state (scalar registers): tap (kh, kw), channel offset inside Cin, weight column ----
    int tap = 0, kh = 0, kw = 0, c0 = 0;
    unsigned wsoff = 0;
    const int nk = p.Kslice / BK;

    u32x4 xr[S][NX], wr[S][NWC];
#define LOAD_TILE(SLOT)                                                                                    \
    {                                                                                                      \
        const unsigned tapbit_ = 1u << (tap & 31);                                                         \
        if (TWO) {                                                                                         \
            const bool in1_ = c0 < p.C1;                         /* uniform: scalar selects, no branch */  \
            const unsigned soff_ = (unsigned)(((kh * p.W + kw) * (in1_ ? p.ldx1 : p.ldx2) + (in1_ ? c0 : c0 - p.C1)) * 2); \
            const rsrc_t rx_ = make_rsrc(in1_ ? xb1 : xb2);                                                \
            _Pragma("unroll") for (int i = 0; i < NX; ++i)                                                 \
                xr[SLOT][i] = bload16(rx_, (vmask[i] & tapbit_) ? (in1_ ? xoff1[i] : xoff2[i]) : BUF_OOB, soff_); \
        } else {                                                                                           \
            const unsigned soff_ = (unsigned)(((kh * p.W + kw) * p.ldx1 + c0) * 2);                        \
            _Pragma("unroll") for (int i = 0; i < NX; ++i)                                                 \
                xr[SLOT][i] = bload16(rx1, (vmask[i] & tapbit_) ? xoff1[i] : BUF_OOB, soff_);              \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NWC; ++i) wr[SLOT][i] = bload16(rw, woff[i], wsoff);         \
        wsoff += BK * 2;                                                                                   \
        const bool wrap_ = c0 + BK >= p.Cin;                                                               \
        c0 = wrap_ ? 0 : c0 + BK;                                                                          \
        tap += wrap_ ? 1 : 0;                                                                              \
        kw += wrap_ ? 1 : 0;                                                                               \
        const bool wrap2_ = kw == p.KW;                                                                    \
        kw = wrap2_ ? 0 : kw;                                                                              \
        kh += wrap2_ ? 1 : 0;                                                                              \
    }
#define STORE_TILE(BUF, SLOT)                                                                              \
    {                                                                                                      \
        u32x4* dst_ = smem_raw + (BUF) * ((BM + BN) * CPR) + wrx;                                          \
        _Pragma("unroll") for (int i = 0; i < NX; ++i) {                                                   \
            u32x4 v = xr[SLOT][i];                                                                         \
            if (RELU) { v.x = relu_bf2_pk(v.x); v.y = relu_bf2_pk(v.y); v.z = relu_bf2_pk(v.z); v.w = relu_bf2_pk(v.w); } \
            dst_[i * RPT * CPR] = v;                                                                       \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NWC; ++i) dst_[(BM + i * RPT) * CPR] = wr[SLOT][i];          \
    }

    const int wm = wave / WN, wn = wave % WN;
    const int pm0 = wm * (BM / WM), cn0 = wn * (BN / WN);
    const int l15 = lane & 15, l4 = lane >> 4;
    int rdx[KSUB], rdw[KSUB];                                        // fragment reads: tile row t adds t*16*CPR
#pragma unroll
    for (int j = 0; j < KSUB; ++j) {
        rdx[j] = (pm0 + l15) * CPR + ((j * 4 + l4) ^ swz<CPR>(pm0 + l15));
        rdw[j] = (BM + cn0 + l15) * CPR + ((j * 4 + l4) ^ swz<CPR>(cn0 + l15));
    }
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // K loop: tile t lives in ring slot t % S; LDS is double-buffered; one barrier per tile (as conv_igemm_kernel).
    // Tiles past nk (only in the prologue of very short K loops) would read weight columns >= Kpad: guarded.
#pragma unroll
    for (int t = 0; t < S - 1; ++t)
        if (t < nk) LOAD_TILE(t);
    STORE_TILE(0, 0);
    __syncthreads();
#define K_ITER(KS, U, DO_LOAD, DO_STORE)                                                                   \
    {                                                                                                      \
        const int buf = (KS) & 1;                                                                          \
        if (DO_LOAD) LOAD_TILE(((U) + S - 1) % S);                                                         \
        _Pragma("unroll") for (int j = 0; j < KSUB; ++j) {                                                 \
            bf16x8 bfr[TM], afr[TN];                                                                       \
            const u32x4* src_ = smem_raw + buf * ((BM + BN) * CPR);                                        \
            _Pragma("unroll") for (int t = 0; t < TM; ++t) bfr[t] = __builtin_bit_cast(bf16x8, src_[rdx[j] + t * 16 * CPR]); \
            _Pragma("unroll") for (int t = 0; t < TN; ++t) afr[t] = __builtin_bit_cast(bf16x8, src_[rdw[j] + t * 16 * CPR]); \
            _Pragma("unroll") for (int a = 0; a < TN; ++a)                                                 \
                _Pragma("unroll") for (int b = 0; b < TM; ++b)                                             \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[a], bfr[b], acc[a][b], 0, 0, 0); \
        }                                                                                                  \
        if (DO_STORE) STORE_TILE(buf ^ 1, ((U) + 1) % S);                                                  \
        __syncthreads();                                                                                   \
    }
    int ks0 = 0;
    for (; ks0 + 2 * S - 1 <= nk; ks0 += S) {
#pragma unroll
        for (int u = 0; u < S; ++u) K_ITER(ks0 + u, u, true, true)
    }
    for (; ks0 < nk; ks0 += S) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int ks = ks0 + u;
            if (ks < nk) K_ITER(ks, u, ks + S - 1 < nk, ks + 1 < nk)
        }
    }
#undef K_ITER
#undef STORE_TILE
#undef LOAD_TILE

    // ---- epilogue: fp32 tile transposed through LDS, 16-B stores along the channel axis (as conv_igemm_kernel) ----
    float* ctile = reinterpret_cast<float*>(smem_raw);
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int px = pm0 + b * 16 + l15, ch = cn0 + a * 16 + l4 * 4;
            *reinterpret_cast<f32x4*>(ctile + px * LDC + ch) = acc[a][b];
        }
    __syncthreads();
    constexpr int CH8 = BN / 8;
    for (int q = threadIdx.x; q < BM * CH8; q += NT) {
        const int px = q / CH8, c8 = q - px * CH8;
        const int m = m0 + px, ch0 = n0 + c8 * 8;
        if (m >= p.M || ch0 >= p.Cout) continue;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(ctile + px * LDC + c8 * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(ctile + px * LDC + c8 * 8 + 4);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        conv_finish(p, v, m, ch0);
    }
}

template <int BM, int BN, int WM, int WN, int BK, int S, bool RELU, bool TWO>
static int launch_buf2(const ConvParams& p, hipStream_t s, int gy) {
    constexpr int lds = conv_lds_bytes<BM, BN, BK, 1>();
    static bool attr_set = false;                        // one flag per instantiation
    if (!attr_set) {
        if (lds > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bufload_kernel<BM, BN, WM, WN, BK, S, RELU, TWO>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            cutie_set_error("conv buffer-load tile: cannot raise the dynamic LDS limit to %d bytes", lds);
            return -2;
        }
        attr_set = true;
    }
    dim3 grid((p.M + BM - 1) / BM, (unsigned)gy);
    hipLaunchKernelGGL((conv_bufload_kernel<BM, BN, WM, WN, BK, S, RELU, TWO>), grid, dim3(WM * WN * 64), lds, s, p);
    return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int BK, int S>
static int launch_buf(ConvParams p, hipStream_t s) {
    const long x1_bytes = (long)p.B * p.H * p.W * p.ldx1 * 2, x2_bytes = p.C2 ? (long)p.B * p.H * p.W * p.ldx2 * 2 : 0;
    const long gy = (p.Cout + BN - 1) / BN, w_bytes = gy * BN * (long)p.Kpad * 2;
    if (p.Kpad % BK || p.Cin % BK || (p.C2 && p.C1 % BK) || p.KH * p.KW > 16 || p.splitk != 1 ||
        x1_bytes >= BUF_RECORDS || x2_bytes >= BUF_RECORDS || w_bytes >= BUF_RECORDS) {
        cutie_set_error("conv buffer-load tile: needs Cin %% %d == 0 (C1 too for two sources), KH*KW <= 16, no split-K, operands < 2 GiB "
                        "(Cin=%d C1=%d Kpad=%d k=%dx%d splitk=%d)", BK, p.Cin, p.C1, p.Kpad, p.KH, p.KW, p.splitk);
        return -2;
    }
    p.Kslice = p.Kpad;
    const bool relu = p.flags & CUTIE_F_RELU_IN, two = p.C2 != 0;
    if (relu) return two ? launch_buf2<BM, BN, WM, WN, BK, S, true, true>(p, s, (int)gy) : launch_buf2<BM, BN, WM, WN, BK, S, true, false>(p, s, (int)gy);
    return two ? launch_buf2<BM, BN, WM, WN, BK, S, false, true>(p, s, (int)gy) : launch_buf2<BM, BN, WM, WN, BK, S, false, false>(p, s, (int)gy);
}

// tile table (mirrored by cutie_amd/ops.py:EXPERIMENTAL_TILES): id -> BM, BN, BK
int launch_conv_bufload(const ConvParams& p, int tile, hipStream_t s) {
    switch (tile) {
        case 50: return launch_buf<128, 64, 2, 2, 64, 3>(p, s);
        case 51: return launch_buf<64, 64, 2, 2, 64, 4>(p, s);
        case 52: return launch_buf<64, 128, 2, 2, 64, 3>(p, s);
        case 53: return launch_buf<32, 64, 2, 2, 128, 3>(p, s);
        case 54: return launch_buf<64, 64, 2, 2, 128, 3>(p, s);
        case 55: return launch_buf<128, 128, 2, 4, 64, 3>(p, s);        // 8 waves
        case 56: return launch_buf<32, 64, 2, 2, 64, 4>(p, s);
        default: cutie_set_error("conv: bad experimental tile id %d", tile); return -2;
    }
}
