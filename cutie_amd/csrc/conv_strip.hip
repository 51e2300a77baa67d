// 3x3 / stride 1 / pad 1 convolution on narrow feature maps with the input STRIP resident in LDS -- tiles 90..
//
// Why: conv_dma_kernel is bound by the L2 -> LDS fill (profiles/r02_conv_ablation.md: removing the MFMAs or the fragment reads
// changes nothing, removing the DMA saves a quarter of the kernel), and a 3x3 conv streams every input row NINE times, once per
// tap.  The BM output pixels of a block are consecutive in memory (pixel index m = (b*H + y)*W + x), so ALL their taps live in the
// contiguous pixel range [m0 - W - 1, m0 + BM + W + 1): this kernel fetches that strip ONCE per 64-channel slice (BM + 2W + 2 rows of
// 128 B, double-buffered across slices) and walks the nine taps over it as row shifts of the fragment reads; only the weight
// tiles stream per tap.  DMA bytes per 128 x 64 output tile at W = 54, Cin = 256: 122 KB of pixels + 295 KB of weights instead of
// 590 + 295 KB.
//
// What a tap is here: output row r of the tile reads strip row r + kh*W + kw (strip row 0 = pixel m0 - W - 1).  Rows that the
// shift takes across the left / right image border, above / below the image or into the neighbouring object are real pixels of the
// wrong place, so they are zeroed per (row, tap) after the LDS read (a 9-bit mask per fragment row, computed once; the test is
// skipped for taps where no lane of the wave needs it).  Strip rows outside the tensor are out-of-range buffer offsets: zeros.
// The swizzle of conv_dma.hip is kept (16-B chunk c of strip row s sits at chunk c ^ (s & 7)): the shift changes s & 7, so the
// XOR term is recomputed per tap (3 VALU).  Pipeline: one K step = (channel slice, tap); the weight ring and the counted-vmcnt /
// raw-barrier protocol are those of conv_dma.hip; the next slice's strip is requested one 1-KiB piece per wave and tap during taps
// 0..7 of the current slice (tap 8 re-requests the last piece, so that nothing the next slice needs is younger than the wait
// count allows).  Epilogue, GAP side jobs, XCD-aware block order: as conv_dma.hip.
// Requirements (launch_conv_strip): k = 3, stride 1, pad 1, Cin % 64 == 0 (both sources), no split-K, strip pieces per wave <= 8 (ring of 3) / 7 (ring of 4).
#include "conv_common.h"

typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define DMA_WORD3 0x00020000
#define DMA_OOB 0x80000000u
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define WAIT_VMCNT_LDS(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | (7 << 4) | (0 << 8))
#define TILE_SYNC(N) { WAIT_VMCNT_LDS(N); asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

typedef short s16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu2_(unsigned w) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_, w), (s16x2_){0, 0}));
}

template <int BM, int BN, int WM, int WN, int NSW, bool RELU, bool TWO>
__global__ __launch_bounds__(WM * WN * 64) void conv_strip_kernel(ConvParams p, int SRP) {
#if __HIP_DEVICE_COMPILE__
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int NWI = BN / 8 / NW;                     // weight pieces (8 rows x 128 B) per wave per K step
    constexpr int LPT = 1 + NWI;                         // DMA instructions per wave per K step: one strip piece + the weight pieces
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int WSTAGE = BN * 128;                     // bytes of one weight stage
    constexpr int LDC = BN + 4;
    static_assert(NWI >= 1 && NWI * 8 * NW == BN && TM >= 1 && TN >= 1 && NSW == 3 && NT % (BN / 8) == 0, "bad strip tile");
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const int XB = SRP * 128;                            // bytes of one strip buffer; layout [strip 0 | strip 1 | weight ring]
    const int wring = 2 * XB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int m0, n0;
    {
        const int nb = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = id & 7, kq = id >> 3, q = nb >> 3, r = nb & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kq;
        const int mt = logical / (int)gridDim.y;
        m0 = mt * BM;
        n0 = (logical - mt * (int)gridDim.y) * BN;
    }
    if (p.zero && blockIdx.x == 0 && blockIdx.y == 0)
        for (int z = tid; z < p.nzero; z += NT) p.zero[z] = 0ull;
    constexpr int CH8 = BN / 8, PSTEP = NT / CH8;
    const int ec8 = tid % CH8, epx0 = tid / CH8, ech0 = n0 + ec8 * 8;
    float bias[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) bias[r] = (p.bias && ech0 + r < p.Cout) ? p.bias[ech0 + r] : 0.f;

    // ---- per-lane DMA offsets ----
    const int lr = lane >> 3;
    const unsigned kcb = (unsigned)(((lane & 7) ^ lr) * 16);
    const int NP = SRP >> 3;                             // strip pieces
    // slot q (0..8) of this wave = piece q*NW + wave, clamped to the last piece (slots past the end re-request it: harmless)
    unsigned xoff1[8], xoff2[8];
    int xdst[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int piece = min(q * NW + wave, NP - 1);
        const int gm = m0 - p.W - 1 + piece * 8 + lr;    // pixel of this strip row (may lie outside the tensor)
        const bool ok = (unsigned)gm < (unsigned)p.M;
        xoff1[q] = ok ? (unsigned)gm * (unsigned)(p.ldx1 * 2) + kcb : DMA_OOB;
        xoff2[q] = TWO ? (ok ? (unsigned)gm * (unsigned)(p.ldx2 * 2) + kcb : DMA_OOB) : 0u;
        xdst[q] = piece * 1024;
    }
    unsigned woff[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) woff[i] = (unsigned)((n0 + (wave * NWI + i) * 8 + lr) * p.Kpad * 2) + kcb;
    const rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, 0x7fffffff, DMA_WORD3);
    const rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x1), 0, 0x7fffffff, DMA_WORD3);
    const rsrc_t rx2 = TWO ? __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x2), 0, 0x7fffffff, DMA_WORD3) : rx1;
    const int wdst = wave * NWI * 1024;

    // ---- fragment geometry ----
    const int wm = wave / WN, wn = wave % WN;
    const int pm0 = wm * (BM / WM), cn0 = wn * (BN / WN);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int cj0 = l4 << 4, cj1 = (4 + l4) << 4;        // unswizzled chunk byte offsets of the two k-steps of a 64-channel slice
    int rdw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) rdw[j] = ((cn0 + l15) * 8 + ((j * 4 + l4) ^ (l15 & 7))) * 16;
    // validity of the nine taps for the TM fragment rows of this lane (bit kh*3 + kw)
    unsigned vm[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int m = m0 + pm0 + t * 16 + l15;
        unsigned mk = 0;
        if (m < p.M) {
            const int rem = m % p.OHW, oy = rem / p.OW, ox = rem - oy * p.OW;
            unsigned cols = 0;
            for (int k = 0; k < 3; ++k) cols |= ((unsigned)(ox - 1 + k) < (unsigned)p.W) ? (1u << k) : 0u;
            for (int k = 0; k < 3; ++k) mk |= ((unsigned)(oy - 1 + k) < (unsigned)p.H) ? (cols << (3 * k)) : 0u;
        }
        vm[t] = mk;
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int cin2 = p.Cin * 2;                          // bytes of one tap in a weight row
    const int nslice = p.Cin / 64, nk = nslice * 9;
    const int ZOFF = wring + NSW * WSTAGE;               // a 128-B row of zeros: what a fragment row reads for a tap outside its image
    if (tid < 8) *reinterpret_cast<u32x4*>(lds + ZOFF + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
    const char* const zrow = lds + ZOFF;

    // slice -> (source descriptor, channel byte offset); wave-uniform
#define X_SRC(SL, R, SOFF, FIRST)                                                                          \
    const bool FIRST = !TWO || (SL) * 64 < p.C1;                                                           \
    const rsrc_t R = FIRST ? rx1 : rx2;                                                                    \
    const int SOFF = (FIRST ? (SL) * 64 : (SL) * 64 - p.C1) * 2;
    // the weight tile of K step (slice SL, tap TP) into ring stage ST (all compile-time but SL)
#define LOAD_W(SL, TP, ST)                                                                                 \
    {                                                                                                      \
        const int ws_ = (TP) * cin2 + (SL) * 128;                                                          \
        _Pragma("unroll") for (int i = 0; i < NWI; ++i)                                                    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(lds + wring + (ST) * WSTAGE + wdst + i * 1024), 16, woff[i], ws_, 0, 0); \
    }

    // ---- prologue: strip of slice 0 (all pieces of this wave), then the weight tiles of taps 0 and 1 ----
    {
        X_SRC(0, r0_, soff0_, first0_)
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q * NW + wave < NP)                      // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r0_, LDS_PTR(lds + xdst[q]), 16, first0_ ? xoff1[q] : xoff2[q], soff0_, 0, 0);
    }
    LOAD_W(0, 0, 0)
    LOAD_W(0, 1, 1)
    TILE_SYNC(NWI)                                       // strip 0 and weight tile 0 have landed; tile 1 may be in flight

    // ---- K loop: slices outside, the nine taps unrolled (tap, ring stage, strip slot and mask bit are compile-time) ----
    const int sW1 = p.W, sW2 = 2 * p.W;
    int xcur = 0;
    for (int slice = 0; slice < nslice; ++slice) {
        const bool more = slice + 1 < nslice;
        X_SRC(more ? slice + 1 : slice, rn_, soffn_, firstn_)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
            const int shift = (kh == 0 ? 0 : kh == 1 ? sW1 : sW2) + kw;
            const int sb = pm0 + l15 + shift;
            const int swz = (sb & 7) << 4;
            const char* const xrow = lds + xcur + (sb << 7);
            const int c0_ = cj0 ^ swz, c1_ = cj1 ^ swz;
            bf16x8 b0[TM], b1[TM], a0[TN], a1[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const char* const ra = ((vm[t] >> tap) & 1u) ? xrow + t * 2048 : zrow;      // outside the image: the zero row
                u32x4 v0 = *reinterpret_cast<const u32x4*>(ra + c0_);
                u32x4 v1 = *reinterpret_cast<const u32x4*>(ra + c1_);
                if (RELU) { v0.x = relu2_(v0.x); v0.y = relu2_(v0.y); v0.z = relu2_(v0.z); v0.w = relu2_(v0.w);
                            v1.x = relu2_(v1.x); v1.y = relu2_(v1.y); v1.z = relu2_(v1.z); v1.w = relu2_(v1.w); }
                b0[t] = __builtin_bit_cast(bf16x8, v0); b1[t] = __builtin_bit_cast(bf16x8, v1);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                a0[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds + wring + (tap % 3) * WSTAGE + rdw[0] + t * 2048));
                a1[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds + wring + (tap % 3) * WSTAGE + rdw[1] + t * 2048));
            }
            // requests of this step: one strip piece of the NEXT slice into the other buffer (slot = tap; tap 8 repeats slot 7; no next
            // slice: a zero fill there) and the weight tile two steps ahead
            const bool has_w = slice * 9 + tap + 2 < nk;
            if (has_w) {
                const int q_ = tap < 8 ? tap : 7;
                const int dst_ = min(q_ * NW + wave, NP - 1) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rn_, LDS_PTR(lds + (xcur ^ XB) + dst_), 16,
                                                         more ? (firstn_ ? xoff1[q_] : xoff2[q_]) : DMA_OOB, soffn_, 0, 0);
                if (tap + 2 < 9) LOAD_W(slice, tap + 2, (tap + 2) % 3)
                else LOAD_W(slice + 1, tap + 2 - 9, (tap + 2) % 3)
            }
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[a], b0[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[a], b1[b], acc[a][b], 0, 0, 0);
                }
            if (has_w) { TILE_SYNC(LPT) } else { TILE_SYNC(0) }
        }
        xcur ^= XB;
    }
#undef LOAD_W
#undef X_SRC

    // ---- epilogue (as conv_dma.hip): fp32 tile through LDS, bias / residual / activation, 16-B stores, GAP side job ----
    float* ctile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int px = pm0 + b * 16 + l15, ch = cn0 + a * 16 + l4 * 4;
            *reinterpret_cast<f32x4*>(ctile + px * LDC + ch) = acc[a][b];
        }
    __syncthreads();
    const bool active = ech0 < p.Cout;
    if (!active && !p.gap) return;
    const int act = (p.flags >> CUTIE_ACT_SHIFT) & 7;
    const bool out_f32 = p.flags & CUTIE_F_OUT_F32, res_bcast = p.flags & CUTIE_F_RES_BCAST;
    const bool fast = ech0 + 7 < p.Cout && (out_f32 ? (p.ldy & 3) == 0 : (p.ldy & 7) == 0) && (!p.res || (p.ldr & 7) == 0);
    if (active)
    for (int px = epx0; px < BM; px += PSTEP) {
        const int m = m0 + px;
        if (m >= p.M) break;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(ctile + px * LDC + ec8 * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(ctile + px * LDC + ec8 * 8 + 4);
        float v[8] = {lo[0] + bias[0], lo[1] + bias[1], lo[2] + bias[2], lo[3] + bias[3],
                      hi[0] + bias[4], hi[1] + bias[5], hi[2] + bias[6], hi[3] + bias[7]};
        if (!fast) {
            ConvParams q = p;
            q.bias = nullptr;
            conv_finish(q, v, m, ech0);
            continue;
        }
        if (p.res) {
            const int mres = res_bcast ? (m % p.OHW) : m;
            const uint4 rr = *reinterpret_cast<const uint4*>(p.res + (long)mres * p.ldr + ech0);
            v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
            v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
            v[4] += __uint_as_float(rr.z << 16); v[5] += __uint_as_float(rr.z & 0xffff0000u);
            v[6] += __uint_as_float(rr.w << 16); v[7] += __uint_as_float(rr.w & 0xffff0000u);
        }
        if (act == CUTIE_ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if (act == CUTIE_ACT_SIGMOID) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = sigmoidf_(v[r]);
        } else if (act == CUTIE_ACT_SQ1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = v[r] * v[r] + 1.f;
        }
        if (out_f32) {
            float* yp = reinterpret_cast<float*>(p.y) + (long)m * p.ldy + ech0;
            *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            bf16_t* yp = reinterpret_cast<bf16_t*>(p.y) + (long)m * p.ldy + ech0;
            const uint4 o = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
            *reinterpret_cast<uint4*>(yp) = o;
            if (p.gap) {
                const f32x4 a = {__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u), __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u)};
                const f32x4 b = {__uint_as_float(o.z << 16), __uint_as_float(o.z & 0xffff0000u), __uint_as_float(o.w << 16), __uint_as_float(o.w & 0xffff0000u)};
                *reinterpret_cast<f32x4*>(ctile + px * LDC + ec8 * 8) = a;
                *reinterpret_cast<f32x4*>(ctile + px * LDC + ec8 * 8 + 4) = b;
            }
        }
    }
    if (p.gap) {
        __syncthreads();
        constexpr int PARTS = NT / BN, RPART = BM / PARTS;
        static_assert(NT % BN == 0 && BM % PARTS == 0, "GAP row parts");
        const int c = tid % BN, part = tid / BN;
        const int rows = min(BM, p.M - m0);
        if (n0 + c < p.Cout) {
            const int r0 = part * RPART, r1 = min(r0 + RPART, rows);
            int m = m0 + r0, obj = m / p.OHW, next_b = (obj + 1) * p.OHW;
            float sum = 0.f;
            for (int r = r0; r < r1; ++r, ++m) {
                if (m == next_b) {
                    atomicAdd(reinterpret_cast<unsigned long long*>(p.gap + (long)obj * p.Cout + n0 + c), (unsigned long long)__float2ll_rn(sum * GAP_FIXED_SCALE));
                    sum = 0.f; ++obj; next_b += p.OHW;
                }
                sum += ctile[r * LDC + c];
            }
            if (r1 > r0) atomicAdd(reinterpret_cast<unsigned long long*>(p.gap + (long)obj * p.Cout + n0 + c), (unsigned long long)__float2ll_rn(sum * GAP_FIXED_SCALE));
        }
    }
#endif
}

template <int BM, int BN, int WM, int WN, int NSW, bool RELU, bool TWO>
static int launch_strip2(const ConvParams& p, hipStream_t s, int SRP, int lds) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_strip_kernel<BM, BN, WM, WN, NSW, RELU, TWO>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            cutie_set_error("conv strip tile: cannot raise the dynamic LDS limit");
            return -2;
        }
        attr_set = true;
    }
    dim3 grid((p.M + BM - 1) / BM, (unsigned)((p.Cout + BN - 1) / BN));
    hipLaunchKernelGGL((conv_strip_kernel<BM, BN, WM, WN, NSW, RELU, TWO>), grid, dim3(WM * WN * 64), lds, s, p, SRP);
    return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int NSW>
static int launch_strip(ConvParams p, hipStream_t s) {
    constexpr int NW = WM * WN;
    const int SRP = (BM + 2 * p.W + 2 + 7) & ~7;         // strip rows, rounded up to whole pieces
    const int ppw = ((SRP >> 3) + NW - 1) / NW;          // pieces per wave
    const int lds_pipe = 2 * SRP * 128 + NSW * BN * 128 + 128, lds_epi = BM * (BN + 4) * 4;
    const int lds = lds_pipe > lds_epi ? lds_pipe : lds_epi;
    const long gy = (p.Cout + BN - 1) / BN;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.OH != p.H || p.OW != p.W || p.Cin % 64 || (p.C2 && p.C1 % 64) ||
        p.splitk != 1 || p.Kpad < 9 * p.Cin || ppw > 11 - NSW || ppw > 8 || lds > 160 * 1024 ||
        (long)p.M * p.ldx1 * 2 >= 0x7fff0000L || (p.C2 && (long)p.M * p.ldx2 * 2 >= 0x7fff0000L) || gy * BN * (long)p.Kpad * 2 >= 0x7fff0000L) {
        cutie_set_error("conv strip tile: needs 3x3 / stride 1 / pad 1, Cin %% 64 == 0 (C1 too), no split-K, a strip of <= %d pieces per wave "
                        "(W=%d: %d) and <= 160 KB of LDS (%d)", 11 - NSW, p.W, ppw, lds);
        return -2;
    }
    const bool relu = p.flags & CUTIE_F_RELU_IN, two = p.C2 != 0;
    if (relu) { if (two) return launch_strip2<BM, BN, WM, WN, NSW, true, true>(p, s, SRP, lds); return launch_strip2<BM, BN, WM, WN, NSW, true, false>(p, s, SRP, lds); }
    if (two) return launch_strip2<BM, BN, WM, WN, NSW, false, true>(p, s, SRP, lds);
    return launch_strip2<BM, BN, WM, WN, NSW, false, false>(p, s, SRP, lds);
}

int launch_conv_strip(const ConvParams& p, int tile, hipStream_t s) {
    switch (tile) {
        case 90: return launch_strip<128, 64, 2, 2, 3>(p, s);
        case 91: return launch_strip<64, 64, 2, 2, 3>(p, s);
        case 92: return launch_strip<128, 128, 2, 4, 3>(p, s);      // 8 waves
        case 93: return launch_strip<64, 128, 2, 2, 3>(p, s);
        case 94: return launch_strip<128, 64, 4, 2, 3>(p, s);       // 8 waves
        case 95: return launch_strip<64, 64, 2, 4, 3>(p, s);        // 8 waves
        default: cutie_set_error("conv: bad strip tile id %d", tile); return -2;
    }
}
