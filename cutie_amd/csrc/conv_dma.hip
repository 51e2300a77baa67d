// Implicit-GEMM convolution with LDS-DMA operand staging (gfx950: buffer_load_dwordx4 ... lds) -- tiles 60..
//
// Why a third conv kernel: the register-staged kernel (conv_igemm.hip) issues 10-27 non-MFMA instructions per MFMA in its
// steady-state loop (per-chunk im2col address arithmetic, halo predicates, zero masking, ds_write staging, LDS index
// arithmetic; profiles/r01_isa_mix.txt) and therefore runs at 8 % of the MFMA roofline: the vector pipe, not the matrix pipe,
// is the bound.  Here nothing of that is left in the loop:
//   * operands go global -> LDS directly (1 KiB per wave-instruction, no VGPR round trip, no ds_write, no staging registers);
//     the LDS image of a DMA is lane-linear (M0 base + lane * 16), so the bank-conflict swizzle is applied to the per-lane SOURCE
//     address: lane l of an 8-row x 128-B piece fetches k-chunk (l & 7) ^ (l >> 3) of row l >> 3 (the 8 lanes of a row still
//     cover one whole 128-B line), and the fragment reads apply the same XOR;
//   * the per-lane byte offset of a chunk (pixel base + swizzled chunk) is a loop-invariant VGPR; everything that changes per
//     K tile -- filter tap, channel offset, source tensor of a virtual concat, weight column -- is wave-uniform and lives in
//     the SGPR soffset / the descriptor;
//   * the halo needs no predicated pointer and no masking: a per-row bit mask over the filter taps is computed once and an
//     invalid chunk is fetched at voffset 0x80000000 >= num_records, for which the buffer unit delivers zeros;
//   * NS-deep LDS ring, loads of tile t+NS-1 in flight while tile t is multiplied, ONE barrier per K tile, counted vmcnt
//     (raw s_barrier behind `s_waitcnt vmcnt(N) lgkmcnt(0)`: __syncthreads() would drain the DMA queue);
//   * the fused input ReLU is applied to the pixel fragments after the LDS read (one v_pk_max_i16 per dword).
// Steady-state loop of the 128x128 tile: 32 MFMA, 16 ds_read_b128, 8 LDS-DMA + 8 s_mov m0, ~12 VALU (3x3 halo select), ~10 SALU
// per wave and K tile: ~1.5 non-MFMA per MFMA.
// Requirements (checked at launch): Cin % 64 == 0 (C1 % 64 == 0 as well for a two-source input), KH*KW <= 32, no split-K,
// every operand < 2 GiB (32-bit buffer offsets).  GEMM view, fragment layout and epilogue are those of conv_igemm.hip:
// D[cout][pixel], weights = MFMA A operand, a lane owns 4 consecutive output channels of one pixel.
#include "conv_common.h"
#include <stdlib.h>

typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define DMA_WORD3 0x00020000          // raw buffer descriptor, dword 3: DATA_FORMAT = 32 bit, no swizzle, no tid
#define DMA_RECORDS 0x7fffffff        // bytes addressable through one descriptor
#define DMA_OOB 0x80000000u           // any voffset >= DMA_RECORDS reads as zero
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
// s_waitcnt with only vmcnt counted (gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14] << 4, expcnt [6:4], lgkmcnt [11:8])
// ... and with lgkmcnt(0): this wave's LDS reads have RETURNED (what __syncthreads() waits for before its s_barrier, minus the
// vmcnt(0) that would drain the DMA queue).  Needed before every barrier behind which another wave may overwrite what this wave
// read: the next DMA into the stage just multiplied and, with no latency cushion at all, the epilogue's ds_write into the ring.
#define WAIT_VMCNT_LDS(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | (7 << 4) | (0 << 8))
// Raw s_barrier (no vmcnt drain), fenced for the compiler: no LDS access or DMA may move across it.
#define TILE_SYNC(N) { WAIT_VMCNT_LDS(N); asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

__device__ __forceinline__ rsrc_t dma_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, DMA_RECORDS, DMA_WORD3);
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu2(unsigned w) {         // ReLU on two packed bf16: v_pk_max_i16 with 0
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), (s16x2){0, 0}));
}

template <int BM, int BN, int NS>
constexpr int dma_lds_bytes() {
    constexpr int pipe = NS * (BM + BN) * 128, epi = BM * (BN + 4) * 4;
    return pipe > epi ? pipe : epi;
}

// HALO: some filter taps can fall outside the image (pad > 0); RELU: fused input ReLU; TWO: two-source (virtual concat) input.
template <int BM, int BN, int WM, int WN, int NS, bool HALO, bool RELU, bool TWO>
__global__ __launch_bounds__(WM * WN * 64) void conv_dma_kernel(ConvParams p) {
#if __HIP_DEVICE_COMPILE__     // (the host pass only needs the launch stub; the LDS-DMA builtin and the LDS address space exist on the device side)
    constexpr int NW = WM * WN, NT = NW * 64, CPR = 8;  // BK = 64: 8 chunks of 16 B per LDS row
    constexpr int NXI = BM / 8 / NW;                    // X pieces (8 rows x 128 B) per wave per K tile
    constexpr int NWI = BN / 8 / NW;                    // W pieces per wave per K tile
    constexpr int LPT = NXI + NWI;                      // DMA instructions per wave per K tile
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int STAGE = (BM + BN) * CPR;              // 16-B units per ring stage
    constexpr int LDC = BN + 4;
    static_assert(NXI >= 1 && NWI >= 1 && NXI * 8 * NW == BM && NWI * 8 * NW == BN && TM >= 1 && TN >= 1 && NS >= 2 && NS <= 4 && NXI <= 4 && NWI <= 4 &&
                  (NS - 2) * LPT < 64, "bad tile");
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // provably wave-uniform (LDS base of the DMA goes to M0)
    int m0, n0;                                          // XCD-aware tile mapping, as in conv_igemm_kernel
    {
        const int nb = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = id & 7, kq = id >> 3, q = nb >> 3, r = nb & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kq;
        const int mt = logical / (int)gridDim.y;
        m0 = mt * BM;
        n0 = (logical - mt * (int)gridDim.y) * BN;
    }
    // ---- loop-invariant per-lane state: byte offset of this lane's chunk in every piece, tap validity mask ----
    // The pieces of one operand share ONE M0 value per stage: piece i is issued with the instruction offset i * 1024, which the
    // hardware adds to the LDS address AND to the global address; the per-lane offset of piece i is lowered by the same amount
    // (the descriptors start PRE = 4096 bytes early, so the lowered offsets stay non-negative).
    constexpr unsigned PRE = 4096;
    const int lr = lane >> 3;                            // row inside the piece
    const unsigned kcb = (unsigned)(((lane & 7) ^ lr) * 16);         // swizzled k-chunk (bytes): LDS slot lane & 7 holds chunk (lane & 7) ^ row
    unsigned xoff1[NXI], xoff2[NXI], vmask[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int m = m0 + (wave * NXI + i) * 8 + lr;
        const bool valid = m < p.M;
        const int mm = valid ? m : p.M - 1;              // rows past the end recompute the last pixel (never stored)
        const int b = mm / p.OHW;
        const int rem = mm - b * p.OHW;
        const int oh = rem / p.OW;
        const int ow = rem - oh * p.OW;
        const int ih0 = oh * p.stride, iw0 = ow * p.stride;           // (the descriptors start at (-pad, -pad))
        const unsigned pix = (unsigned)((b * p.H + ih0) * p.W + iw0);
        xoff1[i] = pix * (unsigned)(p.ldx1 * 2) + kcb + PRE - (unsigned)i * 1024u;
        xoff2[i] = TWO ? pix * (unsigned)(p.ldx2 * 2) + kcb + PRE - (unsigned)i * 1024u : 0u;
        unsigned mk = 0;
        if (HALO) {
            int t = 0;
            for (int kh = 0; kh < p.KH; ++kh)
                for (int kw = 0; kw < p.KW; ++kw, ++t) {
                    const int ih = ih0 - p.pad + kh, iw = iw0 - p.pad + kw;
                    mk |= ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? (1u << t) : 0u;
                }
        }
        vmask[i] = mk;
    }
    unsigned woff[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i)
        woff[i] = (unsigned)((n0 + (wave * NWI + i) * 8 + lr) * p.Kpad * 2) + kcb + PRE - (unsigned)i * 1024u;
    const long shift = (long)p.pad * p.W + p.pad;                    // pixels
    const char* xb1 = reinterpret_cast<const char*>(p.x1 - shift * p.ldx1) - PRE;
    const char* xb2 = TWO ? reinterpret_cast<const char*>(p.x2 - shift * p.ldx2) - PRE : xb1;
    const rsrc_t rw = dma_rsrc(reinterpret_cast<const char*>(p.w) - PRE), rx1 = dma_rsrc(xb1), rx2 = dma_rsrc(xb2);
    constexpr int STAGE_B = STAGE * 16;                              // bytes per ring stage
    char* const lds = reinterpret_cast<char*>(smem);
    const int xdst = wave * NXI * 1024;                              // this wave's first X piece inside a stage (bytes)
    const int wdst = BM * 128 + wave * NWI * 1024;

    // ---- wave-uniform K-tile state (SGPRs): byte offsets of the current tap / channel tile, all advanced incrementally ----
    // Blocks start their K loop at different tiles (and wrap around): all blocks of a launch stream the SAME weight tiles, and
    // started in lockstep they would ask the L2 for the same few lines at the same time.
    const int nk = p.Kslice / 64;
    int kidx = p.kstag ? (int)(((unsigned)(blockIdx.y * gridDim.x + blockIdx.x) * (unsigned)p.kstag) % (unsigned)nk) : 0;
    int tap = (kidx * 64) / p.Cin, kw = tap % p.KW;
    int cc = (kidx * 64 - tap * p.Cin) * 2;              // channel offset inside Cin (bytes)
    int pixA = ((tap / p.KW) * p.W + kw) * p.ldx1 * 2;   // byte offset of the current tap's pixel in source 1 / 2
    int pixB = TWO ? ((tap / p.KW) * p.W + kw) * p.ldx2 * 2 : 0;
    unsigned wsoff = (unsigned)kidx * 128u;
    const int cin2 = p.Cin * 2, c12 = p.C1 * 2;
    const int stepA1 = p.ldx1 * 2, stepA2 = (p.W - p.KW + 1) * p.ldx1 * 2;      // next tap in the row / first tap of the next row
    const int stepB1 = p.ldx2 * 2, stepB2 = (p.W - p.KW + 1) * p.ldx2 * 2;

// (the instruction offset must be a literal: one macro expansion per piece)
#define XPIECE(I, LD)                                                                                      \
    if constexpr ((I) < NXI) {                                                                             \
        const unsigned v_ = in1_ ? xoff1[(I) < NXI ? (I) : 0] : xoff2[(I) < NXI ? (I) : 0];               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, LDS_PTR(lds + (LD) + xdst), 16,                      \
            (!HALO || (vmask[(I) < NXI ? (I) : 0] & tapbit_)) ? v_ : DMA_OOB, soff_, (I) * 1024, 0);       \
    }
#define WPIECE(I, LD)                                                                                      \
    if constexpr ((I) < NWI)                                                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(lds + (LD) + wdst), 16, woff[(I) < NWI ? (I) : 0], wsoff, (I) * 1024, 0);
#define LOAD_TILE(LD)                                                                                      \
    {                                                                                                      \
        const unsigned tapbit_ = 1u << tap;                                                                \
        const bool in1_ = !TWO || cc < c12;                      /* uniform: scalar selects, no branch */  \
        const unsigned soff_ = (unsigned)(in1_ ? pixA + cc : pixB + cc - c12);                             \
        const rsrc_t rx_ = in1_ ? rx1 : rx2;                                                               \
        XPIECE(0, LD) XPIECE(1, LD) XPIECE(2, LD) XPIECE(3, LD)                                            \
        WPIECE(0, LD) WPIECE(1, LD) WPIECE(2, LD) WPIECE(3, LD)                                            \
        wsoff += 128;                                                                                      \
        cc += 128;                                                                                         \
        const bool wrap_ = cc >= cin2;                                                                     \
        cc = wrap_ ? 0 : cc;                                                                               \
        tap += wrap_ ? 1 : 0;                                                                              \
        kw += wrap_ ? 1 : 0;                                                                               \
        const bool wrap2_ = kw == p.KW;                                                                    \
        kw = wrap2_ ? 0 : kw;                                                                              \
        pixA += wrap_ ? (wrap2_ ? stepA2 : stepA1) : 0;                                                    \
        if (TWO) pixB += wrap_ ? (wrap2_ ? stepB2 : stepB1) : 0;                                           \
        const bool last_ = ++kidx == nk;                         /* staggered start: wrap around to tile 0 */ \
        kidx = last_ ? 0 : kidx; wsoff = last_ ? 0u : wsoff; cc = last_ ? 0 : cc; tap = last_ ? 0 : tap;   \
        kw = last_ ? 0 : kw; pixA = last_ ? 0 : pixA; pixB = last_ ? 0 : pixB;                             \
    }

    const int wm = wave / WN, wn = wave % WN;
    const int pm0 = wm * (BM / WM), cn0 = wn * (BN / WN);
    const int l15 = lane & 15, l4 = lane >> 4;
    int rdx[2], rdw[2];                                              // fragment reads of the two k-steps (16-B units); tile row t adds t*16*CPR
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        rdx[j] = (pm0 + l15) * CPR + ((j * 4 + l4) ^ (l15 & 7));
        rdw[j] = (BM + cn0 + l15) * CPR + ((j * 4 + l4) ^ (l15 & 7));
    }
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define COMPUTE_TILE(RD)                                                                                   \
    {                                                                                                      \
        const u32x4* src_ = reinterpret_cast<const u32x4*>(lds + (RD));                                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                    \
            bf16x8 bfr[TM], afr[TN];                                                                       \
            _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                               \
                u32x4 v = src_[rdx[j] + t * 16 * CPR];                                                     \
                if (RELU) { v.x = relu2(v.x); v.y = relu2(v.y); v.z = relu2(v.z); v.w = relu2(v.w); }      \
                bfr[t] = __builtin_bit_cast(bf16x8, v);                                                    \
            }                                                                                              \
            _Pragma("unroll") for (int t = 0; t < TN; ++t) afr[t] = __builtin_bit_cast(bf16x8, src_[rdw[j] + t * 16 * CPR]); \
            _Pragma("unroll") for (int a = 0; a < TN; ++a)                                                 \
                _Pragma("unroll") for (int b = 0; b < TM; ++b)                                             \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[a], bfr[b], acc[a][b], 0, 0, 0); \
        }                                                                                                  \
    }

    // ---- K loop (rolled; the ring stage offsets are wave-uniform run-time values).  Tile t lives in stage t % NS.
    // Iteration t: issue the DMA of tile t+NS-1 (its stage was read in iteration t-1, which every wave has left through the
    // barrier), multiply tile t, then wait until this wave's pieces of tile t+1 have landed (only the pieces of the NS-2 younger
    // tiles may still be in flight) and meet the other waves. ----
    int ld = 0;                                          // stage (byte offset) the next DMA goes to
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) { LOAD_TILE(ld) ld += STAGE_B; }
    if (nk >= NS - 1) { TILE_SYNC((NS - 2) * LPT) } else { TILE_SYNC(0) }
    int rd = 0;                                          // stage being multiplied
    int kt = 0;
    for (; kt < nk - (NS - 1); ++kt) {
        LOAD_TILE(ld)
        COMPUTE_TILE(rd)
        TILE_SYNC((NS - 2) * LPT)
        ld = ld == (NS - 1) * STAGE_B ? 0 : ld + STAGE_B;
        rd = rd == (NS - 1) * STAGE_B ? 0 : rd + STAGE_B;
    }
    for (; kt < nk; ++kt) {                              // the last NS-1 tiles: nothing left to load
        COMPUTE_TILE(rd)
        TILE_SYNC(0)
        rd = rd == (NS - 1) * STAGE_B ? 0 : rd + STAGE_B;
    }
#undef LOAD_TILE
#undef XPIECE
#undef WPIECE
#undef COMPUTE_TILE

    // ---- epilogue: fp32 tile transposed through LDS, 16-B accesses along the channel axis (as conv_igemm_kernel) ----
    float* ctile = reinterpret_cast<float*>(smem);       // the loop ended with a barrier: the operand ring is dead
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int px = pm0 + b * 16 + l15, ch = cn0 + a * 16 + l4 * 4;
            *reinterpret_cast<f32x4*>(ctile + px * LDC + ch) = acc[a][b];
        }
    __syncthreads();
    constexpr int CH8 = BN / 8;
    for (int q = tid; q < BM * CH8; q += NT) {
        const int px = q / CH8, c8 = q - px * CH8;
        const int m = m0 + px, ch0 = n0 + c8 * 8;
        if (m >= p.M || ch0 >= p.Cout) continue;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(ctile + px * LDC + c8 * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(ctile + px * LDC + c8 * 8 + 4);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        conv_finish(p, v, m, ch0);
    }
#endif
}

template <int BM, int BN, int WM, int WN, int NS, bool HALO, bool RELU, bool TWO>
static int launch_dma3(const ConvParams& p, hipStream_t s, int gy) {
    constexpr int lds = dma_lds_bytes<BM, BN, NS>();
    static bool attr_set = false;                        // one flag per instantiation
    if (!attr_set) {
        if (lds > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dma_kernel<BM, BN, WM, WN, NS, HALO, RELU, TWO>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            cutie_set_error("conv DMA tile: cannot raise the dynamic LDS limit to %d bytes", lds);
            return -2;
        }
        attr_set = true;
    }
    dim3 grid((p.M + BM - 1) / BM, (unsigned)gy);
    hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, NS, HALO, RELU, TWO>), grid, dim3(WM * WN * 64), lds, s, p);
    return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int NS>
static int launch_dma(ConvParams p, hipStream_t s) {
    const long x1_bytes = (long)p.B * p.H * p.W * p.ldx1 * 2, x2_bytes = p.C2 ? (long)p.B * p.H * p.W * p.ldx2 * 2 : 0;
    const long gy = (p.Cout + BN - 1) / BN, w_bytes = gy * BN * (long)p.Kpad * 2;
    if (p.Kpad % 64 || p.Cin % 64 || (p.C2 && p.C1 % 64) || p.KH * p.KW > 32 || p.splitk != 1 || p.Kpad < p.KH * p.KW * p.Cin ||
        x1_bytes >= DMA_RECORDS - 8192 || x2_bytes >= DMA_RECORDS - 8192 || w_bytes >= DMA_RECORDS - 8192) {
        cutie_set_error("conv DMA tile: needs Cin %% 64 == 0 (C1 too for two sources), KH*KW <= 32, no split-K, operands < 2 GiB "
                        "(Cin=%d C1=%d Kpad=%d k=%dx%d splitk=%d)", p.Cin, p.C1, p.Kpad, p.KH, p.KW, p.splitk);
        return -2;
    }
    p.Kslice = p.KH * p.KW * p.Cin;                      // a multiple of 64: the zero-padded tail of Kpad is not visited
    static int kstag_env = -1;                           // (experiment switch: CUTIE_DMA_KSTAG=0 disables the staggered K start)
    if (kstag_env < 0) { const char* e = getenv("CUTIE_DMA_KSTAG"); kstag_env = e ? atoi(e) : 5; }
    p.kstag = kstag_env;
    const bool relu = p.flags & CUTIE_F_RELU_IN, two = p.C2 != 0, halo = p.pad > 0;
    const int g = (int)gy;
#define DMA_GO(H_, R_, T_) return launch_dma3<BM, BN, WM, WN, NS, H_, R_, T_>(p, s, g)
    if (halo) {
        if (relu) { if (two) DMA_GO(true, true, true); DMA_GO(true, true, false); }
        if (two) DMA_GO(true, false, true);
        DMA_GO(true, false, false);
    }
    if (relu) { if (two) DMA_GO(false, true, true); DMA_GO(false, true, false); }
    if (two) DMA_GO(false, false, true);
    DMA_GO(false, false, false);
#undef DMA_GO
}

// tile table (mirrored by cutie_amd/ops.py:DMA_TILES): id -> BM, BN, waves, ring depth
int launch_conv_dma(const ConvParams& p, int tile, hipStream_t s) {
    switch (tile) {
        case 60: return launch_dma<128, 128, 2, 2, 3>(p, s);         // 4 waves, 64x64 per wave, 96 KB
        case 61: return launch_dma<128, 128, 2, 4, 3>(p, s);         // 8 waves, 64x32 per wave
        case 62: return launch_dma<128, 128, 2, 2, 2>(p, s);         // 4 waves, 2-deep ring: 2 blocks per CU
        case 63: return launch_dma<128, 64, 2, 2, 3>(p, s);          // 72 KB: 2 blocks per CU
        case 64: return launch_dma<64, 128, 2, 2, 3>(p, s);
        case 65: return launch_dma<64, 64, 2, 2, 4>(p, s);           // 64 KB
        case 66: return launch_dma<64, 64, 2, 2, 3>(p, s);           // 48 KB: 3 blocks per CU
        case 67: return launch_dma<32, 64, 1, 4, 4>(p, s);           // 48 KB
        case 68: return launch_dma<256, 128, 4, 2, 3>(p, s);         // 8 waves, 64x64 per wave, 144 KB
        case 69: return launch_dma<32, 128, 1, 4, 4>(p, s);
        default: cutie_set_error("conv: bad DMA tile id %d", tile); return -2;
    }
}
