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
//   * the DMA pieces of the next tile are issued BETWEEN the MFMAs of the current one (sched_group_barrier): a 64-lane 16-B
//     DMA costs the issuing wave 60-180 cycles of address-pipeline time, which in front of the MFMAs was the longest single
//     item of an iteration (measured round 2: 1600 cycles per K tile against 512 cycles of MFMA on the 128x128 tile);
//   * the fused input ReLU is applied to the pixel fragments after the LDS read (one v_pk_max_i16 per dword);
//   * prologue without per-tap loops or per-piece divisions, epilogue with the bias in registers and hardware bf16 packing:
//     the fixed cost per block was 13-15 us on the 128x128 tile (40 of the 82 us of the 77760-pixel 3x3 layers).
// Requirements (checked at launch): Cin % 64 == 0 (C1 % 64 == 0 as well for a two-source input), KH, KW <= 5, no split-K,
// every operand < 2 GiB (32-bit buffer offsets).  GEMM view, fragment layout and epilogue are those of conv_igemm.hip:
// D[cout][pixel], weights = MFMA A operand, a lane owns 4 consecutive output channels of one pixel.
#include "conv_common.h"

typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define DMA_WORD3 0x00020000          // raw buffer descriptor, dword 3: DATA_FORMAT = 32 bit, no swizzle, no tid
#define DMA_RECORDS 0x7fffffff        // bytes addressable through one descriptor
#define DMA_OOB 0x80000000u           // any voffset >= DMA_RECORDS reads as zero
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
// s_waitcnt vmcnt(N) lgkmcnt(0) (gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14] << 4, expcnt [6:4], lgkmcnt [11:8]).
// lgkmcnt(0): this wave's LDS reads have RETURNED (what __syncthreads() waits for before its s_barrier, minus the vmcnt(0) that
// would drain the DMA queue).  Needed before every barrier behind which another wave may overwrite what this wave read: the next
// DMA into the stage just multiplied and, with no latency cushion at all, the epilogue's ds_write into the ring.
#define WAIT_VMCNT_LDS(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | (7 << 4) | (0 << 8))
// Raw s_barrier (no vmcnt drain), fenced for the compiler: no LDS access or DMA may move across it.
#ifdef DMA_ABL_NO_BARRIER                                 // (ablation, see DMA_ISSUE below)
#define TILE_SYNC(N) { WAIT_VMCNT_LDS(N); asm volatile("" ::: "memory"); }
#else
#define TILE_SYNC(N) { WAIT_VMCNT_LDS(N); asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
#endif
// instruction classes of __builtin_amdgcn_sched_group_barrier
#define SG_MFMA 0x8
#define SG_VMEM 0x20
#define SG_DSR 0x100

__device__ __forceinline__ rsrc_t dma_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, DMA_RECORDS, DMA_WORD3);
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu2(unsigned w) {         // ReLU on two packed bf16: v_pk_max_i16 with 0
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), (s16x2){0, 0}));
}

template <int BM, int BN, int NS, int BK>
constexpr int dma_lds_bytes() {
    constexpr int pipe = NS * (BM + BN) * BK * 2, epi = BM * (BN + 4) * 4;
    return pipe > epi ? pipe : epi;
}

// HALO: the conv has more than one tap and / or padding (tap state + validity masks); RELU: fused input ReLU; TWO: two-source
// (virtual concat) input.
// BK: K tile (64 or 128 channels of one tap): LDS rows of BK * 2 bytes, a DMA piece = 1 KiB = RPP rows.
template <int BM, int BN, int WM, int WN, int NS, int BK, bool HALO, bool RELU, bool TWO>
__global__ __launch_bounds__(WM * WN * 64) void conv_dma_kernel(ConvParams p) {
#if __HIP_DEVICE_COMPILE__     // (the host pass only needs the launch stub; the LDS-DMA builtin and the LDS address space exist on the device side)
    if (p.flags & CUTIE_F_PRIO) __builtin_amdgcn_s_setprio(1);      // a launch of the frame's critical path: see include/cutie_hip.h
    constexpr int NW = WM * WN, NT = NW * 64, CPR = BK / 8;     // 16-B chunks per LDS row
    constexpr int RPP = 64 / CPR;                       // rows per DMA piece (8 at BK = 64, 4 at BK = 128)
    constexpr int ROWB = BK * 2, KSTEPS = BK / 32;      // bytes per LDS row, MFMA k-steps per tile
    constexpr int NXI = BM / RPP / NW;                  // X pieces per wave per K tile
    constexpr int NWI = BN / RPP / NW;                  // W pieces per wave per K tile
    constexpr int LPT = NXI + NWI;                      // DMA instructions per wave per K tile
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int STAGE_B = (BM + BN) * ROWB;           // bytes per ring stage
    constexpr int LDC = BN + 4;
    static_assert((BK == 64 || BK == 128) && NXI >= 1 && NWI >= 1 && NXI * RPP * NW == BM && NWI * RPP * NW == BN && TM >= 1 && TN >= 1 && NS >= 2 && NS <= 8 &&
                  NXI <= 4 && NWI <= 4 && (NS - 2) * LPT < 64 && NT % (BN / 8) == 0, "bad tile");
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // provably wave-uniform (LDS base of the DMA goes to M0)
    TL_DECL(lds + dma_lds_bytes<BM, BN, NS, BK>())
    TL(0)
    int m0, n0;                                          // XCD-aware tile mapping, as in conv_igemm_kernel
    int tl_logical = 0, tl_nb = 0;                       // (timeline builds only)
    {
        const int nb = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = id & 7, kq = id >> 3, q = nb >> 3, r = nb & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kq;
        const int mt = logical / (int)gridDim.y;
        m0 = mt * BM;
        n0 = (logical - mt * (int)gridDim.y) * BN;
        tl_logical = logical; tl_nb = nb;
    }
    if (p.zero) {                                          // side job: clear an accumulator (the next conv's GAP sums; the transformer's fixed-point sums: 0.9 MB, so every block takes a slice)
        for (int z = (blockIdx.y * gridDim.x + blockIdx.x) * NT + tid; z < p.nzero; z += gridDim.x * gridDim.y * NT) p.zero[z] = 0ull;
    }
    // ---- epilogue operands fetched up front (their latency hides behind the whole K loop): this thread's 8-channel column ----
    constexpr int CH8 = BN / 8, PSTEP = NT / CH8;        // a thread keeps its channel chunk and walks down the pixels
    const int ec8 = tid % CH8, epx0 = tid / CH8, ech0 = n0 + ec8 * 8;
    float bias[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) bias[r] = (p.bias && ech0 + r < p.Cout) ? p.bias[ech0 + r] : 0.f;

    // ---- loop-invariant per-lane state: byte offset of this lane's chunk in every piece, tap validity mask ----
    // The pieces of one operand share ONE M0 value per stage: piece i is issued with the instruction offset i * 1024, which the
    // hardware adds to the LDS address AND to the global address; the per-lane offset of piece i is lowered by the same amount
    // (the descriptors start PRE = 4096 bytes early, so the lowered offsets stay non-negative).
    constexpr unsigned PRE = 4096;
    const int lr = lane / CPR;                           // row inside the piece
    // swizzled k-chunk (bytes): LDS slot (lane % CPR) of row r holds chunk (lane % CPR) ^ (r % CPR') -- r % 8 = lr at BK = 64 (pieces
    // start at multiples of 8 rows); at BK = 128 the XOR uses r % 16, which depends on the piece: added per piece below
    const unsigned kcb = BK == 64 ? (unsigned)(((lane & 7) ^ lr) * 16) : 0u;
    unsigned xoff1[NXI], xoff2[NXI], vmask[NXI];
    {
        // (b, oh, ow) of the first piece's row by division, of the following pieces (+8 rows each) by carry
        int m = m0 + wave * NXI * RPP + lr;
        const int mm = m < p.M ? m : p.M - 1;            // rows past the end recompute the last pixel (never stored)
        int b = mm / p.OHW;
        const int rem = mm - b * p.OHW;
        int oh = rem / p.OW;
        int ow = rem - oh * p.OW;
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int ih0 = oh * p.stride, iw0 = ow * p.stride;       // (the descriptors start at (-pad, -pad))
            const unsigned pix = (unsigned)((b * p.H + ih0) * p.W + iw0);
            const unsigned kx = BK == 64 ? kcb : (unsigned)(((lane & 15) ^ (((wave * NXI + i) * RPP + lr) & 15)) * 16);
            xoff1[i] = pix * (unsigned)(p.ldx1 * 2) + kx + PRE - (unsigned)i * 1024u;
            xoff2[i] = TWO ? pix * (unsigned)(p.ldx2 * 2) + kx + PRE - (unsigned)i * 1024u : 0u;
            unsigned mk = 0;
            if (HALO) {
                // valid taps = (valid rows) x (valid columns): KH + KW comparisons instead of KH * KW
                unsigned cols = 0;
                for (int k = 0; k < p.KW; ++k) cols |= ((unsigned)(iw0 - p.pad + k) < (unsigned)p.W) ? (1u << k) : 0u;
                for (int k = 0; k < p.KH; ++k) mk |= ((unsigned)(ih0 - p.pad + k) < (unsigned)p.H) ? (cols << (k * p.KW)) : 0u;
            }
            vmask[i] = mk;
            if (i + 1 < NXI) {                           // next piece: RPP rows further
                m += RPP;
                if (m < p.M) {
                    ow += RPP;
                    while (ow >= p.OW) { ow -= p.OW; ++oh; }
                    while (oh >= p.OH) { oh -= p.OH; ++b; }
                }
            }
        }
    }
    unsigned woff[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i)
        woff[i] = (unsigned)((n0 + (wave * NWI + i) * RPP + lr) * p.Kpad * 2) + PRE - (unsigned)i * 1024u +
                  (BK == 64 ? kcb : (unsigned)(((lane & 15) ^ (((wave * NWI + i) * RPP + lr) & 15)) * 16));
    const long shift = (long)p.pad * p.W + p.pad;                    // pixels
    const char* xb1 = reinterpret_cast<const char*>(p.x1 - shift * p.ldx1) - PRE;
    const char* xb2 = TWO ? reinterpret_cast<const char*>(p.x2 - shift * p.ldx2) - PRE : xb1;
    const rsrc_t rw = dma_rsrc(reinterpret_cast<const char*>(p.w) - PRE), rx1 = dma_rsrc(xb1), rx2 = dma_rsrc(xb2);
    const int xdst = wave * NXI * 1024;                              // this wave's first X piece inside a stage (bytes)
    const int wdst = BM * ROWB + wave * NWI * 1024;

    // ---- wave-uniform K-tile state (SGPRs): byte offsets of the current tap / channel tile, all advanced incrementally ----
#ifdef DMA_ABL_NO_LOOP
    const int nk = 0;
#else
    const int nk = p.Kslice / BK;
#endif
    int tap = 0, kw = 0;
    int cc = 0;                                          // channel offset inside Cin (bytes)
    int pixA = 0, pixB = 0;                              // byte offset of the current tap's pixel in source 1 / 2
    unsigned wsoff = 0;
    const int cin2 = p.Cin * 2, c12 = p.C1 * 2;
    const int stepA1 = p.ldx1 * 2, stepA2 = (p.W - p.KW + 1) * p.ldx1 * 2;      // next tap in the row / first tap of the next row
    const int stepB1 = p.ldx2 * 2, stepB2 = (p.W - p.KW + 1) * p.ldx2 * 2;

    // per-tile scalars of the tile about to be loaded; TILE_BEGIN computes them, XPIECE / WPIECE issue one DMA each (the
    // instruction offset must be a literal: one macro expansion per piece), TILE_END advances the state
#define TILE_BEGIN()                                                                                       \
    const unsigned tapbit_ = 1u << tap;                                                                    \
    const bool in1_ = !TWO || cc < c12;                          /* uniform: scalar selects, no branch */  \
    const unsigned soff_ = (unsigned)(in1_ ? pixA + cc : pixB + cc - c12);                                 \
    const rsrc_t rx_ = in1_ ? rx1 : rx2;
// Ablation switches for tools/ablate_conv.sh (extra diagnostic libraries; never defined in the product build): which of the three
// streams of the K loop -- LDS-DMA fill, fragment reads, MFMA -- bounds a layer.  Results are garbage with any of them.
#ifdef DMA_ABL_NO_DMA
#define DMA_ISSUE(...) ((void)0)
#else
#define DMA_ISSUE(...) __builtin_amdgcn_raw_ptr_buffer_load_lds(__VA_ARGS__)
#endif
#define XPIECE(I, LD)                                                                                      \
    if constexpr ((I) < NXI) {                                                                             \
        const unsigned v_ = in1_ ? xoff1[(I) < NXI ? (I) : 0] : xoff2[(I) < NXI ? (I) : 0];               \
        DMA_ISSUE(rx_, LDS_PTR(lds + (LD) + xdst), 16,                                                     \
            (!HALO || (vmask[(I) < NXI ? (I) : 0] & tapbit_)) ? v_ : DMA_OOB, soff_, (I) * 1024, 0);       \
    }
#define WPIECE(I, LD)                                                                                      \
    if constexpr ((I) < NWI)                                                                               \
        DMA_ISSUE(rw, LDS_PTR(lds + (LD) + wdst), 16, woff[(I) < NWI ? (I) : 0], wsoff, (I) * 1024, 0);
#define TILE_END()                                                                                         \
    {                                                                                                      \
        wsoff += ROWB;                                                                                     \
        cc += ROWB;                                              /* single tap, single source: soff = cc */ \
        if (HALO || TWO) {                                                                                 \
            const bool wrap_ = cc >= cin2;                                                                 \
            cc = wrap_ ? 0 : cc;                                                                           \
            if (HALO) {                                                                                    \
                tap += wrap_ ? 1 : 0;                                                                      \
                kw += wrap_ ? 1 : 0;                                                                       \
                const bool wrap2_ = kw == p.KW;                                                            \
                kw = wrap2_ ? 0 : kw;                                                                      \
                pixA += wrap_ ? (wrap2_ ? stepA2 : stepA1) : 0;                                            \
                if (TWO) pixB += wrap_ ? (wrap2_ ? stepB2 : stepB1) : 0;                                   \
            }                                                                                              \
        }                                                                                                  \
    }
#define LOAD_TILE(LD)                                                                                      \
    {                                                                                                      \
        TILE_BEGIN()                                                                                       \
        XPIECE(0, LD) XPIECE(1, LD) XPIECE(2, LD) XPIECE(3, LD)                                            \
        WPIECE(0, LD) WPIECE(1, LD) WPIECE(2, LD) WPIECE(3, LD)                                            \
        TILE_END()                                                                                         \
    }

    const int wm = wave / WN, wn = wave % WN;
    const int pm0 = wm * (BM / WM), cn0 = wn * (BN / WN);
    const int l15 = lane & 15, l4 = lane >> 4;
    int rdx[KSTEPS], rdw[KSTEPS];                                    // fragment reads of the k-steps (bytes); tile row t adds t*16*ROWB
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) {
        rdx[j] = ((pm0 + l15) * CPR + ((j * 4 + l4) ^ (l15 & (CPR - 1)))) * 16;
        rdw[j] = ((BM + cn0 + l15) * CPR + ((j * 4 + l4) ^ (l15 & (CPR - 1)))) * 16;
    }
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#ifdef DMA_ABL_NO_DSREAD
#define READ_FRAGS(J, RD, BFR, AFR)                                                                        \
    _Pragma("unroll") for (int t = 0; t < TM; ++t) { u32x4 v = {(unsigned)(RD) + t, (unsigned)lane, 3u, (unsigned)(J)}; BFR[t] = __builtin_bit_cast(bf16x8, v); } \
    _Pragma("unroll") for (int t = 0; t < TN; ++t) { u32x4 v = {(unsigned)(RD) + t, (unsigned)lane, 5u, (unsigned)(J)}; AFR[t] = __builtin_bit_cast(bf16x8, v); }
#else
#define READ_FRAGS(J, RD, BFR, AFR)                                                                        \
    _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                                       \
        u32x4 v = *reinterpret_cast<const u32x4*>(lds + (RD) + rdx[J] + t * 16 * ROWB);                     \
        if (RELU) { v.x = relu2(v.x); v.y = relu2(v.y); v.z = relu2(v.z); v.w = relu2(v.w); }              \
        BFR[t] = __builtin_bit_cast(bf16x8, v);                                                            \
    }                                                                                                      \
    _Pragma("unroll") for (int t = 0; t < TN; ++t) AFR[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds + (RD) + rdw[J] + t * 16 * ROWB));
#endif
#ifdef DMA_ABL_NO_MFMA
#define MFMA_STEP(BFR, AFR)                                                                                \
    _Pragma("unroll") for (int a = 0; a < TN; ++a)                                                         \
        _Pragma("unroll") for (int b = 0; b < TM; ++b)                                                     \
            acc[a][b][0] += __uint_as_float(__builtin_bit_cast(u32x4, AFR[a]).x ^ __builtin_bit_cast(u32x4, BFR[b]).y);
#else
#define MFMA_STEP(BFR, AFR)                                                                                \
    _Pragma("unroll") for (int a = 0; a < TN; ++a)                                                         \
        _Pragma("unroll") for (int b = 0; b < TM; ++b)                                                     \
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AFR[a], BFR[b], acc[a][b], 0, 0, 0);
#endif
    // Scheduling of one iteration with a load: fragments of k-step 0, then LPT groups of { MF0 MFMAs of k-step 0, one DMA piece,
    // RDP fragment reads of k-step 1 }, then the MFMAs of k-step 1.  (Program order of the memory operations is already this;
    // the MFMAs, which the compiler may place anywhere, are pinned between them.)
    constexpr int NMF = TM * TN, NRD = TM + TN;
    constexpr int MF0 = NMF / LPT > 0 ? NMF / LPT : 1;  // MFMAs between two DMA pieces
    constexpr int RDP = (NRD + LPT - 1) / LPT;          // fragment reads of k-step 1 per piece
#define SCHED_GROUP(G)                                                                                     \
    if constexpr ((G) < LPT) {                                                                             \
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, MF0, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(SG_VMEM, 1, 0);                                               \
        __builtin_amdgcn_sched_group_barrier(SG_DSR, RDP, 0);                                              \
    }
#define COMPUTE_TILE(RD, LD, DO_LOAD)                                                                      \
    {                                                                                                      \
        bf16x8 b0[TM], a0[TN], b1[TM], a1[TN];                                                             \
        READ_FRAGS(0, RD, b0, a0)                                                                          \
        if (DO_LOAD) LOAD_TILE(LD)                                                                         \
        READ_FRAGS(1, RD, b1, a1)                                                                          \
        MFMA_STEP(b0, a0)                                                                                  \
        MFMA_STEP(b1, a1)                                                                                  \
        if (DO_LOAD) {                                                                                     \
            __builtin_amdgcn_sched_group_barrier(SG_DSR, NRD, 0);                                          \
            SCHED_GROUP(0) SCHED_GROUP(1) SCHED_GROUP(2) SCHED_GROUP(3) SCHED_GROUP(4) SCHED_GROUP(5) SCHED_GROUP(6) SCHED_GROUP(7) \
        }                                                                                                  \
        if constexpr (KSTEPS == 4) {                             /* BK = 128: two more k-steps */           \
            bf16x8 b2[TM], a2[TN], b3[TM], a3[TN];                                                         \
            READ_FRAGS(2, RD, b2, a2)                                                                      \
            READ_FRAGS(3, RD, b3, a3)                                                                      \
            MFMA_STEP(b2, a2)                                                                              \
            MFMA_STEP(b3, a3)                                                                              \
        }                                                                                                  \
    }

    // ---- K loop (rolled; the ring stage offsets are wave-uniform run-time values).  Tile t lives in stage t % NS.
    // Iteration t: multiply tile t while issuing the DMA of tile t+NS-1 (its stage was read in iteration t-1, which every wave
    // has left through the barrier), then wait until this wave's pieces of tile t+1 have landed (only the pieces of the NS-2
    // younger tiles may still be in flight) and meet the other waves. ----
    int ld = 0;                                          // stage (byte offset) the next DMA goes to
    TL(1)
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) { LOAD_TILE(ld) ld += STAGE_B; }
    if (nk >= NS - 1) { TILE_SYNC((NS - 2) * LPT) } else { TILE_SYNC(0) }
    int rd = 0;                                          // stage being multiplied
    int kt = 0;
    TL(2)
    for (; kt < nk - (NS - 1); ++kt) {
        TL(3)
        COMPUTE_TILE(rd, ld, true)
        TL(4)
        WAIT_VMCNT_LDS((NS - 2) * LPT);
        TL(5)
        TILE_SYNC((NS - 2) * LPT)
        ld = ld == (NS - 1) * STAGE_B ? 0 : ld + STAGE_B;
        rd = rd == (NS - 1) * STAGE_B ? 0 : rd + STAGE_B;
    }
    for (; kt < nk; ++kt) {                              // the last NS-1 tiles: nothing left to load
        COMPUTE_TILE(rd, ld, false)
        TILE_SYNC(0)
        rd = rd == (NS - 1) * STAGE_B ? 0 : rd + STAGE_B;
    }
#undef COMPUTE_TILE
#undef SCHED_GROUP
#undef MFMA_STEP
#undef READ_FRAGS
#undef LOAD_TILE
#undef TILE_END
#undef WPIECE
#undef XPIECE
#undef TILE_BEGIN

#ifdef DMA_ABL_NO_EPILOGUE
    if (acc[0][0][0] != 12345.678f) return;
#endif
    // ---- epilogue: fp32 tile transposed through LDS, 16-B accesses along the channel axis; bias already in registers ----
    TLE(6)
    float* ctile = reinterpret_cast<float*>(smem);       // the loop ended with a barrier: the operand ring is dead
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int px = pm0 + b * 16 + l15, ch = cn0 + a * 16 + l4 * 4;
            *reinterpret_cast<f32x4*>(ctile + px * LDC + ch) = acc[a][b];
        }
    __syncthreads();
    TLE(7)
    const bool active = ech0 < p.Cout;
    if (!active && !p.gap) return;                       // (GAP mode: every thread reaches the barrier below)
    const int act = (p.flags >> CUTIE_ACT_SHIFT) & 7;
    const bool out_f32 = p.flags & CUTIE_F_OUT_F32;
    const bool fast = ech0 + 7 < p.Cout && (out_f32 ? (p.ldy & 3) == 0 : (p.ldy & 7) == 0) && (!p.res || (p.ldr & 7) == 0);
    if (active)
    for (int px = epx0; px < BM; px += PSTEP) {
        const int m = m0 + px;
        if (m >= p.M) break;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(ctile + px * LDC + ec8 * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(ctile + px * LDC + ec8 * 8 + 4);
        float v[8] = {lo[0] + bias[0], lo[1] + bias[1], lo[2] + bias[2], lo[3] + bias[3],
                      hi[0] + bias[4], hi[1] + bias[5], hi[2] + bias[6], hi[3] + bias[7]};
        if (!fast) {                                     // ragged channel tail / unaligned strides: the general tail (bias already added)
            ConvParams q = p;
            q.bias = nullptr;
            conv_finish(q, v, m, ech0);
            continue;
        }
        if (p.res) {
            const int mres = conv_res_row(p, m);
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
            if (p.gap) {                                 // the STORED (bf16-rounded) values go back into the tile for the column sums
                const f32x4 a = {__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u), __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u)};
                const f32x4 b = {__uint_as_float(o.z << 16), __uint_as_float(o.z & 0xffff0000u), __uint_as_float(o.w << 16), __uint_as_float(o.w & 0xffff0000u)};
                *reinterpret_cast<f32x4*>(ctile + px * LDC + ec8 * 8) = a;
                *reinterpret_cast<f32x4*>(ctile + px * LDC + ec8 * 8 + 4) = b;
            }
        }
    }
    TLE(8)
    TL_DUMP(tl_logical, tl_nb, NW)
    if (p.gap) {
        // GAP partials of this tile: thread = (channel, row part); rows of one object summed in row order in fp32, each partial
        // converted to fixed point and added with an integer atomic (a tile may straddle objects: flush at every boundary)
        __syncthreads();
        constexpr int PARTS = NT / BN, RPART = BM / PARTS;
        static_assert(NT % BN == 0 && BM % PARTS == 0, "GAP row parts");
        const int c = tid % BN, part = tid / BN;
        const int rows = min(BM, p.M - m0);
        if (n0 + c < p.Cout) {
            const int r0 = part * RPART, r1 = min(r0 + RPART, rows);
            int m = m0 + r0, obj = m / p.OHW, next_b = (obj + 1) * p.OHW;
            long long sum = 0;                               // (fixed point per VALUE, integer sums: conv_common.h conv_gapfx)
            for (int r = r0; r < r1; ++r, ++m) {
                if (m == next_b) {
                    atomicAdd(reinterpret_cast<unsigned long long*>(p.gap + (long)obj * p.Cout + n0 + c), (unsigned long long)(sum << GAP_ELEM_SHIFT));
                    sum = 0; ++obj; next_b += p.OHW;
                }
                sum += conv_gapfx(ctile[r * LDC + c]);
            }
            if (r1 > r0) atomicAdd(reinterpret_cast<unsigned long long*>(p.gap + (long)obj * p.Cout + n0 + c), (unsigned long long)(sum << GAP_ELEM_SHIFT));
        }
    }
#endif
}

template <int BM, int BN, int WM, int WN, int NS, int BK, bool HALO, bool RELU, bool TWO>
static int launch_dma3(const ConvParams& p, hipStream_t s, int gy) {
    constexpr int lds = dma_lds_bytes<BM, BN, NS, BK>() + TL_BYTES;
    static bool attr_set = false;                        // one flag per instantiation
    if (!attr_set) {
        if (lds > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dma_kernel<BM, BN, WM, WN, NS, BK, HALO, RELU, TWO>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            cutie_set_error("conv DMA tile: cannot raise the dynamic LDS limit to %d bytes", lds);
            return -2;
        }
        attr_set = true;
    }
    dim3 grid((p.M + BM - 1) / BM, (unsigned)gy);
    hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, NS, BK, HALO, RELU, TWO>), grid, dim3(WM * WN * 64), lds, s, p);
    return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int NS, int BK = 64>
static int launch_dma(ConvParams p, hipStream_t s) {
    const long x1_bytes = (long)p.B * p.H * p.W * p.ldx1 * 2, x2_bytes = p.C2 ? (long)p.B * p.H * p.W * p.ldx2 * 2 : 0;
    const long gy = (p.Cout + BN - 1) / BN, w_bytes = gy * BN * (long)p.Kpad * 2;
    if (p.Kpad % BK || p.Cin % BK || (p.C2 && p.C1 % BK) || p.KH > 5 || p.KW > 5 || p.splitk != 1 || p.Kpad < p.KH * p.KW * p.Cin ||
        x1_bytes >= DMA_RECORDS - 8192 || x2_bytes >= DMA_RECORDS - 8192 || w_bytes >= DMA_RECORDS - 8192) {
        cutie_set_error("conv DMA tile: needs Cin %% %d == 0 (C1 too for two sources), KH, KW <= 5, no split-K, operands < 2 GiB "
                        "(Cin=%d C1=%d Kpad=%d k=%dx%d splitk=%d)", BK, p.Cin, p.C1, p.Kpad, p.KH, p.KW, p.splitk);
        return -2;
    }
    p.Kslice = p.KH * p.KW * p.Cin;                      // a multiple of 64: the zero-padded tail of Kpad is not visited
    const bool relu = p.flags & CUTIE_F_RELU_IN, two = p.C2 != 0, halo = p.pad > 0 || p.KH * p.KW > 1;
    const int g = (int)gy;
#define DMA_GO(H_, R_, T_) return launch_dma3<BM, BN, WM, WN, NS, BK, H_, R_, T_>(p, s, g)
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
        case 70: return launch_dma<128, 128, 2, 4, 2>(p, s);         // 8 waves, 2-deep ring: 2 blocks = 16 waves per CU
        case 71: return launch_dma<128, 64, 2, 2, 2>(p, s);          // 48 KB: 3 blocks per CU
        case 72: return launch_dma<64, 64, 2, 2, 2>(p, s);           // 32 KB: 5 blocks per CU
        // deep rings for the small-M layers (<= ~1.5 blocks per CU anyway): an iteration there is bound by the DMA latency divided
        // by the tiles in flight (measured: 840 cycles per K tile for 128 cycles of MFMA with 2 tiles in flight)
        case 73: return launch_dma<64, 64, 2, 2, 6>(p, s);           // 96 KB
        case 74: return launch_dma<64, 64, 2, 2, 8>(p, s);           // 128 KB
        case 75: return launch_dma<32, 64, 1, 4, 8>(p, s);           // 96 KB
        case 76: return launch_dma<128, 64, 2, 2, 5>(p, s);          // 120 KB
        case 77: return launch_dma<64, 128, 2, 2, 5>(p, s);          // 120 KB
        case 78: return launch_dma<32, 128, 1, 4, 6>(p, s);          // 120 KB
        // BK = 128: half the K iterations (barrier + wait + scalar bookkeeping per iteration amortised over twice the MFMAs)
        case 80: return launch_dma<64, 64, 2, 2, 3, 128>(p, s);      // 96 KB
        case 81: return launch_dma<64, 64, 2, 2, 2, 128>(p, s);      // 64 KB: 2 blocks per CU
        case 82: return launch_dma<32, 64, 1, 4, 3, 128>(p, s);      // 72 KB: 2 blocks per CU
        case 83: return launch_dma<32, 64, 1, 4, 4, 128>(p, s);      // 96 KB
        case 84: return launch_dma<64, 128, 2, 4, 2, 128>(p, s);     // 8 waves, 96 KB
        case 85: return launch_dma<128, 64, 4, 2, 2, 128>(p, s);     // 8 waves, 96 KB
        // 96 rows: M = 4860 (3 objects at stride 16) gives 51 x 4 = 204 blocks -- one round on 256 CUs where 64-row tiles need two
        case 86: return launch_dma<96, 64, 2, 2, 3>(p, s);           // 60 KB
        case 87: return launch_dma<96, 64, 2, 2, 2>(p, s);           // 40 KB
        // 8 waves on a 64 x 64 tile: one X and one W piece per wave and K tile -- the DMA issue cost (100-200 cycles of the issuing
        // wave per 1-KiB piece) is the longest item of an iteration on the small tiles
        case 88: return launch_dma<64, 64, 2, 4, 3>(p, s);           // 48 KB
        case 89: return launch_dma<64, 64, 2, 4, 4>(p, s);           // 64 KB
        default: cutie_set_error("conv: bad DMA tile id %d", tile); return -2;
    }
}
