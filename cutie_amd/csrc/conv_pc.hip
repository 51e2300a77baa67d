// Producer / consumer implicit-GEMM convolution with LDS-DMA operand staging -- tiles 100..
//
// Why (round 3, profiles/r03_conv_timeline.md): in conv_dma_kernel every wave multiplies AND issues its share of the tile's LDS-DMA
// pieces.  A 1-KiB piece occupies the CU's vector-memory path for >= 16 cycles (64 B/clk) and the issuing wave waits in that queue
// for 60-180 cycles per piece -- in program order, between its own MFMAs.  With one block per CU (the small-M layers of a single
// clip) a K step of the 64x64 tile therefore takes ~570 cycles for 128 cycles of MFMA, whatever the ring depth or the byte count
// (conv_strip: 35 % fewer bytes, same time).  Here the two jobs belong to different waves of the workgroup:
//   * NPW producer waves own the address state and issue every LDS-DMA piece; their queueing time is nobody's MFMA time;
//   * WM x WN consumer waves read fragments and issue MFMAs, nothing else;
//   * one raw s_barrier per K step joins them: producers arrive after a counted vmcnt (tile t+1 has landed), consumers after
//     lgkmcnt(0) (their reads of tile t have returned) -- the ring protocol of conv_dma.hip with the roles split.
// Two operand modes:
//   * stream (TW = 0): im2col rows streamed per K tile as in conv_dma.hip (any kernel size / stride, halo by out-of-range offsets);
//   * halo   (TW > 0): 3x3 / stride 1 / pad 1.  The block owns a TH x TW patch of output pixels of ONE image; the (TH+2) x (TW+2)
//     input patch of a 64-channel slice is fetched ONCE (double-buffered across slices; pixels outside the image are out-of-range
//     offsets = zeros, so the loop has no border logic at all) and the nine taps are fragment reads at shifted patch rows; only the
//     weight tiles stream per tap: 9x fewer activation bytes through the L2 -> LDS path.
// Epilogue: straight from the accumulators (no LDS transpose, no barrier).  The producers fetch the weight rows of each 32-channel
// group in the order 0-3, 8-11, 16-19, 24-27 | 4-7, 12-15, 20-23, 28-31, so the two 16-row MFMA fragments of a group leave every
// lane with 8 CONSECUTIVE output channels of one pixel: bias / residual / activation on registers, one 16-B store (bf16).  The
// residual of the small tiles is requested before the K loop.  GAP side job: per-fragment column sums by lane shuffles, fixed-point
// integer atomics (order-independent, like conv_dma.hip).
// Requirements (checked at launch): Cin % 64 == 0 (C1 too for two sources), no split-K, operands < 2 GiB; halo mode: k = 3,
// stride 1, pad 1.
#include "conv_common.h"

typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define PC_WORD3 0x00020000
#define PC_RECORDS 0x7fffffff
#define PC_OOB 0x80000000u
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
// s_waitcnt with only one counter constrained (gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14] << 4, expcnt [6:4], lgkmcnt [11:8])
#define PC_WAIT_VM(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | (7 << 4) | (15 << 8))
#define PC_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8))
// Ablation switches (diagnostic libraries of tools/build_diag.sh only; results are garbage with any of them): PC_ABL_NO_DMA, _NO_MFMA,
// _NO_READ (no fragment reads), _SAME_TILE (every K step fetches the operands of step 0: L2-hot), _NO_LOOPBAR (no barrier inside the K
// loop), _NO_WAIT (no counted vmcnt wait inside the K loop), _NO_LOOP, _NO_EPILOGUE
#define PC_BARRIER() { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
#ifdef PC_ABL_NO_LOOPBAR
#define PC_LOOP_BARRIER() { asm volatile("" ::: "memory"); }
#else
#define PC_LOOP_BARRIER() PC_BARRIER()
#endif
#ifdef PC_ABL_NO_WAIT
#define PC_LOOP_WAIT_VM(N) ((void)0)
#else
#define PC_LOOP_WAIT_VM(N) PC_WAIT_VM(N)
#endif
#ifdef PC_ABL_SAME_TILE
#define PC_SAME_TILE_GUARD(...)
#else
#define PC_SAME_TILE_GUARD(...) __VA_ARGS__
#endif
#ifdef PC_ABL_NO_DMA
#define PC_DMA(...) ((void)0)
#else
#define PC_DMA(...) __builtin_amdgcn_raw_ptr_buffer_load_lds(__VA_ARGS__)
#endif

__device__ __forceinline__ rsrc_t pc_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, PC_RECORDS, PC_WORD3);
}

typedef short pc_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pc_relu2(unsigned w) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(pc_s16x2, w), (pc_s16x2){0, 0}));
}

// row of the packed weight matrix that LDS weight row R of a tile holds (see the header: 8 consecutive channels per lane)
__device__ __forceinline__ int pc_wrow(int R, bool pair) {
    if (!pair) return R;
    const int ii = R & 15, alo = (R >> 4) & 1;
    return (R & ~31) | ((ii >> 2) << 3) | (alo << 2) | (ii & 3);
}

#define PC_HT(TH, TW) ((TH) * 256 + (TW))              // halo tile TH x TW output pixels (<= BM) as one template argument; 0 = stream mode
template <int BM, int BN, int NS, int HT>
constexpr int pc_lds_bytes(int NPW) {
    if (HT == 0) return NS * (BM + BN) * 128;
    const int TH = HT >> 8, TW = HT & 255, npiece = ((TH + 2) * (TW + 2) + 7) / 8, nxp = (npiece + NPW - 1) / NPW;
    return 2 * nxp * NPW * 1024 + NS * BN * 128;
}

// one slice of the epilogue: NCH (4 or 8) consecutive channels of one pixel
template <int NCH>
__device__ __forceinline__ void pc_finish(const ConvParams& p, float (&v)[NCH], int m, int ch0, bool vec_ok, bool have_res,
                                          const unsigned (&rpre)[NCH / 2], float (&stored)[NCH]) {
#ifdef PC_ABL_FIXED_EPI                                    // (diagnostic: the epilogue of ONE configuration -- relu, bf16, no residual -- only)
    {
#pragma unroll
        for (int r = 0; r < NCH; ++r) v[r] = fmaxf(v[r], 0.f);
        bf16_t* yp = reinterpret_cast<bf16_t*>(p.y) + (long)m * p.ldy + ch0;
        unsigned o[NCH / 2];
#pragma unroll
        for (int r = 0; r < NCH / 2; ++r) o[r] = pack_bf2(v[2 * r], v[2 * r + 1]);
        if constexpr (NCH == 8) *reinterpret_cast<uint4*>(yp) = make_uint4(o[0], o[1], o[NCH / 2 - 2], o[NCH / 2 - 1]);
        else *reinterpret_cast<uint2*>(yp) = make_uint2(o[0], o[1]);
        return;
    }
#endif
    const int act = (p.flags >> CUTIE_ACT_SHIFT) & 7;
    const bool out_f32 = p.flags & CUTIE_F_OUT_F32;
    const bool full = ch0 + NCH <= p.Cout;
    if (p.res) {
        if (have_res) {                                  // requested before the K loop
#pragma unroll
            for (int r = 0; r < NCH / 2; ++r) { v[2 * r] += __uint_as_float(rpre[r] << 16); v[2 * r + 1] += __uint_as_float(rpre[r] & 0xffff0000u); }
        } else {
            const int mres = conv_res_row(p, m);
            const bf16_t* rp = p.res + (long)mres * p.ldr + ch0;
            if (full && vec_ok) {
                unsigned rr[NCH / 2];
                if constexpr (NCH == 8) { const uint4 t = *reinterpret_cast<const uint4*>(rp); rr[0] = t.x; rr[1] = t.y; rr[NCH / 2 - 2] = t.z; rr[NCH / 2 - 1] = t.w; }
                else { const uint2 t = *reinterpret_cast<const uint2*>(rp); rr[0] = t.x; rr[1] = t.y; }
#pragma unroll
                for (int r = 0; r < NCH / 2; ++r) { v[2 * r] += __uint_as_float(rr[r] << 16); v[2 * r + 1] += __uint_as_float(rr[r] & 0xffff0000u); }
            } else {
#pragma unroll
                for (int r = 0; r < NCH; ++r) if (ch0 + r < p.Cout) v[r] += bf2f(rp[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NCH; ++r) {
        if (act == CUTIE_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
        else if (act == CUTIE_ACT_SIGMOID) v[r] = sigmoidf_(v[r]);
        else if (act == CUTIE_ACT_SQ1) v[r] = v[r] * v[r] + 1.f;
    }
    if (out_f32) {
        float* yp = reinterpret_cast<float*>(p.y) + (long)m * p.ldy + ch0;
        if (full && vec_ok) {
#pragma unroll
            for (int r = 0; r < NCH; r += 4) *reinterpret_cast<float4*>(yp + r) = make_float4(v[r], v[r + 1], v[r + 2], v[r + 3]);
        } else {
#pragma unroll
            for (int r = 0; r < NCH; ++r) if (ch0 + r < p.Cout) yp[r] = v[r];
        }
    } else {
        bf16_t* yp = reinterpret_cast<bf16_t*>(p.y) + (long)m * p.ldy + ch0;
        unsigned o[NCH / 2];
#pragma unroll
        for (int r = 0; r < NCH / 2; ++r) o[r] = pack_bf2(v[2 * r], v[2 * r + 1]);
        if (full && vec_ok) {
            if constexpr (NCH == 8) *reinterpret_cast<uint4*>(yp) = make_uint4(o[0], o[1], o[NCH / 2 - 2], o[NCH / 2 - 1]);
            else *reinterpret_cast<uint2*>(yp) = make_uint2(o[0], o[1]);
        } else {
#pragma unroll
            for (int r = 0; r < NCH; ++r) if (ch0 + r < p.Cout) yp[r] = (bf16_t)(r & 1 ? o[r >> 1] >> 16 : o[r >> 1] & 0xffffu);
        }
        {                                                // GAP: the STORED (bf16-rounded) values (dead code without the side job)
#pragma unroll
            for (int r = 0; r < NCH / 2; ++r) { stored[2 * r] = __uint_as_float(o[r] << 16); stored[2 * r + 1] = __uint_as_float(o[r] & 0xffff0000u); }
        }
    }
}

// integer division by a positive runtime divisor on the float pipe (0 <= m < 2^23; rd = 1 / d to an ulp): ~8 instructions instead of the
// ~35 of the exact 32-bit sequence -- the prologue is cold code, every instruction of it costs (profiles/r03_conv_ablation.md)
__device__ __forceinline__ int pc_div(int m, int d, float rd) {
    int q = (int)((float)m * rd);
    int r = m - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// conv_res_row (conv_common.h) on the float pipe
__device__ __forceinline__ int pc_res_row(const ConvParams& p, int m) {
    if (!(p.flags & CUTIE_F_RES_BCAST)) return m;
    int r = m - pc_div(m, p.OHW, __builtin_amdgcn_rcpf((float)p.OHW)) * p.OHW;
    if (p.res_grp_rows > 0) r += pc_div(m, p.res_grp_rows, __builtin_amdgcn_rcpf((float)p.res_grp_rows)) * p.res_grp_stride;
    return r;
}

// GAP side job: column sums of s[0 .. NCH) over the 16 lanes (= 16 pixels) of a row group by a reduce-SCATTER butterfly: after NCH - 1
// + 1 shuffles the lane holds the 16-pixel total of ONE channel (index returned in c; lanes come in pairs -- quadruples for NCH = 4 --
// with the same value, the first of which writes).  One atomic instruction then carries 32 (16) different channels of a wave; issued
// per lane and channel (round 3, first version) the same sums were 8 four-lane instructions per fragment whose requests serialise in
// the L2 on the few lines of the accumulator: +21 us on the 4860 x 256 conv (profiles/r03_conv_ablation.md).
template <int NCH>
__device__ __forceinline__ long long pc_colsum16(const long long (&s)[NCH], int l15, int& c, bool& writer) {
    const bool h8 = l15 & 8, h4 = l15 & 4, h2 = l15 & 2;
    long long w;
    if constexpr (NCH == 8) {
        long long t[4], u[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = (h8 ? s[4 + r] : s[r]) + __shfl_xor(h8 ? s[r] : s[4 + r], 8, 64);
#pragma unroll
        for (int r = 0; r < 2; ++r) u[r] = (h4 ? t[2 + r] : t[r]) + __shfl_xor(h4 ? t[r] : t[2 + r], 4, 64);
        w = (h2 ? u[1] : u[0]) + __shfl_xor(h2 ? u[0] : u[1], 2, 64);
        w += __shfl_xor(w, 1, 64);
        c = (h8 ? 4 : 0) + (h4 ? 2 : 0) + (h2 ? 1 : 0);
        writer = (l15 & 1) == 0;
    } else {
        long long t[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) t[r] = (h8 ? s[2 + r] : s[r]) + __shfl_xor(h8 ? s[r] : s[2 + r], 8, 64);
        w = (h4 ? t[1] : t[0]) + __shfl_xor(h4 ? t[0] : t[1], 4, 64);
        w += __shfl_xor(w, 2, 64);
        w += __shfl_xor(w, 1, 64);
        c = (h8 ? 2 : 0) + (h4 ? 1 : 0);
        writer = (l15 & 3) == 0;
    }
    return w;
}
__device__ __forceinline__ long long pc_gapfx(float v) { return conv_gapfx(v); }      // (conv_common.h: fixed point per VALUE, integer sums)
__device__ __forceinline__ void pc_gap_add(const ConvParams& p, int obj, int ch, long long v) {
    if (ch < p.Cout) atomicAdd(reinterpret_cast<unsigned long long*>(p.gap + (long)obj * p.Cout + ch), (unsigned long long)(v << GAP_ELEM_SHIFT));
}

// Fast epilogue of the common configurations, chosen ONCE per block by wave-uniform tests: whole tile inside Cout, aligned strides, no
// GAP side job, activation none / relu.  Straight-line code per configuration -- the generic slice code (pc_finish) re-tests activation,
// output type, residual kind and raggedness per element; measured on the 64x64 tile: 2.0 us of epilogue against 1.1 us
// (profiles/r03_conv_ablation.md), most of it instruction fetch of code that is executed once per block.
template <int NCH, int TM, int TNP, bool PAIR, bool RELU_OUT, bool F32, bool RES, bool PRE, bool GAPJ = false>
__device__ __forceinline__ void pc_epilogue_fast(const ConvParams& p, const f32x4 (&acc)[PAIR ? 2 * TNP : TNP][TM], const float (&bias)[TNP][NCH],
                                                 const int (&mrow)[TM], const bool (&mval)[TM], int chbase,
                                                 const unsigned (&rpre)[PRE ? TM : 1][PRE ? TNP : 1][NCH / 2], int gap_obj = 0, int l15 = 0) {
    unsigned rr[TM][TNP][NCH / 2];
    if (RES && !PRE) {                                   // all residual loads first: one exposed latency, not one per slice
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const int mres = pc_res_row(p, mrow[b]);
#pragma unroll
            for (int a = 0; a < TNP; ++a) {
                const bf16_t* rp = p.res + (long)mres * p.ldr + chbase + a * (PAIR ? 32 : 16);
#pragma unroll
                for (int r = 0; r < NCH / 2; ++r) rr[b][a][r] = 0u;
                if (mval[b]) {
                    if constexpr (NCH == 8) { const uint4 t = *reinterpret_cast<const uint4*>(rp); rr[b][a][0] = t.x; rr[b][a][1] = t.y; rr[b][a][NCH / 2 - 2] = t.z; rr[b][a][NCH / 2 - 1] = t.w; }
                    else { const uint2 t = *reinterpret_cast<const uint2*>(rp); rr[b][a][0] = t.x; rr[b][a][1] = t.y; }
                }
            }
        }
    }
    // (column group outermost: with GAPJ its 64-bit accumulators -- NCH of them, not TNP x NCH: the BN = 128 tiles crossed 128 VGPRs with all of
    // them alive, four waves per SIMD became three and the 64 x 128 tile 20 % slower -- are reduced and released before the next group)
#pragma unroll
    for (int a = 0; a < TNP; ++a) {
        long long gs[NCH];                               // GAPJ (the wave's rows lie in ONE object, gap_obj): stored values, in fixed point, summed over its fragments
#pragma unroll
        for (int r = 0; r < NCH; ++r) gs[r] = 0;
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            float v[NCH];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[PAIR ? 2 * a : a][b][r] + bias[a][r];
                if (PAIR) v[(NCH - 4) + r] = acc[PAIR ? 2 * a + 1 : a][b][r] + bias[a][(NCH - 4) + r];
            }
            if (RES) {
#pragma unroll
                for (int r = 0; r < NCH / 2; ++r) {
                    const unsigned w = PRE ? rpre[PRE ? b : 0][PRE ? a : 0][r] : rr[b][a][r];
                    v[2 * r] += __uint_as_float(w << 16); v[2 * r + 1] += __uint_as_float(w & 0xffff0000u);
                }
            }
            if (RELU_OUT) {
#pragma unroll
                for (int r = 0; r < NCH; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (mval[b]) {
                const long off = (long)mrow[b] * p.ldy + chbase + a * (PAIR ? 32 : 16);
                if (F32) {
                    float* yp = reinterpret_cast<float*>(p.y) + off;
#pragma unroll
                    for (int r = 0; r < NCH; r += 4) *reinterpret_cast<float4*>(yp + r) = make_float4(v[r], v[r + 1], v[r + 2], v[r + 3]);
                } else {
                    bf16_t* yp = reinterpret_cast<bf16_t*>(p.y) + off;
                    unsigned o[NCH / 2];
#pragma unroll
                    for (int r = 0; r < NCH / 2; ++r) o[r] = pack_bf2(v[2 * r], v[2 * r + 1]);
                    if constexpr (NCH == 8) *reinterpret_cast<uint4*>(yp) = make_uint4(o[0], o[1], o[NCH / 2 - 2], o[NCH / 2 - 1]);
                    else *reinterpret_cast<uint2*>(yp) = make_uint2(o[0], o[1]);
                    if (GAPJ) {                          // the STORED (bf16-rounded) values
#pragma unroll
                        for (int r = 0; r < NCH / 2; ++r) { gs[2 * r] += pc_gapfx(__uint_as_float(o[r] << 16)); gs[2 * r + 1] += pc_gapfx(__uint_as_float(o[r] & 0xffff0000u)); }
                    }
                }
            }
        }
        if (GAPJ) {
            int c; bool writer;
            const long long tot = pc_colsum16<NCH>(gs, l15, c, writer);
            if (writer) pc_gap_add(p, gap_obj, chbase + a * (PAIR ? 32 : 16) + c, tot);
        }
    }
}

// HT == 0: stream mode (TAPS: more than one tap and / or padding); HT = PC_HT(TH, TW): halo mode (TAPS ignored)
// KP: K tiles per barrier (1, or 2 = "pair steps": the ring then holds NS / 2 pairs; see the pair-step notes at the producer loops)
template <int BM, int BN, int WM, int WN, int NPW, int NS, int HT, bool TAPS, bool RELU, bool TWO, int KP = 1>
__global__ __launch_bounds__((WM * WN + NPW) * 64) void conv_pc_kernel(ConvParams p) {
#if __HIP_DEVICE_COMPILE__
    if (p.flags & CUTIE_F_PRIO) __builtin_amdgcn_s_setprio(1);      // a launch of the frame's critical path: see include/cutie_hip.h
    constexpr bool HALO = HT > 0;
    constexpr int TH = HALO ? (HT >> 8) : 1, TW = HALO ? (HT & 255) : 1;
    constexpr int NC = WM * WN, NTC = NC * 64;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr bool PAIR = (TN % 2) == 0;
    constexpr int NCH = PAIR ? 8 : 4, TNP = PAIR ? TN / 2 : TN;          // channels per lane and epilogue slice, slices per pixel
    constexpr int NXI = HALO ? 1 : BM / 8 / NPW;         // stream: X pieces per producer wave per K tile
    constexpr int NWI = BN / 8 / NPW;                    // W pieces per producer wave per K tile
    constexpr int LPT = HALO ? NWI : NXI + NWI;
    constexpr int PH = TH + 2, PW = TW + 2;
    constexpr int NPIECE = (PH * PW + 7) / 8, NXP = (NPIECE + NPW - 1) / NPW, XBUF = NXP * NPW * 1024;
    constexpr int WSTAGE = BN * 128, STAGE = HALO ? WSTAGE : (BM + BN) * 128;
    constexpr int WBASE = HALO ? 2 * XBUF : 0;           // halo: [X slice buffer 0 | 1 | W ring]; stream: ring of [X | W] stages
    constexpr bool PRE_RES = TM * TNP <= 4;              // residual requested before the K loop
    static_assert(TM >= 1 && TN >= 1 && BM % (WM * 16) == 0 && BN % (WN * 16) == 0 && NWI >= 1 && NWI * 8 * NPW == BN && NS >= 2 && NS <= 8, "bad tile");
    static_assert(HALO || (NXI >= 1 && NXI * 8 * NPW == BM), "bad stream tile");
    static_assert(!HALO || (TH * TW <= BM && TH * TW > BM - 16 * WM && (NS == 3 || NS == 4) && NXP <= 11 - NS && NXP <= 8), "bad halo tile");
    static_assert(KP == 1 || (KP == 2 && (NS == 4 || NS == 6)), "pair steps: a ring of 2 or 3 pairs");
    constexpr int NP = NS / 2;                           // (pair steps) pairs the ring holds
    extern __shared__ __attribute__((aligned(16))) u32x4 pc_smem[];
    char* const lds = reinterpret_cast<char*>(pc_smem);
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    TL_DECL(lds + pc_lds_bytes<BM, BN, NS, HT>(NPW))
    TL(0)
    conv_preload_args(p);
    TL(10)
    // XCD-aware tile order (N inner), bijective for any grid
    const int nb = gridDim.x * gridDim.y;
    int logical;
    {
        const int id = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = id & 7, kq = id >> 3, q = nb >> 3, r = nb & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kq;
    }
    const int mt = logical / (int)gridDim.y;
    const int n0 = (logical - mt * (int)gridDim.y) * BN;
    int m0 = 0, bimg = 0, ty0 = 0, tx0 = 0;              // stream: first row; halo: image and first output pixel of the patch
    if (HALO) {
        const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH, per = tiles_x * tiles_y;
        bimg = mt / per;
        const int r = mt - bimg * per, ty = r / tiles_x;
        ty0 = ty * TH;
        tx0 = (r - ty * tiles_x) * TW;
    } else {
        m0 = mt * BM;
    }
    if (p.zero) {                                          // side job: clear an accumulator (the next conv's GAP sums; the transformer's fixed-point sums: 0.9 MB, so every block takes a slice)
        const int nthr = (NC + NPW) * 64;
        for (int z = (blockIdx.y * gridDim.x + blockIdx.x) * nthr + tid; z < p.nzero; z += gridDim.x * gridDim.y * nthr) p.zero[z] = 0ull;
    }
    const int nslice = p.Cin / 64;
#ifdef PC_ABL_NO_LOOP
    const int nk = 0;
#else
    const int nk = HALO ? nslice * 9 : p.KH * p.KW * nslice;
#endif

    if (wave >= NC) {
        // =========================================== producer waves ===========================================
        const int pw = wave - NC;                        // wave-uniform
        const int lr = lane >> 3;
        const unsigned kcb = (unsigned)(((lane & 7) ^ lr) * 16);
        constexpr unsigned PRE = 4096;                   // descriptors start PRE bytes early: instruction offsets i * 1024 stay legal
        unsigned woff[NWI];
#pragma unroll
        for (int i = 0; i < NWI; ++i)
            woff[i] = (unsigned)((n0 + pc_wrow((pw * NWI + i) * 8 + lr, PAIR)) * p.Kpad * 2) + kcb + (HALO ? 0u : PRE - (unsigned)i * 1024u);
        const rsrc_t rw = pc_rsrc(reinterpret_cast<const char*>(p.w) - (HALO ? 0 : PRE));
        const int wdst = WBASE + (HALO ? 0 : BM * 128) + pw * NWI * 1024;
        TL(11)
        if constexpr (!HALO) {
            // the weight tiles of the first NS stages go out BEFORE the pixel offsets are computed (they need nothing but woff)
            const int cin2 = p.Cin * 2, c12 = p.C1 * 2;
            unsigned wsoff = 0;
            int ldw = 0;
#define PCS_WPIECE(I, LD)                                                                                      \
    if constexpr ((I) < NWI) PC_DMA(rw, LDS_PTR(lds + (LD) + wdst), 16, woff[(I) < NWI ? (I) : 0], wsoff, (I) * 1024, 0);
#define PCS_LOAD_W(LD) { PCS_WPIECE(0, LD) PCS_WPIECE(1, LD) PCS_WPIECE(2, LD) PCS_WPIECE(3, LD) PC_SAME_TILE_GUARD(wsoff += 128;) }
            TL(1)
#pragma unroll
            for (int s = 0; s < NS - 1; ++s)
                if (s < nk) { PCS_LOAD_W(ldw) ldw += STAGE; }
            unsigned xoff1[NXI], xoff2[NXI], vmask[NXI];
            {
                int m = m0 + pw * NXI * 8 + lr;
                const int mm = m < p.M ? m : p.M - 1;    // rows past the end recompute the last pixel (never stored)
                const bool linear = !TAPS && p.stride == 1;                     // 1x1 / stride 1: pixel index = row index, no divisions
                int b = 0, oh = 0, ow = mm;
                if (!linear) {
                    b = pc_div(mm, p.OHW, __builtin_amdgcn_rcpf((float)p.OHW));
                    const int rem = mm - b * p.OHW;
                    oh = pc_div(rem, p.OW, __builtin_amdgcn_rcpf((float)p.OW));
                    ow = rem - oh * p.OW;
                }
#pragma unroll
                for (int i = 0; i < NXI; ++i) {
                    const int ih0 = oh * p.stride, iw0 = ow * p.stride;
                    const unsigned pix = linear ? (unsigned)min(mm + 8 * i, p.M - 1) : (unsigned)((b * p.H + ih0) * p.W + iw0);
                    xoff1[i] = pix * (unsigned)(p.ldx1 * 2) + kcb + PRE - (unsigned)i * 1024u;
                    xoff2[i] = TWO ? pix * (unsigned)(p.ldx2 * 2) + kcb + PRE - (unsigned)i * 1024u : 0u;
                    unsigned mk = 0;
                    if (TAPS) {
                        unsigned cols = 0;
                        for (int k = 0; k < p.KW; ++k) cols |= ((unsigned)(iw0 - p.pad + k) < (unsigned)p.W) ? (1u << k) : 0u;
                        for (int k = 0; k < p.KH; ++k) mk |= ((unsigned)(ih0 - p.pad + k) < (unsigned)p.H) ? (cols << (k * p.KW)) : 0u;
                    }
                    vmask[i] = mk;
                    if (i + 1 < NXI && !linear) {
                        m += 8;
                        if (m < p.M) {
                            ow += 8;
                            while (ow >= p.OW) { ow -= p.OW; ++oh; }
                            while (oh >= p.OH) { oh -= p.OH; ++b; }
                        }
                    }
                }
            }
            const long shift = (long)p.pad * p.W + p.pad;
            const char* xb1 = reinterpret_cast<const char*>(p.x1 - shift * p.ldx1) - PRE;
            const char* xb2 = TWO ? reinterpret_cast<const char*>(p.x2 - shift * p.ldx2) - PRE : xb1;
            const rsrc_t rx1 = pc_rsrc(xb1), rx2 = pc_rsrc(xb2);
            const int xdst = pw * NXI * 1024;
            int tap = 0, kw = 0, cc = 0, pixA = 0, pixB = 0;
            const int stepA1 = p.ldx1 * 2, stepA2 = (p.W - p.KW + 1) * p.ldx1 * 2;
            const int stepB1 = p.ldx2 * 2, stepB2 = (p.W - p.KW + 1) * p.ldx2 * 2;
#define PCS_XPIECE(I, LD)                                                                                      \
    if constexpr ((I) < NXI) {                                                                                 \
        const unsigned v_ = in1_ ? xoff1[(I) < NXI ? (I) : 0] : xoff2[(I) < NXI ? (I) : 0];                   \
        PC_DMA(rx_, LDS_PTR(lds + (LD) + xdst), 16, (!TAPS || (vmask[(I) < NXI ? (I) : 0] & tapbit_)) ? v_ : PC_OOB, soff_, (I) * 1024, 0); \
    }
#define PCS_LOAD_X(LD)                                                                                         \
    {                                                                                                          \
        const unsigned tapbit_ = 1u << tap;                                                                    \
        const bool in1_ = !TWO || cc < c12;                                                                    \
        const unsigned soff_ = (unsigned)(in1_ ? pixA + cc : pixB + cc - c12);                                 \
        const rsrc_t rx_ = in1_ ? rx1 : rx2;                                                                   \
        PCS_XPIECE(0, LD) PCS_XPIECE(1, LD) PCS_XPIECE(2, LD) PCS_XPIECE(3, LD)                                \
        PC_SAME_TILE_GUARD(cc += 128;)                                                                         \
        if (TAPS || TWO) {                                                                                     \
            const bool wrap_ = cc >= cin2;                                                                     \
            cc = wrap_ ? 0 : cc;                                                                               \
            if (TAPS) {                                                                                        \
                tap += wrap_ ? 1 : 0;                                                                          \
                kw += wrap_ ? 1 : 0;                                                                           \
                const bool wrap2_ = kw == p.KW;                                                                \
                kw = wrap2_ ? 0 : kw;                                                                          \
                pixA += wrap_ ? (wrap2_ ? stepA2 : stepA1) : 0;                                                \
                if (TWO) pixB += wrap_ ? (wrap2_ ? stepB2 : stepB1) : 0;                                       \
            }                                                                                                  \
        }                                                                                                      \
    }
#define PCS_LOAD_TILE(LD) { PCS_LOAD_X(LD) PCS_LOAD_W(LD) }
            static_assert(NXI <= 4 && NWI <= 4, "pieces per producer wave");
            // Ring protocol: iteration t issues tile t + NS - 1 into the stage of tile t - 1 (read during iteration t - 1: every consumer
            // has passed barrier t - 1 with its reads returned), then waits until tile t + 1 has landed.
            int ld = 0;
#pragma unroll
            for (int s = 0; s < NS - 1; ++s)
                if (s < nk) { PCS_LOAD_X(ld) ld += STAGE; }
            // (queue order: W(0 .. NS-2), X(0 .. NS-2).  NS = 3: tile 0 is complete once only the X pieces of tile 1 are outstanding, and
            // the in-order counts of the loop hold from step 0 on.  Deeper rings: the weights-first order would let X(1) hide behind the
            // younger tiles in the first steps' counts, so the whole prologue burst -- issued back to back -- is awaited here.)
            if constexpr (KP == 1) {
            if (NS == 3 && nk >= NS - 1) { PC_WAIT_VM((NS - 2) * NXI); } else { PC_WAIT_VM(0); }
            TL(2)
            PC_BARRIER()
            int kt = 0;
            for (; kt < nk - (NS - 1); ++kt) {
                TL(3)
                PCS_LOAD_TILE(ld)
                TL(4)
                PC_LOOP_WAIT_VM((NS - 2) * LPT);
                TL(5)
                PC_LOOP_BARRIER()
                ld = ld == (NS - 1) * STAGE ? 0 : ld + STAGE;
            }
            for (; kt < nk; ++kt) {
                PC_WAIT_VM(0);
                PC_LOOP_BARRIER()
            }
            } else {
            // Pair steps.  The consumers multiply tiles 2u and 2u + 1 between barriers u and u + 1 (P = ceil(nk / 2) iterations, P + 1
            // barriers).  The prologues above have issued tiles 0 .. min(nk, NS - 1) - 1 -- one tile more than the NP - 1 pairs this
            // protocol wants in flight; that tile (NS - 2, the first of pair NP - 1) stays counted as issued: `t` below is the next tile
            // to issue, tiles go out strictly in order, so tile t always lands in stage t % NS.  Iteration u runs after barrier u: every
            // consumer has finished reading pair u - 1, whose two stages are the ones tiles 2 (u + NP - 1) and 2 (u + NP - 1) + 1 go
            // to.  Only FULL pairs are issued in the counted part (an odd last tile goes out in the drain, behind vmcnt(0)), so
            // "at most NP - 2 pairs outstanding" really means pair u + 1 has landed.
            PC_WAIT_VM(0);
            TL(2)
            PC_BARRIER()
            int t = nk < NS - 1 ? nk : NS - 1;           // next tile to issue
            const int P = (nk + 1) >> 1;
            int u = 0;
            for (; 2 * (u + NP - 1) + 1 < nk; ++u) {
                TL(3)
                // (tile 2 (u + NP - 1) is already out in iteration 0: the prologue's extra tile)
                for (; t <= 2 * (u + NP - 1) + 1; ++t) { PCS_LOAD_TILE(ld) ld = ld == (NS - 1) * STAGE ? 0 : ld + STAGE; }
                TL(4)
                PC_LOOP_WAIT_VM((NP - 2) * 2 * LPT);
                TL(5)
                PC_LOOP_BARRIER()
            }
            for (; u < P; ++u) {
                for (; t < nk && t <= 2 * (u + NP - 1) + 1; ++t) { PCS_LOAD_TILE(ld) ld = ld == (NS - 1) * STAGE ? 0 : ld + STAGE; }
                PC_WAIT_VM(0);
                PC_LOOP_BARRIER()
            }
            }
#undef PCS_LOAD_TILE
#undef PCS_LOAD_X
#undef PCS_LOAD_W
#undef PCS_WPIECE
#undef PCS_XPIECE
        } else {
            // ---- halo mode: patch pieces of a slice (per-lane pixel offsets, computed once) + one weight tile per K step ----
            unsigned xo1[NXP], xo2[NXP];
#pragma unroll
            for (int i = 0; i < NXP; ++i) {
                const int pp = (pw + NPW * i) * 8 + lr;  // patch pixel of this lane's row
                const int py = pp / PW, px = pp - py * PW;
                const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
                const bool ok = pp < PH * PW && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                const unsigned pix = (unsigned)((bimg * p.H + gy) * p.W + gx);
                const unsigned kc = (unsigned)(((lane & 7) ^ (pp & 7)) * 16);
                xo1[i] = ok ? pix * (unsigned)(p.ldx1 * 2) + kc : PC_OOB;
                xo2[i] = TWO ? (ok ? pix * (unsigned)(p.ldx2 * 2) + kc : PC_OOB) : 0u;
            }
            const rsrc_t rx1 = pc_rsrc(p.x1), rx2 = TWO ? pc_rsrc(p.x2) : rx1;
            const int cin2 = p.Cin * 2;
            // one patch piece (slot I of this wave) of slice SL into slice buffer SL & 1
#define PCH_XPIECE(I, SL)                                                                                      \
    {                                                                                                          \
        const bool f_ = !TWO || (SL) * 64 < p.C1;                                                              \
        const int so_ = (f_ ? (SL) * 64 : (SL) * 64 - p.C1) * 2;                                               \
        PC_DMA(f_ ? rx1 : rx2, LDS_PTR(lds + ((SL) & 1) * XBUF + (pw + NPW * (I)) * 1024), 16, f_ ? xo1[I] : xo2[I], so_, 0, 0); \
    }
            // weight tile of the K step whose state is (wsl, wtap) into ring stage offset wld, then advance that state
            int wsl = 0, wtap = 0, wld = 0;
            unsigned wso = 0;                            // (wtap * Cin + wsl * 64) * 2
#define PCH_WTILE()                                                                                            \
    {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < NWI; ++i) PC_DMA(rw, LDS_PTR(lds + wld + wdst + i * 1024), 16, woff[i], wso, 0, 0); \
        wld = wld == (NS - 1) * WSTAGE ? 0 : wld + WSTAGE;                                                     \
        ++wtap; PC_SAME_TILE_GUARD(wso += (unsigned)cin2;)                                                     \
        if (wtap == 9) { wtap = 0; ++wsl; PC_SAME_TILE_GUARD(wso = (unsigned)(wsl * 128);) }                                       \
    }
            TL(1)
#pragma unroll
            for (int s = 0; s < NS - 1; ++s)
                if (s < nk) PCH_WTILE()
#pragma unroll
            for (int i = 0; i < NXP; ++i) PCH_XPIECE(i, 0)
            PC_WAIT_VM(0);                               // the patch of slice 0 is the youngest request: everything has landed
            TL(2)
            PC_BARRIER()
            if constexpr (KP == 1) {
            // Patch pieces of the NEXT slice: one per K step during taps 0 .. NXP-1 of the current slice, issued BEFORE the step's weight
            // tile; they are older than weight tile 9 (s + 1) (issued in step 9 (s + 1) - (NS - 1): NXP <= 11 - NS), whose counted wait
            // -- before barrier 9 (s + 1) - 1 -- therefore covers them.  The wait of step t leaves the NS - 2 youngest weight tiles and the
            // patch pieces issued in steps t - (NS - 3) .. t in flight.
            int sl = 0, tap = 0;
            int xprev = 0;                               // patch piece issued in the previous step (NS == 4: it may still be in flight)
            int kt = 0;
            for (; kt < nk - (NS - 1); ++kt) {
                TL(3)
                const int xp = (sl + 1 < nslice && tap < NXP) ? 1 : 0;            // wave-uniform
                if (xp) {
                    // slot `tap` of the next slice (a switch keeps the per-lane offsets in registers: no dynamic indexing)
                    switch (tap) {
                        case 0: if constexpr (NXP > 0) PCH_XPIECE(0, sl + 1) break;
                        case 1: if constexpr (NXP > 1) PCH_XPIECE(NXP > 1 ? 1 : 0, sl + 1) break;
                        case 2: if constexpr (NXP > 2) PCH_XPIECE(NXP > 2 ? 2 : 0, sl + 1) break;
                        case 3: if constexpr (NXP > 3) PCH_XPIECE(NXP > 3 ? 3 : 0, sl + 1) break;
                        case 4: if constexpr (NXP > 4) PCH_XPIECE(NXP > 4 ? 4 : 0, sl + 1) break;
                        case 5: if constexpr (NXP > 5) PCH_XPIECE(NXP > 5 ? 5 : 0, sl + 1) break;
                        case 6: if constexpr (NXP > 6) PCH_XPIECE(NXP > 6 ? 6 : 0, sl + 1) break;
                        default: if constexpr (NXP > 7) PCH_XPIECE(NXP > 7 ? 7 : 0, sl + 1) break;
                    }
                }
                PCH_WTILE()
                TL(4)
                const int xin = NS == 3 ? xp : xp + xprev;                        // patch pieces younger than weight tile t + 1
                if (xin == 0) { PC_LOOP_WAIT_VM((NS - 2) * NWI); } else if (xin == 1) { PC_LOOP_WAIT_VM((NS - 2) * NWI + 1); } else { PC_LOOP_WAIT_VM((NS - 2) * NWI + 2); }
                TL(5)
                PC_LOOP_BARRIER()
                xprev = xp;
                ++tap;
                if (tap == 9) { tap = 0; ++sl; }
            }
            for (; kt < nk; ++kt) {
                PC_WAIT_VM(0);
                PC_LOOP_BARRIER()
            }
            } else {
            // Pair steps (see the stream-mode loop).  The prologue above has issued the weight tiles of steps 0 .. min(nk, NS - 1) - 1; `t`
            // is the next one.  Patch pieces of slice s + 1 go to the buffer of slice s - 1: free once the consumers are past step
            // 9 s - 1, i.e. from iteration ceil(9 s / 2) on, and needed at step 9 s + 9, i.e. landed by the end of iteration
            // floor((9 s + 9) / 2) - 1 -- a window of exactly four iterations for every s, two pieces per iteration (NXP <= 8), issued
            // BEFORE the iteration's weight tiles so that the counted wait ("at most NP - 2 weight pairs outstanding") covers them.
            // 9 (nslice - 1) / 2 + 4 <= P - (NP - 1) for NP <= 3: the last window closes inside the counted loop.
            static_assert(NXP <= 8 && NP <= 3, "pair steps: two patch pieces per iteration over four iterations");
            int t = nk < NS - 1 ? nk : NS - 1;
            const int P = (nk + 1) >> 1;
            int ps = 0, pk = 0, pnext = 5;               // slice whose successor is being fetched, iteration inside its window, first iteration of the next window
            int u = 0;
            for (; 2 * (u + NP - 1) + 1 < nk; ++u) {
                TL(3)
                if (u == pnext) { ++ps; pk = 0; pnext += (ps & 1) ? 4 : 5; }           // ceil(9 s / 2) = 0, 5, 9, 14, 18, ...
                if (ps + 1 < nslice && pk < 4) {
                    switch (pk) {
                        case 0: if constexpr (NXP > 0) PCH_XPIECE(0, ps + 1) if constexpr (NXP > 1) PCH_XPIECE(NXP > 1 ? 1 : 0, ps + 1) break;
                        case 1: if constexpr (NXP > 2) PCH_XPIECE(NXP > 2 ? 2 : 0, ps + 1) if constexpr (NXP > 3) PCH_XPIECE(NXP > 3 ? 3 : 0, ps + 1) break;
                        case 2: if constexpr (NXP > 4) PCH_XPIECE(NXP > 4 ? 4 : 0, ps + 1) if constexpr (NXP > 5) PCH_XPIECE(NXP > 5 ? 5 : 0, ps + 1) break;
                        default: if constexpr (NXP > 6) PCH_XPIECE(NXP > 6 ? 6 : 0, ps + 1) if constexpr (NXP > 7) PCH_XPIECE(NXP > 7 ? 7 : 0, ps + 1) break;
                    }
                    ++pk;
                }
                for (; t <= 2 * (u + NP - 1) + 1; ++t) PCH_WTILE()
                TL(4)
                PC_LOOP_WAIT_VM((NP - 2) * 2 * NWI);
                TL(5)
                PC_LOOP_BARRIER()
            }
            for (; u < P; ++u) {
                for (; t < nk && t <= 2 * (u + NP - 1) + 1; ++t) PCH_WTILE()
                PC_WAIT_VM(0);
                PC_LOOP_BARRIER()
            }
            }
#undef PCH_WTILE
#undef PCH_XPIECE
        }
        TLE(6)
        TL_DUMP(logical, nb, NC + NPW)
        if (p.pf_bytes > 0) {
            // side job: touch the next conv's weights.  Hardware block id % 8 = XCD: the blocks of one XCD split the range among them, so
            // every L2 ends up with all of it (what the next launch's first K steps would otherwise wait for); results are discarded
            const int id = blockIdx.y * gridDim.x + blockIdx.x;
            const int nbx = (nb + 7) >> 3, kq = id >> 3;
            u32x4 t = {0u, 0u, 0u, 0u};                      // one destination for all of them, alive until the wait (the data lands late)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned char* base = r ? p.pf2 : p.pf;
                const int bytes = r ? p.pf2_bytes : p.pf_bytes;
                const int chunk = (((bytes + nbx - 1) / nbx) + 1023) & ~1023;
                const int b0 = kq * chunk, b1 = min(b0 + chunk, bytes & ~15);
                for (int o = b0 + (pw * 64 + lane) * 16; o < b1; o += NPW * 1024)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(t) : "v"(base + o) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(t) :: "memory");
        }
        return;
    }

    // =========================================== consumer waves ===========================================
    const int wm = wave / WN, wn = wave % WN;
    const int pm0 = wm * (BM / WM), cn0 = wn * (BN / WN);
    // output pixel of this lane in fragment b
    int mrow[TM];
    bool mval[TM];
    int xbase[TM];                                       // halo: patch pixel of tap (0, 0)
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int q = pm0 + b * 16 + l15;
        if (HALO) {
            const int qq = q < TH * TW ? q : TH * TW - 1;              // rows past the patch (BM > TH * TW) recompute its last pixel
            const int qy = qq / TW, qx = qq - qy * TW;
            const int gy = ty0 + qy, gx = tx0 + qx;
            mval[b] = q < TH * TW && gy < p.H && gx < p.W;
            mrow[b] = (bimg * p.H + min(gy, p.H - 1)) * p.W + min(gx, p.W - 1);
            xbase[b] = qy * PW + qx;
        } else {
            mrow[b] = m0 + q;
            mval[b] = mrow[b] < p.M;
            xbase[b] = 0;
        }
    }
    TL(11)
    // epilogue operands requested up front: bias of this lane's channels, residual (small tiles)
    const bool out_f32 = p.flags & CUTIE_F_OUT_F32;
    const bool vec_ok = (out_f32 ? (p.ldy & 3) == 0 : (p.ldy & (NCH - 1)) == 0) && (!p.res || (p.ldr & (NCH - 1)) == 0);
    float bias[TNP][NCH];
    const float* const bias_p = p.bias ? p.bias : reinterpret_cast<const float*>(p.w);      // (any mapped address when there is no bias)
    const int bias_lim = p.bias ? p.Cout : 0;
    typedef const __attribute__((address_space(1))) float* pc_gf32;       // (a select of two kernel-argument pointers is a flat pointer to hipcc: say global)
    const pc_gf32 bias_g = (pc_gf32)bias_p;
#pragma unroll
    for (int a = 0; a < TNP; ++a) {
        const int ch0 = n0 + cn0 + a * (PAIR ? 32 : 16) + l4 * NCH;
#pragma unroll
        for (int r = 0; r < NCH; ++r) {                  // unconditional (clamped) loads, then a select: a load behind a per-element branch makes hipcc wait for each one
            // (the select must not name p.bias: `(p.bias && ch < Cout) ? t : 0` let hipcc sink every load into its own `if (p.bias)` block,
            // each ending in s_waitcnt vmcnt(0) -- TNP * NCH dependent round trips in front of the consumers' first MFMA, tools/isa_waits.py)
            const float t = bias_g[min(ch0 + r, p.Cout - 1)];
            bias[a][r] = ch0 + r < bias_lim ? t : 0.f;
        }
    }
    unsigned rpre[PRE_RES ? TM : 1][PRE_RES ? TNP : 1][NCH / 2];
    const bool have_res = PRE_RES && p.res && vec_ok;
    if (PRE_RES) {
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int a = 0; a < TNP; ++a) {
                const int ch0 = n0 + cn0 + a * (PAIR ? 32 : 16) + l4 * NCH;
#pragma unroll
                for (int r = 0; r < NCH / 2; ++r) rpre[b][a][r] = 0u;
                if (have_res && mval[b] && ch0 + NCH <= p.Cout) {
                    const int mres = pc_res_row(p, mrow[b]);
                    const bf16_t* rp = p.res + (long)mres * p.ldr + ch0;
                    if constexpr (NCH == 8) { const uint4 t = *reinterpret_cast<const uint4*>(rp); rpre[b][a][0] = t.x; rpre[b][a][1] = t.y; rpre[b][a][NCH / 2 - 2] = t.z; rpre[b][a][NCH / 2 - 1] = t.w; }
                    else { const uint2 t = *reinterpret_cast<const uint2*>(rp); rpre[b][a][0] = t.x; rpre[b][a][1] = t.y; }
                }
            }
    }
    TL(12)
    // fragment read offsets (bytes inside a stage / slice buffer)
    const int c0 = l4 << 4, c1 = (4 + l4) << 4;          // unswizzled chunk byte offsets of the two MFMA k-steps
    const int swl = (l15 & 7) << 4;
    int rdw0, rdx0 = 0;                                  // W fragment a adds a * 2048; stream X fragment b adds b * 2048
    rdw0 = (HALO ? 0 : BM * 128) + (cn0 + l15) * 128;
    if (!HALO) rdx0 = (pm0 + l15) * 128;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#ifdef PC_ABL_NO_MFMA
#define PC_MFMA(AF, BF, ACC) ACC[0] += __uint_as_float(__builtin_bit_cast(u32x4, AF).x ^ __builtin_bit_cast(u32x4, BF).y)
#else
#define PC_MFMA(AF, BF, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AF, BF, ACC, 0, 0, 0)
#endif
#define PC_RELU4(V) { V.x = pc_relu2(V.x); V.y = pc_relu2(V.y); V.z = pc_relu2(V.z); V.w = pc_relu2(V.w); }
    // one K step: read the fragments of tile t (complete since the last barrier), multiply, meet the producers
    int rd = 0;                                          // ring stage of the tile being multiplied (byte offset)
    int sl = 0, tap = 0, toff = 0;                       // halo: slice, tap, patch-pixel offset of the tap
    TL(1)
    PC_BARRIER()                                         // tile 0 (and the patch of slice 0) has landed
    TL(2)
    for (int kt = 0; kt < nk; ++kt) {
        TL(3)
        u32x4 x0[TM], x1[TM];
        bf16x8 w0[TN], w1[TN];
#ifdef PC_ABL_NO_READ
#pragma unroll
        for (int a = 0; a < TN; ++a) { u32x4 v = {(unsigned)rd + a, (unsigned)lane, 5u, 1u}; w0[a] = __builtin_bit_cast(bf16x8, v); v.w = 2u; w1[a] = __builtin_bit_cast(bf16x8, v); }
#pragma unroll
        for (int b = 0; b < TM; ++b) { x0[b] = (u32x4){(unsigned)rd + b, (unsigned)lane, 3u, 1u}; x1[b] = (u32x4){(unsigned)rd + b, (unsigned)lane, 3u, 2u}; }
#else
        const char* const wst = lds + WBASE + rd + rdw0;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            w0[a] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wst + a * 2048 + (c0 ^ swl)));
            w1[a] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wst + a * 2048 + (c1 ^ swl)));
        }
        if (HALO) {
            const char* const xs = lds + (sl & 1) * XBUF;
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int pp = xbase[b] + toff;
                const int sw = (pp & 7) << 4;
                x0[b] = *reinterpret_cast<const u32x4*>(xs + pp * 128 + (c0 ^ sw));
                x1[b] = *reinterpret_cast<const u32x4*>(xs + pp * 128 + (c1 ^ sw));
            }
        } else {
            const char* const xs = lds + rd + rdx0;
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                x0[b] = *reinterpret_cast<const u32x4*>(xs + b * 2048 + (c0 ^ swl));
                x1[b] = *reinterpret_cast<const u32x4*>(xs + b * 2048 + (c1 ^ swl));
            }
        }
#endif
        if (RELU) {
#pragma unroll
            for (int b = 0; b < TM; ++b) { PC_RELU4(x0[b]) PC_RELU4(x1[b]) }
        }
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) PC_MFMA(w0[a], __builtin_bit_cast(bf16x8, x0[b]), acc[a][b]);
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) PC_MFMA(w1[a], __builtin_bit_cast(bf16x8, x1[b]), acc[a][b]);
        TL(4)
        if (KP == 1 || (kt & 1) || kt == nk - 1) {       // pair steps: tiles 2u and 2u + 1 between two barriers
            PC_WAIT_LGKM0();
            PC_LOOP_BARRIER()
        }
        rd = rd == (NS - 1) * STAGE ? 0 : rd + STAGE;
        if (HALO) {
            ++tap;
            toff += (tap == 3 || tap == 6) ? PW - 2 : 1;
            if (tap == 9) { tap = 0; toff = 0; ++sl; }
        }
    }
    TLE(5)
#ifdef PC_ABL_NO_EPILOGUE
    if (acc[0][0][0] != 12345.678f) { TL_DUMP(logical, nb, NC + NPW) return; }
#endif
    // ---- epilogue straight from the accumulators: lane = (pixel l15 of fragment b, channels l4 * NCH .. + NCH of slice a) ----
    // GAP side job: do all rows of this wave lie in one object?  (halo mode: always; stream mode: first and last valid row)
    int gap_obj = bimg;
    bool gap_one = true;
    if (p.gap && !HALO) {
        const int mf = m0 + pm0, ml = min(mf + BM / WM - 1, p.M - 1);
        const float rd = __builtin_amdgcn_rcpf((float)p.OHW);
        gap_obj = pc_div(min(mf, p.M - 1), p.OHW, rd);
        gap_one = pc_div(ml, p.OHW, rd) == gap_obj;
    }
#ifndef PC_ABL_GENERIC_EPI
    {
        const int act_ = (p.flags >> CUTIE_ACT_SHIFT) & 7;
        const bool gapfast = p.gap && gap_one && !act_ && !p.res && !out_f32;
        if (vec_ok && n0 + BN <= p.Cout && (!p.gap || gapfast) && act_ <= CUTIE_ACT_RELU && !(out_f32 && (act_ || p.res))) {       // wave-uniform
            const int chbase = n0 + cn0 + l4 * NCH;
#define PC_FAST(RL, F, RS) pc_epilogue_fast<NCH, TM, TNP, PAIR, RL, F, RS, PRE_RES>(p, acc, bias, mrow, mval, chbase, rpre)
            if (gapfast) pc_epilogue_fast<NCH, TM, TNP, PAIR, false, false, false, PRE_RES, true>(p, acc, bias, mrow, mval, chbase, rpre, gap_obj, l15);
            else if (out_f32) PC_FAST(false, true, false);
            else if (p.res) { if (act_) PC_FAST(true, false, true); else PC_FAST(false, false, true); }
            else { if (act_) PC_FAST(true, false, false); else PC_FAST(false, false, false); }
#undef PC_FAST
            TLE(6)
            TL_DUMP(logical, nb, NC + NPW)
            return;
        }
    }
#endif
#pragma unroll
    for (int a = 0; a < TNP; ++a) {                      // (column group outermost: see pc_epilogue_fast)
        long long gsum[NCH];                             // generic path, GAP with the wave in one object: stored values, in fixed point, summed over its fragments
#pragma unroll
        for (int r = 0; r < NCH; ++r) gsum[r] = 0;
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const int ch0 = n0 + cn0 + a * (PAIR ? 32 : 16) + l4 * NCH;
            const bool live = mval[b] && ch0 < p.Cout;
            float v[NCH], st[NCH];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[PAIR ? 2 * a : a][b][r] + bias[a][r];
                if (PAIR) v[(NCH - 4) + r] = acc[PAIR ? 2 * a + 1 : a][b][r] + bias[a][(NCH - 4) + r];
            }
#pragma unroll
            for (int r = 0; r < NCH; ++r) st[r] = 0.f;
            if (live) pc_finish<NCH>(p, v, mrow[b], ch0, vec_ok, have_res && ch0 + NCH <= p.Cout, rpre[PRE_RES ? b : 0][PRE_RES ? a : 0], st);
            TLE(20 + b * TNP + a)
#ifndef PC_ABL_FIXED_EPI
            if (p.gap) {
                if (gap_one) {
#pragma unroll
                    for (int r = 0; r < NCH; ++r) gsum[r] += pc_gapfx(st[r]);
                } else {
                    // a wave that straddles an object boundary (at most K - 1 row tiles of a launch): per fragment, the lanes of the first
                    // object and the lanes of the second are reduced separately (same-address atomics from 16 lanes of one instruction
                    // serialise in the L2: ~10 us for one such wave); more than two objects in 16 rows (maps smaller than 16 pixels): per lane
                    const unsigned long long vb = __ballot(mval[b]);                                  // wave-uniform
                    if (vb) {
                        const int ob = mval[b] ? pc_div(mrow[b], p.OHW, __builtin_amdgcn_rcpf((float)p.OHW)) : -1;
                        const int o1 = __builtin_amdgcn_readfirstlane(__shfl(ob, __ffsll((long long)vb) - 1, 64));
                        const unsigned long long rest = __ballot(mval[b] && ob != o1);
                        const int o2 = rest ? __builtin_amdgcn_readfirstlane(__shfl(ob, __ffsll((long long)rest) - 1, 64)) : o1;
                        if (__ballot(mval[b] && ob != o1 && ob != o2)) {
                            if (live) {
#pragma unroll
                                for (int r = 0; r < NCH; ++r) pc_gap_add(p, ob, ch0 + r, pc_gapfx(st[r]));
                            }
                        } else {
#pragma unroll
                            for (int pass = 0; pass < 2; ++pass) {
                                const int oo = pass ? o2 : o1;
                                if (pass && !rest) break;
                                long long part[NCH];
#pragma unroll
                                for (int r = 0; r < NCH; ++r) part[r] = (live && ob == oo) ? pc_gapfx(st[r]) : 0;
                                int c; bool writer;
                                const long long tot = pc_colsum16<NCH>(part, l15, c, writer);
                                if (writer) pc_gap_add(p, oo, ch0 + c, tot);
                            }
                        }
                    }
                }
            }
#endif
        }
#ifndef PC_ABL_FIXED_EPI
        if (p.gap && gap_one) {
            int c; bool writer;
            const long long tot = pc_colsum16<NCH>(gsum, l15, c, writer);
            if (writer) pc_gap_add(p, gap_obj, n0 + cn0 + a * (PAIR ? 32 : 16) + l4 * NCH + c, tot);
        }
#endif
    }
    TLE(6)
    TL_DUMP(logical, nb, NC + NPW)
#undef PC_RELU4
#undef PC_MFMA
#endif
}

template <int BM, int BN, int WM, int WN, int NPW, int NS, int HT, bool TAPS, bool RELU, bool TWO, int KP>
static int launch_pc3(const ConvParams& p, hipStream_t s) {
    constexpr int lds = pc_lds_bytes<BM, BN, NS, HT>(NPW) + TL_BYTES;
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (lds > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pc_kernel<BM, BN, WM, WN, NPW, NS, HT, TAPS, RELU, TWO, KP>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            cutie_set_error("conv pc tile: cannot raise the dynamic LDS limit to %d bytes", lds);
            return -2;
        }
        attr_set = true;
    }
    unsigned gx;
    if (HT > 0) {
        constexpr int TH = HT > 0 ? (HT >> 8) : 1, TW = HT > 0 ? (HT & 255) : 1;
        gx = (unsigned)(p.B * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW));
    } else {
        gx = (unsigned)((p.M + BM - 1) / BM);
    }
    hipLaunchKernelGGL((conv_pc_kernel<BM, BN, WM, WN, NPW, NS, HT, TAPS, RELU, TWO, KP>), dim3(gx, (unsigned)((p.Cout + BN - 1) / BN)),
                       dim3((WM * WN + NPW) * 64), lds, s, p);
    return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int NPW, int NS, int HT, int KP = 1>
static int launch_pc(ConvParams p, hipStream_t s) {
    const long x1_bytes = (long)p.B * p.H * p.W * p.ldx1 * 2, x2_bytes = p.C2 ? (long)p.B * p.H * p.W * p.ldx2 * 2 : 0;
    const long gy = (p.Cout + BN - 1) / BN, w_bytes = gy * BN * (long)p.Kpad * 2;
    if (p.Cin % 64 || (p.C2 && p.C1 % 64) || p.KH > 5 || p.KW > 5 || p.splitk != 1 || p.Kpad < p.KH * p.KW * p.Cin ||
        x1_bytes >= PC_RECORDS - 8192 || x2_bytes >= PC_RECORDS - 8192 || w_bytes >= PC_RECORDS - 8192) {
        cutie_set_error("conv pc tile: needs Cin %% 64 == 0 (C1 too for two sources), KH, KW <= 5, no split-K, operands < 2 GiB "
                        "(Cin=%d C1=%d Kpad=%d k=%dx%d splitk=%d)", p.Cin, p.C1, p.Kpad, p.KH, p.KW, p.splitk);
        return -2;
    }
    if (HT > 0 && (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.OH != p.H || p.OW != p.W)) {
        cutie_set_error("conv pc halo tile: needs 3x3 / stride 1 / pad 1 (k=%dx%d stride=%d pad=%d)", p.KH, p.KW, p.stride, p.pad);
        return -2;
    }
    p.Kslice = p.KH * p.KW * p.Cin;
    const bool relu = p.flags & CUTIE_F_RELU_IN, two = p.C2 != 0, taps = p.pad > 0 || p.KH * p.KW > 1;
#define PC_GO(T_, R_, W_) return launch_pc3<BM, BN, WM, WN, NPW, NS, HT, T_, R_, W_, KP>(p, s)
    if constexpr (HT > 0) {
        if (relu) { if (two) PC_GO(true, true, true); PC_GO(true, true, false); }
        if (two) PC_GO(true, false, true);
        PC_GO(true, false, false);
    } else {
        if (taps) {
            if (relu) { if (two) PC_GO(true, true, true); PC_GO(true, true, false); }
            if (two) PC_GO(true, false, true);
            PC_GO(true, false, false);
        }
        if (relu) { if (two) PC_GO(false, true, true); PC_GO(false, true, false); }
        if (two) PC_GO(false, false, true);
        PC_GO(false, false, false);
    }
#undef PC_GO
}

// tile table (mirrored by cutie_amd/ops.py:PC_TILES): id -> BM, BN, consumer waves WM x WN, producer waves, ring depth, halo tile width
int launch_conv_pc(const ConvParams& p, int tile, hipStream_t s) {
    switch (tile) {
        // ---- stream mode (any kernel size / stride) ----
        case 100: return launch_pc<64, 64, 2, 2, 4, 3, 0>(p, s);        // 48 KB
#ifndef PC_QUICK                                                         // (PC_QUICK: two tiles only, for quick compile checks)
        case 101: return launch_pc<64, 64, 2, 2, 4, 4, 0>(p, s);        // 64 KB
        case 102: return launch_pc<32, 64, 2, 2, 4, 4, 0>(p, s);        // 16 x 32 per consumer wave, 48 KB
        case 103: return launch_pc<128, 64, 2, 2, 4, 3, 0>(p, s);       // 72 KB
        case 104: return launch_pc<64, 128, 2, 2, 4, 3, 0>(p, s);       // 72 KB
        case 105: return launch_pc<128, 128, 2, 2, 4, 3, 0>(p, s);      // 64 x 64 per consumer wave, 96 KB
        case 106: return launch_pc<128, 128, 2, 2, 8, 3, 0>(p, s);      // 8 producers
        case 107: return launch_pc<64, 64, 2, 2, 8, 3, 0>(p, s);        // 8 producers: one X + one W piece each
        case 108: return launch_pc<128, 128, 2, 4, 8, 3, 0>(p, s);      // 8 consumers (64 x 32) + 8 producers
        case 109: return launch_pc<32, 64, 2, 2, 4, 6, 0>(p, s);        // deep ring, 72 KB
        case 110: return launch_pc<96, 64, 2, 2, 4, 3, 0>(p, s);        // 48 x 32 per consumer wave: M = 4860 in 51 row tiles
        case 111: return launch_pc<96, 128, 2, 2, 4, 3, 0>(p, s);
#endif
        // ---- halo mode (3x3 / stride 1 / pad 1): TH x TW output patch ----
        case 120: return launch_pc<64, 64, 2, 2, 4, 3, PC_HT(8, 8)>(p, s);
#ifndef PC_QUICK
        case 121: return launch_pc<64, 64, 2, 2, 4, 3, PC_HT(4, 16)>(p, s);
        case 122: return launch_pc<128, 64, 2, 2, 4, 3, PC_HT(8, 16)>(p, s);
        case 123: return launch_pc<128, 128, 2, 2, 4, 3, PC_HT(8, 16)>(p, s);     // 64 x 64 per consumer wave
        case 124: return launch_pc<64, 128, 2, 2, 4, 3, PC_HT(8, 8)>(p, s);
        case 125: return launch_pc<32, 64, 2, 2, 4, 3, PC_HT(4, 8)>(p, s);
        case 126: return launch_pc<128, 128, 2, 4, 8, 3, PC_HT(8, 16)>(p, s);     // 8 consumers + 8 producers
        case 127: return launch_pc<64, 128, 2, 2, 4, 3, PC_HT(4, 16)>(p, s);
        // patch shapes that tile the 30 x 54 (stride-16, 480p) maps into <= 256 workgroups
        case 129: return launch_pc<96, 64, 2, 2, 4, 3, PC_HT(10, 9)>(p, s);       // 3 x 6 patches per object: 216 blocks at K = 3, Cout = 256
        case 130: return launch_pc<32, 64, 2, 2, 4, 3, PC_HT(5, 6)>(p, s);        // 6 x 9 patches per object: 216 blocks at K = 1
        case 131: return launch_pc<64, 128, 2, 2, 4, 3, PC_HT(6, 9)>(p, s);       // 5 x 6 patches: 180 blocks at K = 3, Cout = 256
        case 132: return launch_pc<64, 64, 2, 2, 4, 3, PC_HT(6, 9)>(p, s);
        case 133: return launch_pc<96, 128, 2, 2, 4, 3, PC_HT(10, 9)>(p, s);
        // 20 x 16 patches: the 120 x 216 (stride-4, 480p) maps of 3 objects in 3 x 6 x 14 = 252 workgroups per 64 channels (one round of the 256 CUs)
        case 134: return launch_pc<320, 64, 4, 2, 8, 3, PC_HT(20, 16)>(p, s);     // 80 x 32 per consumer wave, 8 + 8 waves, 136 KB
        // ---- pair steps (two K tiles per barrier; UNMEASURED, branch next/pc-pairstep) ----
        case 140: return launch_pc<64, 64, 2, 2, 4, 4, 0, 2>(p, s);               // tile 101's ring as 2 pairs
        case 141: return launch_pc<96, 64, 2, 2, 4, 4, 0, 2>(p, s);               // tile 110 + one stage
        case 142: return launch_pc<96, 64, 2, 2, 4, 6, 0, 2>(p, s);               // 3 pairs, 120 KB
        case 143: return launch_pc<32, 64, 2, 2, 4, 4, 0, 2>(p, s);               // tile 102
        case 144: return launch_pc<128, 64, 2, 2, 4, 4, 0, 2>(p, s);              // tile 103 + one stage
        case 145: return launch_pc<96, 64, 2, 2, 4, 4, PC_HT(10, 9), 2>(p, s);    // tile 129 + one weight stage
        case 146: return launch_pc<128, 64, 2, 2, 4, 4, PC_HT(8, 16), 2>(p, s);   // tile 122 + one weight stage
        case 147: return launch_pc<32, 64, 2, 2, 4, 4, PC_HT(5, 6), 2>(p, s);     // tile 130 + one weight stage
#endif
        default: cutie_set_error("conv: bad pc tile id %d", tile); return -2;
    }
}
