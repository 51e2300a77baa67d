// Implicit-GEMM convolution on MFMA (gfx950), NHWC bf16 activations, fp32 accumulate.
//
// One kernel family covers every nn.Conv2d / GConv2d / pixel-token nn.Linear of the hot path
// (CUTIE_OP_CONV in include/cutie_hip.h).  GEMM view:  D[cout][pixel] = sum_k W[cout][k] * X[pixel][k],
// k = (kh*KW + kw)*Cin + c.  The weights are the MFMA "A" operand and the im2col pixels the "B"
// operand, so each lane of v_mfma_f32_16x16x32_bf16 ends up holding 4 *consecutive output channels*
// of one pixel -> 8-byte bf16 (16-byte f32) NHWC stores and residual loads.
//
// Tiling: 256 threads = 4 waves, block tile BM pixels x BN channels x BK (32/64/128) per step.
//   * global -> registers -> LDS staging (the im2col gather needs per-chunk zero fill for the halo, the optional
//     fused input ReLU and the 2-source channel concat), with a register *prefetch ring* of S tiles: the loads of
//     tile ks+S-1 are issued before the MFMAs of tile ks.  At the small M of the stride-16 layers (1620 pixels per
//     object) the kernel is latency-, not bandwidth-bound, so the ring (counted vmcnt waits, kept branch-free so
//     hipcc does not fall back to vmcnt(0)) and a larger BK (fewer barriers per MFMA) are what matter.
//   * LDS double-buffered, one barrier per K step; rows are BK*2 bytes and the 16-B chunk index is XOR-swizzled
//     (row&(CPR-1), or ((row>>3)&1)*3 for 64-B rows) so ds_read_b128 fragment reads spread over the bank rows.
//   * the tile shape is chosen per layer by the host (autotuned at plan build, cutie_amd/model/plans.py).
#include "conv_common.h"

// OCC: waves per SIMD the register allocator must leave room for (pinned per tile: hipcc's occupancy heuristic is
// otherwise chaotic w.r.t. small source changes, e.g. 156 -> 208 VGPRs on the 32x64x128 tile = 3 -> 2 resident blocks).
// WK: K groups per block.  WK > 1 = intra-block split-K: the block holds WK independent copies of the whole 4-wave (or
// 8-wave) pipeline, each with its own LDS buffers and accumulators, each streaming 1/WK of the K tiles of the SAME output
// tile; the accumulators are summed through LDS before the epilogue.  The L2->CU fill rate of this chip scales with the
// number of waves that have loads in flight (~11 GB/s per wave, tools/membench.hip), and the small-M layers of this model
// cannot put more than ~1 block on a CU; splitting the OUTPUT tile over more waves (the 8-wave tiles) leaves the bytes in
// flight per block unchanged, splitting K doubles them.
template <int BM, int BN, int WM, int WN, int BK, int S, int OCC, int WK>
__global__ __launch_bounds__(WM * WN * WK * 64) void conv_igemm_kernel(ConvParams p) {
    if (p.flags & CUTIE_F_PRIO) __builtin_amdgcn_s_setprio(1);      // a launch of the frame's critical path: see include/cutie_hip.h
    constexpr int NTB = WM * WN * WK * 64;              // threads per block
    constexpr int NT = WM * WN * 64;                    // threads per K group: 4 or 8 waves
    constexpr int CPR = BK / 8;                         // 16-B chunks per LDS row
    constexpr int RPT = NT / CPR;                       // rows covered by one pass of the group's threads
    constexpr int NX = (BM * CPR) / NT;                 // X chunks per thread per tile
    constexpr int NWC = (BN * CPR + NT - 1) / NT;       // W chunks per thread per tile
    constexpr int TM = BM / WM / 16;                    // 16-pixel MFMA tiles per wave
    constexpr int TN = BN / WN / 16;                    // 16-channel MFMA tiles per wave
    constexpr int KSUB = BK / 32;                       // MFMA k-steps per tile
    constexpr int NWRAP = BK == 32 ? 4 : BK / 32;       // tap wraps per tile advance (BK > 32 requires Cin >= 32)
    static_assert((NT == 256 || NT == 512) && NTB <= 1024 && NX >= 1 && TM >= 1 && TN >= 1 && S >= 2 && (BM * CPR) % NT == 0, "bad tile");
    constexpr int LDC = BN + 4;                         // fp32 row stride of the epilogue tile (pad: bank spread)
    constexpr int GRP_CHUNKS = 2 * (BM + BN) * CPR;     // 16-B chunks of one group's double-buffered operand tiles
    extern __shared__ __attribute__((aligned(16))) u32x4 smem_raw[];    // max(WK * GRP_CHUNKS, epilogue tile): conv_lds_bytes()
    const int grp = threadIdx.x / NT;                   // K group of this wave (wave-uniform)
    u32x4 (*smem)[(BM + BN) * CPR] = reinterpret_cast<u32x4 (*)[(BM + BN) * CPR]>(smem_raw + grp * GRP_CHUNKS);

    const int tid = threadIdx.x % NT, lane = tid & 63, wave = tid >> 6;   // position inside the K group
    // XCD-aware tile mapping: hardware block b runs on XCD b % 8 (8 private L2s).  Give every XCD a contiguous run of
    // logical tiles (channel tile fastest) so the blocks sharing an activation row-band hit the same L2 instead of each
    // XCD pulling its own copy through the fabric (rocprof FETCH_SIZE was ~8x the unique bytes without it).  Bijective
    // for any grid size; affects speed only.
    int m0, n0;
    {
        const int nb = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = id & 7, kq = id >> 3, q = nb >> 3, r = nb & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kq;
        const int mt = logical / (int)gridDim.y;
        m0 = mt * BM;
        n0 = (logical - mt * (int)gridDim.y) * BN;
    }
    const int kc = tid % CPR;                            // this thread's 8-element chunk inside BK
    const int trow = tid / CPR;                          // first row handled (then + RPT per extra chunk)
    const bool relu_in = p.flags & CUTIE_F_RELU_IN;

    // ---- per-thread im2col row state (fixed over the K loop) ----
    int rb[NX], rih[NX], riw[NX];
    bool rvalid[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        int m = m0 + trow + i * RPT;
        rvalid[i] = m < p.M;
        int mm = rvalid[i] ? m : 0;
        int b = mm / p.OHW;
        int rem = mm - b * p.OHW;
        int oh = rem / p.OW;
        int ow = rem - oh * p.OW;
        rb[i] = b * p.H;
        rih[i] = oh * p.stride - p.pad;
        riw[i] = ow * p.stride - p.pad;
    }
    // split-K: grid.z slices the K tiles; this block owns tiles [kbeg, kbeg + nk)
    // (nk is deliberately written as <kernel argument> / BK, like the unsplit kernel's Kpad / BK: hipcc's register
    // allocation for this kernel is chaotic in the form of this expression -- any other spelling costs 40-50 VGPRs)
    const int nk = p.Kslice / (BK * WK);
    const int kbeg = ((int)blockIdx.z * WK + grp) * nk;
    const bf16_t* wp[NWC];                               // this thread's weight chunks of the next tile to load
#pragma unroll
    for (int i = 0; i < NWC; ++i) {
        int n = trow + i * RPT;
        if (BN < RPT) n = n < BN ? n : BN - 1;
        wp[i] = p.w + (long)(n0 + n) * p.Kpad + (long)kbeg * BK + kc * 8;
    }
    int kcur, kh, kw;                                    // channel / tap of this thread's chunk of the next tile to load
    {
        const int kabs = kbeg * BK + kc * 8;
        const int tap = kabs / p.Cin;
        kcur = kabs - tap * p.Cin;
        kh = tap / p.KW;
        kw = tap - kh * p.KW;
    }

    u32x4 xr[S][NX], wr[S][NWC];
    unsigned okmask[S];                                  // bit i of okmask[slot]: chunk i of that tile is real data (else zero)
    // Branch-free on purpose: loads behind exec-masked branches (or data-dependent loops between them) make hipcc's
    // waitcnt insertion fall back to vmcnt(0), which would drain the whole prefetch ring at every LDS write.
#define ADVANCE_TAP()                                                                                      \
    {                                                                                                      \
        kcur += BK;                                                                                        \
        _Pragma("unroll") for (int r_ = 0; r_ < NWRAP; ++r_) {                                             \
            const bool w_ = kcur >= p.Cin;                                                                 \
            kcur -= w_ ? p.Cin : 0;                                                                        \
            kw += w_ ? 1 : 0;                                                                              \
            const bool w2_ = kw == p.KW;                                                                   \
            kw = w2_ ? 0 : kw;                                                                             \
            kh += w2_ ? 1 : 0;                                                                             \
        }                                                                                                  \
    }
#define LOAD_TILE(KS, SLOT)                                                                                \
    {                                                                                                      \
        unsigned ok_ = 0;                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NX; ++i) {                                                   \
            const int ih = rih[i] + kh, iw = riw[i] + kw;                                                  \
            const bool v_ = rvalid[i] && kh < p.KH && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W; \
            const long pix = ((long)(rb[i] + ih)) * p.W + iw;                                              \
            const bf16_t* src = (kcur < p.C1) ? p.x1 + pix * p.ldx1 + kcur : p.x2 + pix * p.ldx2 + (kcur - p.C1); \
            src = v_ ? src : p.x1;                           /* any mapped address; the value is discarded */ \
            xr[SLOT][i] = GLOAD16(src);                                            \
            ok_ |= v_ ? (1u << i) : 0u;                                                                    \
        }                                                                                                  \
        okmask[SLOT] = ok_;                                                                                \
        _Pragma("unroll") for (int i = 0; i < NWC; ++i) {                                                  \
            wr[SLOT][i] = GLOAD16(wp[i]);                                                                  \
            wp[i] += BK;                                                                                   \
        }                                                                                                  \
        ADVANCE_TAP();                                                                                     \
    }
#define STORE_TILE(BUF, SLOT)                                                                              \
    {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < NX; ++i) {                                                   \
            const int row = trow + i * RPT;                                                                \
            u32x4 v = xr[SLOT][i];                                                                         \
            const unsigned keep = (okmask[SLOT] >> i) & 1u ? 0xffffffffu : 0u;                             \
            v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;                                            \
            if (relu_in) { v.x = relu_bf2(v.x); v.y = relu_bf2(v.y); v.z = relu_bf2(v.z); v.w = relu_bf2(v.w); } \
            smem[BUF][row * CPR + (kc ^ swz<CPR>(row))] = v;                                               \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NWC; ++i) {                                                  \
            const int n = trow + i * RPT;                                                                  \
            if (BN >= RPT || n < BN) smem[BUF][(BM + n) * CPR + (kc ^ swz<CPR>(n))] = wr[SLOT][i];         \
        }                                                                                                  \
    }

    const int wm = wave / WN, wn = wave % WN;
    const int pm0 = wm * (BM / WM), cn0 = wn * (BN / WN);
    const int l15 = lane & 15, l4 = lane >> 4;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // K loop.  Tile t lives in ring slot t % S; the loads of tile ks+S-1 are issued before the MFMAs of tile ks, so a
    // global/L2 round trip has S-2 whole iterations to land.  LDS is double-buffered; one barrier per tile.
#pragma unroll
    for (int t = 0; t < S - 1; ++t)
        if (t < nk) LOAD_TILE(t, t);
    STORE_TILE(0, 0);
    __syncthreads();
#define K_ITER(KS, U, DO_LOAD, DO_STORE)                                                                   \
    {                                                                                                      \
        const int buf = (KS) & 1;                                                                          \
        if (DO_LOAD) LOAD_TILE((KS) + S - 1, ((U) + S - 1) % S);                                           \
        _Pragma("unroll") for (int j = 0; j < KSUB; ++j) {                                                 \
            bf16x8 bfr[TM], afr[TN];                                                                       \
            _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                               \
                const int row = pm0 + t * 16 + l15;                                                        \
                bfr[t] = __builtin_bit_cast(bf16x8, smem[buf][row * CPR + ((j * 4 + l4) ^ swz<CPR>(row))]); \
            }                                                                                              \
            _Pragma("unroll") for (int t = 0; t < TN; ++t) {                                               \
                const int row = cn0 + t * 16 + l15;                                                        \
                afr[t] = __builtin_bit_cast(bf16x8, smem[buf][(BM + row) * CPR + ((j * 4 + l4) ^ swz<CPR>(row))]); \
            }                                                                                              \
            _Pragma("unroll") for (int a = 0; a < TN; ++a)                                                 \
                _Pragma("unroll") for (int b = 0; b < TM; ++b)                                             \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[a], bfr[b], acc[a][b], 0, 0, 0); \
        }                                                                                                  \
        if (DO_STORE) STORE_TILE(buf ^ 1, ((U) + 1) % S);                                                  \
        __syncthreads();                                                                                   \
    }
    int ks0 = 0;
    // steady state: straight-line body (no guards), so the compiler can keep counted vmcnt waits
    for (; ks0 + 2 * S - 1 <= nk; ks0 += S) {
#pragma unroll
        for (int u = 0; u < S; ++u) K_ITER(ks0 + u, u, true, true)
    }
    // tail: the last < 2S tiles, guarded
    for (; ks0 < nk; ks0 += S) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int ks = ks0 + u;
            if (ks < nk) K_ITER(ks, u, ks + S - 1 < nk, ks + 1 < nk)
        }
    }

    // ---- epilogue.  The MFMA layout gives each lane 4 consecutive channels of one pixel (8-B pieces); the tile is
    // transposed through LDS (fp32) so that global traffic is whole 16-B chunks along the channel axis: threads
    // walk [pixel][8-channel chunk], i.e. full 128/256-B lines per pixel for the residual loads and the stores. ----
    float* ctile = reinterpret_cast<float*>(smem_raw);   // the K loop ended with a barrier: operand tiles are dead
#pragma unroll
    for (int g = WK - 1; g >= 0; --g) {                  // K groups add their accumulators in a fixed order
        if (grp == g) {
#pragma unroll
            for (int b = 0; b < TM; ++b)
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    const int px = pm0 + b * 16 + l15, ch = cn0 + a * 16 + l4 * 4;
                    f32x4* dst = reinterpret_cast<f32x4*>(ctile + px * LDC + ch);
                    if (g == WK - 1) *dst = acc[a][b];
                    else { f32x4 t = *dst; t[0] += acc[a][b][0]; t[1] += acc[a][b][1]; t[2] += acc[a][b][2]; t[3] += acc[a][b][3]; *dst = t; }
                }
        }
        __syncthreads();
    }
    constexpr int CH8 = BN / 8;                          // 8-channel chunks per tile row
    if (p.splitk == 1) {
        for (int q = threadIdx.x; q < BM * CH8; q += NTB) {
            const int px = q / CH8, c8 = q - px * CH8;
            const int m = m0 + px, ch0 = n0 + c8 * 8;
            if (m >= p.M || ch0 >= p.Cout) continue;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(ctile + px * LDC + c8 * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(ctile + px * LDC + c8 * 8 + 4);
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            conv_finish(p, v, m, ch0);
        }
        return;
    }
    // split-K: park the raw fp32 partial tile; conv_splitk_reduce_kernel (next launch on the stream) sums the slices in
    // slice order and runs the epilogue.  (A single-kernel "last arriver reduces" variant was measured first: the
    // device-scope release/acquire fences it needs write back / invalidate the XCD's L2 and cost 25-60 us per launch.)
    float* part = p.part + (long)blockIdx.z * p.M * p.ldp;
    for (int q = threadIdx.x; q < BM * CH8; q += NTB) {
        const int px = q / CH8, c8 = q - px * CH8;
        const int m = m0 + px, ch0 = n0 + c8 * 8;
        if (m >= p.M || ch0 >= p.Cout) continue;
        float* dst = part + (long)m * p.ldp + ch0;
        *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(ctile + px * LDC + c8 * 8);
        *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(ctile + px * LDC + c8 * 8 + 4);
    }
}

// Second half of a split-K conv: y = epilogue(sum_z part[z]) ; one thread per (pixel, 8-channel chunk).
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(ConvParams p) {
    const int c8n = p.ldp >> 3;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= (long)p.M * c8n) return;
    const int m = (int)(q / c8n), ch0 = (int)(q - (long)m * c8n) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.splitk; ++z) {
        const float* src = p.part + ((long)z * p.M + m) * p.ldp + ch0;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(src + 4);
        v[0] += lo[0]; v[1] += lo[1]; v[2] += lo[2]; v[3] += lo[3];
        v[4] += hi[0]; v[5] += hi[1]; v[6] += hi[2]; v[7] += hi[3];
    }
    conv_finish(p, v, m, ch0);
}

// ---- Cout == 1 convolutions (mask_pred 1x1, d_proj / pred 3x3): a per-pixel dot product, HBM/L2-bound.  An MFMA tile
// would waste 15/16 of its columns and (at M = 1620) leave 249 CUs idle, so: LP = Cin/8 lanes per output pixel, each lane
// owns 8 channels (one 16-B load per tap), weights staged once per block in LDS, shuffle reduction over the LP lanes.
__global__ __launch_bounds__(256) void conv_cout1_kernel(ConvParams p) {
    if (p.flags & CUTIE_F_PRIO) __builtin_amdgcn_s_setprio(1);
    extern __shared__ __attribute__((aligned(16))) unsigned char wlds_raw[];
    u32x4* wlds = reinterpret_cast<u32x4*>(wlds_raw);                   // [KH*KW][Cin/8] chunks of the single filter
    const int LP = p.Cin >> 3;                                           // lanes per pixel (power of two, <= 64)
    const int ntap = p.KH * p.KW;
    const int ppb = 256 / LP;                                            // pixels per block
    const int sub = threadIdx.x / LP, cl = threadIdx.x % LP;
    const int m = blockIdx.x * ppb + sub;
    const bool is33 = p.KH == 3 && p.KW == 3;
    const bool relu_in = p.flags & CUTIE_F_RELU_IN;
    const float bias0 = p.bias ? p.bias[0] : 0.f;                        // requested with everything else, not after the reduction
    // 3x3: the nine taps of this pixel are requested FIRST (clamped addresses, invalid taps multiplied by zero), then the filter is staged:
    // staging + barrier in front of them made every launch two dependent memory round trips (8.2 us per launch at 4860 pixels)
    u32x4 xv[9];
    float ok[9];
    if (is33) {
        const int mc = min(m, p.M - 1);
        const int b = mc / p.OHW, rem = mc - b * p.OHW, oh = rem / p.OW, ow = rem - oh * p.OW;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ih = oh * p.stride - p.pad + t / 3, iw = ow * p.stride - p.pad + t % 3;
            const bool v = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
            xv[t] = *reinterpret_cast<const u32x4*>(p.x1 + (((long)b * p.H + ihc) * p.W + iwc) * p.ldx1 + cl * 8);
            ok[t] = v ? 1.f : 0.f;
        }
    }
    for (int t = threadIdx.x; t < ntap * LP; t += 256) wlds[t] = *reinterpret_cast<const u32x4*>(p.w + (long)t * 8);
    __syncthreads();
    float acc = 0.f;
    if (m < p.M) {
        const int b = m / p.OHW, rem = m - b * p.OHW, oh = rem / p.OW, ow = rem - oh * p.OW;
        if (is33) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                u32x4 x = xv[t];
                if (relu_in) { x.x = relu_bf2(x.x); x.y = relu_bf2(x.y); x.z = relu_bf2(x.z); x.w = relu_bf2(x.w); }
                const u32x4 wv = wlds[t * LP + cl];
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a += __uint_as_float(x[i] << 16) * __uint_as_float(wv[i] << 16);
                    a += __uint_as_float(x[i] & 0xffff0000u) * __uint_as_float(wv[i] & 0xffff0000u);
                }
                acc += a * ok[t];
            }
        } else
        for (int kh = 0; kh < p.KH; ++kh) {
            const int ih = oh * p.stride - p.pad + kh;
            if ((unsigned)ih >= (unsigned)p.H) continue;
            for (int kw = 0; kw < p.KW; ++kw) {
                const int iw = ow * p.stride - p.pad + kw;
                if ((unsigned)iw >= (unsigned)p.W) continue;
                u32x4 xv = *reinterpret_cast<const u32x4*>(p.x1 + (((long)b * p.H + ih) * p.W + iw) * p.ldx1 + cl * 8);
                if (relu_in) { xv.x = relu_bf2(xv.x); xv.y = relu_bf2(xv.y); xv.z = relu_bf2(xv.z); xv.w = relu_bf2(xv.w); }
                const u32x4 wv = wlds[(kh * p.KW + kw) * LP + cl];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc += __uint_as_float(xv[i] << 16) * __uint_as_float(wv[i] << 16);
                    acc += __uint_as_float(xv[i] & 0xffff0000u) * __uint_as_float(wv[i] & 0xffff0000u);
                }
            }
        }
    }
    for (int o = LP >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (cl == 0 && m < p.M) {
        float v = acc + bias0;
        const int act = (p.flags >> CUTIE_ACT_SHIFT) & 7;
        if (act == CUTIE_ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == CUTIE_ACT_SIGMOID) v = sigmoidf_(v);
        else if (act == CUTIE_ACT_SQ1) v = v * v + 1.f;
        if (p.flags & CUTIE_F_OUT_F32) reinterpret_cast<float*>(p.y)[(long)m * p.ldy] = v;
        else reinterpret_cast<bf16_t*>(p.y)[(long)m * p.ldy] = f2bf(v);
    }
}

// The same layer with the input TILE in LDS and the dot products on the matrix cores (Cin = 128; round 4: LDS tile + VALU, round 5: MFMA).
// A block owns TR x 16 output pixels and stages the (TR + 2) x 18 input pixels once (coalesced 256-byte rows, fused input ReLU applied on
// the way in, zeros outside the image = the conv's padding).  The VALU form then spent 72 x (unpack + 16 FMA) per thread -- 8.4 us of pure
// VALU issue per launch at 2.5 blocks per CU, for 90 MFLOP -- so the nine taps x 128 channels run as 36 k-steps of
// v_mfma_f32_16x16x32_bf16: M = the 16 pixels of an output row, K = 32 channels of one tap, N = 16 with ONE real column (the filter; the
// other fifteen B columns are zero): 15/16 of the matrix work is wasted and it is still ~10x cheaper than the VALU loop.  A fragment =
// one ds_read_b128 of the pixel the tap lands on; the 16-B chunks of a pixel are XOR-swizzled with the pixel index (the 16 pixels of a
// fragment read sit 256 B apart: without it, 8 lanes per bank group).  fp32 accumulation in MFMA order (k ascending per tap): results agree
// with conv_cout1_rows_kernel / conv_cout1_kernel to fp32 rounding, not bit for bit.
template <int TR>
__global__ __launch_bounds__(256) void conv_cout1_tile_kernel(ConvParams p) {
    if (p.flags & CUTIE_F_PRIO) __builtin_amdgcn_s_setprio(1);
    static_assert(TR % 4 == 0, "rows per wave");
    constexpr int TC = 16, PW = TC + 2, NPIX = (TR + 2) * PW, RW = TR / 4;
    __shared__ u32x4 tile[NPIX * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const int tiles_x = (p.W + TC - 1) / TC, tiles_y = (p.H + TR - 1) / TR;
    const int b = blockIdx.x / (tiles_x * tiles_y), rem = blockIdx.x - b * tiles_x * tiles_y;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int x0 = tx * TC, y0 = ty * TR;
    const bool relu_in = p.flags & CUTIE_F_RELU_IN;
    const float bias0 = p.bias ? p.bias[0] : 0.f;
    // the filter as the B operand: k-step s = (tap s >> 2, channels 32 (s & 3) ..), lane (column n, k group g) holds channels + 8 g .. + 7 of
    // column n -- the filter for n == 0, zeros otherwise.
    // (kept in LDS, 2.3 KB, not in 144 registers per lane: with them the kernel held 240 VGPRs = two blocks per CU, and 630 blocks on 512
    // slots is two rounds; three blocks per CU take the 480p head in one)
    __shared__ u32x4 wl[36 * 4];
    const u32x4 wmine = tid < 144 ? *reinterpret_cast<const u32x4*>(p.w + (long)((tid >> 4) * 16 + (tid & 15)) * 8) : (u32x4){0u, 0u, 0u, 0u};
    constexpr int NCH = (NPIX * 16 + 255) / 256;
    u32x4 st[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int q = tid + 256 * i, px = q >> 4, c = q & 15;
        const int iy = y0 + px / PW - 1, ix = x0 + px % PW - 1;
        const bool in = q < NPIX * 16 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const long o = in ? (((long)b * p.H + iy) * p.W + ix) * p.ldx1 + c * 8 : 0;
        const u32x4 v = *reinterpret_cast<const u32x4*>(p.x1 + o);       // unconditional (clamped) loads: all in flight at once
        st[i] = in ? v : (u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int q = tid + 256 * i, px = q >> 4, c = q & 15;
        u32x4 v = st[i];
        if (relu_in) { v.x = relu_bf2(v.x); v.y = relu_bf2(v.y); v.z = relu_bf2(v.z); v.w = relu_bf2(v.w); }
        if (q < NPIX * 16) tile[px * 16 + (c ^ (px & 15))] = v;
    }
    if (tid < 144) wl[tid] = wmine;                           // [tap][16 chunks of 8 channels] = [k-step][k group] in order
    __syncthreads();
    f32x4 acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int sidx = 0; sidx < 36; ++sidx) {
        const int tap = sidx >> 2, ky = tap / 3, kx = tap - 3 * ky, ch = (sidx & 3) * 4 + g;
        const u32x4 wv = wl[sidx * 4 + g];
        const bf16x8 wb = __builtin_bit_cast(bf16x8, n == 0 ? wv : zero4);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int px = (wave * RW + r + ky) * PW + n + kx;          // A: row m = lane & 15 = the output column, its tap pixel
            const bf16x8 a = __builtin_bit_cast(bf16x8, tile[px * 16 + (ch ^ (px & 15))]);
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wb, acc[r], 0, 0, 0);
        }
    }
    // D[m = 4 g + q][n]: the lanes of column n == 0 hold four output pixels of each of the wave's rows
    if (n == 0) {
        const int act = (p.flags >> CUTIE_ACT_SHIFT) & 7;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int oy = y0 + wave * RW + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int x = x0 + 4 * g + q;
                if (x < p.W && oy < p.H) {
                    float v = acc[r][q] + bias0;
                    if (act == CUTIE_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (act == CUTIE_ACT_SIGMOID) v = sigmoidf_(v);
                    else if (act == CUTIE_ACT_SQ1) v = v * v + 1.f;
                    const long m = ((long)b * p.H + oy) * p.W + x;
                    if (p.flags & CUTIE_F_OUT_F32) reinterpret_cast<float*>(p.y)[m * p.ldy] = v;
                    else reinterpret_cast<bf16_t*>(p.y)[m * p.ldy] = f2bf(v);
                }
            }
        }
    }
}

// Cout == 1, 3x3 / stride 1 / pad 1 on a large map (the decoder's logits head: 3 x 120 x 216 pixels x 128 channels): a thread column.
// conv_cout1_kernel fetches the nine taps of every pixel separately -- 9 vector requests per output, 180 MB through the texture path for
// 20 MB of input.  Here a thread (pixel column x, 8-channel lane) walks R output rows: each of the R + 2 input rows is requested ONCE
// per column tap (3 (R + 2) requests for R outputs: 3.75 instead of 9 at R = 8), all of them up front, the nine filter chunks of the lane
// live in registers (no LDS, no barrier).  Per output the products are summed in conv_cout1_kernel's order (tap by tap, padding taps
// as a * 0, then the same lane butterfly): bit-identical results.
template <int R>
__global__ __launch_bounds__(256) void conv_cout1_rows_kernel(ConvParams p) {
    if (p.flags & CUTIE_F_PRIO) __builtin_amdgcn_s_setprio(1);
    const int LP = p.Cin >> 3;                                           // lanes per pixel (8, 16 or 32)
    const int ppb = 256 / LP;                                            // pixel columns per block
    const int sub = threadIdx.x / LP, cl = threadIdx.x % LP;
    const int tiles_x = (p.W + ppb - 1) / ppb, tiles_y = (p.H + R - 1) / R;
    const int b = blockIdx.x / (tiles_x * tiles_y), rem = blockIdx.x - b * tiles_x * tiles_y;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int x = tx * ppb + sub, y0 = ty * R;
    const bool relu_in = p.flags & CUTIE_F_RELU_IN;
    const float bias0 = p.bias ? p.bias[0] : 0.f;
    u32x4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const u32x4*>(p.w + (long)(t * LP + cl) * 8);
    u32x4 xv[R + 2][3];
    float okc[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) okc[kx] = ((unsigned)(x + kx - 1) < (unsigned)p.W) ? 1.f : 0.f;
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
        const int iy = min(max(y0 + r - 1, 0), p.H - 1);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = min(max(x + kx - 1, 0), p.W - 1);
            xv[r][kx] = *reinterpret_cast<const u32x4*>(p.x1 + (((long)b * p.H + iy) * p.W + ix) * p.ldx1 + cl * 8);
        }
    }
    if (relu_in) {
#pragma unroll
        for (int r = 0; r < R + 2; ++r)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) { u32x4& v = xv[r][kx]; v.x = relu_bf2(v.x); v.y = relu_bf2(v.y); v.z = relu_bf2(v.z); v.w = relu_bf2(v.w); }
    }
    const int act = (p.flags >> CUTIE_ACT_SHIFT) & 7;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int oy = y0 + j;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float okr = ((unsigned)(oy + ky - 1) < (unsigned)p.H) ? 1.f : 0.f;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const u32x4 xx = xv[j + ky][kx], ww = wv[ky * 3 + kx];
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a += __uint_as_float(xx[i] << 16) * __uint_as_float(ww[i] << 16);
                    a += __uint_as_float(xx[i] & 0xffff0000u) * __uint_as_float(ww[i] & 0xffff0000u);
                }
                acc += a * (okr * okc[kx]);
            }
        }
        for (int o = LP >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (cl == 0 && x < p.W && oy < p.H) {
            float v = acc + bias0;
            if (act == CUTIE_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == CUTIE_ACT_SIGMOID) v = sigmoidf_(v);
            else if (act == CUTIE_ACT_SQ1) v = v * v + 1.f;
            const long m = ((long)b * p.H + oy) * p.W + x;
            if (p.flags & CUTIE_F_OUT_F32) reinterpret_cast<float*>(p.y)[m * p.ldy] = v;
            else reinterpret_cast<bf16_t*>(p.y)[m * p.ldy] = f2bf(v);
        }
    }
}

template <int BM, int BN, int WM, int WN, int BK, int S, int OCC, int WK = 1>
static int launch_cfg(ConvParams p, hipStream_t s) {
    if (p.Kpad % BK) { cutie_set_error("conv: Kpad %d not a multiple of BK %d", p.Kpad, BK); return -2; }
    if (BK > 32 && p.Cin < 32) { cutie_set_error("conv: BK %d needs Cin >= 32 (Cin=%d)", BK, p.Cin); return -2; }
    const int nk_all = p.Kpad / BK;
    if (p.splitk > nk_all) { cutie_set_error("conv: splitk %d > K tiles %d", p.splitk, nk_all); return -2; }
    if (nk_all % (p.splitk * WK)) { cutie_set_error("conv: splitk %d x %d K groups must divide the %d K tiles", p.splitk, WK, nk_all); return -2; }
    p.Kslice = p.Kpad / p.splitk;
    constexpr int lds = conv_lds_bytes<BM, BN, BK, WK>();
    static bool attr_set = false;                        // one flag per instantiation
    if (!attr_set) {
        if (lds > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WM, WN, BK, S, OCC, WK>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            cutie_set_error("conv: cannot raise the dynamic LDS limit to %d bytes", lds);
            return -2;
        }
        attr_set = true;
    }
    dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, p.splitk);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, BK, S, OCC, WK>), grid, dim3(WM * WN * WK * 64), lds, s, p);
    if (p.splitk > 1) {
        const long nq = (long)p.M * (p.ldp >> 3);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, p);
    }
    return (int)hipGetLastError();
}

// tile table (mirrored by cutie_amd/ops.py:TILES): id -> BM, BN, BK
int launch_conv(const cutie_op* op, hipStream_t s) {
    ConvParams p;
    p.x1 = (const bf16_t*)op->p[0]; p.x2 = (const bf16_t*)op->p[1]; p.w = (const bf16_t*)op->p[2];
    p.bias = (const float*)op->p[3]; p.res = (const bf16_t*)op->p[4]; p.y = (void*)op->p[5];
    const int32_t* i = op->i;
    p.B = i[0]; p.H = i[1]; p.W = i[2]; p.C1 = i[3]; p.C2 = i[4]; p.ldx1 = i[5]; p.ldx2 = i[6];
    p.OH = i[7]; p.OW = i[8]; p.Cout = i[9]; p.ldy = i[10]; p.KH = i[11]; p.KW = i[12];
    p.stride = i[13]; p.pad = i[14]; p.ldr = i[15]; p.Kpad = i[16];
    p.flags = op->flags; p.OHW = p.OH * p.OW; p.M = p.B * p.OHW; p.Cin = p.C1 + p.C2;
    p.splitk = i[19] > 1 ? i[19] : 1; p.part = (float*)op->p[6];
    p.ldp = (p.Cout + 7) & ~7;
    p.gap = (long long*)op->p[7]; p.zero = (unsigned long long*)op->p[8]; p.nzero = i[21];
    p.pf = (const unsigned char*)op->p[9]; p.pf_bytes = op->p[9] ? i[22] : 0;
    p.pf2 = (const unsigned char*)op->p[10]; p.pf2_bytes = op->p[10] ? i[23] : 0;
    p.res_grp_rows = p.res_grp_stride = 0;
    if ((p.flags & CUTIE_F_RES_BCAST) && op->f[0] > 0.f) {  // clips in lock step: objects in groups of f0, one residual map per group, f1 rows apart
        const int kg = (int)op->f[0];
        if (kg <= 0 || p.B % kg || op->f[1] < (float)p.OHW || op->f[1] != (float)(int)op->f[1]) { cutie_set_error("conv: residual groups of %d objects do not divide B = %d (or stride %g < OH*OW)", kg, p.B, (double)op->f[1]); return -2; }
        p.res_grp_rows = kg * p.OHW; p.res_grp_stride = (int)op->f[1];
    }
    if ((p.gap || p.zero) && (i[17] < 60 || i[17] >= 200 || ((p.flags & CUTIE_F_OUT_F32) && p.gap) || (p.Cout & 7) || (p.ldy & 7) || (p.res && (p.ldr & 7)))) {
        cutie_set_error("conv: GAP accumulation / zero job need an LDS-DMA tile (60..199), bf16 output, Cout %% 8 == 0 (tile %d)", i[17]);
        return -2;
    }
    if (p.splitk > 1 && (!p.part || (long)p.splitk * p.M * p.ldp > (long)i[20] * 1024)) {
        cutie_set_error("conv: split-K %d needs the fp32 partial scratch (p6, capacity i20 KiB-floats)", p.splitk);
        return -2;
    }
    if ((p.C1 & 7) || (p.C2 & 7) || (p.ldx1 & 7) || (p.C2 && (p.ldx2 & 7)) || (p.Kpad & 31) ||
        p.Kpad < p.KH * p.KW * p.Cin || p.M <= 0 || p.Cout <= 0) {
        cutie_set_error("conv: bad geometry C1=%d C2=%d ldx1=%d ldx2=%d Kpad=%d M=%d Cout=%d",
                        p.C1, p.C2, p.ldx1, p.ldx2, p.Kpad, p.M, p.Cout);
        return -2;
    }
    if (i[17] == 19) {                                   // dedicated Cout == 1 kernel
        const int LP = p.Cin >> 3;
        if (p.Cout != 1 || p.C2 != 0 || p.res || LP > 64 || (LP & (LP - 1)) || LP < 1) {
            cutie_set_error("conv cout1: needs Cout=1, single source, no residual, Cin/8 a power of two <= 64 (Cin=%d)", p.Cin);
            return -2;
        }
        const int ppb = 256 / LP;
        if (!(p.flags & CUTIE_F_PLAIN) && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.OH == p.H && p.OW == p.W && LP >= 8 && LP <= 32 && p.H * p.W >= 4096) {
            if (LP == 16 && !(p.flags & CUTIE_F_TILE_OFF)) {   // Cin = 128 (the decoder's logits head): the input tile in LDS
                constexpr int TR = 8;
                const int blocks = p.B * ((p.H + TR - 1) / TR) * ((p.W + 15) / 16);
                hipLaunchKernelGGL(conv_cout1_tile_kernel<TR>, dim3(blocks), dim3(256), 0, s, p);
                return (int)hipGetLastError();
            }
            constexpr int R = 4;                         // large maps: one thread per (column, 8-channel lane) walks R output rows
            const int blocks = p.B * ((p.H + R - 1) / R) * ((p.W + ppb - 1) / ppb);
            hipLaunchKernelGGL(conv_cout1_rows_kernel<R>, dim3(blocks), dim3(256), 0, s, p);
            return (int)hipGetLastError();
        }
        hipLaunchKernelGGL(conv_cout1_kernel, dim3((p.M + ppb - 1) / ppb), dim3(256), (size_t)p.KH * p.KW * LP * 16, s, p);
        return (int)hipGetLastError();
    }
    if (i[17] >= 60 && i[17] < 90) return launch_conv_dma(p, i[17], s);         // LDS-DMA tiles (conv_dma.hip)
    if (i[17] >= 100 && i[17] < 200) return launch_conv_pc(p, i[17], s);         // producer / consumer LDS-DMA tiles (conv_pc.hip)
    switch (i[17]) {
        case 0: return launch_cfg<128, 128, 2, 2, 32, 4, 2>(p, s);
        case 1: return launch_cfg<128, 64, 2, 2, 32, 4, 3>(p, s);
        case 2: return launch_cfg<64, 64, 2, 2, 32, 4, 4>(p, s);
        case 3: return launch_cfg<256, 16, 4, 1, 32, 4, 3>(p, s);
        case 4: return launch_cfg<64, 128, 2, 2, 32, 4, 3>(p, s);
        case 5: return launch_cfg<64, 64, 2, 2, 64, 4, 3>(p, s);
        case 6: return launch_cfg<64, 128, 2, 2, 64, 3, 2>(p, s);
        case 7: return launch_cfg<128, 128, 2, 2, 64, 2, 2>(p, s);
        case 8: return launch_cfg<64, 64, 2, 2, 128, 3, 2>(p, s);
        case 9: return launch_cfg<32, 64, 2, 2, 64, 4, 4>(p, s);
        case 10: return launch_cfg<128, 64, 2, 2, 64, 3, 2>(p, s);
        case 11: return launch_cfg<32, 64, 2, 2, 128, 3, 3>(p, s);
        case 12: return launch_cfg<32, 128, 2, 2, 64, 3, 2>(p, s);
        case 13: return launch_cfg<64, 64, 2, 4, 64, 4, 5>(p, s);        // 8 waves
        case 14: return launch_cfg<128, 64, 4, 2, 64, 4, 4>(p, s);
        case 15: return launch_cfg<64, 128, 2, 4, 64, 4, 4>(p, s);
        case 16: return launch_cfg<128, 128, 2, 4, 64, 3, 3>(p, s);
        case 17: return launch_cfg<64, 64, 2, 4, 128, 3, 4>(p, s);
        case 18: return launch_cfg<128, 128, 2, 4, 32, 4, 4>(p, s);
        // 20..: two (or four) K groups per block (intra-block split-K), 8 (16) waves
        case 20: return launch_cfg<64, 64, 2, 2, 64, 4, 3, 2>(p, s);
        case 21: return launch_cfg<64, 64, 2, 2, 128, 3, 2, 2>(p, s);
        case 22: return launch_cfg<32, 64, 2, 2, 128, 3, 3, 2>(p, s);
        case 23: return launch_cfg<128, 64, 2, 2, 64, 3, 2, 2>(p, s);
        case 24: return launch_cfg<64, 128, 2, 2, 64, 3, 2, 2>(p, s);
        case 25: return launch_cfg<128, 128, 2, 2, 64, 2, 2, 2>(p, s);
        case 26: return launch_cfg<32, 64, 2, 2, 64, 4, 4, 2>(p, s);
        case 27: return launch_cfg<64, 64, 2, 2, 64, 3, 3, 4>(p, s);
        case 28: return launch_cfg<32, 64, 2, 2, 64, 4, 4, 4>(p, s);
        // 40..: 3x3 convs with the input patch resident in LDS (small feature maps)
        case 29: return launch_cfg<64, 128, 2, 2, 32, 4, 3, 2>(p, s);
        case 30: return launch_cfg<64, 64, 2, 2, 32, 4, 4, 2>(p, s);
        default: cutie_set_error("conv: bad tile id %d", i[17]); return -2;
    }
}
