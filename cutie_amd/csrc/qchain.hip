// The query side of a QueryTransformerBlock (reference object_transformer.py:12-73, transformer_layers.py:12-118) in FOUR launches:
//   ATTN_Q2P  (masked cross attention queries <- pixels, q projection and LayerNorm in front, per-head output projection behind)
//   ATTN_SELF (LayerNorm + q|k|v projection + 16 x 16 attention + per-head output projection)
//   QFFN      (LayerNorm + linear1 + relu + linear2, split over slices of the hidden layer)
//   ATTN_P2Q  (k|v projection of the queries + cross attention pixels <- queries)
// instead of the seven of round 2 (three LINEAR launches in between).  What the attention.hip timeline showed (tools/attn_timeline.py,
// profiles/r03_qchain.md): these launches are bound by how many bytes ONE compute unit has to pull (about 35 GB/s per CU with a few
// waves in flight) and by dependent memory round trips, not by arithmetic.  Hence:
//   * every launch issues ALL its global loads at entry, in the order they are consumed (vmcnt retires in order: a late-needed big load
//     issued first delays an early-needed small one behind it);
//   * products that mix the blocks of a launch (the output projections over the 8 heads, linear2 over the hidden slices) are summed by
//     the PRODUCER into one fixed-point accumulator (int64, x 2^32, integer atomics: the sum does not depend on the order of arrival,
//     so replays stay bit-identical) -- the consumer reads 32 bytes per value-quad instead of 8 partial tensors;
//   * the 16 x 16 and 16 x 256-pixel attentions run on MFMA in the transposed form (a lane's accumulator column is "its" query or
//     pixel), with the free k-slot <-> index assignment of the second product chosen so that the first product's D layout IS the second
//     product's B operand (no LDS transposes, no shuffles beyond the 4-lane softmax reduction);
//   * V of ATTN_Q2P is fetched with 16-byte loads and transposed through a wave-private LDS tile (the 2-byte gathers of the round-2
//     kernel cost 16 load instructions per 32 pixels).
// The accumulators are cleared by QUERY_INIT at the start of the plan (flags&1: p9, i2).
#include "attention_common.h"

#define QACC_SCALE 4294967296.f
#define QACC_INV (1.f / 4294967296.f)

struct QIn {
    const float* x;                                      // [K*16, 256] fp32 rows
    const long long* acc;                                // fixed-point sum that belongs to the rows (x_eff = x + abias + acc / 2^32) or null
    const float* abias;                                  // the bias of the linear whose products acc holds (or null)
    const float* add;                                    // query embedding [K*16, 256] or null
    const float* ln_g; const float* ln_b;                // LayerNorm in front (null: none)
    float* ln_out;                                       // normalised rows, written once per object (null: not kept)
    float* x_out;                                        // x_eff, written once per object (null: not kept)
    const bf16_t* W; const float* bias;                  // the packed projection of this launch [N][256] bf16, bias [N]
    unsigned long long* tl;                              // diagnostic builds: stamp buffer (tools/attn_timeline.py)
    int prio;                                            // CUTIE_F_PRIO: the launch belongs to the frame's critical path -- s_setprio 1
};
struct QOut { const bf16_t* W; long long* acc; };        // output projection Wo [256][256] bf16; acc [K*16, 256] fixed point

// ATL(id): cycle stamp of every wave (lane 0) into slot id (0..13); slot 14 / 15 = the 100 MHz wall clock at stamp 0 / at the last stamp
#ifdef ATT_TIMELINE
#define ATL(ID) { if (in.tl && (threadIdx.x & 63) == 0) { unsigned long long* o_ = in.tl + ((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 16; \
                  o_[ID] = __builtin_readcyclecounter(); if ((ID) == 0) o_[14] = wall_clock64(); else o_[15] = wall_clock64(); } }
#else
#define ATL(ID)
#endif

// Block-wide barrier for LDS hand-overs.  __syncthreads() also drains EVERY outstanding global load of the wave (s_waitcnt vmcnt(0) in
// front of s_barrier): with all loads of a launch issued at entry, the first barrier then waits for the K / V prefetch or the weights
// of a later phase (measured: 6 us from entry to the first hand-over of ATTN_Q2P although it needed 2 KB).  Only the LDS / scalar
// counter is drained here; registers fed by global loads are waited for by the compiler where they are used.
#define QSYNC() { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8)); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

__device__ __forceinline__ void qacc_add(long long* a, float v) {
    atomicAdd(reinterpret_cast<unsigned long long*>(a), (unsigned long long)__float2ll_rn(v * QACC_SCALE));
}

// ---- the 16 rows of an object: loads (issue) and sum + LayerNorm + LDS copy (finish) ----------------------------------------------
// ACC / ADD / LN are compile-time: with run-time tests ("if (in.acc) load") the compiler folded each row's accumulator load and its
// use into ONE branch -- load, s_waitcnt vmcnt(0), convert -- i.e. a dependent memory round trip per row (4 per wave in QFFN), and a
// load inside a branch makes every later counted wait conservative (the ISA showed vmcnt(0) in front of ATTN_Q2P's mask pass, which
// then waited for the whole K / V prefetch).  Straight-line code keeps all loads in flight together.
template <int RPW> struct QRows { float4 x[RPW], add[RPW]; longlong2 a[RPW][2]; float4 ab, gg, bb; };

template <int NW, int RPW, bool ACC, bool ADD, bool LN>
__device__ __forceinline__ void qrows_issue(const QIn& in, int k, QRows<RPW>& R) {
    static_assert(NW * RPW == 16, "16 rows per object");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const long off = ((long)k * 16 + wave + j * NW) * 256 + lane * 4;
        R.x[j] = *reinterpret_cast<const float4*>(in.x + off);
        if (ACC) {
            const longlong2* ap = reinterpret_cast<const longlong2*>(in.acc + off);
            R.a[j][0] = ap[0]; R.a[j][1] = ap[1];
        }
        if (ADD) R.add[j] = *reinterpret_cast<const float4*>(in.add + off);
    }
    if (ACC) R.ab = *reinterpret_cast<const float4*>(in.abias + lane * 4);
    if (LN) { R.gg = *reinterpret_cast<const float4*>(in.ln_g + lane * 4); R.bb = *reinterpret_cast<const float4*>(in.ln_b + lane * 4); }
}

// xs_add: LN(x_eff) + add (or x_eff + add); xs_plain (nullable): LN(x_eff) (or x_eff).  Row pitch PROJ_XLD.
template <int NW, int RPW, bool ACC, bool ADD, bool LN>
__device__ __forceinline__ void qrows_finish(const QIn& in, int k, const QRows<RPW>& R, float* xs_add, float* xs_plain, bool writer) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int r = wave + j * NW;
        const long off = ((long)k * 16 + r) * 256 + lane * 4;
        float4 v = R.x[j];
        if (ACC) {
            v.x += R.ab.x + __ll2float_rn(R.a[j][0].x) * QACC_INV; v.y += R.ab.y + __ll2float_rn(R.a[j][0].y) * QACC_INV;
            v.z += R.ab.z + __ll2float_rn(R.a[j][1].x) * QACC_INV; v.w += R.ab.w + __ll2float_rn(R.a[j][1].y) * QACC_INV;
        }
        if (in.x_out && writer) *reinterpret_cast<float4*>(in.x_out + off) = v;
        if (LN) {
            const float mean = wave_sum_dpp((v.x + v.y) + (v.z + v.w)) * (1.f / 256.f);
            const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
            const float rstd = rsqrtf(wave_sum_dpp((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.f / 256.f) + 1e-5f);
            v.x = dx * rstd * R.gg.x + R.bb.x; v.y = dy * rstd * R.gg.y + R.bb.y; v.z = dz * rstd * R.gg.z + R.bb.z; v.w = dw * rstd * R.gg.w + R.bb.w;
            if (in.ln_out && writer) *reinterpret_cast<float4*>(in.ln_out + off) = v;
        }
        if (xs_plain) *reinterpret_cast<float4*>(xs_plain + r * PROJ_XLD + lane * 4) = v;
        if (ADD) { v.x += R.add[j].x; v.y += R.add[j].y; v.z += R.add[j].z; v.w += R.add[j].w; }
        *reinterpret_cast<float4*>(xs_add + r * PROJ_XLD + lane * 4) = v;
    }
}

__device__ __forceinline__ bf16x8 as_frag(const proj_u4& u) { return __builtin_bit_cast(bf16x8, u); }
__device__ __forceinline__ f32x4 mfma3(const proj_u4& ahi, const proj_u4& alo, const proj_u4& bhi, const proj_u4& blo, f32x4 acc) {
    // (ahi + alo) . (bhi + blo) without the lo . lo term: fp32-class products from bf16 MFMA
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(alo), as_frag(bhi), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(ahi), as_frag(blo), acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(ahi), as_frag(bhi), acc, 0, 0, 0);
}
// 8 (or 4 + zero padding) fp32 values -> hi / lo bf16 fragments
__device__ __forceinline__ void split8(const float* v, proj_u4& hi, proj_u4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { uint32_t h_, l_; split_bf2(v[2 * j], v[2 * j + 1], h_, l_); hi[j] = h_; lo[j] = l_; }
}
__device__ __forceinline__ void split4(const float* v, proj_u4& hi, proj_u4& lo) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { uint32_t h_, l_; split_bf2(v[2 * j], v[2 * j + 1], h_, l_); hi[j] = h_; lo[j] = l_; }
    hi[2] = hi[3] = lo[2] = lo[3] = 0u;
}

// =====================================================================================================================================
// ATTN_Q2P, chain form.  grid (heads, K), block 1024 = 16 waves; wave w owns the 32-pixel chunks w, w + 16, ...
// S^T[pixel][query] = K[pixel][dim] . Q^T[dim][query];  O^T[dim][query] = V^T[dim][pixel] . P^T[pixel][query]  (as attention.hip's kernel)
// =====================================================================================================================================
#define Q2C_PF 4
#define Q2C_VLD 40                                       // bf16 pitch of the V tile (80 bytes: the four pixel groups of a read hit disjoint banks)
struct Q2CChunk { q2p_u32x4 k[2], v[2]; };
__device__ __forceinline__ void q2c_load(Q2CChunk& L, const bf16_t* __restrict__ kvb, int p0, int HW, int ldkv, int voff, int lane) {
    const int c16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int p = min(p0 + t * 16 + c16, HW - 1);
        L.k[t] = *reinterpret_cast<const q2p_u32x4*>(kvb + (long)p * ldkv + 8 * g);
    }
    const bf16_t* vp = kvb + (long)min(p0 + (lane >> 1), HW - 1) * ldkv + voff + (lane & 1) * 16;
    L.v[0] = *reinterpret_cast<const q2p_u32x4*>(vp);
    L.v[1] = *reinterpret_cast<const q2p_u32x4*>(vp + 8);
}

// Foreground flag of object k at one pixel (object_transformer.py:179-205): L_k >= max(L_bg, max_j L_j) with L = logit(clamp(p, 1e-7,
// 1 - 1e-7)).  logit is strictly increasing, so the clamped probabilities are compared directly -- no log, and the sigmoid through
// v_exp / v_rcp: the exact form cost ~200 instructions per pixel and object, x 1620 pixels in EVERY head's block (3 us of a 20 us launch,
// profiles/r03_qchain.md).  Decisions differ from the exact form only where two probabilities agree to fp32 rounding.
template <int KT>
__device__ __forceinline__ bool aux_fg_vals(const float* v, int K, int k) {
    float bg = 1.f, mx = 0.f, mine_l = 0.f;
#pragma unroll
    for (int j = 0; j < KT; ++j) {                         // slots >= K hold a copy of plane K - 1: selected away, no branches
        const float pr = __frcp_rn(1.f + __expf(-v[j]));
        bg *= j < K ? (1.f - pr) : 1.f;
        mx = fmaxf(mx, j < K ? pr : 0.f);
        mine_l = j == k ? v[j] : mine_l;
    }
    const float lo = 1e-7f, hi = 1.f - 1e-7f;
    const float mine = fminf(fmaxf(__frcp_rn(1.f + __expf(-mine_l)), lo), hi);
    return mine >= fminf(fmaxf(mx, lo), hi) && mine >= fminf(fmaxf(bg, lo), hi);
}
__device__ __forceinline__ bool aux_fg_late(const float* __restrict__ lg, int K, int HW, int k, int p) {      // K > 8 or pixels >= 2048
    float bg = 1.f, mx = 0.f, mine = 0.f;
    for (int j = 0; j < K; ++j) {
        const float pr = __frcp_rn(1.f + __expf(-lg[(long)j * HW + p]));
        bg *= (1.f - pr);
        mx = fmaxf(mx, pr);
        mine = j == k ? pr : mine;
    }
    const float lo = 1e-7f, hi = 1.f - 1e-7f;
    mine = fminf(fmaxf(mine, lo), hi);
    return mine >= fminf(fmaxf(mx, lo), hi) && mine >= fminf(fmaxf(bg, lo), hi);
}

// qpre != null (flags&16): q of this launch was projected by the ATTN_P2Q launch of the previous block (its extra blocks, see below) --
// the 80 KB of rows + weights and three block-wide barriers in front of the pixel loop are gone.
// NW waves per block (8: half the per-wave prologue instructions of the 16-wave form contend for the SIMDs; the pixel loop is bound by
// the K / V bytes of the block either way).  KT: the mask logits of this thread's pixels are fetched at entry for K <= KT objects (0: late).
template <bool QPRE, bool ACC, int KT, int NW>
__global__ __launch_bounds__(NW * 64) void q2p_chain_kernel(QIn in, QOut out, const bf16_t* __restrict__ kv, const float* __restrict__ lg,
                                                           int HW, int HWp, int ldkv, int voff, int hstride, const float* __restrict__ qpre, int Kg) {
    if (in.prio) __builtin_amdgcn_s_setprio(1);            // CUTIE_F_PRIO: a launch of the frame's critical path (the plan decides, ops.OpList.prio)
    constexpr int Q = 16, NT = NW * 64;
    constexpr bool EARLY = KT > 0;
    constexpr int PPT = 2048 / NT;                         // pixels per thread with logits fetched at entry
    constexpr int UPW = 16 / NW;                           // (column tile, k step) units of the q projection per wave; rows per wave; Wo tiles per wave
    constexpr int SO_BYTES = NW * 16 * 33 * 4, ST_BYTES = 16 * PROJ_XLD * 4 + NW * 64 * 16;
    __shared__ __attribute__((aligned(16))) unsigned char sbuf[SO_BYTES > ST_BYTES ? SO_BYTES : ST_BYTES];
    float (*sO)[16][33] = reinterpret_cast<float (*)[16][33]>(sbuf);       // [wave][query][dim]; the projection staging aliases it
    __shared__ float sM[NW][16], sL[NW][16];
    __shared__ float sQ[16][33];                           // this head's 32 query columns, scaled; later the head's output
    __shared__ int sCnt;
    extern __shared__ uint8_t dynlds[];                    // [HWp foreground flags][NW waves x 32 pixels x 80 B of V]
    // Kg: objects per clip (clips in lock step: the grid holds gridDim.y / Kg clips).  The foreground mask is decided among the Kg objects of
    // object k's clip (planes lg[clip * Kg ..]); everything else of the launch is per object.
    const int hh = blockIdx.x, k = blockIdx.y, K = Kg, kin = k % Kg;
    lg += (long)(k - kin) * HW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c16 = lane & 15, g = lane >> 4;
    uint8_t* sFg = dynlds;
    bf16_t* sV = reinterpret_cast<bf16_t*>(dynlds + HWp) + wave * 32 * Q2C_VLD;
    const float scale = rsqrtf(32.f);
    ATL(0)
    // ---- every global load of the launch, in the order of use ----
    QRows<UPW> R;
    float4 qp0, qp1;
    if (QPRE) {
        const float* qr = qpre + ((long)k * 16 + c16) * 256 + hh * 32 + 8 * g;
        qp0 = *reinterpret_cast<const float4*>(qr); qp1 = *reinterpret_cast<const float4*>(qr + 4);
    } else {
        qrows_issue<NW, UPW, ACC, true, true>(in, k, R);
    }
    float lgv[EARLY ? PPT : 1][EARLY ? KT : 1];
    if (EARLY) {                                           // unconditional loads (plane index clamped): no branch around a load
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int p = min((int)threadIdx.x + NT * i, HW - 1);
#pragma unroll
            for (int j = 0; j < KT; ++j) lgv[i][j] = lg[(long)min(j, K - 1) * HW + p];
        }
    }
    const int tile = wave & 1, ks0 = (wave >> 1) * UPW;
    proj_u4 wq[UPW], wo[UPW];
    if (!QPRE) proj16_load<UPW>(in.W, hh * 32 + tile * 16, ks0, wq);
#pragma unroll
    for (int t = 0; t < UPW; ++t) proj16_load<1>(out.W, (wave * UPW + t) * 16, hh, &wo[t]);   // Wo rows of this wave's column tiles, the columns of this head
    const bf16_t* kvb = kv + (long)k * HW * ldkv + hh * hstride;
    const int nchunk = (HW + 31) >> 5;
    if (threadIdx.x == 0) sCnt = 0;
    // The K / V prefetch goes out only after EVERY wave has issued its small loads: the CU's memory pipeline serves requests in issue
    // order across waves.  The barrier only waits for the issue, not for the data.
    QSYNC();
    Q2CChunk pf[Q2C_PF];                                   // unconditional (chunk index clamped): a load inside a branch costs the counted waits
#pragma unroll
    for (int j = 0; j < Q2C_PF; ++j) q2c_load(pf[j], kvb, min(wave + NW * j, nchunk - 1) * 32, HW, ldkv, voff, lane);
    // ---- q = (LN(x_eff) + emb) . Wq[head]^T + b: 2 column tiles x 8 k-steps over the waves, summed through LDS (staging aliases sO) ----
    q2p_frag qh, ql;
    if (!QPRE) {
        float* xs = reinterpret_cast<float*>(sbuf);
        f32x4* red = reinterpret_cast<f32x4*>(xs + 16 * PROJ_XLD);
        qrows_finish<NW, UPW, ACC, true, true>(in, k, R, xs, nullptr, hh == 0);
        ATL(1)
        QSYNC();
        red[wave * 64 + lane] = proj16_mma<UPW>(xs, ks0, wq);
        QSYNC();
        if (threadIdx.x < 128) {
            const int t = threadIdx.x >> 6;
            f32x4 a = red[t * 64 + lane];
#pragma unroll
            for (int j = 1; j < NW / 2; ++j) { const f32x4 b = red[(t + 2 * j) * 64 + lane]; a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3]; }
            const float bv = in.bias[hh * 32 + t * 16 + c16];
#pragma unroll
            for (int r = 0; r < 4; ++r) sQ[4 * g + r][t * 16 + c16] = (a[r] + bv) * scale;
        }
    }
    {
        // ---- foreground flags of object k (AUX_MASK fused, object_transformer.py:179-205) while the reduction settles ----
        int cnt = 0;
        if (EARLY) {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const int p = threadIdx.x + NT * i;
                if (p < HW) { const bool f = aux_fg_vals<EARLY ? KT : 1>(lgv[i], K, kin); sFg[p] = f ? 1 : 0; cnt += f ? 1 : 0; }
            }
        }
        for (int p = threadIdx.x + (EARLY ? 2048 : 0); p < HW; p += NT) {
            const bool f = aux_fg_late(lg, K, HW, kin, p);
            sFg[p] = f ? 1 : 0;
            cnt += f ? 1 : 0;
        }
        cnt = wave_sum_i32(cnt);
        if (lane == 0 && cnt) atomicAdd(&sCnt, cnt);
        QSYNC();                                           // sQ, sFg, sCnt complete; xs / red (= sO) are free again
        ATL(2)
        if (QPRE) {
            const float qv[8] = {qp0.x, qp0.y, qp0.z, qp0.w, qp1.x, qp1.y, qp1.z, qp1.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { uint32_t h_, l_; split_bf2(qv[2 * j], qv[2 * j + 1], h_, l_); qh.u[j] = h_; ql.u[j] = l_; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { uint32_t h_, l_; split_bf2(sQ[c16][8 * g + 2 * j], sQ[c16][8 * g + 2 * j + 1], h_, l_); qh.u[j] = h_; ql.u[j] = l_; }
        }
    }
    const int n_fg = sCnt;
    const bool is_fg_query = c16 < Q / 2;                  // queries 0..7 attend foreground only
    const bool masked = is_fg_query ? (n_fg != 0) : (n_fg != HW);          // row fully blocked -> unblocked (object_transformer.py:203)
    float m = -INFINITY, l = 0.f;
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    auto chunk = [&](const Q2CChunk& L, int p0) {
        // V tile -> LDS (row = pixel), read back transposed: lane (dim c16, group g) gets its 8 pixels of dims c16 and c16 + 16
        bf16_t* vrow = sV + (lane >> 1) * Q2C_VLD + (lane & 1) * 16;
        *reinterpret_cast<q2p_u32x4*>(vrow) = L.v[0];
        *reinterpret_cast<q2p_u32x4*>(vrow + 8) = L.v[1];
        uint8_t fgv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) fgv[j] = sFg[min(p0 + (j >> 2) * 16 + 4 * g + (j & 3), HW - 1)];
        f32x4 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            q2p_frag ka; ka.u = L.k[t];
            z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka.b, qh.b, z, 0, 0, 0);
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka.b, ql.b, z, 0, 0, 0);
        }
        float sv[8], tm = -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = p0 + (j >> 2) * 16 + 4 * g + (j & 3);
            const bool ok = p < HW && (!masked || ((fgv[j] != 0) == is_fg_query));
            sv[j] = ok ? s[j >> 2][j & 3] : -INFINITY;
            tm = fmaxf(tm, sv[j]);
        }
        tm = rows_max(tm);
        const float mn = fmaxf(m, tm);
        const float mref = (mn == -INFINITY) ? 0.f : mn;
        const float alpha = __expf(m - mref);
        float pe[8], ps = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { pe[j] = (sv[j] == -INFINITY) ? 0.f : __expf(sv[j] - mref); ps += pe[j]; }
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        q2p_frag ph, pl, va0, va1;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            uint32_t h_, l_; split_bf2(pe[2 * jj], pe[2 * jj + 1], h_, l_); ph.u[jj] = h_; pl.u[jj] = l_;
            const int ja = 2 * jj, jb = 2 * jj + 1;
            const bf16_t* ra = sV + ((ja >> 2) * 16 + 4 * g + (ja & 3)) * Q2C_VLD + c16;
            const bf16_t* rb = sV + ((jb >> 2) * 16 + 4 * g + (jb & 3)) * Q2C_VLD + c16;
            va0.u[jj] = (uint32_t)ra[0] | ((uint32_t)rb[0] << 16);
            va1.u[jj] = (uint32_t)ra[16] | ((uint32_t)rb[16] << 16);
        }
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va0.b, ph.b, o0, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va0.b, pl.b, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va1.b, ph.b, o1, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va1.b, pl.b, o1, 0, 0, 0);
    };
#pragma unroll
    for (int j = 0; j < Q2C_PF; ++j)
        if (wave + NW * j < nchunk) chunk(pf[j], (wave + NW * j) * 32);
    for (int ch = wave + NW * Q2C_PF; ch < nchunk; ch += NW) {
        Q2CChunk L;
        q2c_load(L, kvb, ch * 32, HW, ldkv, voff, lane);
        chunk(L, ch * 32);
    }
    ATL(3)
    l = rows_sum(l);
    if (g == 0) { sM[wave][c16] = m; sL[wave][c16] = l; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { sO[wave][c16][4 * g + r] = o0[r]; sO[wave][c16][16 + 4 * g + r] = o1[r]; }
    QSYNC();
    ATL(4)
    for (int e = threadIdx.x; e < 512; e += NT) {          // (query i, dim d): merge the waves
        const int i = e >> 5, d = e & 31;
        float Mg = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) Mg = fmaxf(Mg, sM[w][i]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float f = (sM[w][i] == -INFINITY) ? 0.f : __expf(sM[w][i] - Mg);
            num += sO[w][i][d] * f;
            den += sL[w][i] * f;
        }
        sQ[i][d] = num / den;
    }
    QSYNC();
    ATL(5)
    {   // per-head output projection: o (16 x 32) . Wo[:, 32 hh ..]^T -- UPW 16-column tiles per wave, summed over the heads
        proj_u4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) { uint32_t h_, l_; split_bf2(sQ[c16][8 * g + 2 * j], sQ[c16][8 * g + 2 * j + 1], h_, l_); hi[j] = h_; lo[j] = l_; }
#pragma unroll
        for (int t = 0; t < UPW; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(lo), as_frag(wo[t]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(hi), as_frag(wo[t]), acc, 0, 0, 0);
            long long* ap = out.acc + ((long)k * 16 + 4 * g) * 256 + (wave * UPW + t) * 16 + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) qacc_add(ap + r * 256, acc[r]);
        }
    }
    ATL(6)
}

// =====================================================================================================================================
// ATTN_SELF, chain form.  grid (heads, K), block 512.  Waves (kq = w & 3, th = w >> 2): column tiles 3 th .. 3 th + 2 of [q | k | v] (two
// tiles each) over k-steps 2 kq, 2 kq + 1; summed through LDS.  Every wave then runs the 16 x 16 attention of the head on MFMA (a few
// instructions: cheaper than handing one wave's result around) and projects it onto ITS two 16-column tiles of Wo.
// =====================================================================================================================================
__global__ __launch_bounds__(512) void self_chain_kernel(QIn in, QOut out) {
    if (in.prio) __builtin_amdgcn_s_setprio(1);            // CUTIE_F_PRIO: a launch of the frame's critical path (the plan decides, ops.OpList.prio)
    __shared__ float sX[2][16 * PROJ_XLD];                 // [LN(x)+emb | LN(x)]
    __shared__ f32x4 sRed[4][6][64];
    __shared__ float sP[3][16][33];                        // q (scaled) | k | v of this head
    const int hh = blockIdx.x, k = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int kq = wave & 3, th = wave >> 2;
    const float scale = rsqrtf(32.f);
    ATL(0)
    QRows<2> R;
    qrows_issue<8, 2, true, true, true>(in, k, R);
    proj_u4 wv[3][2];                                      // tiles 0,1: q  2,3: k  4,5: v (weight rows (t/2)*256 + head*32 + (t&1)*16)
#pragma unroll
    for (int t = 0; t < 3; ++t) { const int tl_ = th * 3 + t; proj16_load<2>(in.W, (tl_ >> 1) * 256 + hh * 32 + (tl_ & 1) * 16, 2 * kq, wv[t]); }
    // Wo for the out-projection, k-slots in the order the attention leaves its output in: dims 4g..4g+3 and 16+4g..16+4g+3
    proj_u4 wo[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const bf16_t* wr = out.W + (long)((wave * 2 + t) * 16 + c) * 256 + hh * 32 + 4 * g;
        const uint2 a = *reinterpret_cast<const uint2*>(wr), b = *reinterpret_cast<const uint2*>(wr + 16);
        wo[t][0] = a.x; wo[t][1] = a.y; wo[t][2] = b.x; wo[t][3] = b.y;
    }
    qrows_finish<8, 2, true, true, true>(in, k, R, sX[0], sX[1], hh == 0);
    ATL(1)
    QSYNC();
#pragma unroll
    for (int t = 0; t < 3; ++t) sRed[kq][th * 3 + t][lane] = proj16_mma<2>(sX[th * 3 + t < 4 ? 0 : 1], 2 * kq, wv[t]);
    QSYNC();
    ATL(2)
    if (threadIdx.x < 6 * 64) {
        const int t = threadIdx.x >> 6;
        f32x4 a = sRed[0][t][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) { const f32x4 b = sRed[w][t][lane]; a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3]; }
        const int col = (t & 1) * 16 + c;
        const float bv = in.bias[(t >> 1) * 256 + hh * 32 + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) sP[t >> 1][4 * g + r][col] = (a[r] + bv) * (t < 2 ? scale : 1.f);
    }
    QSYNC();
    ATL(3)
    // S^T[key][query] = K . Q^T: A = K (lane: key c, dims 8g..), B = Q^T (lane: query c, dims 8g..); D: keys 4g..4g+3 of query c
    float kf[8], qf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { kf[j] = sP[1][c][8 * g + j]; qf[j] = sP[0][c][8 * g + j]; }
    proj_u4 khi, klo, qhi, qlo;
    split8(kf, khi, klo);
    split8(qf, qhi, qlo);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    s = mfma3(khi, klo, qhi, qlo, s);
    const float mx = rows_max(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
    float pe[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pe[r] = __expf(s[r] - mx);
    const float inv = 1.f / rows_sum((pe[0] + pe[1]) + (pe[2] + pe[3]));
#pragma unroll
    for (int r = 0; r < 4; ++r) pe[r] *= inv;
    // O^T[dim][query] = V^T . P^T over the 16 keys: k-slot j < 4 of group g <-> key 4g + j (the D layout of S^T), slots 4..7 empty
    proj_u4 phi, plo;
    split4(pe, phi, plo);
    float ot[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        float vf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) vf[j] = sP[2][4 * g + j][dt * 16 + c];
        proj_u4 vhi, vlo;
        split4(vf, vhi, vlo);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        o = mfma3(vhi, vlo, phi, plo, o);
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[dt][r] = o[r];      // dim dt*16 + 4g + r of query c
    }
    ATL(4)
    // out-projection: A = o (lane: query c; k-slots = the 8 dims this lane holds), B = Wo in the same slot order
    {
        const float of[8] = {ot[0][0], ot[0][1], ot[0][2], ot[0][3], ot[1][0], ot[1][1], ot[1][2], ot[1][3]};
        proj_u4 ohi, olo;
        split8(of, ohi, olo);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(olo), as_frag(wo[t]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(ohi), as_frag(wo[t]), acc, 0, 0, 0);
            long long* ap = out.acc + ((long)k * 16 + 4 * g) * 256 + (wave * 2 + t) * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) qacc_add(ap + r * 256, acc[r]);
        }
    }
    ATL(5)
}

// =====================================================================================================================================
// QFFN: x_eff -> LayerNorm -> linear1 -> relu -> linear2 (transformer_layers.py:101-118), split over FF / HS slices of the hidden layer.
// grid (FF / HS, K), block 256.  Block (s, k): hidden columns HS s .. HS s + HS - 1 of object k's 16 rows, and their contribution
// h_s . W2[:, HS s ..]^T to all 256 outputs, added to the fixed-point accumulator.  linear2's bias and the residual x_eff (written to
// x_out by slice 0) are added by the consumer.  HS = 64: 96 blocks at K = 3, 112 KB pulled per block.
// =====================================================================================================================================
struct QFfn { const bf16_t* W1; const float* b1; const bf16_t* W2; long long* acc; int FF; };
template <int HS>
__global__ __launch_bounds__(256) void qffn_kernel(QIn in, QFfn a) {
    if (in.prio) __builtin_amdgcn_s_setprio(1);            // CUTIE_F_PRIO: a launch of the frame's critical path (the plan decides, ops.OpList.prio)
    constexpr int T1 = HS / 64;                            // hidden 16-column tiles per wave
    constexpr int KS2 = HS / 32;                           // k steps of the second product
    constexpr int HLD = HS + 4;
    __shared__ float sX[16 * PROJ_XLD];
    __shared__ float sH[16 * HLD];
    const int sl = blockIdx.x, k = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    ATL(0)
    QRows<4> R;
    qrows_issue<4, 4, true, false, true>(in, k, R);
    proj_u4 w1[T1][8], w2[4][KS2];
#pragma unroll
    for (int t = 0; t < T1; ++t) proj16_load<8>(a.W1, sl * HS + (wave * T1 + t) * 16, 0, w1[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) proj16_load<KS2>(a.W2, (wave * 4 + t) * 16, sl * KS2, w2[t], a.FF);
    qrows_finish<4, 4, true, false, true>(in, k, R, sX, nullptr, sl == 0);
    ATL(1)
    QSYNC();
#pragma unroll
    for (int t = 0; t < T1; ++t) {
        const f32x4 acc = proj16_mma<8>(sX, 0, w1[t]);
        const int col = (wave * T1 + t) * 16 + c;
        const float bv = a.b1[sl * HS + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) sH[(4 * g + r) * HLD + col] = fmaxf(acc[r] + bv, 0.f);
    }
    ATL(2)
    QSYNC();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f32x4 acc = proj16_mma<KS2>(sH, 0, w2[t], HLD);
        long long* ap = a.acc + ((long)k * 16 + 4 * g) * 256 + (wave * 4 + t) * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) qacc_add(ap + r * 256, acc[r]);
    }
    ATL(3)
}

// =====================================================================================================================================
// ATTN_P2Q, chain form.  grid (ceil(HW / 256), heads, K), block 256: wave w projects one 16-column tile of [k | v] of the head's 16
// queries (k from x_eff + emb, v from x_eff), then attends its 64 pixels on MFMA:
//     S^T[query][pixel] = K[query][dim] . Q^T[dim][pixel]   (B = the pixel rows as they lie in memory: one 16-byte load per lane)
//     O^T[dim][pixel]   = V^T[dim][query] . P^T[query][pixel]
// =====================================================================================================================================
// Extra blocks (flags&16, blockIdx.x == gridDim.x - 1): the q projection of the NEXT transformer block's ATTN_Q2P for (head, object) --
// its input rows are the very rows staged here (x_eff), normalised with the next block's read_from_pixel.norm: q_out = ((LN(x_eff) +
// emb) Wq^T + b) / sqrt(32), xn_out = LN(x_eff) (the residual of that attention).  24 blocks next to 168: free.
struct NextQ { const float* ln_g; const float* ln_b; const bf16_t* W; const float* bias; float* q_out; float* xn_out; };
__global__ __launch_bounds__(256) void p2q_chain_kernel(QIn in, const bf16_t* __restrict__ q, bf16_t* __restrict__ y, int HW, int ldq, NextQ nq) {
    if (in.prio) __builtin_amdgcn_s_setprio(1);            // CUTIE_F_PRIO: a launch of the frame's critical path (the plan decides, ops.OpList.prio)
    constexpr int C = 256;
    __shared__ float sX[2][16 * PROJ_XLD];
    __shared__ float ks[16][36], vs[16][36];
    const int hh = blockIdx.y, k = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int isv = wave >> 1, col = (wave & 1) * 16;
    const float scale = rsqrtf(32.f);
    ATL(0)
    if (nq.W && blockIdx.x == gridDim.x - 1) {
        QIn in2 = in;
        in2.ln_g = nq.ln_g; in2.ln_b = nq.ln_b; in2.ln_out = nq.xn_out;
        QRows<4> R2;
        qrows_issue<4, 4, true, true, true>(in2, k, R2);
        proj_u4 wq[4];                                     // wave: column tile (w & 1), k steps 4 (w >> 1) ..
        proj16_load<4>(nq.W, hh * 32 + col, 4 * isv, wq);
        qrows_finish<4, 4, true, true, true>(in2, k, R2, sX[0], nullptr, hh == 0);
        QSYNC();
        f32x4* red = reinterpret_cast<f32x4*>(sX[1]);
        red[wave * 64 + lane] = proj16_mma<4>(sX[0], 4 * isv, wq);
        QSYNC();
        if (wave < 2) {
            const f32x4 a = red[wave * 64 + lane], b = red[(wave + 2) * 64 + lane];
            const float bv = nq.bias[hh * 32 + col + c];
#pragma unroll
            for (int r = 0; r < 4; ++r) nq.q_out[((long)k * 16 + 4 * g + r) * C + hh * 32 + col + c] = (a[r] + b[r] + bv) * scale;
        }
        return;
    }
    QRows<4> R;
    qrows_issue<4, 4, true, true, false>(in, k, R);
    proj_u4 wv[8];
    proj16_load<8>(in.W, isv * C + hh * 32 + col, 0, wv);
    const int pbase = blockIdx.x * 256 + wave * 64;
    proj_u4 qv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int p = min(pbase + t * 16 + c, HW - 1);
        qv[t] = *reinterpret_cast<const proj_u4*>(q + ((long)k * HW + p) * ldq + hh * 32 + 8 * g);
    }
    qrows_finish<4, 4, true, true, false>(in, k, R, sX[0], sX[1], false);
    ATL(1)
    QSYNC();
    {
        const f32x4 a = proj16_mma<8>(sX[isv], 0, wv);
        const float bv = in.bias[isv * C + hh * 32 + col + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (isv) vs[4 * g + r][col + c] = a[r] + bv;
            else ks[4 * g + r][col + c] = (a[r] + bv) * scale;
        }
    }
    QSYNC();
    ATL(2)
    if (pbase >= HW) return;
    float kf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) kf[j] = ks[c][8 * g + j];
    proj_u4 khi, klo, vhi[2], vlo[2];
    split8(kf, khi, klo);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        float vf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) vf[j] = vs[4 * g + j][dt * 16 + c];
        split4(vf, vhi[dt], vlo[dt]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int p = pbase + t * 16 + c;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(klo), as_frag(qv[t]), s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(khi), as_frag(qv[t]), s, 0, 0, 0);
        const float mx = rows_max(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
        float pe[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pe[r] = __expf(s[r] - mx);
        const float inv = 1.f / rows_sum((pe[0] + pe[1]) + (pe[2] + pe[3]));
#pragma unroll
        for (int r = 0; r < 4; ++r) pe[r] *= inv;
        proj_u4 phi, plo;
        split4(pe, phi, plo);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            o = mfma3(vhi[dt], vlo[dt], phi, plo, o);      // dims dt*16 + 4g .. + 3 of pixel c
            if (p < HW)
                *reinterpret_cast<uint2*>(y + ((long)k * HW + p) * C + hh * 32 + dt * 16 + 4 * g) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
        }
    }
    ATL(3)
}

#ifdef CUTIE_DIAG     // a measured-and-lost variant: only in the diagnostic library (make DIAG=1), never in the product .so
// =====================================================================================================================================
// ATTN_P2Q + the output projection of read_from_query + the residual (flags&32): pixel = pixel + Wo . attn + bo in the SAME launch -- the
// 1x1 conv behind ATTN_P2Q (6.6 us + a launch boundary, three times per frame) is gone, and so is the bf16 round trip of the attention
// output through HBM.  grid (ceil(HW / 64) [+ 8 with flags&16], K), block 512: wave h IS head h for the block's 64 pixels --
//   1. the 16 rows of the object are staged once (8 waves x 2 rows); wave h projects ITS head's k (2 column tiles, from x_eff + emb) and v
//      (2 tiles, from x_eff): per tile the MFMA sequence of p2q_chain_kernel (lo, hi per 32-wide k step), so k / v are bit-identical;
//   2. the attention of p2q_chain_kernel on four 16-pixel tiles; O^T (dims x pixels) goes to an LDS tile [64 pixels][256] in bf16 --
//      the value ATTN_P2Q stores;
//   3. Y^T[out][pixel] = Wo[out][:] . O[pixel][:]: wave h computes output channels 32h .. 32h + 31 of all 64 pixels, eight 32-wide k steps
//      in ascending order into one accumulator per output (the 'stream' K order of the conv kernels, ops.korder_class), then + bias,
//      + residual, round -- the conv epilogue's order: bit-identical to ATTN_P2Q followed by the conv on a stream tile.
// MEASURED AND NOT USED BY THE FRAME (plans.P2Q_OUT = 0): a block pulls 64 KB of rows + 256 KB of Wkv + 128 KB of Wo + 64 KB of pixels
// through ONE compute unit -- at the ~25 GB/s a CU sustains on fragment-shaped loads that is 21.6 us (rows staged after 7 us, k / v projected
// after 14 us; tools/attn_timeline.py) against 7.0 us + 4.8 us + one launch boundary for the head-split kernel and the conv: frame -1.8 %.
// Kept as the tested reference point of "a fusion that concentrates weights in few workgroups loses" (profiles/r04_frame_chain.md section 6).
// =====================================================================================================================================
struct P2QOut { const bf16_t* Wo; const float* bo; const bf16_t* res; };   // Wo [256][256] (out, in); res / y: [K, HW, 256]
#define P2O_LD 264                                       // bf16 pitch of the attention tile: the 16 rows of a fragment read start 4 banks apart
#define P2O_LDS_BYTES (2 * 16 * PROJ_XLD * 4 + 8 * 2 * 16 * 36 * 4 + 64 * P2O_LD * 2)
__global__ __launch_bounds__(512) void p2q_out_kernel(QIn in, const bf16_t* __restrict__ q, bf16_t* __restrict__ y, int HW, int ldq, NextQ nq, P2QOut po, int ntiles) {
    constexpr int C = 256;
    extern __shared__ uint8_t dynlds[];
    float* const sX0 = reinterpret_cast<float*>(dynlds);
    float* const sX1 = sX0 + 16 * PROJ_XLD;
    float (*sKV)[16][36] = reinterpret_cast<float (*)[16][36]>(dynlds + 2 * 16 * PROJ_XLD * 4);        // [head * 2 + isv][query][dim]
    bf16_t* const sO = reinterpret_cast<bf16_t*>(dynlds + 2 * 16 * PROJ_XLD * 4 + 8 * 2 * 16 * 36 * 4);
    const int k = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const float scale = rsqrtf(32.f);
    ATL(0)
    if ((int)blockIdx.x >= ntiles) {                       // the next block's q projection for head blockIdx.x - ntiles (see p2q_chain_kernel)
        const int hh = blockIdx.x - ntiles, isv = (wave >> 1) & 1, col = (wave & 1) * 16;
        QIn in2 = in;
        in2.ln_g = nq.ln_g; in2.ln_b = nq.ln_b; in2.ln_out = nq.xn_out;
        QRows<2> R2;
        qrows_issue<8, 2, true, true, true>(in2, k, R2);
        proj_u4 wq[4];
        if (wave < 4) proj16_load<4>(nq.W, hh * 32 + col, 4 * isv, wq);
        qrows_finish<8, 2, true, true, true>(in2, k, R2, sX0, nullptr, hh == 0);
        QSYNC();
        f32x4* red = reinterpret_cast<f32x4*>(sX1);
        if (wave < 4) red[wave * 64 + lane] = proj16_mma<4>(sX0, 4 * isv, wq);
        QSYNC();
        if (wave < 2) {
            const f32x4 a = red[wave * 64 + lane], b = red[(wave + 2) * 64 + lane];
            const float bv = nq.bias[hh * 32 + col + c];
#pragma unroll
            for (int r = 0; r < 4; ++r) nq.q_out[((long)k * 16 + 4 * g + r) * C + hh * 32 + col + c] = (a[r] + b[r] + bv) * scale;
        }
        return;
    }
    const int hh = wave;
    QRows<2> R;
    qrows_issue<8, 2, true, true, false>(in, k, R);
    proj_u4 wk[2][8], wv[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t) proj16_load<8>(in.W, hh * 32 + t * 16, 0, wk[t]);
#pragma unroll
    for (int t = 0; t < 2; ++t) proj16_load<8>(in.W, C + hh * 32 + t * 16, 0, wv[t]);
    const int pbase = blockIdx.x * 64;
    proj_u4 qv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int p = min(pbase + t * 16 + c, HW - 1);
        qv[t] = *reinterpret_cast<const proj_u4*>(q + ((long)k * HW + p) * ldq + hh * 32 + 8 * g);
    }
    float bk[2], bv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { bk[t] = in.bias[hh * 32 + t * 16 + c]; bv[t] = in.bias[C + hh * 32 + t * 16 + c]; }
    qrows_finish<8, 2, true, true, false>(in, k, R, sX0, sX1, false);
    ATL(1)
    QSYNC();
    {
        f32x4 ak[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, av[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            const int kw = s_ * 32 + 8 * g;
#pragma unroll
            for (int isv = 0; isv < 2; ++isv) {
                const float* xs = isv ? sX1 : sX0;
                const float4 a = *reinterpret_cast<const float4*>(xs + c * PROJ_XLD + kw), b = *reinterpret_cast<const float4*>(xs + c * PROJ_XLD + kw + 4);
                const float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                proj_u4 hi, lo;
                split8(xv, hi, lo);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4& acc = isv ? av[t] : ak[t];
                    const proj_u4& w = isv ? wv[t][s_] : wk[t][s_];
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(lo), as_frag(w), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(hi), as_frag(w), acc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sKV[hh * 2][4 * g + r][t * 16 + c] = (ak[t][r] + bk[t]) * scale;
                sKV[hh * 2 + 1][4 * g + r][t * 16 + c] = av[t][r] + bv[t];
            }
    }
    // the out-projection's operands: requested now, consumed after the attention
    proj_u4 wo[2][8];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) proj16_load<8>(po.Wo, hh * 32 + ct * 16, 0, wo[ct]);
    uint2 rs[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int p = min(pbase + t * 16 + c, HW - 1);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) rs[ct][t] = *reinterpret_cast<const uint2*>(po.res + ((long)k * HW + p) * C + hh * 32 + ct * 16 + 4 * g);
    }
    float4 bo[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) bo[ct] = *reinterpret_cast<const float4*>(po.bo + hh * 32 + ct * 16 + 4 * g);
    ATL(2)
    // k / v of this head were written by this wave: the LDS queue of a wave is in order, no barrier
    asm volatile("" ::: "memory");
    {
        float kf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[j] = sKV[hh * 2][c][8 * g + j];
        proj_u4 khi, klo, vhi[2], vlo[2];
        split8(kf, khi, klo);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            float vf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) vf[j] = sKV[hh * 2 + 1][4 * g + j][dt * 16 + c];
            split4(vf, vhi[dt], vlo[dt]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(klo), as_frag(qv[t]), s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(khi), as_frag(qv[t]), s, 0, 0, 0);
            const float mx = rows_max(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
            float pe[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pe[r] = __expf(s[r] - mx);
            const float inv = 1.f / rows_sum((pe[0] + pe[1]) + (pe[2] + pe[3]));
#pragma unroll
            for (int r = 0; r < 4; ++r) pe[r] *= inv;
            proj_u4 phi, plo;
            split4(pe, phi, plo);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                o = mfma3(vhi[dt], vlo[dt], phi, plo, o);      // dims dt*16 + 4g .. + 3 of pixel c
                *reinterpret_cast<uint2*>(sO + (t * 16 + c) * P2O_LD + hh * 32 + dt * 16 + 4 * g) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
        }
    }
    ATL(3)
    QSYNC();
    {
        f32x4 acc[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[ct][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const proj_u4 b = *reinterpret_cast<const proj_u4*>(sO + (t * 16 + c) * P2O_LD + s_ * 32 + 8 * g);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wo[ct][s_]), as_frag(b), acc[ct][t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int p = pbase + t * 16 + c;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                float v[4] = {acc[ct][t][0] + bo[ct].x, acc[ct][t][1] + bo[ct].y, acc[ct][t][2] + bo[ct].z, acc[ct][t][3] + bo[ct].w};
                v[0] += __uint_as_float(rs[ct][t].x << 16); v[1] += __uint_as_float(rs[ct][t].x & 0xffff0000u);
                v[2] += __uint_as_float(rs[ct][t].y << 16); v[3] += __uint_as_float(rs[ct][t].y & 0xffff0000u);
                if (p < HW)
                    *reinterpret_cast<uint2*>(y + ((long)k * HW + p) * C + hh * 32 + ct * 16 + 4 * g) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
        }
    }
    ATL(4)
}

#endif  // CUTIE_DIAG

// ---- host side ------------------------------------------------------------------------------------------------------------------------
static bool qin_from_op(const cutie_op* op, QIn& in, QOut& out, const char* who, int xslot, int lnout_slot, bool no_proj = false) {
    const uint64_t* p = op->p;
    in = QIn{};
    out = QOut{};
    in.prio = (op->flags & CUTIE_F_PRIO) ? 1 : 0;
    in.x = (const float*)p[xslot];
    in.ln_out = lnout_slot >= 0 ? (float*)p[lnout_slot] : nullptr;
    in.W = (const bf16_t*)p[5]; in.bias = (const float*)p[6]; in.add = (const float*)p[7];
    in.ln_g = (const float*)p[8]; in.ln_b = (const float*)p[9];
    if (op->flags & 4) {
        if (!p[10]) { cutie_set_error("%s: flags&4 needs p10 = the fixed-point accumulator", who); return false; }
        in.acc = (const long long*)p[10]; in.abias = (const float*)p[11];
    }
    if (op->flags & 8) {
        if (!p[12] || !p[13]) { cutie_set_error("%s: flags&8 needs p12 = Wo and p13 = the accumulator", who); return false; }
        out.W = (const bf16_t*)p[12]; out.acc = (long long*)p[13];
    }
#ifdef ATT_TIMELINE
    in.tl = (unsigned long long*)p[15];
#endif
    if (!in.x || (!in.W && !no_proj)) { cutie_set_error("%s: chain form needs the rows and the projection weight", who); return false; }
    return true;
}

int launch_qchain(const cutie_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const uint64_t* p = op->p;
    QIn in; QOut out;
    switch (op->kind) {
        case CUTIE_OP_ATTN_Q2P: {
            if (i[1] != 16 || i[3] != 256 || i[4] != 8 || (i[7] > 0 && i[7] != 256)) { cutie_set_error("attn_q2p (chain form): Q = 16, C = 256, 8 heads, dense rows"); return -2; }
            if ((op->flags & 11) != 11) { cutie_set_error("attn_q2p (chain form): flags 1 | 2 | 8 required"); return -2; }
            if (!qin_from_op(op, in, out, "attn_q2p", 0, 3, (op->flags & 16) != 0)) return -2;
            if (!p[1] || !p[2]) { cutie_set_error("attn_q2p (chain form): kv and the mask logits required"); return -2; }
            const int HWp = (i[2] + 15) & ~15;
            const bool qp = (op->flags & 16) != 0, ac = !qp && (op->flags & 4);
            const float* qpre = nullptr;
            if (qp) { qpre = (const float*)p[0]; in.x = nullptr; in.W = nullptr; in.ln_out = nullptr; }   // p0 = q [K*16, 256], projected and scaled
            else if (!in.add || !in.ln_g || !in.ln_b || !in.bias || (ac && !in.abias)) {
                cutie_set_error("attn_q2p (chain form): query embedding, LayerNorm, the projection bias and (flags&4) the accumulator's bias are required");
                return -2;
            }
            // one instantiation per (q handed in, accumulator input, logits fetched at entry for <= 4 / <= 8 objects / late)
            const int Kg = i[9] > 0 ? i[9] : i[0];            // i9: objects per clip (0: one clip holds all i0 objects)
            if (i[0] % Kg) { cutie_set_error("attn_q2p (chain form): clips of %d objects do not divide K = %d", Kg, i[0]); return -2; }
            const int kt = Kg <= 4 ? 1 : Kg <= 8 ? 2 : 0;
            const int var = (qp ? 6 : ac ? 3 : 0) + kt;
            constexpr int NWV = 8;
            typedef void (*Kern)(QIn, QOut, const bf16_t*, const float*, int, int, int, int, int, const float*, int);
            static const Kern kerns[9] = {
                q2p_chain_kernel<false, false, 0, NWV>, q2p_chain_kernel<false, false, 4, NWV>, q2p_chain_kernel<false, false, 8, NWV>,
                q2p_chain_kernel<false, true, 0, NWV>, q2p_chain_kernel<false, true, 4, NWV>, q2p_chain_kernel<false, true, 8, NWV>,
                q2p_chain_kernel<true, false, 0, NWV>, q2p_chain_kernel<true, false, 4, NWV>, q2p_chain_kernel<true, false, 8, NWV>};
            const Kern kern = kerns[var];
            const size_t dyn = (size_t)HWp + NWV * 32 * Q2C_VLD * 2;
            if (dyn > 96 * 1024) { cutie_set_error("attn_q2p (chain form): HW = %d does not fit the LDS flag array", i[2]); return -2; }
            static bool attr_set[9] = {};
            if (!attr_set[var]) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) {
                    cutie_set_error("attn_q2p (chain form): cannot raise the dynamic LDS limit");
                    return -2;
                }
                attr_set[var] = true;
            }
            const int hstride = i[8] > 0 ? i[8] : 32;       // elements between the heads of k / v inside a pixel row (32: [k | v | ..] by kind; 64: k | v interleaved per head)
            hipLaunchKernelGGL(kern, dim3(8, i[0]), dim3(NWV * 64), dyn, s, in, out, (const bf16_t*)p[1], (const float*)p[2], i[2], HWp, i[5], i[6], hstride, qpre, Kg);
            break;
        }
        case CUTIE_OP_ATTN_SELF:
            if (i[1] != 16 || i[2] != 256 || i[3] != 8 || (i[6] > 0 && i[6] != 256)) { cutie_set_error("attn_self (chain form): Q = 16, C = 256, 8 heads, dense rows"); return -2; }
            if ((op->flags & 14) != 14) { cutie_set_error("attn_self (chain form): flags 2 | 4 | 8 required"); return -2; }
            if (!qin_from_op(op, in, out, "attn_self", 0, 3)) return -2;
            if (!in.add || !in.ln_g || !in.ln_b || !in.bias || !in.abias) { cutie_set_error("attn_self (chain form): query embedding, LayerNorm and both biases are required"); return -2; }
            hipLaunchKernelGGL(self_chain_kernel, dim3(8, i[0]), dim3(512), 0, s, in, out);
            break;
        case CUTIE_OP_ATTN_P2Q:
            if (i[1] != 16 || i[3] != 256 || i[4] != 8 || (i[7] > 0 && i[7] != 256)) { cutie_set_error("attn_p2q (chain form): Q = 16, C = 256, 8 heads, dense rows"); return -2; }
            if ((op->flags & 6) != 6) { cutie_set_error("attn_p2q (chain form): flags 2 | 4 required"); return -2; }
            if (!qin_from_op(op, in, out, "attn_p2q", 1, -1)) return -2;
            in.ln_g = in.ln_b = nullptr;
            if (!p[0] || !p[3]) { cutie_set_error("attn_p2q (chain form): q and y required"); return -2; }
            if (!in.add || !in.bias || !in.abias) { cutie_set_error("attn_p2q (chain form): query embedding and both biases are required"); return -2; }
        {
            NextQ nq = {};
            if (op->flags & 16) {                            // p8/p9 = LayerNorm of the next block, p12 = its Wq, p13 = bias, p14 = q_out, p15 = xn_out
                if (!p[8] || !p[9] || !p[12] || !p[13] || !p[14] || !p[15]) { cutie_set_error("attn_p2q (chain form): flags&16 needs p8, p9, p12, p13, p14, p15"); return -2; }
                nq = NextQ{(const float*)p[8], (const float*)p[9], (const bf16_t*)p[12], (const float*)p[13], (float*)p[14], (float*)p[15]};
            }
#ifdef CUTIE_DIAG
            if (op->flags & 32) {                            // + read_from_query's output projection and the residual: p2 = [Wo bf16 [256,256] | bias f32 [256]], p4 = residual
                if (!p[2] || !p[4] || (p[2] & 15) || (p[4] & 7) || (p[3] & 7) || p[3] == p[0]) { cutie_set_error("attn_p2q (flags&32): p2 = Wo | bias (16-byte aligned), p4 = residual, y 8-byte aligned required"); return -2; }
                static bool attr_set = false;
                if (!attr_set) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(p2q_out_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, P2O_LDS_BYTES) != hipSuccess) {
                        cutie_set_error("attn_p2q (flags&32): cannot raise the dynamic LDS limit");
                        return -2;
                    }
                    attr_set = true;
                }
                const bf16_t* Wo = (const bf16_t*)p[2];
                const P2QOut po = {Wo, reinterpret_cast<const float*>(Wo + 256 * 256), (const bf16_t*)p[4]};
                const int ntiles = (i[2] + 63) / 64;
                hipLaunchKernelGGL(p2q_out_kernel, dim3(ntiles + (nq.W ? 8 : 0), i[0]), dim3(512), P2O_LDS_BYTES, s, in, (const bf16_t*)p[0], (bf16_t*)p[3], i[2], i[5], nq, po, ntiles);
                break;
            }
#else
            if (op->flags & 32) { cutie_set_error("attn_p2q (flags&32: output projection inside): only in the diagnostic library (make DIAG=1)"); return -2; }
#endif
            hipLaunchKernelGGL(p2q_chain_kernel, dim3((i[2] + 255) / 256 + (nq.W ? 1 : 0), 8, i[0]), dim3(256), 0, s, in, (const bf16_t*)p[0], (bf16_t*)p[3], i[2], i[5], nq);
        }
            break;
        case CUTIE_OP_QFFN: {
            const int HS = i[2] > 0 ? i[2] : 64;
            if ((i[0] & 15) || i[1] < HS || (i[1] % HS) || (HS != 64 && HS != 128) || !p[0] || !p[2] || !p[3] || !p[4] || !p[6] || !p[7]) {
                cutie_set_error("qffn: rows %% 16, FF %% slice == 0, slice 64 | 128 and x, gamma, beta, W1, W2, accumulator required (rows=%d FF=%d slice=%d)", i[0], i[1], HS);
                return -2;
            }
            in = QIn{};
            in.prio = (op->flags & CUTIE_F_PRIO) ? 1 : 0;
            in.x = (const float*)p[0]; in.x_out = (float*)p[1]; in.ln_g = (const float*)p[2]; in.ln_b = (const float*)p[3];
            if (!p[10] || !p[11] || !p[5]) { cutie_set_error("qffn: the input accumulator, its bias and linear1's bias are required"); return -2; }
            in.acc = (const long long*)p[10]; in.abias = (const float*)p[11];
#ifdef ATT_TIMELINE
            in.tl = (unsigned long long*)p[15];
#endif
            QFfn a = {(const bf16_t*)p[4], (const float*)p[5], (const bf16_t*)p[6], (long long*)p[7], i[1]};
            if (HS == 64) hipLaunchKernelGGL(qffn_kernel<64>, dim3(i[1] / 64, i[0] / 16), dim3(256), 0, s, in, a);
            else hipLaunchKernelGGL(qffn_kernel<128>, dim3(i[1] / 128, i[0] / 16), dim3(256), 0, s, in, a);
            break;
        }
        default:
            cutie_set_error("qchain: unknown op kind %d", op->kind);
            return -3;
    }
    return (int)hipGetLastError();
}
