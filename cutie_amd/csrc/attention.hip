// Object-transformer attention kernels (QueryTransformerBlock, reference object_transformer.py:12-73).
// 16 object queries per object, 8 heads x 32: latency-bound (a few MFLOP per launch).  The query->pixel product runs
// on 16x16x32 MFMA tiles with split-bf16 operands (flash-style online softmax); the pixel->query side streams.
// The pixel-side linear projections that feed them are MFMA convs (conv_igemm.hip).
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

// AUX_MASK: fg bytes + nfg[k] += count.  (The frame's plans use the fused form inside ATTN_Q2P; this op stays for callers
// that want the mask itself.)
__global__ void aux_mask_kernel(const float* __restrict__ lg, uint8_t* __restrict__ fg, int* __restrict__ nfg, int K, int HW) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = p < HW;
    for (int k = 0; k < K; ++k) {
        bool f = false;
        if (valid) {
            f = aux_fg_(lg, K, HW, k, p);
            fg[(long)k * HW + p] = f ? 1 : 0;
        }
        unsigned long long b = __ballot(f);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(&nfg[k], __popcll(b));
    }
}

// ATTN_Q2P: masked cross-attention of the 16 object queries over the HW pixels (object_transformer.py:176-206).
// grid (heads, K), block 1024 = 16 waves; wave w owns the 32-pixel chunks w, w+16, ...  Both products run on MFMA in
// the transposed form, so that a lane's accumulator column is always "its" query (lane & 15):
//     S^T[pixel][query] = K[pixel][dim] . Q^T[dim][query]        (Q split hi+lo bf16: ~fp32 scores)
//     O^T[dim][query]   = V^T[dim][pixel] . P^T[pixel][query]    (P split hi+lo bf16)
// The D layout of S^T (4 consecutive pixels of one query per lane) is used directly as the B operand of the second
// product: the k-slot <-> pixel assignment of that MFMA is free, and V^T is gathered to match it.  Online softmax
// state (m, l) is per query, shared by the 4 lanes of a query through two xor-shuffles.  Waves merge through LDS.
// lg != null: the foreground mask of object k is derived here from the mask_pred logits of all K objects (AUX_MASK fused: every
// block recomputes its object's HW flags into LDS and counts them -- two launches and a global counter less per transformer block).
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
    // flags&4: the rows are a sum, x_eff = x + pbias + sum_s part[s] (part: [npart][prows][256] fp32 partial products of the launch in
    // front -- QFFN's hidden-layer slices or the per-head out-projection parts; summed in slice order: deterministic)
    const float* part; const float* pbias; int npart; int prows;
#ifdef ATT_TIMELINE
    unsigned long long* tl;                              // diagnostic builds (tools/attn_timeline.py): p15 = stamp buffer [blocks][16 waves][16]
#endif
};
// ATL(id): cycle stamp of every wave (lane 0) into slot id (0..13); slot 14 / 15 = the 100 MHz wall clock at stamp 0 / at the last stamp
#ifdef ATT_TIMELINE
#define ATL(ID) { if (pi.tl && (threadIdx.x & 63) == 0) { unsigned long long* o_ = pi.tl + ((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 16; \
                  o_[ID] = __builtin_readcyclecounter(); if ((ID) == 0) o_[14] = wall_clock64(); else o_[15] = wall_clock64(); } }
#define ATL_SET(PI, OP) (PI).tl = (unsigned long long*)(OP)->p[15]
#else
#define ATL(ID)
#define ATL_SET(PI, OP)
#endif
struct ProjOut {                                         // flags&8: per-head output projection, part[h] = att_h . Wo[:, 32h..32h+31]^T
    const bf16_t* W; float* part; int prows;
};

__device__ __forceinline__ float4 rows_partial_sum(const ProjIn& pi, long row, int lane, float4 v) {
    if (pi.pbias) { const float4 b = *reinterpret_cast<const float4*>(pi.pbias + lane * 4); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    const float* pr = pi.part + row * 256 + lane * 4;
    const long ps = (long)pi.prows * 256;
    if (pi.npart == 8) {                                 // the frame's case: all eight loads in flight together
        float4 t[8];
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) t[s_] = *reinterpret_cast<const float4*>(pr + s_ * ps);
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) { v.x += t[s_].x; v.y += t[s_].y; v.z += t[s_].z; v.w += t[s_].w; }
    } else {
        for (int s_ = 0; s_ < pi.npart; ++s_) { const float4 t = *reinterpret_cast<const float4*>(pr + s_ * ps); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    }
    return v;
}

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
    if (pi.part) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) xv[j] = rows_partial_sum(pi, (long)k * 16 + wave + j * NW, lane, xv[j]);
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

// one 32-pixel chunk of K and (transposed) V for a wave: issued ahead of use (the first Q2P_PF chunks of every wave are requested
// at kernel entry, in front of the projection prologue and the mask pass: the pixel loop was a chain of dependent memory round trips)
struct Q2PChunk { q2p_frag ka[2], va0, va1; };
#define Q2P_PF 4
__device__ __forceinline__ void q2p_load(Q2PChunk& L, const bf16_t* __restrict__ kvb, int p0, int HW, int ldkv, int voff, int c16, int g) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int p = min(p0 + t * 16 + c16, HW - 1);
        L.ka[t].u = *reinterpret_cast<const q2p_u32x4*>(kvb + (long)p * ldkv + 8 * g);
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {                       // V^T rows c16 / c16+16, k-slots = this lane group's 8 pixels
        const int ja = 2 * jj, jb = 2 * jj + 1;
        const int pa = min(p0 + (ja >> 2) * 16 + 4 * g + (ja & 3), HW - 1);
        const int pb = min(p0 + (jb >> 2) * 16 + 4 * g + (jb & 3), HW - 1);
        const bf16_t* ra = kvb + (long)pa * ldkv + voff + c16;
        const bf16_t* rb = kvb + (long)pb * ldkv + voff + c16;
        L.va0.u[jj] = (uint32_t)ra[0] | ((uint32_t)rb[0] << 16);
        L.va1.u[jj] = (uint32_t)ra[16] | ((uint32_t)rb[16] << 16);
    }
}

__global__ __launch_bounds__(1024) void attn_q2p_kernel(const float* __restrict__ q, const bf16_t* __restrict__ kv,
                                                        const uint8_t* __restrict__ fg, const int* __restrict__ nfg,
                                                        float* __restrict__ y, int Q, int HW, int C, int ldkv, int voff,
                                                        const float* __restrict__ lg, ProjIn pi, ProjOut po) {
    __shared__ float sO[16][16][33];                       // [wave][query][dim]
    __shared__ float sM[16][16], sL[16][16];
    __shared__ float sQ[16][33];                           // fused projection: this head's 32 query columns, scaled (later: the head's output)
    __shared__ int sCnt;
    extern __shared__ uint8_t sFg[];                       // HW flags (fused form only)
    const int hh = blockIdx.x, k = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c16 = lane & 15, g = lane >> 4;              // c16: query (B/D column) or pixel/dim (A row)
    const float scale = rsqrtf(32.f);
    const bf16_t* kvb = kv + (long)k * HW * ldkv + hh * 32;
    const int nchunk = (HW + 31) >> 5;
    ATL(0)
    Q2PChunk pf[Q2P_PF];
#pragma unroll
    for (int j = 0; j < Q2P_PF; ++j)
        if (wave + 16 * j < nchunk) q2p_load(pf[j], kvb, (wave + 16 * j) * 32, HW, ldkv, voff, c16, g);
    proj_u4 wo[1];
    if (po.W) proj16_load<1>(po.W, wave * 16, hh, wo);     // Wo rows 16 wave.., columns of this head (one 32-wide k step)
    q2p_frag qh, ql;
    if (pi.W) {
        // q = (LN(x) + emb) . Wq[head]^T + b computed here (flags&2: the LINEAR launch in front of this one is gone): the staging
        // buffers alias sO, which is only used after the pixel loop.  16 waves = 2 column tiles x 8 k-steps, summed through LDS.
        float* xs = &sO[0][0][0];
        f32x4* red = reinterpret_cast<f32x4*>(xs + 16 * PROJ_XLD);
        static_assert(sizeof(float) * 16 * PROJ_XLD + sizeof(f32x4) * 16 * 64 <= sizeof(float) * 16 * 16 * 33, "projection staging fits in sO");
        const int tile = wave & 1, ks = wave >> 1;
        proj_u4 wq[1];
        proj16_load<1>(pi.W, hh * 32 + tile * 16, ks, wq);         // in flight while the rows are normalised
        stage_rows16<16>(pi, k, xs, nullptr, hh == 0);
        ATL(1)
        __syncthreads();
        ATL(2)
        red[wave * 64 + lane] = proj16_mma<1>(xs, ks, wq);
        ATL(3)
        __syncthreads();
        if (threadIdx.x < 128) {
            const int t = threadIdx.x >> 6;
            f32x4 a = red[t * 64 + lane];
#pragma unroll
            for (int j = 1; j < 8; ++j) { const f32x4 b = red[(t + 2 * j) * 64 + lane]; a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3]; }
            const float bv = pi.bias ? pi.bias[hh * 32 + t * 16 + c16] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) sQ[4 * g + r][t * 16 + c16] = (a[r] + bv) * scale;
        }
        __syncthreads();                                   // sQ complete; xs / red (= sO) are free again
        ATL(4)
#pragma unroll
        for (int j = 0; j < 4; ++j) { uint32_t h_, l_; split_bf2(sQ[c16][8 * g + 2 * j], sQ[c16][8 * g + 2 * j + 1], h_, l_); qh.u[j] = h_; ql.u[j] = l_; }
    } else {
        const float* qr = q + ((long)k * Q + c16) * C + hh * 32 + 8 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) { uint32_t h_, l_; split_bf2(qr[2 * j] * scale, qr[2 * j + 1] * scale, h_, l_); qh.u[j] = h_; ql.u[j] = l_; }
    }
    int n_fg;
    if (lg) {
        if (threadIdx.x == 0) sCnt = 0;
        __syncthreads();
        int cnt = 0;
        for (int p = threadIdx.x; p < HW; p += 1024) {
            const bool f = aux_fg_(lg, gridDim.y, HW, k, p);
            sFg[p] = f ? 1 : 0;
            cnt += f ? 1 : 0;
        }
        cnt = wave_sum_i32(cnt);
        if (lane == 0 && cnt) atomicAdd(&sCnt, cnt);
        __syncthreads();
        ATL(5)
        n_fg = sCnt;
    } else {
        n_fg = nfg[k];
    }
    const bool is_fg_query = c16 < Q / 2;                  // queries 0..7 attend foreground only
    // row fully blocked -> unblocked (object_transformer.py:203)
    const bool masked = is_fg_query ? (n_fg != 0) : (n_fg != HW);
    float m = -INFINITY, l = 0.f;
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    const uint8_t* fgb = lg ? sFg : fg + (long)k * HW;
    auto chunk = [&](const Q2PChunk& L, int p0) {
        uint8_t fgv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) fgv[j] = fgb[min(p0 + (j >> 2) * 16 + 4 * g + (j & 3), HW - 1)];
        f32x4 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L.ka[t].b, qh.b, z, 0, 0, 0);
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L.ka[t].b, ql.b, z, 0, 0, 0);
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
        q2p_frag ph, pl;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { uint32_t h_, l_; split_bf2(pe[2 * jj], pe[2 * jj + 1], h_, l_); ph.u[jj] = h_; pl.u[jj] = l_; }
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L.va0.b, ph.b, o0, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L.va0.b, pl.b, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L.va1.b, ph.b, o1, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L.va1.b, pl.b, o1, 0, 0, 0);
    };
#pragma unroll
    for (int j = 0; j < Q2P_PF; ++j) {
        if (wave + 16 * j < nchunk) chunk(pf[j], (wave + 16 * j) * 32);
        if (j == 0) { ATL(6) }
    }
    for (int ch = wave + 16 * Q2P_PF; ch < nchunk; ch += 16) {
        Q2PChunk L;
        q2p_load(L, kvb, ch * 32, HW, ldkv, voff, c16, g);
        chunk(L, ch * 32);
    }
    ATL(7)
    l = rows_sum(l);
    if (g == 0) { sM[wave][c16] = m; sL[wave][c16] = l; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { sO[wave][c16][4 * g + r] = o0[r]; sO[wave][c16][16 + 4 * g + r] = o1[r]; }
    __syncthreads();
    ATL(8)
    if (threadIdx.x < 512) {                               // (query i, dim d): merge the 16 waves
        const int i = threadIdx.x >> 5, d = threadIdx.x & 31;
        float Mg = -INFINITY;
#pragma unroll
        for (int w = 0; w < 16; ++w) Mg = fmaxf(Mg, sM[w][i]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            float f = (sM[w][i] == -INFINITY) ? 0.f : __expf(sM[w][i] - Mg);
            num += sO[w][i][d] * f;
            den += sL[w][i] * f;
        }
        const float o = num / den;
        if (y) y[((long)k * Q + i) * C + hh * 32 + d] = o;
        if (po.W) sQ[i][d] = o;
    }
    ATL(9)
    if (!po.W) return;
    // per-head output projection (flags&8): part[hh] = o (16 x 32) . Wo[:, 32 hh ..]^T -- wave w: output columns 16 w .. 16 w + 15
    __syncthreads();
    ATL(10)
    {
        proj_u4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) { uint32_t h_, l_; split_bf2(sQ[c16][8 * g + 2 * j], sQ[c16][8 * g + 2 * j + 1], h_, l_); hi[j] = h_; lo[j] = l_; }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, lo), __builtin_bit_cast(bf16x8, wo[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hi), __builtin_bit_cast(bf16x8, wo[0]), acc, 0, 0, 0);
        float* pr = po.part + ((long)hh * po.prows + (long)k * 16 + 4 * g) * 256 + wave * 16 + c16;
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[r * 256] = acc[r];
    }
    ATL(11)
}

// ATTN_SELF: grid (heads, K), block 64: lane = query*4 + part (8 dims each).
// Fused form (pi.W != 0, block 256): q | k | v of this head are projected here from the object's 16 rows -- q and k from
// LN(x) + emb, v from LN(x) (transformer_layers.py:28-41) -- instead of by a LINEAR launch: 4 waves x 2 k-steps x 6 column tiles,
// summed through LDS; wave 0 then runs the 16 x 16 attention on the LDS copies.
// flags&8: the output projection of this head follows in the same launch (ProjOut, as in ATTN_Q2P): 4 waves x 4 column tiles.
__global__ __launch_bounds__(256) void attn_self_kernel(const float* __restrict__ qk, const float* __restrict__ v, float* __restrict__ y, int Q, int C,
                                                        int ldqk, int ldv, ProjIn pi, ProjOut po) {
    __shared__ float sX[2][16 * PROJ_XLD];                 // [LN(x)+emb | LN(x)]
    __shared__ f32x4 sRed[4][6][64];
    __shared__ float sP[3][16][33];                        // q (scaled) | k | v of this head
    const int hh = blockIdx.x, k = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, qi = lane >> 2, part = lane & 3;
    const float scale = rsqrtf(32.f);
    ATL(0)
    proj_u4 wo[4];
    if (po.W) {
#pragma unroll
        for (int t = 0; t < 4; ++t) proj16_load<1>(po.W, (wave * 4 + t) * 16, hh, &wo[t]);
    }
    if (pi.W) {
        proj_u4 wv[6][2];                                  // tiles 0,1: q  2,3: k  4,5: v (weight rows (t/2)*C + head*32 + (t&1)*16)
#pragma unroll
        for (int t = 0; t < 6; ++t) proj16_load<2>(pi.W, (t >> 1) * C + hh * 32 + (t & 1) * 16, 2 * wave, wv[t]);
        stage_rows16<4>(pi, k, sX[0], sX[1], hh == 0);
        ATL(1)
        __syncthreads();
        ATL(2)
#pragma unroll
        for (int t = 0; t < 6; ++t) sRed[wave][t][lane] = proj16_mma<2>(sX[t < 4 ? 0 : 1], 2 * wave, wv[t]);
        ATL(3)
        __syncthreads();
        ATL(4)
        for (int e = threadIdx.x; e < 6 * 64; e += 256) {
            const int t = e >> 6, l = e & 63, c = l & 15, g = l >> 4;
            f32x4 a = sRed[0][t][l];
#pragma unroll
            for (int w = 1; w < 4; ++w) { const f32x4 b = sRed[w][t][l]; a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3]; }
            const int col = (t & 1) * 16 + c;
            const float bv = pi.bias ? pi.bias[(t >> 1) * C + hh * 32 + col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) sP[t >> 1][4 * g + r][col] = (a[r] + bv) * (t < 2 ? scale : 1.f);
        }
        ATL(5)
        __syncthreads();
        ATL(6)
    }
    if (wave == 0) {
        float qf[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) qf[d] = pi.W ? sP[0][qi][part * 8 + d] : qk[((long)k * Q + qi) * ldqk + hh * 32 + part * 8 + d] * scale;
        float s[16], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float* kr = pi.W ? &sP[1][j][part * 8] : qk + ((long)k * Q + j) * ldqk + C + hh * 32 + part * 8;
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
            const float* vr = pi.W ? &sP[2][j][part * 8] : v + ((long)k * Q + j) * ldv + hh * 32 + part * 8;
#pragma unroll
            for (int d = 0; d < 8; ++d) o[d] += s[j] * vr[d];
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            if (y) y[((long)k * Q + qi) * C + hh * 32 + part * 8 + d] = o[d] * inv;
            if (po.W) sP[0][qi][part * 8 + d] = o[d] * inv;        // q is dead (every lane holds its own in registers)
        }
    }
    ATL(7)
    if (!po.W) return;
    __syncthreads();
    ATL(8)
    {
        const int c = lane & 15, g = lane >> 4;
        proj_u4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) { uint32_t h_, l_; split_bf2(sP[0][c][8 * g + 2 * j], sP[0][c][8 * g + 2 * j + 1], h_, l_); hi[j] = h_; lo[j] = l_; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, lo), __builtin_bit_cast(bf16x8, wo[t]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hi), __builtin_bit_cast(bf16x8, wo[t]), acc, 0, 0, 0);
            float* pr = po.part + ((long)hh * po.prows + (long)k * 16 + 4 * g) * 256 + (wave * 4 + t) * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[r * 256] = acc[r];
        }
    }
    ATL(9)
}

// ATTN_P2Q: grid (ceil(HW/256), heads, K): one thread per (pixel, head)
// Fused form (pi.W != 0): k | v of the 16 object queries for this head are projected here (k from x + emb, v from x; the packed
// [k | v] weight of read_from_query) instead of by a LINEAR launch: wave w computes column tile w (k: 0, 1; v: 2, 3) over all of K.
__global__ __launch_bounds__(256) void attn_p2q_kernel(const bf16_t* __restrict__ q, const float* __restrict__ kq,
                                                       const float* __restrict__ vq, bf16_t* __restrict__ y, int Q, int HW,
                                                       int C, int ldq, int ldkv, ProjIn pi) {
    __shared__ float ks[16][32], vs[16][32];
    __shared__ float sX[2][16 * PROJ_XLD];
    const int hh = blockIdx.y, k = blockIdx.z;
    ATL(0)
    if (pi.W) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
        const int isv = wave >> 1, col = (wave & 1) * 16;
        proj_u4 wv[8];
        proj16_load<8>(pi.W, isv * C + hh * 32 + col, 0, wv);
        stage_rows16<4>(pi, k, sX[0], sX[1], false);
        ATL(1)
        __syncthreads();
        ATL(2)
        const f32x4 a = proj16_mma<8>(sX[isv], 0, wv);
        ATL(3)
        const float bv = pi.bias ? pi.bias[isv * C + hh * 32 + col + c] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) (isv ? vs : ks)[4 * g + r][col + c] = a[r] + bv;
    } else
    for (int t = threadIdx.x; t < 512; t += 256) {
        int j = t >> 5, d = t & 31;
        ks[j][d] = kq[((long)k * Q + j) * ldkv + hh * 32 + d];
        vs[j][d] = vq[((long)k * Q + j) * ldkv + hh * 32 + d];
    }
    __syncthreads();
    ATL(4)
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
    ATL(5)
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
    ATL(6)
}

// QFFN: the FFN of one transformer block (transformer_layers.py:101-118) in ONE launch instead of three (out-projection LINEAR of the
// self attention, linear1, linear2): grid (S = FF/256, K), block 512.  Block (s, k) sums the 16 rows of object k from the residual and
// the out-projection parts (ProjIn partial form), normalises them, computes the hidden columns 256 s .. 256 s + 255 (relu) and their
// contribution to all 256 outputs: part[s] = h_s . W2[:, 256 s ..]^T.  Eight waves x two 16-column tiles per layer; both layers'
// weight fragments are requested at entry (128 VGPRs in flight over the staging), so the launch is one memory round trip deep.
// linear2's bias and the residual are added by whoever reads the parts (flags&4 of the attention ops): no second pass, fixed order.
struct QFfn { const float* ln_g; const float* ln_b; float* x_out; const bf16_t* W1; const float* b1; const bf16_t* W2; float* part; int FF; };
#define QFFN_HLD 264
__global__ __launch_bounds__(512) void qffn_kernel(ProjIn pi, QFfn a) {
    __shared__ float sX[16 * PROJ_XLD];
    __shared__ float sH[16 * QFFN_HLD];
    const int sl = blockIdx.x, k = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    ATL(0)
    proj_u4 w1[2][8], w2[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t) proj16_load<8>(a.W1, sl * 256 + (wave * 2 + t) * 16, 0, w1[t]);
#pragma unroll
    for (int t = 0; t < 2; ++t) proj16_load<8>(a.W2, (wave * 2 + t) * 16, sl * 8, w2[t], a.FF);
    {   // rows wave, wave + 8: x_eff = x + pbias + sum of parts; LayerNorm
        float4 xv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long row = (long)k * 16 + wave + 8 * j;
            xv[j] = *reinterpret_cast<const float4*>(pi.x + row * pi.ldx + lane * 4);
        }
        if (pi.part) {
#pragma unroll
            for (int j = 0; j < 2; ++j) xv[j] = rows_partial_sum(pi, (long)k * 16 + wave + 8 * j, lane, xv[j]);
        }
        const float4 gg = *reinterpret_cast<const float4*>(a.ln_g + lane * 4), bb = *reinterpret_cast<const float4*>(a.ln_b + lane * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = wave + 8 * j;
            float4 v = xv[j];
            if (a.x_out && sl == 0) *reinterpret_cast<float4*>(a.x_out + ((long)k * 16 + r) * 256 + lane * 4) = v;
            const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.f / 256.f);
            const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
            const float rstd = rsqrtf(wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.f / 256.f) + 1e-5f);
            v.x = dx * rstd * gg.x + bb.x; v.y = dy * rstd * gg.y + bb.y; v.z = dz * rstd * gg.z + bb.z; v.w = dw * rstd * gg.w + bb.w;
            *reinterpret_cast<float4*>(sX + r * PROJ_XLD + lane * 4) = v;
        }
    }
    ATL(1)
    __syncthreads();
    ATL(2)
#ifdef ATT_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATL(3)
#endif
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x4 acc = proj16_mma<8>(sX, 0, w1[t]);
        const int col = (wave * 2 + t) * 16 + c;
        const float bv = a.b1 ? a.b1[sl * 256 + col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) sH[(4 * g + r) * QFFN_HLD + col] = fmaxf(acc[r] + bv, 0.f);
    }
    ATL(4)
    __syncthreads();
    ATL(5)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x4 acc = proj16_mma<8>(sH, 0, w2[t], QFFN_HLD);
        float* pr = a.part + ((long)sl * pi.prows + (long)k * 16 + 4 * g) * 256 + (wave * 2 + t) * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[r * 256] = acc[r];
    }
    ATL(6)
}

// QUERY_INIT with its two linears (flags&1): per object, x = sums / (area + 1e-4) for the 16 summaries (object_transformer.py:125-132)
// staged in LDS, then query = x Wi^T + bi + query_init and query_emb = x We^T + be + query_emb (:137-138) -- three launches in one.
// grid (K, 4), block 256: a block computes 8 of the 32 column tiles (2 x 16: query_init | query_emb), 2 per wave over all of K = 256;
// the rows are staged by every block (16 x 257 floats).  (One block per object with 4 tiles per wave took 16 us: 3 blocks, 208 VGPRs.)
struct QInit2 { const float* om; float* y[2]; const bf16_t* W[2]; const float* b[2]; const float* res[2]; };
__global__ __launch_bounds__(256) void query_init2_kernel(QInit2 a) {
    __shared__ float sX[16 * PROJ_XLD];
    const int k = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    proj_u4 wv[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tile = blockIdx.y * 8 + wave * 2 + t;     // 0..15: query_init columns, 16..31: query_emb columns
        proj16_load<8>(a.W[tile >> 4], (tile & 15) * 16, 0, wv[t]);
    }
    for (int r = wave; r < 16; r += 4) {
        const float* row = a.om + ((long)k * 16 + r) * 257;
        const float inv = 1.f / (row[256] + 1e-4f);
#pragma unroll
        for (int j = 0; j < 4; ++j) sX[r * PROJ_XLD + lane * 4 + j] = row[lane * 4 + j] * inv;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tile = blockIdx.y * 8 + wave * 2 + t, which = tile >> 4, col = (tile & 15) * 16 + c;
        const f32x4 acc = proj16_mma<8>(sX, 0, wv[t]);
        const float bv = a.b[which] ? a.b[which][col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long m = (long)k * 16 + 4 * g + r;
            a.y[which][m * 256 + col] = acc[r] + bv + (a.res[which] ? a.res[which][m * 256 + col] : 0.f);
        }
    }
}

// flags&4 / flags&8 of the attention ops: the partial-sum input (p10, p11, i8) and the per-head output projection (p12, p13)
static int proj_extras(const cutie_op* op, ProjIn& pi, ProjOut& po, const char* who, int rows) {
    if (!(op->flags & 12)) return 0;
    if (!(op->flags & 2)) { cutie_set_error("%s: flags 4 / 8 need the fused projection (flags&2)", who); return -2; }
    if (op->flags & 4) {
        if (!op->p[10] || op->i[8] < 1) { cutie_set_error("%s: flags&4 needs p10 = parts and i8 >= 1", who); return -2; }
        pi.part = (const float*)op->p[10]; pi.pbias = (const float*)op->p[11]; pi.npart = op->i[8]; pi.prows = rows;
    }
    if (op->flags & 8) {
        if (!op->p[12] || !op->p[13]) { cutie_set_error("%s: flags&8 needs p12 = Wo and p13 = parts", who); return -2; }
        po.W = (const bf16_t*)op->p[12]; po.part = (float*)op->p[13]; po.prows = rows;
    }
    return 0;
}

int launch_attention(const cutie_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const uint64_t* p = op->p;
    switch (op->kind) {
        case CUTIE_OP_QUERY_INIT: {                          // fused form only (flags&1): p0=obj_mem [K*16,257] p1=query p2=query_emb p3..5 = Wi, bi, res_i  p6..8 = We, be, res_e
            if (i[1] != 256 || (i[0] & 15)) { cutie_set_error("query_init (fused): C == 256, rows %% 16 == 0"); return -2; }
            QInit2 a = {(const float*)p[0], {(float*)p[1], (float*)p[2]}, {(const bf16_t*)p[3], (const bf16_t*)p[6]}, {(const float*)p[4], (const float*)p[7]},
                        {(const float*)p[5], (const float*)p[8]}};
            hipLaunchKernelGGL(query_init2_kernel, dim3(i[0] / 16, 4), dim3(256), 0, s, a);
            break;
        }
        case CUTIE_OP_AUX_MASK:
            hipLaunchKernelGGL(aux_mask_kernel, dim3((i[1] + 255) / 256), dim3(256), 0, s, (const float*)p[0], (uint8_t*)p[1], (int*)p[2], i[0], i[1]);
            break;
        case CUTIE_OP_ATTN_Q2P: {
            if (i[1] != 16 || i[3] != i[4] * 32) { cutie_set_error("attn_q2p: Q=16, head dim 32 only"); return -2; }
            if ((op->flags & 1) && i[2] > 24576) { cutie_set_error("attn_q2p: fused aux mask holds HW <= 24576 flags in LDS (HW=%d)", i[2]); return -2; }
            ProjIn pi = {};
            if (op->flags & 2) {                             // q projection fused: p0 = x (rows of i[7] floats), p3 = ln_out, p5..p9 = Wq, bq, emb, gamma, beta
                if (!(op->flags & 1) || i[3] != 256) { cutie_set_error("attn_q2p: the fused q projection needs the fused mask form and C == 256"); return -2; }
                pi.x = (const float*)p[0]; pi.ln_out = (float*)p[3]; pi.W = (const bf16_t*)p[5]; pi.bias = (const float*)p[6];
                pi.add = (const float*)p[7]; pi.ln_g = (const float*)p[8]; pi.ln_b = (const float*)p[9]; pi.ldx = i[7] > 0 ? i[7] : 256;
            }
            ProjOut po = {};
            ATL_SET(pi, op);
            if (int rc = proj_extras(op, pi, po, "attn_q2p", i[0] * 16)) return rc;
            if (!po.W && !p[4]) { cutie_set_error("attn_q2p: no output"); return -2; }
            if (op->flags & 1)                               // p2 = mask_pred logits f32 [K,HW]; fg / nfg are not read
                hipLaunchKernelGGL(attn_q2p_kernel, dim3(i[4], i[0]), dim3(1024), (size_t)((i[2] + 15) & ~15), s, (const float*)p[0], (const bf16_t*)p[1],
                                   (const uint8_t*)nullptr, (const int*)nullptr, (float*)p[4], i[1], i[2], i[3], i[5], i[6], (const float*)p[2], pi, po);
            else
                hipLaunchKernelGGL(attn_q2p_kernel, dim3(i[4], i[0]), dim3(1024), 0, s, (const float*)p[0], (const bf16_t*)p[1], (const uint8_t*)p[2],
                                   (const int*)p[3], (float*)p[4], i[1], i[2], i[3], i[5], i[6], (const float*)nullptr, pi, po);
            break;
        }
        case CUTIE_OP_ATTN_SELF:
            if (i[1] != 16 || i[2] != i[3] * 32) { cutie_set_error("attn_self: Q=16, head dim 32 only"); return -2; }
            if (op->flags & 2) {                             // qkv projection fused: p0 = x, p3 = ln_out, p5..p9 = Wqkv, b, emb, gamma, beta
                if (i[2] != 256) { cutie_set_error("attn_self: the fused projection needs C == 256"); return -2; }
                ProjIn pi = {};
                pi.x = (const float*)p[0]; pi.ln_out = (float*)p[3]; pi.W = (const bf16_t*)p[5]; pi.bias = (const float*)p[6];
                pi.add = (const float*)p[7]; pi.ln_g = (const float*)p[8]; pi.ln_b = (const float*)p[9]; pi.ldx = i[6] > 0 ? i[6] : 256;
                ProjOut po = {};
                ATL_SET(pi, op);
                if (int rc = proj_extras(op, pi, po, "attn_self", i[0] * 16)) return rc;
                if (!po.W && !p[2]) { cutie_set_error("attn_self: no output"); return -2; }
                hipLaunchKernelGGL(attn_self_kernel, dim3(i[3], i[0]), dim3(256), 0, s, (const float*)nullptr, (const float*)nullptr, (float*)p[2], i[1], i[2], 0, 0, pi, po);
                break;
            }
            if (op->flags & 12) { cutie_set_error("attn_self: flags 4 / 8 need the fused projection (flags&2)"); return -2; }
            hipLaunchKernelGGL(attn_self_kernel, dim3(i[3], i[0]), dim3(64), 0, s, (const float*)p[0], (const float*)p[1], (float*)p[2], i[1], i[2],
                               i[4] > 0 ? i[4] : 2 * i[2], i[5] > 0 ? i[5] : i[2], ProjIn{}, ProjOut{});
            break;
        case CUTIE_OP_ATTN_P2Q:
            if (i[1] != 16 || i[3] != i[4] * 32) { cutie_set_error("attn_p2q: Q=16, head dim 32 only"); return -2; }
        {
            ProjIn pi = {};
            if (op->flags & 2) {                             // kv projection fused: p1 = x (query rows), p5..p7 = Wkv, b, emb
                if (i[3] != 256) { cutie_set_error("attn_p2q: the fused projection needs C == 256"); return -2; }
                pi.x = (const float*)p[1]; pi.W = (const bf16_t*)p[5]; pi.bias = (const float*)p[6]; pi.add = (const float*)p[7]; pi.ldx = i[7] > 0 ? i[7] : 256;
            }
            ProjOut po = {};
            ATL_SET(pi, op);
            if (int rc = proj_extras(op, pi, po, "attn_p2q", i[0] * 16)) return rc;
            if (po.W) { cutie_set_error("attn_p2q: flags&8 is not defined for this op"); return -2; }
            hipLaunchKernelGGL(attn_p2q_kernel, dim3((i[2] + 255) / 256, i[4], i[0]), dim3(256), 0, s, (const bf16_t*)p[0], (const float*)p[1],
                               (const float*)p[2], (bf16_t*)p[3], i[1], i[2], i[3], i[5], i[6] > 0 ? i[6] : i[3], pi);
            break;
        }
        case CUTIE_OP_QFFN: {
            if ((i[0] & 15) || i[1] < 256 || (i[1] & 255) || !p[0] || !p[2] || !p[3] || !p[4] || !p[6] || !p[7]) {
                cutie_set_error("qffn: rows %% 16, FF %% 256 == 0 and x, gamma, beta, W1, W2, part required (rows=%d FF=%d)", i[0], i[1]);
                return -2;
            }
            ProjIn pi = {};
            pi.x = (const float*)p[0]; pi.ldx = 256;
            if (p[10]) { pi.part = (const float*)p[10]; pi.pbias = (const float*)p[11]; pi.npart = i[8]; pi.prows = i[0]; }
            else if (p[11]) { cutie_set_error("qffn: a bias without parts is not supported"); return -2; }
            if (pi.part && pi.npart < 1) { cutie_set_error("qffn: i8 = number of input parts must be >= 1"); return -2; }
            pi.prows = i[0];
            ATL_SET(pi, op);
            QFfn a = {(const float*)p[2], (const float*)p[3], (float*)p[1], (const bf16_t*)p[4], (const float*)p[5], (const bf16_t*)p[6], (float*)p[7], i[1]};
            hipLaunchKernelGGL(qffn_kernel, dim3(i[1] / 256, i[0] / 16), dim3(512), 0, s, pi, a);
            break;
        }
        default:
            cutie_set_error("attention: unknown op kind %d", op->kind);
            return -3;
    }
    return (int)hipGetLastError();
}
