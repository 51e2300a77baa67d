// Object-transformer attention kernels (QueryTransformerBlock, reference object_transformer.py:12-73).
// 16 object queries per object, 8 heads x 32: latency-bound (a few MFLOP per launch).  The query->pixel product runs
// on 16x16x32 MFMA tiles with split-bf16 operands (flash-style online softmax); the pixel->query side streams.
// The pixel-side linear projections that feed them are MFMA convs (conv_igemm.hip).
#include "attention_common.h"

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
__global__ __launch_bounds__(1024) void attn_q2p_kernel(const float* __restrict__ q, const bf16_t* __restrict__ kv,
                                                        const uint8_t* __restrict__ fg, const int* __restrict__ nfg,
                                                        float* __restrict__ y, int Q, int HW, int C, int ldkv, int voff,
                                                        const float* __restrict__ lg, ProjIn pi) {
    __shared__ float sO[16][16][33];                       // [wave][query][dim]
    __shared__ float sM[16][16], sL[16][16];
    __shared__ float sQ[16][33];                           // fused projection: this head's 32 query columns, scaled
    __shared__ int sCnt;
    extern __shared__ uint8_t sFg[];                       // HW flags (fused form only)
    const int hh = blockIdx.x, k = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c16 = lane & 15, g = lane >> 4;              // c16: query (B/D column) or pixel/dim (A row)
    const float scale = rsqrtf(32.f);
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
        __syncthreads();
        red[wave * 64 + lane] = proj16_mma<1>(xs, ks, wq);
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
        n_fg = sCnt;
    } else {
        n_fg = nfg[k];
    }
    const bool is_fg_query = c16 < Q / 2;                  // queries 0..7 attend foreground only
    // row fully blocked -> unblocked (object_transformer.py:203)
    const bool masked = is_fg_query ? (n_fg != 0) : (n_fg != HW);
    float m = -INFINITY, l = 0.f;
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    const bf16_t* kvb = kv + (long)k * HW * ldkv + hh * 32;
    const uint8_t* fgb = lg ? sFg : fg + (long)k * HW;
    const int nchunk = (HW + 31) >> 5;
    for (int ch = wave; ch < nchunk; ch += 16) {
        const int p0 = ch * 32;
        q2p_frag ka[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int p = min(p0 + t * 16 + c16, HW - 1);
            ka[t].u = *reinterpret_cast<const q2p_u32x4*>(kvb + (long)p * ldkv + 8 * g);
        }
        q2p_frag va0, va1;                                 // V^T rows c16 / c16+16, k-slots = this lane group's 8 pixels
        uint8_t fgv[8];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int ja = 2 * jj, jb = 2 * jj + 1;
            int pa = min(p0 + (ja >> 2) * 16 + 4 * g + (ja & 3), HW - 1);
            int pb = min(p0 + (jb >> 2) * 16 + 4 * g + (jb & 3), HW - 1);
            const bf16_t* ra = kvb + (long)pa * ldkv + voff + c16;
            const bf16_t* rb = kvb + (long)pb * ldkv + voff + c16;
            va0.u[jj] = (uint32_t)ra[0] | ((uint32_t)rb[0] << 16);
            va1.u[jj] = (uint32_t)ra[16] | ((uint32_t)rb[16] << 16);
            fgv[ja] = fgb[pa]; fgv[jb] = fgb[pb];
        }
        f32x4 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[t].b, qh.b, z, 0, 0, 0);
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[t].b, ql.b, z, 0, 0, 0);
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
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va0.b, ph.b, o0, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va0.b, pl.b, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va1.b, ph.b, o1, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va1.b, pl.b, o1, 0, 0, 0);
    }
    l = rows_sum(l);
    if (g == 0) { sM[wave][c16] = m; sL[wave][c16] = l; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { sO[wave][c16][4 * g + r] = o0[r]; sO[wave][c16][16 + 4 * g + r] = o1[r]; }
    __syncthreads();
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
        y[((long)k * Q + i) * C + hh * 32 + d] = num / den;
    }
}

// ATTN_SELF: grid (heads, K), block 64: lane = query*4 + part (8 dims each).
// Fused form (pi.W != 0, block 256): q | k | v of this head are projected here from the object's 16 rows -- q and k from
// LN(x) + emb, v from LN(x) (transformer_layers.py:28-41) -- instead of by a LINEAR launch: 4 waves x 2 k-steps x 6 column tiles,
// summed through LDS; wave 0 then runs the 16 x 16 attention on the LDS copies.
__global__ __launch_bounds__(256) void attn_self_kernel(const float* __restrict__ qk, const float* __restrict__ v, float* __restrict__ y, int Q, int C,
                                                        int ldqk, int ldv, ProjIn pi) {
    __shared__ float sX[2][16 * PROJ_XLD];                 // [LN(x)+emb | LN(x)]
    __shared__ f32x4 sRed[4][6][64];
    __shared__ float sP[3][16][33];                        // q (scaled) | k | v of this head
    const int hh = blockIdx.x, k = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, qi = lane >> 2, part = lane & 3;
    const float scale = rsqrtf(32.f);
    if (pi.W) {
        proj_u4 wv[6][2];                                  // tiles 0,1: q  2,3: k  4,5: v (weight rows (t/2)*C + head*32 + (t&1)*16)
#pragma unroll
        for (int t = 0; t < 6; ++t) proj16_load<2>(pi.W, (t >> 1) * C + hh * 32 + (t & 1) * 16, 2 * wave, wv[t]);
        stage_rows16<4>(pi, k, sX[0], sX[1], hh == 0);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 6; ++t) sRed[wave][t][lane] = proj16_mma<2>(sX[t < 4 ? 0 : 1], 2 * wave, wv[t]);
        __syncthreads();
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
        __syncthreads();
    }
    if (wave != 0) return;
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
    for (int d = 0; d < 8; ++d) y[((long)k * Q + qi) * C + hh * 32 + part * 8 + d] = o[d] * inv;
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
    if (pi.W) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
        const int isv = wave >> 1, col = (wave & 1) * 16;
        proj_u4 wv[8];
        proj16_load<8>(pi.W, isv * C + hh * 32 + col, 0, wv);
        stage_rows16<4>(pi, k, sX[0], sX[1], false);
        __syncthreads();
        const f32x4 a = proj16_mma<8>(sX[isv], 0, wv);
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

// QUERY_INIT with its two linears (flags&1): per object, x = sums / (area + 1e-4) for the 16 summaries (object_transformer.py:125-132)
// staged in LDS, then query = x Wi^T + bi + query_init and query_emb = x We^T + be + query_emb (:137-138) -- three launches in one.
// grid (K, 8), block 256: a block computes 4 of the 32 column tiles (2 x 16: query_init | query_emb), 1 per wave over all of K = 256;
// the rows are staged by every block (16 x 257 floats).  (One block per object with 4 tiles per wave took 16 us: 3 blocks, 208 VGPRs.)
struct QInit2 { const float* om; float* y[2]; const bf16_t* W[2]; const float* b[2]; const float* res[2]; uint4* zero; int nzero; };
__global__ __launch_bounds__(256) void query_init2_kernel(QInit2 a) {
    __shared__ float sX[16 * PROJ_XLD];
    const int k = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    // side job (p9, i2): clear the fixed-point accumulators of the transformer blocks behind this launch (qchain.hip)
    for (int e = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; e < a.nzero; e += gridDim.x * gridDim.y * 256) a.zero[e] = make_uint4(0u, 0u, 0u, 0u);
    // one 16-column tile per wave (grid (K, 8): 24 blocks at K = 3, 32 KB of weights each -- a block with two tiles per wave pulled 64 KB
    // through its CU); bias and residual are requested with the weights, not after the product
    const int tile = blockIdx.y * 4 + wave, which = tile >> 4, col = (tile & 15) * 16 + c;     // 0..15: query_init columns, 16..31: query_emb columns
    proj_u4 wv[8];
    proj16_load<8>(a.W[which], (tile & 15) * 16, 0, wv);
    const float bv = a.b[which] ? a.b[which][col] : 0.f;
    float rv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rv[r] = a.res[which] ? a.res[which][((long)k * 16 + 4 * g + r) * 256 + col] : 0.f;
    for (int r = wave; r < 16; r += 4) {
        const float* row = a.om + ((long)k * 16 + r) * 257;
        const float inv = 1.f / (row[256] + 1e-4f);
#pragma unroll
        for (int j = 0; j < 4; ++j) sX[r * PROJ_XLD + lane * 4 + j] = row[lane * 4 + j] * inv;
    }
    __syncthreads();
    const f32x4 acc = proj16_mma<8>(sX, 0, wv);
#pragma unroll
    for (int r = 0; r < 4; ++r) a.y[which][((long)k * 16 + 4 * g + r) * 256 + col] = acc[r] + bv + rv[r];
}

int launch_attention(const cutie_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const uint64_t* p = op->p;
    switch (op->kind) {
        case CUTIE_OP_QUERY_INIT: {                          // fused form only (flags&1): p0=obj_mem [K*16,257] p1=query p2=query_emb p3..5 = Wi, bi, res_i  p6..8 = We, be, res_e
            if (i[1] != 256 || (i[0] & 15)) { cutie_set_error("query_init (fused): C == 256, rows %% 16 == 0"); return -2; }
            QInit2 a = {(const float*)p[0], {(float*)p[1], (float*)p[2]}, {(const bf16_t*)p[3], (const bf16_t*)p[6]}, {(const float*)p[4], (const float*)p[7]},
                        {(const float*)p[5], (const float*)p[8]}, (uint4*)p[9], p[9] ? i[2] : 0};
            if (p[9] && (i[2] < 0 || (p[9] & 15))) { cutie_set_error("query_init (fused): the range to clear must be 16-byte aligned"); return -2; }
            hipLaunchKernelGGL(query_init2_kernel, dim3(i[0] / 16, 8), dim3(256), 0, s, a);
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
            if (op->flags & 1)                               // p2 = mask_pred logits f32 [K,HW]; fg / nfg are not read
                hipLaunchKernelGGL(attn_q2p_kernel, dim3(i[4], i[0]), dim3(1024), (size_t)((i[2] + 15) & ~15), s, (const float*)p[0], (const bf16_t*)p[1],
                                   (const uint8_t*)nullptr, (const int*)nullptr, (float*)p[4], i[1], i[2], i[3], i[5], i[6], (const float*)p[2], pi);
            else
                hipLaunchKernelGGL(attn_q2p_kernel, dim3(i[4], i[0]), dim3(1024), 0, s, (const float*)p[0], (const bf16_t*)p[1], (const uint8_t*)p[2],
                                   (const int*)p[3], (float*)p[4], i[1], i[2], i[3], i[5], i[6], (const float*)nullptr, pi);
            break;
        }
        case CUTIE_OP_ATTN_SELF:
            if (i[1] != 16 || i[2] != i[3] * 32) { cutie_set_error("attn_self: Q=16, head dim 32 only"); return -2; }
            if (op->flags & 2) {                             // qkv projection fused: p0 = x, p3 = ln_out, p5..p9 = Wqkv, b, emb, gamma, beta
                if (i[2] != 256) { cutie_set_error("attn_self: the fused projection needs C == 256"); return -2; }
                ProjIn pi = {};
                pi.x = (const float*)p[0]; pi.ln_out = (float*)p[3]; pi.W = (const bf16_t*)p[5]; pi.bias = (const float*)p[6];
                pi.add = (const float*)p[7]; pi.ln_g = (const float*)p[8]; pi.ln_b = (const float*)p[9]; pi.ldx = i[6] > 0 ? i[6] : 256;
                hipLaunchKernelGGL(attn_self_kernel, dim3(i[3], i[0]), dim3(256), 0, s, (const float*)nullptr, (const float*)nullptr, (float*)p[2], i[1], i[2], 0, 0, pi);
                break;
            }
            hipLaunchKernelGGL(attn_self_kernel, dim3(i[3], i[0]), dim3(64), 0, s, (const float*)p[0], (const float*)p[1], (float*)p[2], i[1], i[2],
                               i[4] > 0 ? i[4] : 2 * i[2], i[5] > 0 ? i[5] : i[2], ProjIn{});
            break;
        case CUTIE_OP_ATTN_P2Q:
            if (i[1] != 16 || i[3] != i[4] * 32) { cutie_set_error("attn_p2q: Q=16, head dim 32 only"); return -2; }
        {
            ProjIn pi = {};
            if (op->flags & 2) {                             // kv projection fused: p1 = x (query rows), p5..p7 = Wkv, b, emb
                if (i[3] != 256) { cutie_set_error("attn_p2q: the fused projection needs C == 256"); return -2; }
                pi.x = (const float*)p[1]; pi.W = (const bf16_t*)p[5]; pi.bias = (const float*)p[6]; pi.add = (const float*)p[7]; pi.ldx = i[7] > 0 ? i[7] : 256;
            }
            hipLaunchKernelGGL(attn_p2q_kernel, dim3((i[2] + 255) / 256, i[4], i[0]), dim3(256), 0, s, (const bf16_t*)p[0], (const float*)p[1],
                               (const float*)p[2], (bf16_t*)p[3], i[1], i[2], i[3], i[5], i[6] > 0 ? i[6] : i[3], pi);
            break;
        }
        default:
            cutie_set_error("attention: unknown op kind %d", op->kind);
            return -3;
    }
    return (int)hipGetLastError();
}
