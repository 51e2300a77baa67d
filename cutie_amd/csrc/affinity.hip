// Fused pixel-memory affinity readout (reference: MemoryManager.read -> get_similarity -> do_softmax(top_k)
// -> _readout; memory_manager.py:112-208, memory_utils.py:7-77).
//
// The reference materialises S[N,HW] in fp32, runs torch.topk over N, scatters the 30 weights back into a
// dense [N,HW] matrix and multiplies it with V[K*256,N].  Here the N x HW matrix never exists:
//
//   KEY_PREP     memory side (once per memorised frame): A_i = [k_i^2 | k_i] split into bf16 hi/lo, scale_i
//                query side  (per frame):                B_j = [-e_j | 2 k_j e_j] hi/lo, c_j = sum e_j k_j^2
//                so that S_ij = scale_i * (A_i . B_j - c_j) is ONE K=128 contraction.
//   AFF_SCORE/0  S tiles on MFMA (v_mfma_f32_16x16x32_bf16, 3 split terms hi*hi + hi*lo + lo*hi ~ fp32
//                accuracy), reduced to per-(16-token tile, query) maxima.
//   AFF_SELECT   tau_j = top_k-th largest tile maximum: a lower bound of the top_k-th largest score with
//                only ~top_k..1.2*top_k scores above it (radix select, exact).
//   AFF_SCORE/1  S tiles again (cheap: 3*2*128 flop per score), append the few scores >= tau_j.
//   AFF_READOUT  exact top-k of the candidates (ties -> lower slot), softmax, usage, sparse V gather.
//
// Memory tokens are addressed by *physical bank slot*; the bank is contiguous per bucket (keys) and per
// object (values), so the valid tokens are at most 3 slot ranges [long-term | permanent | working ring].
#include "common.h"
#include <math.h>

__global__ void key_prep_kernel(const float* __restrict__ key, const float* __restrict__ aux, bf16_t* __restrict__ hi,
                                bf16_t* __restrict__ lo, float* __restrict__ sc, int n, int query) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * 16) return;
    int row = idx >> 4, c8 = idx & 15;                  // 16 chunks of 8 -> 128 operand channels
    const float* k = key + (long)row * 64 + (c8 & 7) * 8;
    float v[8];
    if (!query) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (c8 < 8) ? k[i] * k[i] : k[i];
        if (c8 == 0) sc[row] = aux[row] * 0.125f;       // shrinkage / sqrt(64)
    } else {
        const float* e = aux + (long)row * 64 + (c8 & 7) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (c8 < 8) ? -e[i] : 2.f * k[i] * e[i];
        if (c8 == 0) {
            const float* kk = key + (long)row * 64; const float* ee = aux + (long)row * 64;
            float c = 0.f;
            for (int i = 0; i < 64; ++i) c += ee[i] * kk[i] * kk[i];
            sc[row] = c;
        }
    }
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf16_t h0 = f2bf(v[2 * i]), h1 = f2bf(v[2 * i + 1]);
        bf16_t l0 = f2bf(v[2 * i] - bf2f(h0)), l1 = f2bf(v[2 * i + 1] - bf2f(h1));
        h[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        l[i] = (uint32_t)l0 | ((uint32_t)l1 << 16);
    }
    *reinterpret_cast<uint4*>(hi + (long)row * 128 + c8 * 8) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(lo + (long)row * 128 + c8 * 8) = make_uint4(l[0], l[1], l[2], l[3]);
}

struct ScoreParams {
    const bf16_t* Ahi; const bf16_t* Alo; const float* scale;
    const bf16_t* Bhi; const bf16_t* Blo; const float* c;
    float* gmax_or_tau; float* cand_val; int* cand_idx; int* count;
    int HW, HWp, nranges, rs[3], rn[3], G, cap, mode, tiles_per_block, Gld;
};

typedef __attribute__((ext_vector_type(4))) unsigned int au32x4;

// grid (HWp/64, ceil(G/tiles_per_block)); wave w of the block owns queries j0 + 16w .. +15.
// The memory operands of 4 consecutive 16-token tiles (64 rows x [hi|lo] x 256 B = 32 KB) are staged ONCE per block in
// LDS (double-buffered, register prefetch of the next group) and shared by the 4 waves; rows are XOR-swizzled by
// (row & 15) so the ds_read_b128 fragment reads are conflict-free.  B fragments of the 16 queries stay in registers.
#define AFF_TG 4                                      // tiles per LDS group
__global__ __launch_bounds__(256) void aff_score_kernel(ScoreParams p) {
    __shared__ au32x4 lds[2][2 * 64 * 16];            // [buffer][hi/lo][row][16 chunks]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    // XCD-aware mapping (see conv_igemm.hip): consecutive logical blocks share the token chunk (query block fastest)
    int bx, by;
    {
        const int nb = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = id & 7, kq = id >> 3, q = nb >> 3, r = nb & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kq;
        by = logical / (int)gridDim.x;
        bx = logical - by * (int)gridDim.x;
    }
    const int j = bx * 64 + wave * 16 + l15;                           // query column of this lane
    bf16x8 bh[4], bl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bh[ks] = *reinterpret_cast<const bf16x8*>(p.Bhi + (long)j * 128 + ks * 32 + l4 * 8);
        bl[ks] = *reinterpret_cast<const bf16x8*>(p.Blo + (long)j * 128 + ks * 32 + l4 * 8);
    }
    const bool jvalid = j < p.HW;
    const float cj = jvalid ? p.c[j] : 0.f;
    float thr = INFINITY;
    if (p.mode == 1 && jvalid) {
        float tau = p.gmax_or_tau[j];
        thr = tau - fabsf(tau) * 1e-6f - 1e-30f;                        // never lose the k-th element to 1 ulp
    }
    const int g0 = by * p.tiles_per_block;
    const int g1 = min(g0 + p.tiles_per_block, p.G);
    const int T0 = (p.rn[0] + 15) >> 4, T1 = (p.nranges > 1) ? ((p.rn[1] + 15) >> 4) : 0;
    // this thread stages chunk (row = tid >> 2 .. , 4 chunks) : 64 rows x 16 chunks x 2 arrays = 2048 chunks / 256 threads = 8
    const int srow = tid >> 2, sc0 = (tid & 3) * 4;                    // row 0..63, chunks sc0..sc0+3 of hi and of lo
    au32x4 st[8];
    auto tile_slot = [&](int g, int& start, int& n, int& lt) {
        int r;
        if (g < T0) { r = 0; lt = g; } else if (g < T0 + T1) { r = 1; lt = g - T0; } else { r = 2; lt = g - T0 - T1; }
        start = p.rs[r]; n = p.rn[r];
    };
#define AFF_LOAD(GRP)                                                                                      \
    {                                                                                                      \
        int gt = (GRP) + (srow >> 4);                      /* tile of this staging row */                  \
        gt = gt < g1 ? gt : g1 - 1;                        /* clamp: rows of missing tiles are never used */ \
        int start_, n_, lt_;                                                                               \
        tile_slot(gt, start_, n_, lt_);                                                                    \
        const long off_ = (long)(start_ + lt_ * 16 + (srow & 15)) * 128 + sc0 * 8;                         \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                    \
            st[c] = *reinterpret_cast<const au32x4*>(p.Ahi + off_ + c * 8);                                \
            st[4 + c] = *reinterpret_cast<const au32x4*>(p.Alo + off_ + c * 8);                            \
        }                                                                                                  \
    }
#define AFF_STORE(BUF)                                                                                     \
    {                                                                                                      \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                    \
            lds[BUF][srow * 16 + ((sc0 + c) ^ (srow & 15))] = st[c];                                       \
            lds[BUF][1024 + srow * 16 + ((sc0 + c) ^ (srow & 15))] = st[4 + c];                            \
        }                                                                                                  \
    }
    AFF_LOAD(g0);
    AFF_STORE(0);
    __syncthreads();
    int buf = 0;
    for (int gg = g0; gg < g1; gg += AFF_TG) {
        const bool more = gg + AFF_TG < g1;
        if (more) AFF_LOAD(gg + AFF_TG);
#pragma unroll
        for (int t = 0; t < AFF_TG; ++t) {
            const int g = gg + t;
            if (g < g1) {                                               // block-uniform
                int start, n, lt;
                tile_slot(g, start, n, lt);
                const int row = t * 16 + l15;
                bf16x8 ah[4], al[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    ah[ks] = __builtin_bit_cast(bf16x8, lds[buf][row * 16 + ((ks * 4 + l4) ^ l15)]);
                    al[ks] = __builtin_bit_cast(bf16x8, lds[buf][1024 + row * 16 + ((ks * 4 + l4) ^ l15)]);
                }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {                        // small cross terms first
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], bl[ks], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ks], bh[ks], acc, 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], bh[ks], acc, 0, 0, 0);
                // lane holds tokens lt*16 + l4*4 + q (q = 0..3) of query j
                float s[4];
                float mx = -INFINITY;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int local = lt * 16 + l4 * 4 + q;
                    bool valid = local < n;
                    float sc = valid ? p.scale[start + local] : 0.f;
                    s[q] = valid ? sc * (acc[q] - cj) : -INFINITY;
                    mx = fmaxf(mx, s[q]);
                }
                if (p.mode == 0) {
                    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    if (l4 == 0) p.gmax_or_tau[(long)j * p.Gld + g] = mx;
                } else if (jvalid) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (s[q] >= thr && s[q] > -INFINITY) {
                            int pos = atomicAdd(&p.count[j], 1);
                            if (pos < p.cap) {
                                p.cand_val[(long)j * p.cap + pos] = s[q];
                                p.cand_idx[(long)j * p.cap + pos] = start + lt * 16 + l4 * 4 + q;
                            }
                        }
                    }
                }
            }
        }
        if (more) AFF_STORE(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
}

__device__ __forceinline__ uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// one wave per query column; 4 waves per block.  Exact k-th largest by 4x8-bit radix select.
__global__ __launch_bounds__(256) void aff_select_kernel(const float* __restrict__ gmax, float* __restrict__ tau,
                                                         int HW, int Gld, int G, int k) {
    __shared__ int hist[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    if (j >= HW) return;                                    // whole wave exits together
    if (G < k) { if (lane == 0) tau[j] = -INFINITY; return; }
    uint32_t prefix = 0, mask = 0;
    int remaining = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = lane; b < 256; b += 64) hist[wave][b] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int g = lane; g < G; g += 64) {
            uint32_t key = f2key(gmax[(long)j * Gld + g]);
            if ((key & mask) == prefix) atomicAdd(&hist[wave][(key >> shift) & 255], 1);
        }
        __builtin_amdgcn_wave_barrier();
        int h0 = hist[wave][4 * lane], h1 = hist[wave][4 * lane + 1], h2 = hist[wave][4 * lane + 2], h3 = hist[wave][4 * lane + 3];
        int sl = h0 + h1 + h2 + h3;
        int x = sl;                                          // inclusive suffix scan over lanes (high bins = high lanes)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int yv = __shfl_down(x, off, 64);
            if (lane + off < 64) x += yv;
        }
        int above = x - sl;
        bool mine = (above < remaining) && (remaining <= x);
        int bin = 0, rem_new = 0;
        if (mine) {
            int cum = above;
            int hh[4] = {h0, h1, h2, h3};
            bin = 4 * lane; rem_new = remaining - cum;
#pragma unroll
            for (int t = 3; t >= 0; --t) {
                if (cum + hh[t] >= remaining) { bin = 4 * lane + t; rem_new = remaining - cum; break; }
                cum += hh[t];
            }
        }
        unsigned long long bal = __ballot(mine);
        int src = __ffsll((long long)bal) - 1;
        bin = __shfl(bin, src, 64);
        remaining = __shfl(rem_new, src, 64);
        prefix |= ((uint32_t)bin) << shift;
        mask |= 0xffu << shift;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) tau[j] = key2f(prefix);
}

#define RO_THREADS 128
#define RO_MAXK 64
// one block per query column
__global__ __launch_bounds__(RO_THREADS) void aff_readout_kernel(const float* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                                                 const int* __restrict__ count, const uint64_t* __restrict__ vptrs,
                                                                 float* __restrict__ usage, bf16_t* __restrict__ y, int* __restrict__ overflow,
                                                                 int HW, int cap, int topk, int K, int CV) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* cv = reinterpret_cast<float*>(lds_raw);             // [cap]
    int* ci = reinterpret_cast<int*>(lds_raw + (size_t)cap * 4);  // [cap]
    __shared__ float sel_v[RO_MAXK];
    __shared__ int sel_i[RO_MAXK];
    __shared__ float sel_w[RO_MAXK];
    const int j = blockIdx.x, tid = threadIdx.x;
    int cnt = count[j];
    if (cnt > cap) { if (tid == 0) atomicAdd(overflow, 1); cnt = cap; }
    for (int t = tid; t < cnt; t += RO_THREADS) { cv[t] = cand_val[(long)j * cap + t]; ci[t] = cand_idx[(long)j * cap + t]; }
    const int nsel = min(cnt, topk);
    if (tid < RO_MAXK) { sel_v[tid] = -INFINITY; sel_i[tid] = cnt > 0 ? cand_idx[(long)j * cap] : 0; sel_w[tid] = 0.f; }
    __syncthreads();
    for (int t = tid; t < cnt; t += RO_THREADS) {
        float v = cv[t]; int id = ci[t];
        int rank = 0;
        for (int u = 0; u < cnt; ++u) {
            float w = cv[u];
            rank += (w > v) || (w == v && ci[u] < id);
        }
        if (rank < nsel) { sel_v[rank] = v; sel_i[rank] = id; }
    }
    __syncthreads();
    if (tid < 64) {                                            // softmax over the selected scores (wave 0)
        float e = (tid < nsel && sel_v[tid] > -INFINITY) ? expf(sel_v[tid] - sel_v[0]) : 0.f;   // unassigned rank (duplicate entries): weight 0
        float sum = wave_sum(e);
        if (tid < nsel) {
            float w = e / sum;
            sel_w[tid] = w;
            if (usage) atomicAdd(&usage[sel_i[tid]], w);
        }
    }
    __syncthreads();
    const int C8 = CV >> 3;
    for (int u = tid; u < K * C8; u += RO_THREADS) {
        int o = u / C8, c8 = u - o * C8;
        const bf16_t* V = reinterpret_cast<const bf16_t*>(vptrs[o]);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 6
        for (int t = 0; t < nsel; ++t) {
            uint4 v = *reinterpret_cast<const uint4*>(V + (long)sel_i[t] * CV + c8 * 8);
            const uint32_t* vu = &v.x;
            float w = sel_w[t];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] += w * __uint_as_float(vu[i] << 16);
                acc[2 * i + 1] += w * __uint_as_float(vu[i] & 0xffff0000u);
            }
        }
        *reinterpret_cast<uint4*>(y + ((long)o * HW + j) * CV + c8 * 8) =
            make_uint4(pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3]), pack_bf2(acc[4], acc[5]), pack_bf2(acc[6], acc[7]));
    }
}

int launch_affinity(const cutie_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const uint64_t* p = op->p;
    switch (op->kind) {
        case CUTIE_OP_KEY_PREP: {
            long n = (long)i[0] * 16;
            hipLaunchKernelGGL(key_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)p[0], (const float*)p[1],
                               (bf16_t*)p[2], (bf16_t*)p[3], (float*)p[4], i[0], op->flags & 1);
            break;
        }
        case CUTIE_OP_AFF_SCORE: {
            ScoreParams sp;
            sp.Ahi = (const bf16_t*)p[0]; sp.Alo = (const bf16_t*)p[1]; sp.scale = (const float*)p[2];
            sp.Bhi = (const bf16_t*)p[3]; sp.Blo = (const bf16_t*)p[4]; sp.c = (const float*)p[5];
            sp.gmax_or_tau = (float*)p[6]; sp.cand_val = (float*)p[7]; sp.cand_idx = (int*)p[8]; sp.count = (int*)p[9];
            sp.HW = i[0]; sp.HWp = i[1]; sp.nranges = i[2];
            for (int r = 0; r < 3; ++r) { sp.rs[r] = i[3 + 2 * r]; sp.rn[r] = (r < sp.nranges) ? i[4 + 2 * r] : 0; }
            sp.G = i[9]; sp.cap = i[10]; sp.mode = i[11]; sp.Gld = (sp.G + 63) / 64 * 64;
            int G = 0;
            for (int r = 0; r < 3; ++r) G += (sp.rn[r] + 15) / 16;
            if (G != sp.G || (sp.HWp & 63) || sp.nranges < 1 || sp.nranges > 3) { cutie_set_error("aff_score: bad ranges (G=%d vs %d, HWp=%d)", G, sp.G, sp.HWp); return -2; }
            int qb = sp.HWp / 64;
            // enough blocks to fill 256 CUs a few times over, at least 8 tiles per block
            int tpb = (int)(((long)G * qb + 1023) / 1024);
            if (tpb < 8) tpb = 8;
            tpb = (tpb + AFF_TG - 1) / AFF_TG * AFF_TG;
            sp.tiles_per_block = tpb;
            hipLaunchKernelGGL(aff_score_kernel, dim3(qb, (G + tpb - 1) / tpb), dim3(256), 0, s, sp);
            break;
        }
        case CUTIE_OP_AFF_SELECT:
            hipLaunchKernelGGL(aff_select_kernel, dim3((i[0] + 3) / 4), dim3(256), 0, s, (const float*)p[0], (float*)p[1], i[0], (i[2] + 63) / 64 * 64, i[2], i[3]);
            break;
        case CUTIE_OP_AFF_READOUT: {
            if (i[2] > RO_MAXK || (i[4] & 7)) { cutie_set_error("aff_readout: top_k <= %d, CV %% 8", RO_MAXK); return -2; }
            size_t lds = (size_t)i[1] * 8;
            hipLaunchKernelGGL(aff_readout_kernel, dim3(i[0]), dim3(RO_THREADS), lds, s, (const float*)p[0], (const int*)p[1], (const int*)p[2],
                               (const uint64_t*)p[3], (float*)p[4], (bf16_t*)p[5], (int*)p[6], i[0], i[1], i[2], i[3], i[4]);
            break;
        }
        default:
            cutie_set_error("affinity: unknown op kind %d", op->kind);
            return -3;
    }
    return (int)hipGetLastError();
}
