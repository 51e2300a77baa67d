// Long-term memory maintenance kernels (reference: MemoryManager.consolidation / compress_features,
// memory_manager.py:309-358; KeyValueMemoryStore.remove_obsolete_features, kv_memory_store.py:209-242).
// They run once every ~25 frames on <= 10^4 tokens -- on the caller's stream, in front of the next frame's read-out: rounds 1-4 had
// them as plain fp32 loops (0.55 ms per consolidation); the ranking that decides which prototypes survive stays
// exact (integer counts), the potentiation runs on the matrix cores.
#include "common.h"
#include <math.h>

// ---- RANK_SELECT: order[rank(i)] = i for rank < k, rank by descending use/life, ties -> lower index ---------------------------------
// (torch.topk(usage, k) of memory_manager.py:339 / kv_memory_store.py:222).  Round 5: the all-pairs count is split over RS_SPLIT chunks of
// the compared elements (grid (ceil(n/256), RS_SPLIT): 512 blocks at n = 8100 instead of 32 blocks walking all n elements with a
// division per compared element -- 223 us per call), integer partial ranks to scratch, and a second launch that adds them up in a fixed
// order and scatters -- the same exact order as before.  That second launch also carries the side jobs of a consolidation: up to two
// row gathers through the order it has just produced (dst[rank] = src[i]) and the clearing of a small buffer.
#define RS_SPLIT 16
__global__ __launch_bounds__(256) void rank_part_kernel(const float* __restrict__ use, const float* __restrict__ life,
                                                        int* __restrict__ part, int n) {
    __shared__ __attribute__((aligned(16))) float tile[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ic = min(i, n - 1);
    const float ui = use[ic] / life[ic];
    const int chunk = ((n + RS_SPLIT - 1) / RS_SPLIT + 3) & ~3;
    const int j0 = blockIdx.y * chunk, j1 = min(n, j0 + chunk);
    int rank = 0;
    for (int base = j0; base < j1; base += 256) {
        const int l = base + threadIdx.x, lc = min(l, n - 1);
        const float ul = use[lc] / life[lc];
        tile[threadIdx.x] = l < j1 ? ul : -INFINITY;         // padding never counts: -inf is neither above nor equal to a usage
        __syncthreads();
        const int lim = min(256, j1 - base);
        for (int t = 0; t < lim; t += 4) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(tile + t);
#pragma unroll
            for (int e = 0; e < 4; ++e) rank += (int)(u[e] > ui) | ((int)(u[e] == ui) & (int)(base + t + e < i));
        }
        __syncthreads();
    }
    if (i < n) part[(long)blockIdx.y * n + i] = rank;
}

struct RankSide { const uint32_t* src[2]; uint32_t* dst[2]; int roww[2]; uint32_t* zero; int nzero; };
__global__ __launch_bounds__(256) void rank_scatter_kernel(const int* __restrict__ part, int* __restrict__ order, int n, int k, RankSide sd) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (sd.zero) for (int q = i; q < sd.nzero; q += gridDim.x * 256) sd.zero[q] = 0u;
    if (i >= n) return;
    int rank = 0;
#pragma unroll
    for (int sp = 0; sp < RS_SPLIT; ++sp) rank += part[(long)sp * n + i];
    if (rank >= k) return;
    order[rank] = i;
#pragma unroll
    for (int a = 0; a < 2; ++a)
        if (sd.src[a]) {
            const uint4* s4 = reinterpret_cast<const uint4*>(sd.src[a] + (long)i * sd.roww[a]);
            uint4* d4 = reinterpret_cast<uint4*>(sd.dst[a] + (long)rank * sd.roww[a]);
            for (int c = 0; c < sd.roww[a] / 4; ++c) d4[c] = s4[c];
        }
}

__global__ void gather_rows_kernel(const uint32_t* __restrict__ src, const int* __restrict__ order, uint32_t* __restrict__ dst,
                                   int k, int roww, long ss, long ds) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)k * roww) return;
    int r = idx / roww, c = idx - (long)r * roww;
    dst[(long)r * ds + c] = src[(long)order[r] * ss + c];
}

// ---- long-term consolidation (memory_manager.py:329-358): prototype p = softmax over the candidates of sim(cand_i, proto_p), applied to
// the candidates' values and shrinkages.  Round 5: three launches for all objects instead of 1 + 2 K + 1 --
//   CONSOL_AFF   S[p][i] = sim on v_mfma_f32_16x16x4_f32 (fp32 products, fp32 sums) + the column maxima by integer atomicMax on
//                order-preserving keys (exact and order-independent)
//   CONSOL_READ  a) per chunk of 256 candidates, object and half of the channels: w = exp(S - max) split into bf16 hi + lo, O^T = V^T . w
//                   on v_mfma_f32_16x16x32_bf16 (V is bf16 in the bank: the products are exact, the split keeps 16 bits of w), partial sums
//                   of w and of w . shrinkage;  b) the partials added up in chunk order, divided by the sum of w, stored (deterministic).
__device__ __forceinline__ uint32_t bank_f2key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float bank_key2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// grid ceil(n / 32); block = 32 candidates x all prototypes; wave w owns the 16-prototype tiles w, w + 4, ...  The MFMA k slot (step c,
// lane group g) carries channel g * 16 + c: every lane reads 16 consecutive floats of its row.
__global__ __launch_bounds__(256) void consol_sim_kernel(const float* __restrict__ ckey, const float* __restrict__ cshr,
                                                         const float* __restrict__ pkey, const float* __restrict__ psel,
                                                         float* __restrict__ S, uint32_t* __restrict__ colmax, int n, int P, int ldS) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int i0 = blockIdx.x * 32;
    float ka[2][16];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const float4* src = reinterpret_cast<const float4*>(ckey + (long)min(i0 + rt * 16 + r, n - 1) * 64 + g * 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float4 v = src[c]; ka[rt][4 * c] = v.x; ka[rt][4 * c + 1] = v.y; ka[rt][4 * c + 2] = v.z; ka[rt][4 * c + 3] = v.w; }
    }
    f32x4 shr[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) shr[rt][q] = cshr[min(i0 + rt * 16 + 4 * g + q, n - 1)];
    for (int pt = wave; pt * 16 < P; pt += 4) {
        const int p = pt * 16 + r, pc = min(p, P - 1);
        float kp[16], ep[16];
        const float4* k4 = reinterpret_cast<const float4*>(pkey + (long)pc * 64 + g * 16);
        const float4* e4 = reinterpret_cast<const float4*>(psel + (long)pc * 64 + g * 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 a = k4[c], b = e4[c];
            kp[4 * c] = a.x; kp[4 * c + 1] = a.y; kp[4 * c + 2] = a.z; kp[4 * c + 3] = a.w;
            ep[4 * c] = b.x; ep[4 * c + 1] = b.y; ep[4 * c + 2] = b.z; ep[4 * c + 3] = b.w;
        }
        float bsq = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) bsq += ep[c] * kp[c] * kp[c];
        bsq = rows_sum(bsq);                                     // over the four channel groups of this prototype
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[rt][c] * ka[rt][c], -ep[c], acc[rt], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[rt][c], 2.f * kp[c] * ep[c], acc[rt], 0, 0, 0);
        float mx = -INFINITY;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int ib = i0 + rt * 16 + 4 * g;                 // lane holds candidates ib .. ib + 3 for prototype p
            f32x4 sv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = (acc[rt][q] - bsq) * shr[rt][q] * 0.125f;
                sv[q] = ib + q < n ? v : -INFINITY;              // rows of the padding up to ldS: weight 0
                mx = fmaxf(mx, sv[q]);
            }
            if (p < P && ib < ldS) *reinterpret_cast<f32x4*>(S + (long)p * ldS + ib) = sv;
        }
        mx = rows_max(mx);
        if (g == 0 && p < P) atomicMax(colmax + p, bank_f2key(mx));
    }
}

#define CR_CH 256                                      // candidates per block of CONSOL_READ's first launch (8 steps of 32)
#define CR_VT 40                                       // LDS pitch (bf16 elements) of a transposed value row: 32 candidates + pad, 80 B = 5 x 16 B
struct ConsolRead {
    const float* S; const uint32_t* colmax; const uint64_t* vptrs; const float* cshr;
    float* opart; float* zpart; float* spart;
    int n, P, C, K, ldS, src, nchunk, p0;
};
// grid (nchunk, K * C / 128): block = one chunk of candidates x one object x 128 channels x the prototypes p0 .. p0 + 127.
// wave w: channels w * 32 .. + 31 of the half (two 16-channel MFMA tiles) x 8 prototype tiles.
__global__ __launch_bounds__(256) void consol_read_part_kernel(ConsolRead a) {
    __shared__ __attribute__((aligned(16))) bf16_t vt[2][128 * CR_VT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int chunk = blockIdx.x, halves = a.C / 128, o = blockIdx.y / halves, half = blockIdx.y - o * halves;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.vptrs[o]) + (long)a.src * a.C + half * 128;
    const bool lead = blockIdx.y == 0 && wave == 0;         // writes the sums of w and of w . shrinkage
    const int np = min(128, a.P - a.p0);
    float m[8];
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) m[pt] = bank_key2f(a.colmax[min(a.p0 + pt * 16 + r, a.P - 1)]);
    f32x4 acc[2][8];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int ii = tid >> 3, cseg = tid & 7;                 // staging: candidate row of the step, 16-channel segment
    const int i_first = chunk * CR_CH;
    const int nsteps = min(CR_CH, a.ldS - i_first) / 32;     // (ldS is a multiple of 32)
    uint4 st0, st1;
    auto load = [&](int stp) {
        const bf16_t* src = V + (long)min(i_first + stp * 32 + ii, a.n - 1) * a.C + cseg * 16;
        st0 = *reinterpret_cast<const uint4*>(src);
        st1 = *reinterpret_cast<const uint4*>(src + 8);
    };
    auto store = [&](int buf) {                              // transposed: vt[channel][candidate]
        const uint32_t w[8] = {st0.x, st0.y, st0.z, st0.w, st1.x, st1.y, st1.z, st1.w};
        // (16 channel rows x 80 B = 20 x 64 banks: the eight channel segments of a wave would all land on one bank -- the PMC showed
        // SQ_LDS_BANK_CONFLICT / IDX_ACTIVE = 0.83 for this kernel; the 8-candidate column groups of a row are XOR-ed with the row's segment)
        const int col = (((ii >> 3) ^ (cseg & 3)) << 3) | (ii & 7);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            vt[buf][(cseg * 16 + 2 * e) * CR_VT + col] = (bf16_t)(w[e] & 0xffffu);
            vt[buf][(cseg * 16 + 2 * e + 1) * CR_VT + col] = (bf16_t)(w[e] >> 16);
        }
    };
    load(0);
    store(0);
    __syncthreads();
    for (int stp = 0; stp < nsteps; ++stp) {
        const int buf = stp & 1, i0 = i_first + stp * 32;
        if (stp + 1 < nsteps) load(stp + 1);
        bf16x8 av[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
            av[ct] = *reinterpret_cast<const bf16x8*>(&vt[buf][(wave * 32 + ct * 16 + r) * CR_VT + 8 * (g ^ ((wave * 2 + ct) & 3))]);   // (row's segment = channel >> 4)
        f32x4 sr0 = {0.f, 0.f, 0.f, 0.f}, sr1 = {0.f, 0.f, 0.f, 0.f};
        if (lead) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { sr0[q] = a.cshr[min(i0 + 8 * g + q, a.n - 1)]; sr1[q] = a.cshr[min(i0 + 8 * g + 4 + q, a.n - 1)]; }
        }
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) {
            if (pt * 16 >= np) break;                          // block-uniform
            const float* srow = a.S + (long)min(a.p0 + pt * 16 + r, a.P - 1) * a.ldS + i0 + 8 * g;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(srow), s1 = *reinterpret_cast<const f32x4*>(srow + 4);
            float w[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { w[q] = expf(s0[q] - m[pt]); w[4 + q] = expf(s1[q] - m[pt]); }
            bf16x8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) { const __bf16 h = (__bf16)w[q]; hi[q] = h; lo[q] = (__bf16)(w[q] - (float)h); z[pt] += w[q]; }
            if (lead) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { sh[pt] += w[q] * sr0[q]; sh[pt] += w[4 + q] * sr1[q]; }
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ct], hi, acc[ct][pt], 0, 0, 0);
                acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ct], lo, acc[ct][pt], 0, 0, 0);
            }
        }
        if (stp + 1 < nsteps) store(buf ^ 1);
        __syncthreads();
    }
    // partial sums: opart [chunk][object][prototype][channel]; lane holds channels 4 g .. 4 g + 3 of its tile for prototype pt * 16 + r
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
        const int p = a.p0 + pt * 16 + r;
        if (pt * 16 >= np || p >= a.P) continue;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
            *reinterpret_cast<f32x4*>(a.opart + (((long)chunk * a.K + o) * a.P + p) * a.C + half * 128 + wave * 32 + ct * 16 + 4 * g) = acc[ct][pt];
        if (lead) {
            const float zz = rows_sum(z[pt]), ss = rows_sum(sh[pt]);
            if (g == 0) { a.zpart[(long)chunk * a.P + p] = zz; a.spart[(long)chunk * a.P + p] = ss; }
        }
    }
}

// one thread per (object, prototype, 8 channels) + one per prototype for the shrinkage; chunk order = fixed order
__global__ void consol_read_comb_kernel(const float* __restrict__ opart, const float* __restrict__ zpart, const float* __restrict__ spart,
                                        const uint64_t* __restrict__ vptrs, float* __restrict__ out_shr, int P, int C, int K, int nchunk, int dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, C8 = C >> 3;
    if (idx < K * P * C8) {
        const int o = idx / (P * C8), rem = idx - o * P * C8, p = rem / C8, c8 = rem - p * C8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, zs = 0.f;
        for (int ch = 0; ch < nchunk; ++ch) {
            const float* src = opart + (((long)ch * K + o) * P + p) * C + c8 * 8;
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
            zs += zpart[(long)ch * P + p];
        }
        const float inv = 1.f / zs;
        bf16_t* out = reinterpret_cast<bf16_t*>(vptrs[o]) + (long)(dst + p) * C + c8 * 8;
        *reinterpret_cast<uint4*>(out) = make_uint4(pack_bf2(acc[0] * inv, acc[1] * inv), pack_bf2(acc[2] * inv, acc[3] * inv),
                                                    pack_bf2(acc[4] * inv, acc[5] * inv), pack_bf2(acc[6] * inv, acc[7] * inv));
    } else if (idx < K * P * C8 + P && out_shr) {
        const int p = idx - K * P * C8;
        float zs = 0.f, ss = 0.f;
        for (int ch = 0; ch < nchunk; ++ch) { zs += zpart[(long)ch * P + p]; ss += spart[(long)ch * P + p]; }
        out_shr[p] = ss / zs;
    }
}

int launch_bank(const cutie_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const uint64_t* p = op->p;
    switch (op->kind) {
        case CUTIE_OP_RANK_SELECT: {
            if (!p[3]) { cutie_set_error("rank_select: needs its partial-rank scratch (p3, %d x n ints)", RS_SPLIT); return -2; }
            RankSide sd = {{(const uint32_t*)p[4], (const uint32_t*)p[6]}, {(uint32_t*)p[5], (uint32_t*)p[7]}, {i[2], i[3]}, (uint32_t*)p[8], i[4]};
            if ((p[4] && (i[2] & 3)) || (p[6] && (i[3] & 3))) { cutie_set_error("rank_select: gathered rows must be whole 16-B words"); return -2; }
            const int nb = (i[0] + 255) / 256;
            hipLaunchKernelGGL(rank_part_kernel, dim3(nb, RS_SPLIT), dim3(256), 0, s, (const float*)p[0], (const float*)p[1], (int*)p[3], i[0]);
            hipLaunchKernelGGL(rank_scatter_kernel, dim3(nb), dim3(256), 0, s, (const int*)p[3], (int*)p[2], i[0], i[1], sd);
            break;
        }
        case CUTIE_OP_GATHER_ROWS: {
            int roww = i[1] / 4;
            long n = (long)i[0] * roww;
            hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint32_t*)p[0], (const int*)p[1],
                               (uint32_t*)p[2], i[0], roww, (long)i[2] / 4, (long)i[3] / 4);
            break;
        }
        case CUTIE_OP_CONSOL_AFF: {
            if ((i[2] & 31) || i[2] < i[0]) { cutie_set_error("consol_aff: ldS=%d must be a multiple of 32 >= n=%d", i[2], i[0]); return -2; }
            hipLaunchKernelGGL(consol_sim_kernel, dim3(i[2] / 32), dim3(256), 0, s, (const float*)p[0], (const float*)p[1], (const float*)p[2],
                               (const float*)p[3], (float*)p[4], (uint32_t*)p[5], i[0], i[1], i[2]);
            break;
        }
        case CUTIE_OP_CONSOL_READ: {
            ConsolRead a;
            a.S = (const float*)p[0]; a.colmax = (const uint32_t*)p[1]; a.vptrs = (const uint64_t*)p[2]; a.cshr = (const float*)p[3];
            a.n = i[0]; a.P = i[1]; a.C = i[2]; a.K = i[3]; a.ldS = i[4]; a.src = i[5];
            a.nchunk = (a.ldS + CR_CH - 1) / CR_CH;
            if ((a.C & 127) || (a.ldS & 31) || a.ldS < a.n || a.K < 1) { cutie_set_error("consol_read: C %% 128, ldS %% 32 (C=%d ldS=%d n=%d)", a.C, a.ldS, a.n); return -2; }
            float* part = (float*)p[4];                          // [nchunk][K][P][C] + 2 x [nchunk][P]
            a.opart = part; a.zpart = part + (long)a.nchunk * a.K * a.P * a.C; a.spart = a.zpart + (long)a.nchunk * a.P;
            for (a.p0 = 0; a.p0 < a.P; a.p0 += 128)
                hipLaunchKernelGGL(consol_read_part_kernel, dim3(a.nchunk, a.K * (a.C / 128)), dim3(256), 0, s, a);
            const int nt = a.K * a.P * (a.C >> 3) + a.P;
            hipLaunchKernelGGL(consol_read_comb_kernel, dim3((nt + 255) / 256), dim3(256), 0, s, (const float*)a.opart, (const float*)a.zpart,
                               (const float*)a.spart, a.vptrs, (float*)p[5], a.P, a.C, a.K, a.nchunk, i[6]);
            break;
        }
        default:
            cutie_set_error("bank: unknown op kind %d", op->kind);
            return -3;
    }
    return (int)hipGetLastError();
}
