// Long-term memory maintenance kernels (reference: MemoryManager.consolidation / compress_features,
// memory_manager.py:309-358; KeyValueMemoryStore.remove_obsolete_features, kv_memory_store.py:209-242).
// These run once every ~25 frames on <= 10^4 tokens, so they are plain fp32 kernels (exact arithmetic matters
// more than speed here: they decide which prototypes survive).
#include "common.h"
#include <math.h>

// order[rank(i)] = i for rank < k, rank by descending use/life, ties -> lower index
__global__ __launch_bounds__(256) void rank_select_kernel(const float* __restrict__ use, const float* __restrict__ life,
                                                          int* __restrict__ order, int n, int k) {
    __shared__ float tile[256];
    int i = blockIdx.x * 256 + threadIdx.x;
    float ui = (i < n) ? use[i] / life[i] : 0.f;
    int rank = 0;
    for (int base = 0; base < n; base += 256) {
        int l = base + threadIdx.x;
        tile[threadIdx.x] = (l < n) ? use[l] / life[l] : -INFINITY;
        __syncthreads();
        int lim = min(256, n - base);
        for (int t = 0; t < lim; ++t) {
            float u = tile[t];
            rank += (u > ui) || (u == ui && (base + t) < i);
        }
        __syncthreads();
    }
    if (i < n && rank < k) order[rank] = i;
}

__global__ void gather_rows_kernel(const uint32_t* __restrict__ src, const int* __restrict__ order, uint32_t* __restrict__ dst,
                                   int k, int roww, long ss, long ds) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)k * roww) return;
    int r = idx / roww, c = idx - (long)r * roww;
    dst[(long)r * ds + c] = src[(long)order[r] * ss + c];
}

// one block per prototype: aff[p,:] = softmax_i(sim(cand_i, proto_p))
__global__ __launch_bounds__(256) void consol_aff_kernel(const float* __restrict__ ckey, const float* __restrict__ cshr,
                                                         const float* __restrict__ pkey, const float* __restrict__ psel,
                                                         float* __restrict__ aff, int n) {
    __shared__ float pk[64], pe[64], red[256];
    __shared__ float bsq_s;
    const int p = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) { pk[tid] = pkey[(long)p * 64 + tid]; pe[tid] = psel[(long)p * 64 + tid]; }
    __syncthreads();
    if (tid == 0) {
        float b = 0.f;
        for (int c = 0; c < 64; ++c) b += pe[c] * pk[c] * pk[c];
        bsq_s = b;
    }
    __syncthreads();
    const float bsq = bsq_s;
    float* row = aff + (long)p * n;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 256) {
        const float* k = ckey + (long)i * 64;
        float asq = 0.f, ab = 0.f;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) { float kv = k[c]; asq += kv * kv * pe[c]; ab += kv * pk[c] * pe[c]; }
        float s = (-asq + 2.f * ab - bsq) * cshr[i] * 0.125f;
        row[i] = s;
        mx = fmaxf(mx, s);
    }
    red[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) { float e = expf(row[i] - mx); row[i] = e; sum += e; }
    red[tid] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    float inv = 1.f / red[0];
    for (int i = tid; i < n; i += 256) row[i] *= inv;
}

// grid (ceil(C/64), ceil(P/4)); block 256 = 4 prototypes x 64 channels
template <bool F32>
__global__ __launch_bounds__(256) void consol_read_kernel(const float* __restrict__ aff, const void* __restrict__ V, void* __restrict__ out,
                                                          int n, int P, int C, int ldv, int ldo) {
    __shared__ float a[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), pp = threadIdx.x >> 6, p = blockIdx.y * 4 + pp;
    float acc = 0.f;
    for (int base = 0; base < n; base += 64) {
        int l = base + (threadIdx.x & 63);
        a[pp][threadIdx.x & 63] = (p < P && l < n) ? aff[(long)p * n + l] : 0.f;
        __syncthreads();
        int lim = min(64, n - base);
        if (c < C)
            for (int t = 0; t < lim; ++t) {
                float v = F32 ? ((const float*)V)[(long)(base + t) * ldv + c] : bf2f(((const bf16_t*)V)[(long)(base + t) * ldv + c]);
                acc += a[pp][t] * v;
            }
        __syncthreads();
    }
    if (p < P && c < C) {
        if (F32) ((float*)out)[(long)p * ldo + c] = acc;
        else ((bf16_t*)out)[(long)p * ldo + c] = f2bf(acc);
    }
}

// Fast path of CONSOL_READ for bf16 value banks (C % 8 == 0, C <= 256): the n candidates are split over CR_NS blocks per group
// of 8 prototypes (256 blocks instead of 128 threads-bound ones looping over all 8100 candidates: 840 us -> ~10 us per
// object), 16-B value loads shared by the 8 prototypes of a block, fp32 partial sums, then a fixed-order sum (deterministic).
#define CR_NS 16
__global__ __launch_bounds__(256) void consol_read_part_kernel(const float* __restrict__ aff, const bf16_t* __restrict__ V,
                                                               float* __restrict__ part, int n, int P, int C, int ldv) {
    const int c8 = threadIdx.x & 31, pq = threadIdx.x >> 5, p = blockIdx.x * 8 + pq, ns = blockIdx.y;
    const int chunk = (n + CR_NS - 1) / CR_NS, t0 = ns * chunk, t1 = min(n, t0 + chunk);
    const bool live = p < P && c8 * 8 < C;
    const float* ar = aff + (long)(live ? p : 0) * n;
    const bf16_t* vr = V + (live ? c8 * 8 : 0);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int t = t0; t < t1; ++t) {
        const float a = ar[t];
        const uint4 v = *reinterpret_cast<const uint4*>(vr + (long)t * ldv);
        const uint32_t* vu = &v.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[2 * i] += a * __uint_as_float(vu[i] << 16);
            acc[2 * i + 1] += a * __uint_as_float(vu[i] & 0xffff0000u);
        }
    }
    if (live) {
        float* dst = part + ((long)ns * P + p) * C + c8 * 8;
        *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

__global__ void consol_read_sum_kernel(const float* __restrict__ part, bf16_t* __restrict__ out, int P, int C, int ldo) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, C8 = C >> 3;
    if (idx >= P * C8) return;
    const int p = idx / C8, c8 = idx - p * C8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int ns = 0; ns < CR_NS; ++ns) {
        const float* src = part + ((long)ns * P + p) * C + c8 * 8;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
    }
    *reinterpret_cast<uint4*>(out + (long)p * ldo + c8 * 8) =
        make_uint4(pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3]), pack_bf2(acc[4], acc[5]), pack_bf2(acc[6], acc[7]));
}

// CONSOL_READ with C == 1 (prototype shrinkage): one block per prototype, strided partial sums + fixed-order tree (deterministic)
__global__ __launch_bounds__(256) void consol_dot_kernel(const float* __restrict__ aff, const float* __restrict__ v, float* __restrict__ out,
                                                         int n, int ldv, int ldo) {
    __shared__ float red[256];
    const int p = blockIdx.x;
    float acc = 0.f;
    for (int t = threadIdx.x; t < n; t += 256) acc += aff[(long)p * n + t] * v[(long)t * ldv];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(long)p * ldo] = red[0];
}

int launch_bank(const cutie_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const uint64_t* p = op->p;
    switch (op->kind) {
        case CUTIE_OP_RANK_SELECT:
            hipLaunchKernelGGL(rank_select_kernel, dim3((i[0] + 255) / 256), dim3(256), 0, s, (const float*)p[0], (const float*)p[1], (int*)p[2], i[0], i[1]);
            break;
        case CUTIE_OP_GATHER_ROWS: {
            int roww = i[1] / 4;
            long n = (long)i[0] * roww;
            hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint32_t*)p[0], (const int*)p[1],
                               (uint32_t*)p[2], i[0], roww, (long)i[2] / 4, (long)i[3] / 4);
            break;
        }
        case CUTIE_OP_CONSOL_AFF:
            hipLaunchKernelGGL(consol_aff_kernel, dim3(i[1]), dim3(256), 0, s, (const float*)p[0], (const float*)p[1], (const float*)p[2],
                               (const float*)p[3], (float*)p[4], i[0]);
            break;
        case CUTIE_OP_CONSOL_READ: {
            if (!(op->flags & 1) && p[3] && (i[2] & 7) == 0 && i[2] <= 256 && (i[3] & 7) == 0 && (i[4] & 7) == 0) {
                hipLaunchKernelGGL(consol_read_part_kernel, dim3((i[1] + 7) / 8, CR_NS), dim3(256), 0, s, (const float*)p[0], (const bf16_t*)p[1],
                                   (float*)p[3], i[0], i[1], i[2], i[3]);
                const int nq = i[1] * (i[2] >> 3);
                hipLaunchKernelGGL(consol_read_sum_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, (const float*)p[3], (bf16_t*)p[2], i[1], i[2], i[4]);
                break;
            }
            if ((op->flags & 1) && i[2] == 1) {
                hipLaunchKernelGGL(consol_dot_kernel, dim3(i[1]), dim3(256), 0, s, (const float*)p[0], (const float*)p[1], (float*)p[2], i[0], i[3], i[4]);
                break;
            }
            dim3 grid((i[2] + 63) / 64, (i[1] + 3) / 4);
            if (op->flags & 1)
                hipLaunchKernelGGL(consol_read_kernel<true>, grid, dim3(256), 0, s, (const float*)p[0], (const void*)p[1], (void*)p[2], i[0], i[1], i[2], i[3], i[4]);
            else
                hipLaunchKernelGGL(consol_read_kernel<false>, grid, dim3(256), 0, s, (const float*)p[0], (const void*)p[1], (void*)p[2], i[0], i[1], i[2], i[3], i[4]);
            break;
        }
        default:
            cutie_set_error("bank: unknown op kind %d", op->kind);
            return -3;
    }
    return (int)hipGetLastError();
}
