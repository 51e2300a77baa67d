// HBM-bound kernels of the hot path: layout/normalisation, pooling, resampling, gates, epilogues.
// All are coalesced along the channel (NHWC) or pixel (planar masks) axis with 16-byte accesses
// where the layout allows; see include/cutie_hip.h for the op contracts and reference citations.
#include "common.h"
#include <math.h>

#define GRID1D(n, bs) dim3((unsigned)(((long)(n) + (bs) - 1) / (bs)))
__global__ void memset32_kernel(uint32_t* d, long n, uint32_t v);

// ---------------------------------------------------------------------------------------------
__global__ void maxpool_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int B, int H, int W, int C8,
                               int OH, int OW, int relu) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * OH * OW * C8;
    if (idx >= total) return;
    int c = idx % C8; long t = idx / C8;
    int ow = t % OW; t /= OW; int oh = t % OH; int b = t / OH;
    float m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = relu ? 0.f : -INFINITY;
    for (int dy = 0; dy < 3; ++dy) {
        int ih = oh * 2 - 1 + dy;
        if ((unsigned)ih >= (unsigned)H) continue;
        for (int dx = 0; dx < 3; ++dx) {
            int iw = ow * 2 - 1 + dx;
            if ((unsigned)iw >= (unsigned)W) continue;
            uint4 v = x[(((long)b * H + ih) * W + iw) * C8 + c];
            uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                m[2 * i] = fmaxf(m[2 * i], __uint_as_float(u[i] << 16));
                m[2 * i + 1] = fmaxf(m[2 * i + 1], __uint_as_float(u[i] & 0xffff0000u));
            }
        }
    }
    y[idx] = make_uint4(pack_bf2(m[0], m[1]), pack_bf2(m[2], m[3]), pack_bf2(m[4], m[5]), pack_bf2(m[6], m[7]));
}

// ---------------------------------------------------------------------------------------------
__global__ void img_prep_kernel(const float* __restrict__ img, const float* __restrict__ masks, uint4* __restrict__ y,
                                int h0, int w0, int H, int W, int pl, int pt, int K,
                                float m0, float m1, float m2, float s0, float s1, float s2) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long HW = (long)H * W;
    if (idx >= HW * K) return;
    int k = idx / HW; long pix = idx - (long)k * HW;
    int yy = pix / W, xx = pix - (long)yy * W;
    int sy = yy - pt, sx = xx - pl;
    float r = 0.f, g = 0.f, b = 0.f;
    if ((unsigned)sy < (unsigned)h0 && (unsigned)sx < (unsigned)w0) {
        long o = (long)sy * w0 + sx, plane = (long)h0 * w0;
        r = img[o]; g = img[plane + o]; b = img[2 * plane + o];
    }
    r = (r - m0) / s0; g = (g - m1) / s1; b = (b - m2) / s2;
    float mk = 0.f, others = 0.f;
    if (masks) {
        float sum = 0.f;
        for (int j = 0; j < K; ++j) sum += masks[(long)j * HW + pix];
        mk = masks[(long)k * HW + pix];
        others = fminf(fmaxf(sum - mk, 0.f), 1.f);
    }
    y[idx] = make_uint4(pack_bf2(r, g), pack_bf2(b, mk), pack_bf2(others, 0.f), 0u);
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void up_coord(int o, int n_in, float scale, int& i0, int& i1, float& lam) {
    float src = ((float)o + 0.5f) * scale - 0.5f;          // torch: area_pixel_compute_source_index
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    lam = src - (float)i0;
}

// Kg > 0 (clips in lock step): the objects come in groups of Kg, group q adds the skip map at skip + q * gstride pixels (Kg = 0: one map for all)
__global__ void upsample2x_add_kernel(const uint4* __restrict__ g, const uint4* __restrict__ skip, uint4* __restrict__ y,
                                      int B, int h, int w, int C8, int Kg, long gstride, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(1);               // CUTIE_F_PRIO: a launch of the frame's critical path (the plan decides, ops.OpList.prio)
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int OH = 2 * h, OW = 2 * w;
    long total = (long)B * OH * OW * C8;
    if (idx >= total) return;
    int c = idx % C8; long t = idx / C8;
    int ox = t % OW; t /= OW; int oy = t % OH; int b = t / OH;
    int y0, y1, x0, x1; float ly, lx;
    up_coord(oy, h, 0.5f, y0, y1, ly);
    up_coord(ox, w, 0.5f, x0, x1, lx);
    const uint4* gb = g + (long)b * h * w * C8;
    uint4 v00 = gb[((long)y0 * w + x0) * C8 + c], v01 = gb[((long)y0 * w + x1) * C8 + c];
    uint4 v10 = gb[((long)y1 * w + x0) * C8 + c], v11 = gb[((long)y1 * w + x1) * C8 + c];
    uint4 sk = skip[((Kg > 0 ? (b / Kg) * gstride : 0l) + (long)oy * OW + ox) * C8 + c];
    const uint32_t* a = &v00.x; const uint32_t* bq = &v01.x; const uint32_t* cq = &v10.x; const uint32_t* d = &v11.x;
    const uint32_t* s = &sk.x;
    float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    uint32_t out[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float lo = w00 * __uint_as_float(a[i] << 16) + w01 * __uint_as_float(bq[i] << 16) +
                   w10 * __uint_as_float(cq[i] << 16) + w11 * __uint_as_float(d[i] << 16) + __uint_as_float(s[i] << 16);
        float hi = w00 * __uint_as_float(a[i] & 0xffff0000u) + w01 * __uint_as_float(bq[i] & 0xffff0000u) +
                   w10 * __uint_as_float(cq[i] & 0xffff0000u) + w11 * __uint_as_float(d[i] & 0xffff0000u) +
                   __uint_as_float(s[i] & 0xffff0000u);
        out[i] = pack_bf2(lo, hi);
    }
    y[idx] = make_uint4(out[0], out[1], out[2], out[3]);
}

// ---------------------------------------------------------------------------------------------
// R > 0: the pooling ratio as a compile-time constant (2, 4): the R x R loads are issued together and summed in the same (dy, dx) order.
// With a run-time r the compiler keeps `load; s_waitcnt vmcnt(0); add` per tap -- r * r dependent round trips per thread
// (tools/isa_waits.py); same bits either way.
template <int R = 0>
__device__ __forceinline__ void area_down_bf16_body(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int C,
                                                    int ldx, int ldy, int rrt, long idx) {
    const int r = R > 0 ? R : rrt;
    int C8 = C >> 3, OH = H / r, OW = W / r;
    long total = (long)B * OH * OW * C8;
    if (idx >= total) return;
    int c = idx % C8; long t = idx / C8;
    int ox = t % OW; t /= OW; int oy = t % OH; int b = t / OH;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (R > 0) {
        uint4 v[R > 0 ? R * R : 1];
#pragma unroll
        for (int dy = 0; dy < R; ++dy)
#pragma unroll
            for (int dx = 0; dx < R; ++dx)
                v[dy * R + dx] = *reinterpret_cast<const uint4*>(x + (((long)b * H + oy * R + dy) * W + ox * R + dx) * ldx + c * 8);
#pragma unroll
        for (int e = 0; e < R * R; ++e) {
            const uint32_t* u = &v[e].x;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] += __uint_as_float(u[i] << 16);
                acc[2 * i + 1] += __uint_as_float(u[i] & 0xffff0000u);
            }
        }
    } else
    for (int dy = 0; dy < r; ++dy)
        for (int dx = 0; dx < r; ++dx) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + (((long)b * H + oy * r + dy) * W + ox * r + dx) * ldx + c * 8);
            const uint32_t* u = &v.x;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] += __uint_as_float(u[i] << 16);
                acc[2 * i + 1] += __uint_as_float(u[i] & 0xffff0000u);
            }
        }
    float inv = 1.f / (float)(r * r);
    uint4 o = make_uint4(pack_bf2(acc[0] * inv, acc[1] * inv), pack_bf2(acc[2] * inv, acc[3] * inv),
                         pack_bf2(acc[4] * inv, acc[5] * inv), pack_bf2(acc[6] * inv, acc[7] * inv));
    *reinterpret_cast<uint4*>(y + (((long)b * OH + oy) * OW + ox) * ldy + c * 8) = o;
}
__global__ void area_down_bf16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int C,
                                      int ldx, int ldy, int r) {
    area_down_bf16_body(x, y, B, H, W, C, ldx, ldy, r, (long)blockIdx.x * blockDim.x + threadIdx.x);
}

template <int R = 0>
__device__ __forceinline__ void area_down_f32_body(const float* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int C,
                                                   int ldx, int ldy, int rrt, int Cz, long idx) {
    const int r = R > 0 ? R : rrt;
    int OH = H / r, OW = W / r;
    long total = (long)B * OH * OW;
    if (idx >= total) return;
    int ox = idx % OW; long t = idx / OW; int oy = t % OH; int b = t / OH;
    float inv = 1.f / (float)(r * r);
    bf16_t* yp = y + idx * ldy;
    for (int c = 0; c < C; ++c) {
        float acc = 0.f;
        if (R > 0) {
            float v[R > 0 ? R * R : 1];
#pragma unroll
            for (int dy = 0; dy < R; ++dy)
#pragma unroll
                for (int dx = 0; dx < R; ++dx) v[dy * R + dx] = x[(((long)b * H + oy * R + dy) * W + ox * R + dx) * ldx + c];
#pragma unroll
            for (int e = 0; e < R * R; ++e) acc += v[e];
        } else
        for (int dy = 0; dy < r; ++dy)
            for (int dx = 0; dx < r; ++dx) acc += x[(((long)b * H + oy * r + dy) * W + ox * r + dx) * ldx + c];
        yp[c] = f2bf(acc * inv);
    }
    for (int c = C; c < Cz; ++c) yp[c] = 0;
}
__global__ void area_down_f32_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int C,
                                     int ldx, int ldy, int r, int Cz) {
    area_down_f32_body(x, y, B, H, W, C, ldx, ldy, r, Cz, (long)blockIdx.x * blockDim.x + threadIdx.x);
}
// AREA_DOWN3: three area poolings in one launch (the g8 / g4 / logits inputs of the sensory update): thread ranges [0, n0), [n0, n0+n1), ...
struct AreaSeg { const void* x; bf16_t* y; int B, H, W, C, ldx, ldy, r, Cz, f32; long n; };
struct Area3 { AreaSeg s[3]; int rt; int prio; };
__global__ void area_down3_kernel(Area3 a) {
    const int prio = a.prio;
    if (prio) __builtin_amdgcn_s_setprio(1);               // CUTIE_F_PRIO: a launch of the frame's critical path (the plan decides, ops.OpList.prio)
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const AreaSeg& g = a.s[q];
        if (idx < g.n) {
            const int rr = a.rt ? 0 : g.r;                  // (a.rt: the run-time-r bodies, A/B switch)
            if (g.f32) {
                if (rr == 2) area_down_f32_body<2>((const float*)g.x, g.y, g.B, g.H, g.W, g.C, g.ldx, g.ldy, 2, g.Cz, idx);
                else if (rr == 4) area_down_f32_body<4>((const float*)g.x, g.y, g.B, g.H, g.W, g.C, g.ldx, g.ldy, 4, g.Cz, idx);
                else area_down_f32_body<0>((const float*)g.x, g.y, g.B, g.H, g.W, g.C, g.ldx, g.ldy, g.r, g.Cz, idx);
            } else {
                if (rr == 2) area_down_bf16_body<2>((const bf16_t*)g.x, g.y, g.B, g.H, g.W, g.C, g.ldx, g.ldy, 2, idx);
                else if (rr == 4) area_down_bf16_body<4>((const bf16_t*)g.x, g.y, g.B, g.H, g.W, g.C, g.ldx, g.ldy, 4, idx);
                else area_down_bf16_body<0>((const bf16_t*)g.x, g.y, g.B, g.H, g.W, g.C, g.ldx, g.ldy, g.r, idx);
            }
            return;
        }
        idx -= g.n;
    }
}

// masks f32 [K,H,W] -> m16 f32 [K,h,w]: one wave per output pixel (r*r = 256 inputs, 4 per lane, row-coalesced)
__global__ void mask_down_mean_kernel(const float* __restrict__ m, float* __restrict__ m16, int K, int H, int W, int r) {
    int h = H / r, w = W / r;
    long idx = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (idx >= (long)K * h * w) return;
    int ox = idx % w; long t = idx / w; int oy = t % h; int k = t / h;
    float acc = 0.f;
    for (int e = lane; e < r * r; e += 64) {
        int dy = e / r, dx = e - dy * r;
        acc += m[((long)k * H + oy * r + dy) * W + ox * r + dx];
    }
    acc = wave_sum(acc);
    if (lane == 0) m16[idx] = acc / (float)(r * r);
}
// Both steps in one launch: one wave per output pixel walks the K objects (means in object order, then the pairs from the means it
// has just written).  Same sums in the same order as the two kernels below.
__global__ void mask_down_pair_kernel(const float* __restrict__ m, float* __restrict__ m16, uint4* __restrict__ y, int K, int H, int W, int r,
                                      int ld8) {          // ld8: 16-B units per pixel of y (1 = 8 channels; 8 = 64, channels 8.. stay as they are)
    const int h = H / r, w = W / r, hw = h * w;
    const long px = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (px >= hw) return;                                   // whole wave
    const int ox = px % w, oy = px / w;
    float sum = 0.f;
    // four objects per round with all their loads in flight (one object after the other cost K dependent round trips: 21 us at K = 3)
    for (int k0 = 0; k0 < K; k0 += 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int e = lane; e < r * r; e += 64) {
            const int dy = e / r, dx = e - dy * r;
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = m[((long)min(k0 + u, K - 1) * H + oy * r + dy) * W + ox * r + dx];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += v[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + u >= K) break;                       // wave-uniform
            const float a = wave_sum(acc[u]) / (float)(r * r);
            sum += a;
            if (lane == 0) m16[(long)(k0 + u) * hw + px] = a;
        }
    }
    if (lane == 0)
        for (int k = 0; k < K; ++k) {
            const float mk = m16[(long)k * hw + px];             // written by this lane above
            const float others = fminf(fmaxf(sum - mk, 0.f), 1.f);
            y[((long)k * hw + px) * ld8] = make_uint4(pack_bf2(mk, others), 0u, 0u, 0u);
        }
}
// m16 f32 [K,hw] -> pair bf16 [K,hw,8] = (mask, others, 0...)
__global__ void mask_pair_kernel(const float* __restrict__ m16, uint4* __restrict__ y, int K, int hw) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)K * hw) return;
    int k = idx / hw; int p = idx - (long)k * hw;
    float sum = 0.f;
    for (int j = 0; j < K; ++j) sum += m16[(long)j * hw + p];
    float mk = m16[idx];
    float others = fminf(fmaxf(sum - mk, 0.f), 1.f);
    y[idx] = make_uint4(pack_bf2(mk, others), 0u, 0u, 0u);
}

// ---------------------------------------------------------------------------------------------
// GAP, two deterministic stages (no float atomics: results are bit-reproducible run to run):
//  1. grid (ceil(HW/64), B): block 256 = row groups x channel octets over a 64-pixel chunk -> part[b][chunk][c]
//  2. grid (B): sums the chunks in order and scales by 1/HW
__global__ __launch_bounds__(256) void gap_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, int HW, int C) {
    __shared__ float red[2048 + 32];                      // [row group][C + 1]
    const int b = blockIdx.y, p0 = blockIdx.x * 64;
    const int C8 = C >> 3;
    const int c8 = threadIdx.x % C8, rg = threadIdx.x / C8, nrg = 256 / C8;       // C=256: 32 octets x 8 row groups
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = rg; r < 64; r += nrg) {
        int p = p0 + r;
        if (p < HW) {
            uint4 v = *reinterpret_cast<const uint4*>(x + ((long)b * HW + p) * C + c8 * 8);
            acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xffff0000u);
            acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xffff0000u);
            acc[4] += __uint_as_float(v.z << 16); acc[5] += __uint_as_float(v.z & 0xffff0000u);
            acc[6] += __uint_as_float(v.w << 16); acc[7] += __uint_as_float(v.w & 0xffff0000u);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[rg * (C + 1) + c8 * 8 + i] = acc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float sum = 0.f;
        for (int g = 0; g < nrg; ++g) sum += red[g * (C + 1) + c];
        part[((long)b * gridDim.x + blockIdx.x) * C + c] = sum;
    }
}
__global__ void gap_final_kernel(const float* __restrict__ part, float* __restrict__ y, int nchunk, int C, float inv) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float sum = 0.f;
        for (int k = 0; k < nchunk; ++k) sum += part[((long)b * nchunk + k) * C + c];
        y[(long)b * C + c] = sum * inv;
    }
}

// ECA_APPLY: y = x * sigmoid(conv1d_k5(mean_hw(x)))[c] + r.  grid (ceil(HW/32), B), block 256.  The prologue finishes the
// deterministic GAP reduction (sum of the per-chunk partials in chunk order) for all C channels, applies the 5-tap
// channel conv + sigmoid once per block into LDS, then the block streams 32 pixels x C channels (16-B accesses).
__global__ __launch_bounds__(256) void eca_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ part,
                                                        const float* __restrict__ wk, const bf16_t* __restrict__ r,
                                                        bf16_t* __restrict__ y, float* __restrict__ gap_out, int HW, int C, int nchunk,
                                                        const long long* __restrict__ fixed, const bf16_t* __restrict__ pw,
                                                        const float* __restrict__ pb, float* __restrict__ plog, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(1);               // CUTIE_F_PRIO: a launch of the frame's critical path (the plan decides, ops.OpList.prio)
    __shared__ float gap[256 + 4], sc[256];
    const int b = blockIdx.y, tid = threadIdx.x;
    const float inv = 1.f / (float)HW;
    // the activation / residual chunks of this thread are requested first: they do not depend on the channel scale, so their
    // latency overlaps the reduction of the partial sums (C <= 256: at most 4 chunks of 8 channels per thread and block)
    const int C8 = C >> 3;
    const int p0 = blockIdx.x * 32;
    uint4 xv[4], rv[4];
    long offs[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int q = tid + it * 256;
        const int pp = q / C8, c8 = q - pp * C8, p = p0 + pp;
        const bool ok = q < 32 * C8 && p < HW;
        offs[it] = ok ? ((long)b * HW + p) * C + c8 * 8 : -1;
        const long o = ok ? offs[it] : 0;
        xv[it] = *reinterpret_cast<const uint4*>(x + o);
        rv[it] = *reinterpret_cast<const uint4*>(r + o);
    }
    if (fixed) {                                          // sums accumulated by the producing conv (fixed point x 2^24, conv_dma.hip)
        for (int c = tid; c < C; c += 256) {
            const float mean = (float)((double)fixed[(long)b * C + c] * (1.0 / 16777216.0)) * inv;
            gap[c + 2] = mean;
            if (gap_out && blockIdx.x == 0) gap_out[(long)b * C + c] = mean;
        }
    } else
    for (int c = tid; c < C; c += 256) {
        // (8 partials in flight per round trip, summed in a fixed order: the plain loop ran nchunk DEPENDENT round trips -- 26 at
        // 480p -- and was 2/3 of this kernel's 11 us)
        float sum = 0.f;
        for (int k0 = 0; k0 < nchunk; k0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[((long)b * nchunk + min(k0 + u, nchunk - 1)) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += k0 + u < nchunk ? v[u] : 0.f;
        }
        gap[c + 2] = sum * inv;
        if (gap_out && blockIdx.x == 0) gap_out[(long)b * C + c] = sum * inv;
    }
    if (tid < 2) { gap[tid] = 0.f; gap[C + 2 + tid] = 0.f; }            // zero padding of the conv1d
    __syncthreads();
    // sc is stored TRANSPOSED -- scale of channel c at [(c & 7) * 32 + (c >> 3)] -- so that the streaming loop below, where lane l reads the
    // eight scales of channel octet c8 = l & 31, hits 32 consecutive banks per read (PMC r05: channel-major, stride 8 words between the lanes,
    // every ds_read_b32 was a 4-way conflict: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.43)
    for (int c = tid; c < C; c += 256) {
        float a = wk[0] * gap[c] + wk[1] * gap[c + 1] + wk[2] * gap[c + 2] + wk[3] * gap[c + 3] + wk[4] * gap[c + 4];
        sc[(c & 7) * 32 + (c >> 3)] = sigmoidf_(a);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (offs[it] < 0) continue;
        const int q = tid + it * 256;
        const int c8 = q % C8;
        const uint32_t* xu = &xv[it].x; const uint32_t* ru = &rv[it].x;
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float lo = __uint_as_float(xu[i] << 16) * sc[(2 * i) * 32 + c8] + __uint_as_float(ru[i] << 16);
            float hi = __uint_as_float(xu[i] & 0xffff0000u) * sc[(2 * i + 1) * 32 + c8] + __uint_as_float(ru[i] & 0xffff0000u);
            o[i] = pack_bf2(lo, hi);
        }
        *reinterpret_cast<uint4*>(y + offs[it]) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    // Optional head riding on the store (pw != 0; C = 256): logit[b, p] = bias + sum_c relu(y[b, p, c]) * w[c] -- the 1x1 conv with fused
    // input ReLU that the object transformer applies to a block's output (mask_pred, object_transformer.py:151-164) on what was just
    // STORED (bf16-rounded, like the conv launch it replaces).  A pixel's 256 channels sit in the 32 lanes of half a wave.
    if (pw) {
        const int c8 = tid & 31;
        const uint4 wv = *reinterpret_cast<const uint4*>(pw + c8 * 8);
        const uint32_t* wu = &wv.x;
        const float bias = pb ? pb[0] : 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            float part = 0.f;
            if (offs[it] >= 0) {
                const uint32_t* xu = &xv[it].x; const uint32_t* ru = &rv[it].x;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float lo = __uint_as_float(xu[i] << 16) * sc[(2 * i) * 32 + c8] + __uint_as_float(ru[i] << 16);
                    const float hi = __uint_as_float(xu[i] & 0xffff0000u) * sc[(2 * i + 1) * 32 + c8] + __uint_as_float(ru[i] & 0xffff0000u);
                    const uint32_t st = pack_bf2(lo, hi);                  // the stored pair
                    part += fmaxf(__uint_as_float(st << 16), 0.f) * __uint_as_float(wu[i] << 16);
                    part += fmaxf(__uint_as_float(st & 0xffff0000u), 0.f) * __uint_as_float(wu[i] & 0xffff0000u);
                }
            }
            // sum over the 32 lanes of the pixel: 16-lane rows on the DPP path, then the row pair {l, l ^ 16}
            part += __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(part), 0xB1, 0xf, 0xf, false));
            part += __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(part), 0x4E, 0xf, 0xf, false));
            part += __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(part), 0x141, 0xf, 0xf, false));
            part += __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(part), 0x140, 0xf, 0xf, false));
            const unsigned pu = __float_as_uint(part);
            const auto rr = __builtin_amdgcn_permlane16_swap(pu, pu, false, false);
            const float tot = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
            const int q = tid + it * 256, p = p0 + q / C8;
            if (c8 == 0 && offs[it] >= 0) plog[(long)b * HW + p] = tot + bias;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// four channels per thread: 16-byte loads of the three gate rows and of h, a 16-byte store of h and an 8-byte store of its bf16 shadow
// (one channel per thread wrote the shadow in 2-byte pieces); the same expression per element as gru_kernel
__global__ void gru4_kernel(const float* __restrict__ v, float* __restrict__ h, bf16_t* __restrict__ hb, long n, int C, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(1);               // CUTIE_F_PRIO: a launch of the frame's critical path (the plan decides, ops.OpList.prio)
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;             // (row, channel quad)
    const int C4 = C >> 2;
    if (idx >= n * C4) return;
    const long row = idx / C4; const int c = (int)(idx - row * C4) * 4;
    const float* vr = v + row * 3 * C + c;
    const float4 f4 = *reinterpret_cast<const float4*>(vr), u4 = *reinterpret_cast<const float4*>(vr + C), n4 = *reinterpret_cast<const float4*>(vr + 2 * C);
    const float4 h4 = *reinterpret_cast<const float4*>(h + row * C + c);
    const float fa[4] = {f4.x, f4.y, f4.z, f4.w}, ua[4] = {u4.x, u4.y, u4.z, u4.w}, na[4] = {n4.x, n4.y, n4.z, n4.w}, ha[4] = {h4.x, h4.y, h4.z, h4.w};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float f = sigmoidf_(fa[i]), u = sigmoidf_(ua[i]), nv = tanhf(na[i]);
        o[i] = f * ha[i] * (1.f - u) + u * nv;
    }
    *reinterpret_cast<float4*>(h + row * C + c) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<uint2*>(hb + row * C + c) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
}

__global__ void gru_kernel(const float* __restrict__ v, float* __restrict__ h, bf16_t* __restrict__ hb, long n, int C) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * C) return;
    long row = idx / C; int c = idx - row * C;
    const float* vr = v + row * 3 * C;
    float f = sigmoidf_(vr[c]), u = sigmoidf_(vr[C + c]), nv = tanhf(vr[2 * C + c]);
    float nh = f * h[idx] * (1.f - u) + u * nv;
    h[idx] = nh;
    hb[idx] = f2bf(nh);
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float clamp_logit(float p) {
    p = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
    return logf(p / (1.f - p));
}

__global__ void seg_agg_kernel(const float* __restrict__ lg, float* __restrict__ agg, int K, int hw) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    float bg = 1.f;
    for (int k = 0; k < K; ++k) {
        float pr = 1.f / (1.f + expf(-lg[(long)k * hw + p]));
        bg *= (1.f - pr);
        agg[(long)(k + 1) * hw + p] = clamp_logit(pr);
    }
    agg[p] = clamp_logit(bg);
}

__global__ void up4_softmax_kernel(const float* __restrict__ agg, float* __restrict__ prob, float* __restrict__ lup,
                                   int P, int h, int w) {
    int OH = 4 * h, OW = 4 * w;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)OH * OW) return;
    int ox = idx % OW, oy = idx / OW;
    int y0, y1, x0, x1; float ly, lx;
    up_coord(oy, h, 0.25f, y0, y1, ly);
    up_coord(ox, w, 0.25f, x0, x1, lx);
    float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    long o00 = (long)y0 * w + x0, o01 = (long)y0 * w + x1, o10 = (long)y1 * w + x0, o11 = (long)y1 * w + x1;
    long hw = (long)h * w, OHW = (long)OH * OW;
    float mx = -INFINITY;
    for (int c = 0; c < P; ++c) {
        const float* a = agg + c * hw;
        float v = w00 * a[o00] + w01 * a[o01] + w10 * a[o10] + w11 * a[o11];
        if (lup) lup[c * OHW + idx] = v;
        mx = fmaxf(mx, v);
    }
    float sum = 0.f;
    for (int c = 0; c < P; ++c) {
        const float* a = agg + c * hw;
        float v = w00 * a[o00] + w01 * a[o01] + w10 * a[o10] + w11 * a[o11];
        sum += expf(v - mx);
    }
    float inv = 1.f / sum;
    for (int c = 0; c < P; ++c) {
        const float* a = agg + c * hw;
        float v = w00 * a[o00] + w01 * a[o01] + w10 * a[o10] + w11 * a[o11];
        prob[c * OHW + idx] = expf(v - mx) * inv;
    }
}

// SEG_AGG + UP4_SOFTMAX in one launch (flags&1 of UP4_SOFTMAX, P <= 16): the aggregated logits of the 4 bilinear taps are
// recomputed per output pixel from the K raw logits (K sigmoids + K+1 logs per tap) instead of being staged in a [P,h,w] buffer by
// a launch of their own; the P interpolated values stay in registers for max / sum / normalise.  Same operations in the same
// order as the two-launch form: bit-identical.
template <int PMAX>
__global__ void up4_softmax_fused_kernel(const float* __restrict__ lg, float* __restrict__ prob, float* __restrict__ lup,
                                         int K, int h, int w) {
    const int OH = 4 * h, OW = 4 * w;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)OH * OW) return;
    const int ox = idx % OW, oy = idx / OW;
    int y0, y1, x0, x1; float ly, lx;
    up_coord(oy, h, 0.25f, y0, y1, ly);
    up_coord(ox, w, 0.25f, x0, x1, lx);
    const float wt[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
    const long off[4] = {(long)y0 * w + x0, (long)y0 * w + x1, (long)y1 * w + x0, (long)y1 * w + x1};
    const long hw = (long)h * w, OHW = (long)OH * OW;
    float v[PMAX];
#pragma unroll
    for (int c = 0; c < PMAX; ++c) v[c] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float bg = 1.f;
#pragma unroll
        for (int k = 0; k < PMAX - 1; ++k)
            if (k < K) {
                const float pr = 1.f / (1.f + expf(-lg[(long)k * hw + off[t]]));
                bg *= (1.f - pr);
                const float a = wt[t] * clamp_logit(pr);
                v[k + 1] = t == 0 ? a : v[k + 1] + a;
            }
        const float a = wt[t] * clamp_logit(bg);
        v[0] = t == 0 ? a : v[0] + a;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < PMAX; ++c)
        if (c <= K) {
            if (lup) lup[c * OHW + idx] = v[c];
            mx = fmaxf(mx, v[c]);
        }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < PMAX; ++c)
        if (c <= K) { v[c] = expf(v[c] - mx); sum += v[c]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int c = 0; c < PMAX; ++c)
        if (c <= K) prob[c * OHW + idx] = v[c] * inv;
}

// The same, four consecutive output pixels of a row per thread: their taps lie in two rows x three columns of the stride-4 map, so the
// logits are fetched (and turned into clamped logits) once per source pixel instead of once per tap -- 6 K loads for 4 pixels instead
// of 48 -- and every plane is written with 16-byte stores.  Per pixel the arithmetic and its order are those of the kernel above
// (bit-identical results; flags&2 of the op keeps the one-pixel form for the comparison in tests).
// The four output pixels (oy, 4j .. 4j + 3) of the fused up-sampling + softmax: out[plane][q] = probabilities, lo[plane][q] = up-sampled logits.
// KC > 0: the object count as a compile-time constant.  With a run-time K every `if (k < K)` around a load is a (uniform) branch that ends
// in its own s_waitcnt vmcnt(0): the 6 x K source logits came in one after the other, 18 dependent L2 round trips in front of the
// first exp at K = 3 (tools/isa_waits.py).  With KC the loop bodies are straight-line code and all loads are in flight together; the
// arithmetic and its order are the same, so are the bits.
// The aggregated, clamped logits of ONE source pixel from its K raw logits (SEG_AGG, cutie.py:199-200): Lp[0] = background.
template <int PMAX>
__device__ __forceinline__ void up4_src(const float (&raw)[PMAX - 1], int K, float (&Lp)[PMAX]) {
    float bg = 1.f;
#pragma unroll
    for (int k = 0; k < PMAX - 1; ++k)
        if (k < K) {
            const float pr = 1.f / (1.f + expf(-raw[k]));
            bg *= (1.f - pr);
            Lp[k + 1] = clamp_logit(pr);
        }
    Lp[0] = clamp_logit(bg);
}

// SHARED: the source pixels' aggregated logits come from Lsh[plane][6 x 6] (the 4 x 4 source pixels under a 16 x 16 output cell and one
// ring around them, first pixel (sy0, sx0), coordinates clamped to the map), computed once per wave by up4_softmax_md_kernel -- every lane
// recomputing its six source pixels was 18 exp + 24 log + 42 divisions per thread at K = 3, 1800 VALU instructions in all: the kernel
// ran at the VALU, not at its 6.6 MB of stores.  Same values from the same expression, whoever computes them.
template <int PMAX, int KC = 0, bool SHARED = false>
__device__ __forceinline__ void up4_four(const float* __restrict__ lg, int Krt, int h, int w, int oy, int j, float (&out)[PMAX][4], float (&lo)[PMAX][4],
                                         const float* Lsh = nullptr, int sy0 = 0, int sx0 = 0) {
    const int K = KC > 0 ? KC : Krt;
    int y0, y1; float ly;
    up_coord(oy, h, 0.25f, y0, y1, ly);
    const int col[3] = {max(j - 1, 0), j, min(j + 1, w - 1)};
    const long hw = (long)h * w;
    // clamped logits of the six source pixels: L[r][c][plane], plane 0 = background
    float L[2][3][PMAX];
    if (SHARED) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int slot = ((r ? y1 : y0) - sy0) * 6 + (col[c] - sx0);
#pragma unroll
                for (int p = 0; p < PMAX; ++p)
                    if (p <= K) L[r][c][p] = Lsh[p * 36 + slot];
            }
    } else {
        float raw[2][3][PMAX - 1];                       // (all 6 x K loads first: one round trip)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const long o = (long)(r ? y1 : y0) * w + col[c];
#pragma unroll
                for (int k = 0; k < PMAX - 1; ++k)
                    if (k < K) raw[r][c][k] = lg[(long)k * hw + o];
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) up4_src<PMAX>(raw[r][c], K, L[r][c]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int x0, x1; float lx;
        up_coord(4 * j + q, w, 0.25f, x0, x1, lx);
        const int c0 = x0 == j ? 1 : (x0 < j ? 0 : 2), c1 = x1 == j ? 1 : (x1 < j ? 0 : 2);
        const float wt[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
        float v[PMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int p = 0; p < PMAX; ++p)
            if (p <= K) {
                // (selects instead of dynamic indexing: c0, c1 are 0..2)
                const float a00 = c0 == 0 ? L[0][0][p] : (c0 == 1 ? L[0][1][p] : L[0][2][p]);
                const float a01 = c1 == 0 ? L[0][0][p] : (c1 == 1 ? L[0][1][p] : L[0][2][p]);
                const float a10 = c0 == 0 ? L[1][0][p] : (c0 == 1 ? L[1][1][p] : L[1][2][p]);
                const float a11 = c1 == 0 ? L[1][0][p] : (c1 == 1 ? L[1][1][p] : L[1][2][p]);
                float t = wt[0] * a00;
                t = t + wt[1] * a01;
                t = t + wt[2] * a10;
                t = t + wt[3] * a11;
                v[p] = t;
                lo[p][q] = t;
                mx = fmaxf(mx, t);
            }
        float sum = 0.f;
#pragma unroll
        for (int p = 0; p < PMAX; ++p)
            if (p <= K) { v[p] = expf(v[p] - mx); sum += v[p]; }
        const float inv = 1.f / sum;
#pragma unroll
        for (int p = 0; p < PMAX; ++p)
            if (p <= K) out[p][q] = v[p] * inv;
    }
}

template <int PMAX, int KC = 0>
__global__ __launch_bounds__(256) void up4_softmax_fused4_kernel(const float* __restrict__ lg, float* __restrict__ prob, float* __restrict__ lup,
                                                                 int Krt, int h, int w, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(1);               // CUTIE_F_PRIO: a launch of the frame's critical path (the plan decides, ops.OpList.prio)
    const int K = KC > 0 ? KC : Krt;
    const int OH = 4 * h, OW = 4 * w;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;             // (oy, j): output pixels (oy, 4j .. 4j + 3)
    if (idx >= (long)OH * w) return;
    const int j = idx % w, oy = idx / w;
    const long OHW = (long)OH * OW;
    {                                                      // blockIdx.y = clip (clips in lock step: K logit planes in, K + 1 planes out per clip)
        const long cl = blockIdx.y;
        lg += cl * K * h * w; prob += cl * (K + 1) * OHW;
        if (lup) lup += cl * (K + 1) * OHW;
    }
    float out[PMAX][4];
    float lo[PMAX][4];
    up4_four<PMAX, KC>(lg, K, h, w, oy, j, out, lo);
    const long base = (long)oy * OW + 4 * j;
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
        if (p <= K) {
            if (lup) *reinterpret_cast<float4*>(lup + p * OHW + base) = make_float4(lo[p][0], lo[p][1], lo[p][2], lo[p][3]);
            *reinterpret_cast<float4*>(prob + p * OHW + base) = make_float4(out[p][0], out[p][1], out[p][2], out[p][3]);
        }
}

// The same, one WAVE per 16 x 16 output cell (= one stride-16 pixel), which lets the launch also produce what MASK_DOWN derives from
// these probabilities for the NEXT frame's pixel fusion: m16[k] = mean of object plane k over the cell, pair = (m16[k], clamp(sum of
// the others)).  The cell's values go through LDS so that every lane sums the same four entries in the same order as
// mask_down_pair_kernel (entries lane, lane + 64, +128, +192 of the row-major cell) and the same wave_sum follows: bit-identical to the
// MASK_DOWN launch it replaces.  4 waves (cells) per block; needs H, W multiples of 16 and K + 1 <= PMAX.
template <int PMAX, int KC = 0, bool SH = false>
__global__ __launch_bounds__(256) void up4_softmax_md_kernel(const float* __restrict__ lg, float* __restrict__ prob, float* __restrict__ lup,
                                                             int Krt, int h, int w, float* __restrict__ m16, uint4* __restrict__ pair, int ld8, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(1);               // CUTIE_F_PRIO: a launch of the frame's critical path (the plan decides, ops.OpList.prio)
    const int K = KC > 0 ? KC : Krt;
    __shared__ float cell[4][PMAX - 1][256];
    __shared__ float srcL[4][SH ? PMAX : 1][36];
    const int OH = 4 * h, OW = 4 * w, ch = OH >> 4, cw = OW >> 4, ncell = ch * cw;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cid = blockIdx.x * 4 + wave;
    if (cid >= ncell) return;                               // whole wave
    {                                                      // blockIdx.y = clip (clips in lock step: every per-clip array advances by one clip's extent)
        const long cl = blockIdx.y;
        lg += cl * K * h * w; prob += cl * (K + 1) * (long)OH * OW;
        if (lup) lup += cl * (K + 1) * (long)OH * OW;
        m16 += cl * K * ncell; pair += cl * K * ncell * ld8;
    }
    const int cy = cid / cw, cx = cid - cy * cw;
    const int r = lane >> 2, q4 = lane & 3;
    const int oy = cy * 16 + r, j = cx * 4 + q4;
    const long OHW = (long)OH * OW;
    float out[PMAX][4];
    float lo[PMAX][4];
    if (SH) {
        // the 6 x 6 source pixels of this cell, one per lane (36 of 64), aggregated once
        if (lane < 36) {
            const int sly = lane / 6, slx = lane - sly * 6;
            const long o = (long)min(max(4 * cy - 1 + sly, 0), h - 1) * w + min(max(4 * cx - 1 + slx, 0), w - 1);
            float raw[PMAX - 1], Lp[PMAX];
#pragma unroll
            for (int k = 0; k < PMAX - 1; ++k)
                if (k < K) raw[k] = lg[(long)k * h * w + o];
            up4_src<PMAX>(raw, K, Lp);
#pragma unroll
            for (int p = 0; p < PMAX; ++p)
                if (p <= K) srcL[wave][SH ? p : 0][lane] = Lp[p];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): srcL is wave-private
        __builtin_amdgcn_wave_barrier();
    }
    up4_four<PMAX, KC, SH>(lg, K, h, w, oy, j, out, lo, &srcL[wave][0][0], 4 * cy - 1, 4 * cx - 1);
    const long base = (long)oy * OW + 4 * j;
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
        if (p <= K) {
            if (lup) *reinterpret_cast<float4*>(lup + p * OHW + base) = make_float4(lo[p][0], lo[p][1], lo[p][2], lo[p][3]);
            *reinterpret_cast<float4*>(prob + p * OHW + base) = make_float4(out[p][0], out[p][1], out[p][2], out[p][3]);
            if (p >= 1) *reinterpret_cast<float4*>(&cell[wave][p - 1][r * 16 + 4 * q4]) = make_float4(out[p][0], out[p][1], out[p][2], out[p][3]);
        }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): the cell is wave-private
    float sum = 0.f, mk[PMAX - 1];
#pragma unroll
    for (int k = 0; k < PMAX - 1; ++k)
        if (k < K) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += cell[wave][k][lane + 64 * i];
            const float a = wave_sum(acc) / 256.f;
            sum += a;
            mk[k] = a;
        }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < PMAX - 1; ++k)
            if (k < K) {
                m16[(long)k * ncell + cid] = mk[k];
                const float others = fminf(fmaxf(sum - mk[k], 0.f), 1.f);
                pair[((long)k * ncell + cid) * ld8] = make_uint4(pack_bf2(mk[k], others), 0u, 0u, 0u);
            }
    }
}

// ---------------------------------------------------------------------------------------------
__global__ void mask_merge_kernel(const void* __restrict__ inmask, const float* __restrict__ pred, const int* __restrict__ src,
                                  float* __restrict__ planes, int h0, int w0, int H, int W, int pl, int pt, int Knew,
                                  int Kold, int nfloat, int fmode) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long HW = (long)H * W;
    if (idx >= HW) return;
    int yy = idx / W, xx = idx - (long)yy * W;
    int sy = yy - pt, sx = xx - pl;
    bool inside = (unsigned)sy < (unsigned)h0 && (unsigned)sx < (unsigned)w0;
    long so = (long)sy * w0 + sx, plane = (long)h0 * w0;
    int iv = 0; bool covered = false;
    if (!fmode) {
        iv = inside ? ((const int*)inmask)[so] : 0;
        covered = iv > 0;
    } else {
        float mx = 0.f;   // F.pad pads with 0
        if (inside) { mx = -INFINITY; for (int c = 0; c < nfloat; ++c) mx = fmaxf(mx, ((const float*)inmask)[c * plane + so]); }
        covered = mx > 0.5f;
    }
    for (int t = 0; t < Knew; ++t) {
        int s = src[t];
        float v;
        if (s >= 0) {
            if (!fmode) v = (iv == s) ? 1.f : 0.f;
            else v = inside ? ((const float*)inmask)[s * plane + so] : 0.f;
        } else {
            v = (pred && t < Kold && !covered) ? pred[(long)(t + 1) * HW + idx] : 0.f;
        }
        planes[(long)t * HW + idx] = v;
    }
}

__global__ void agg_softmax_kernel(const float* __restrict__ planes, float* __restrict__ prob, int K, long HW) {
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    float bg = 1.f, mx;
    for (int k = 0; k < K; ++k) bg *= (1.f - planes[k * HW + p]);
    float l0 = clamp_logit(bg);
    mx = l0;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, clamp_logit(planes[k * HW + p]));
    float sum = expf(l0 - mx);
    for (int k = 0; k < K; ++k) sum += expf(clamp_logit(planes[k * HW + p]) - mx);
    float inv = 1.f / sum;
    prob[p] = expf(l0 - mx) * inv;
    for (int k = 0; k < K; ++k) prob[(k + 1) * HW + p] = expf(clamp_logit(planes[k * HW + p]) - mx) * inv;
}

// ---------------------------------------------------------------------------------------------
// LINEAR (small M = K_objects*16 rows): grid (ceil(N / (4*CPW)), ceil(M/16)); block = 4 waves.  LPC = min(64, Kd/8)
// lanes cooperate on one output column (each lane owns 8 consecutive k per 512-wide sweep: 16-B weight loads,
// 32-B activation loads), so a wave computes CPW = 64/LPC columns x 16 rows.  The 16 per-row partial sums are
// combined with a reduce-scatter butterfly (8+4+2+1 exchanges, then log2(LPC/16) plain steps) instead of 16 full
// wave reductions: 17 shuffles instead of 96.
template <int LPC>
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, const float* __restrict__ xadd,
                                                           const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                           const float* __restrict__ res, float* __restrict__ y, int M, int N,
                                                           int Kd, int ldx, int ldy, int add_rows, int relu) {
    constexpr int CPW = 64 / LPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPC, cl = lane % LPC;             // column slot inside the wave, lane inside the column group
    const int col = (blockIdx.x * 4 + wave) * CPW + sub, m0 = blockIdx.y * 16;
    const bool cvalid = col < N;
    const bf16_t* wrow = W + (long)(cvalid ? col : 0) * Kd;
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = cl * 8; k < Kd; k += LPC * 8) {
        uint4 wv = *reinterpret_cast<const uint4*>(wrow + k);
        float wf[8];
        wf[0] = __uint_as_float(wv.x << 16); wf[1] = __uint_as_float(wv.x & 0xffff0000u);
        wf[2] = __uint_as_float(wv.y << 16); wf[3] = __uint_as_float(wv.y & 0xffff0000u);
        wf[4] = __uint_as_float(wv.z << 16); wf[5] = __uint_as_float(wv.z & 0xffff0000u);
        wf[6] = __uint_as_float(wv.w << 16); wf[7] = __uint_as_float(wv.w & 0xffff0000u);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = m0 + r;
            m = m < M ? m : M - 1;                           // clamp (rows >= M are never stored)
            const float* xr = x + (long)m * ldx + k;
            float4 a = *reinterpret_cast<const float4*>(xr), b = *reinterpret_cast<const float4*>(xr + 4);
            if (xadd) {
                const float* ar = xadd + (long)(m % add_rows) * Kd + k;
                float4 c = *reinterpret_cast<const float4*>(ar), d = *reinterpret_cast<const float4*>(ar + 4);
                a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w; b.x += d.x; b.y += d.y; b.z += d.z; b.w += d.w;
            }
            acc[r] += a.x * wf[0] + a.y * wf[1] + a.z * wf[2] + a.w * wf[3] + b.x * wf[4] + b.y * wf[5] + b.z * wf[6] + b.w * wf[7];
        }
    }
    // reduce-scatter over the top 4 bits of cl: after step s the lane keeps 16 >> (s+1) rows
    constexpr int TOP = LPC / 2;                              // 32 (LPC 64) or 16 (LPC 32)
    float v8[8], v4[4], v2[2], v1;
    {
        const bool up = cl & TOP;
#pragma unroll
        for (int r = 0; r < 8; ++r) { float send = up ? acc[r] : acc[r + 8]; float recv = __shfl_xor(send, TOP, 64); v8[r] = (up ? acc[r + 8] : acc[r]) + recv; }
    }
    {
        const bool up = cl & (TOP / 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) { float send = up ? v8[r] : v8[r + 4]; float recv = __shfl_xor(send, TOP / 2, 64); v4[r] = (up ? v8[r + 4] : v8[r]) + recv; }
    }
    {
        const bool up = cl & (TOP / 4);
#pragma unroll
        for (int r = 0; r < 2; ++r) { float send = up ? v4[r] : v4[r + 2]; float recv = __shfl_xor(send, TOP / 4, 64); v2[r] = (up ? v4[r + 2] : v4[r]) + recv; }
    }
    {
        const bool up = cl & (TOP / 8);
        float send = up ? v2[0] : v2[1];
        float recv = __shfl_xor(send, TOP / 8, 64);
        v1 = (up ? v2[1] : v2[0]) + recv;
    }
#pragma unroll
    for (int o = TOP / 16; o > 0; o >>= 1) v1 += __shfl_xor(v1, o, 64);
    // row owned by this lane: the four butterfly bits, most significant first
    const int row = ((cl & TOP) ? 8 : 0) + ((cl & (TOP / 2)) ? 4 : 0) + ((cl & (TOP / 4)) ? 2 : 0) + ((cl & (TOP / 8)) ? 1 : 0);
    const int m = m0 + row;
    if ((cl & (TOP / 8 - 1)) == 0 && cvalid && m < M) {
        float v = v1;
        if (bias) v += bias[col];
        if (relu) v = fmaxf(v, 0.f);
        if (res) v += res[(long)m * N + col];
        y[(long)m * ldy + col] = v;
    }
}

// LINEAR on MFMA: one block per 16 rows x 16 columns, its 4 waves split K and are summed through LDS.  x is fp32 (the
// query side of the transformer is an fp32 island) and is split into bf16 hi + lo on the fly, W is bf16: two
// v_mfma_f32_16x16x32_bf16 per 32 k (fp32-class accuracy), no cross-lane reduction, one memory round trip per wave.
// A = x rows (lane: row l&15, k 8*(l>>4)..+7), B = W^T (lane: column l&15, same k: a contiguous 16-B piece of W's row).
template <int NW>                                       // waves per block: the K range is split over them (16 for long K: FFN linear2)
__global__ __launch_bounds__(NW * 64) void linear_mfma_kernel(const float* __restrict__ x, const float* __restrict__ xadd,
                                                          const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                          const float* __restrict__ res, float* __restrict__ y, int M, int N,
                                                          int Kd, int ldx, int ldy, int add_rows, int relu, int add_cols,
                                                          const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                          float* __restrict__ ln_out, float eps) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    __shared__ f32x4 red[NW - 1][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int m = min(m0 + c, M - 1), n = min(n0 + c, N - 1);
    const int per = (Kd >> 5) / NW;                          // 32-wide k steps per wave (Kd % (32 NW) == 0)
    const float* xrow = x + (long)m * ldx;
    // fused LayerNorm over the Kd inputs of row m (flag: ln_g != 0; Kd == 256): lane (c, g) reads the 64 values
    // k = 64 g .. of its row, the 4 lanes of a row combine through permlane swaps; every wave does this for itself
    float mean = 0.f, rstd = 1.f;
    if (ln_g) {
        float v[64], sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(xrow + g * 64 + i * 4);
            v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
            sum += (t.x + t.y) + (t.z + t.w);
        }
        mean = rows_sum(sum) * (1.f / 256.f);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) { const float d = v[i] - mean; sq += d * d; }
        rstd = rsqrtf(rows_sum(sq) * (1.f / 256.f) + eps);
    }
    const bool use_add = xadd && (add_cols <= 0 || n0 < add_cols);       // block-uniform: x_add feeds only the first add_cols outputs
    const int kw = (wave * per) * 32 + 8 * g;
    const float* xr = xrow + kw;
    const float* ar = use_add ? xadd + (long)(m % add_rows) * Kd + kw : nullptr;
    const bf16_t* wr = W + (long)n * Kd + kw;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int ks = 0; ks < per; ++ks) {
        float4 a = *reinterpret_cast<const float4*>(xr + ks * 32), b = *reinterpret_cast<const float4*>(xr + ks * 32 + 4);
        if (ln_g) {
            const float4 g0 = *reinterpret_cast<const float4*>(ln_g + kw + ks * 32), g1 = *reinterpret_cast<const float4*>(ln_g + kw + ks * 32 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(ln_b + kw + ks * 32), b1 = *reinterpret_cast<const float4*>(ln_b + kw + ks * 32 + 4);
            a.x = (a.x - mean) * rstd * g0.x + b0.x; a.y = (a.y - mean) * rstd * g0.y + b0.y;
            a.z = (a.z - mean) * rstd * g0.z + b0.z; a.w = (a.w - mean) * rstd * g0.w + b0.w;
            b.x = (b.x - mean) * rstd * g1.x + b1.x; b.y = (b.y - mean) * rstd * g1.y + b1.y;
            b.z = (b.z - mean) * rstd * g1.z + b1.z; b.w = (b.w - mean) * rstd * g1.w + b1.w;
            if (ln_out && blockIdx.x == 0 && m0 + c < M) {                // the normalised rows are a side output (residuals)
                *reinterpret_cast<float4*>(ln_out + (long)m * Kd + kw + ks * 32) = a;
                *reinterpret_cast<float4*>(ln_out + (long)m * Kd + kw + ks * 32 + 4) = b;
            }
        }
        if (ar) {
            const float4 p = *reinterpret_cast<const float4*>(ar + ks * 32), q = *reinterpret_cast<const float4*>(ar + ks * 32 + 4);
            a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w; b.x += q.x; b.y += q.y; b.z += q.z; b.w += q.w;
        }
        const u4 wv = *reinterpret_cast<const u4*>(wr + ks * 32);
        const float xs[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        u4 hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t h0 = f2bf(xs[2 * i]), h1 = f2bf(xs[2 * i + 1]);
            hi[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
            lo[i] = (uint32_t)f2bf(xs[2 * i] - bf2f(h0)) | ((uint32_t)f2bf(xs[2 * i + 1] - bf2f(h1)) << 16);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, lo), __builtin_bit_cast(bf16x8, wv), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hi), __builtin_bit_cast(bf16x8, wv), acc, 0, 0, 0);
    }
    if (wave) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) { const f32x4 t = red[w][lane]; acc[0] += t[0]; acc[1] += t[1]; acc[2] += t[2]; acc[3] += t[3]; }
        const int nn = n0 + c;
        if (nn < N) {
            const float bv = bias ? bias[nn] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mm = m0 + 4 * g + r;
                if (mm < M) {
                    float v = acc[r] + bv;
                    if (relu) v = fmaxf(v, 0.f);
                    if (res) v += res[(long)mm * N + nn];
                    y[(long)mm * ldy + nn] = v;
                }
            }
        }
    }
}

// LAYERNORM: one wave per row
__global__ void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                 float* __restrict__ y, int M, int C) {
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (long)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { float d = xr[c] - mean; v += d * d; }
    float rstd = rsqrtf(wave_sum(v) / (float)C + 1e-5f);
    for (int c = lane; c < C; c += 64) y[(long)row * C + c] = (xr[c] - mean) * rstd * g[c] + b[c];
}

__global__ void query_init_kernel(const float* __restrict__ om, float* __restrict__ y, int rows, int C) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)rows * C) return;
    int r = idx / C, c = idx - (long)r * C;
    y[idx] = om[(long)r * (C + 1) + c] / (om[(long)r * (C + 1) + C] + 1e-4f);
}

// SUMMARIZE: grid (C/64, K, ceil(HW/128)); block 256 = 4 pixel slices x 64 channels over a 128-pixel chunk; the
// per-pixel weights sigmoid(logit)*mask are computed once per block into LDS; per-chunk partial sums go to
// part[k][chunk][q][C+1] and are summed in chunk order by summarize_final_kernel (deterministic, no float atomics).
// F32: the features are fp32 with a row stride of ldf elements (one conv output [feature | weight logits], ldw = its row stride too)
template <bool F32>
__global__ __launch_bounds__(256) void summarize_kernel(const void* __restrict__ feat_, const float* __restrict__ wl,
                                                        const float* __restrict__ m16, float* __restrict__ y, int HW, int C, int Q, int ldf, int ldw) {
    __shared__ float wsm[128][17];
    __shared__ float red[4][64][17];
    const int k = blockIdx.y, cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, s = threadIdx.x >> 6;
    const int p0 = blockIdx.z * 128;
    for (int t = threadIdx.x; t < 128 * 16; t += 256) {
        int pp = t >> 4, q = t & 15, p = p0 + pp;
        float wgt = 0.f;
        if (p < HW) {
            long row = (long)k * HW + p;
            float m = m16[row];
            wgt = (1.f / (1.f + expf(-wl[row * ldw + q]))) * (q < 8 ? m : 1.f - m);
        }
        wsm[pp][q] = wgt;
    }
    __syncthreads();
    float acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float area = 0.f;
    // the thread's 32 feature values are requested together (clamped rows, looked at behind the loads): the loop used to issue one
    // global load per iteration and wait for it -- 32 dependent round trips, 40 us for a launch that moves 2.5 MB (profiles/r05_frame_chain.md)
    float fv[32];
#pragma unroll
    for (int it = 0; it < 32; ++it) {
        const int p = min(p0 + s + 4 * it, HW - 1);
        const long fo = ((long)k * HW + p) * ldf + c;
        fv[it] = F32 ? reinterpret_cast<const float*>(feat_)[fo] : bf2f(reinterpret_cast<const bf16_t*>(feat_)[fo]);
    }
#pragma unroll
    for (int it = 0; it < 32; ++it) {
        const int pp = s + 4 * it;
        if (p0 + pp < HW) {                                     // (same additions in the same order as the rolled loop)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] += wsm[pp][q] * fv[it];
            if (cl < 16) area += wsm[pp][cl];
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) red[s][cl][q] = acc[q];
    red[s][cl][16] = area;
    __syncthreads();
    if (s == 0) {
        float* yp = y + ((long)k * gridDim.z + blockIdx.z) * Q * (C + 1);
        for (int q = 0; q < 16; ++q)
            yp[(long)q * (C + 1) + c] = red[0][cl][q] + red[1][cl][q] + red[2][cl][q] + red[3][cl][q];
        if (blockIdx.x == 0 && cl < Q)
            yp[(long)cl * (C + 1) + C] = red[0][cl][16] + red[1][cl][16] + red[2][cl][16] + red[3][cl][16];
    }
}

__global__ void summarize_final_kernel(const float* __restrict__ part, float* __restrict__ y, int nchunk, int n) {
    const int k = blockIdx.y;
    int e = blockIdx.x * blockDim.x + threadIdx.x;          // element of [Q, C+1]
    if (e >= n) return;
    float sum = 0.f;
    for (int c0 = 0; c0 < nchunk; c0 += 8) {                 // eight partials in flight, added in chunk order (was: one dependent round trip per chunk)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[((long)k * nchunk + min(c0 + u, nchunk - 1)) * n + e];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c0 + u < nchunk) sum += v[u];
    }
    y[(long)k * n + e] = sum;
}

__global__ void add_pe_kernel(const uint4* __restrict__ x, const uint4* __restrict__ pe, uint4* __restrict__ y, int B, long n8) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * n8) return;
    uint4 a = x[idx], b = pe[idx % n8];
    const uint32_t* au = &a.x; const uint32_t* bu = &b.x;
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        o[i] = pack_bf2(__uint_as_float(au[i] << 16) + __uint_as_float(bu[i] << 16),
                        __uint_as_float(au[i] & 0xffff0000u) + __uint_as_float(bu[i] & 0xffff0000u));
    y[idx] = make_uint4(o[0], o[1], o[2], o[3]);
}

// ---------------------------------------------------------------------------------------------
__global__ void memset32_kernel(uint32_t* d, long n, uint32_t v) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = v;
}
__global__ void copy2d_kernel(const uint32_t* __restrict__ s, uint32_t* __restrict__ d, long rows, int roww, long ss, long ds) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * roww) return;
    long r = i / roww; int c = i - r * roww;
    d[r * ds + c] = s[r * ss + c];
}
// BANK_WRITE: blockIdx.y = segment (0..5 copies, 6..7 fills), blockIdx.x strides over its words
struct BankWrite { const uint32_t* src[6]; uint32_t* dst[8]; int words[8]; uint32_t pattern[2]; };
__global__ __launch_bounds__(256) void bank_write_kernel(BankWrite w) {
    const int seg = blockIdx.y;
    const int n = w.words[seg];
    uint32_t* const d = w.dst[seg];
    if (seg < 6) {
        const uint32_t* const s = w.src[seg];
        for (int i = blockIdx.x * 1024 + threadIdx.x; i < n; i += gridDim.x * 1024) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (i + r * 256 < n) d[i + r * 256] = s[i + r * 256];
        }
    } else {
        const uint32_t v = w.pattern[seg - 6];
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = v;
    }
}
__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float a) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}
// life counters of up to two token ranges advance by one; optionally use[i] += delta[i] (the usage a look-ahead read-out parked
// in a side buffer, applied when -- and only when -- that read-out is consumed)
// dfx & 1: delta holds unsigned 64-bit fixed-point sums (2^-40, AFF_READOUT flags&1): converted with one rounding; dfx & 2: and is cleared behind it
__global__ void tick_kernel(float* lifeA, long nA, float* lifeB, long nB, float* use, float* delta, long nU, int dfx) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (lifeA && i < nA) lifeA[i] += 1.f;
    if (lifeB && i < nB) lifeB[i] += 1.f;
    if (use && i < nU) {
        if (dfx & 1) {
            unsigned long long* d = reinterpret_cast<unsigned long long*>(delta);
            use[i] += (float)((double)d[i] * 9.094947017729282e-13);
            if (dfx & 2) d[i] = 0ull;
        } else use[i] += delta[i];
    }
}
__global__ void cast_kernel(const void* s, void* d, long n, int to_f32) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (to_f32) ((float*)d)[i] = bf2f(((const bf16_t*)s)[i]);
    else ((bf16_t*)d)[i] = f2bf(((const float*)s)[i]);
}

// RESIZE: F.interpolate(size=(OH,OW)) of C planes, mode bilinear (align_corners=False, no antialias) or nearest-exact.
__global__ void resize_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W, int OH, int OW,
                              long splane, int sld, int nearest) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)C * OH * OW) return;
    const int ox = (int)(idx % OW), oy = (int)((idx / OW) % OH), c = (int)(idx / ((long)OW * OH));
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    const float* p = src + (long)c * splane;
    if (nearest) {
        const int iy = min((int)floorf((oy + 0.5f) * sy), H - 1), ix = min((int)floorf((ox + 0.5f) * sx), W - 1);
        dst[idx] = p[(long)iy * sld + ix];
        return;
    }
    const float fy = fmaxf((oy + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf((ox + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float a = p[(long)y0 * sld + x0], b = p[(long)y0 * sld + x1], cc = p[(long)y1 * sld + x0], d = p[(long)y1 * sld + x1];
    dst[idx] = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * cc + lx * d);
}

// FLIP_W: dst[c,y,x] = alpha * src[c,y,W-1-x] + beta * dst[c,y,x]  (beta == 0: dst is not read)
__global__ void flip_w_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int W, int slds, int dlds,
                              float alpha, float beta) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * W) return;
    const long r = idx / W;
    const int x = (int)(idx - r * W);
    const float v = alpha * src[r * slds + (W - 1 - x)];
    float* d = dst + r * dlds + x;
    *d = beta != 0.f ? v + beta * *d : v;
}

// PROB_TO_ID: id[y,x] = lut[argmax_p prob[p,y,x]] (first maximum wins, like torch.argmax); 4 pixels per thread when aligned.
template <typename OUT>
__global__ void prob_to_id_kernel(const float* __restrict__ prob, const int* __restrict__ lut, OUT* __restrict__ out,
                                  int P, int H, int W, long plane, int ldrow) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)H * W) return;
    const int yy = (int)(idx / W), xx = (int)(idx - (long)yy * W);
    const float* src = prob + (long)yy * ldrow + xx;
    float best = src[0];
    int arg = 0;
    for (int q = 1; q < P; ++q) {
        const float v = src[(long)q * plane];
        if (v > best) { best = v; arg = q; }
    }
    out[idx] = (OUT)lut[arg];
}

// ---------------------------------------------------------------------------------------------
int launch_elementwise(const cutie_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const uint64_t* p = op->p;
    const int BS = 256;
    const int PRIO = (op->flags & CUTIE_F_PRIO) ? 1 : 0;     // the launch belongs to the frame's critical path (kernels that honour it: see include/cutie_hip.h)
    switch (op->kind) {
        case CUTIE_OP_MAXPOOL: {
            int C8 = i[3] / 8;
            long n = (long)i[0] * i[4] * i[5] * C8;
            hipLaunchKernelGGL(maxpool_kernel, GRID1D(n, BS), dim3(BS), 0, s, (const uint4*)p[0], (uint4*)p[1], i[0], i[1], i[2], C8, i[4], i[5], op->flags & 1);
            break;
        }
        case CUTIE_OP_IMG_PREP: {
            int K = p[1] ? i[6] : 1;
            long n = (long)i[2] * i[3] * K;
            hipLaunchKernelGGL(img_prep_kernel, GRID1D(n, BS), dim3(BS), 0, s, (const float*)p[0], (const float*)p[1], (uint4*)p[2],
                               i[0], i[1], i[2], i[3], i[4], i[5], K, op->f[0], op->f[1], op->f[2], op->f[3], op->f[4], op->f[5]);
            break;
        }
        case CUTIE_OP_UPSAMPLE2X_ADD: {
            int C8 = i[3] / 8;
            long n = (long)i[0] * 4 * i[1] * i[2] * C8;
            if (i[4] < 0 || (i[4] > 0 && (i[0] % i[4] || i[5] < 4 * i[1] * i[2]))) { cutie_set_error("upsample2x_add: skip groups of %d objects do not divide B = %d (or their stride is below one map)", i[4], i[0]); return -2; }
            hipLaunchKernelGGL(upsample2x_add_kernel, GRID1D(n, BS), dim3(BS), 0, s, (const uint4*)p[0], (const uint4*)p[1], (uint4*)p[2], i[0], i[1], i[2], C8, i[4], (long)i[5], PRIO);
            break;
        }
        case CUTIE_OP_AREA_DOWN: {
            int r = i[6];
            long np = (long)i[0] * (i[1] / r) * (i[2] / r);
            if (op->flags & 1)
                hipLaunchKernelGGL(area_down_f32_kernel, GRID1D(np, BS), dim3(BS), 0, s, (const float*)p[0], (bf16_t*)p[1], i[0], i[1], i[2], i[3], i[4], i[5], r, i[7]);
            else
                hipLaunchKernelGGL(area_down_bf16_kernel, GRID1D(np * (i[3] / 8), BS), dim3(BS), 0, s, (const bf16_t*)p[0], (bf16_t*)p[1], i[0], i[1], i[2], i[3], i[4], i[5], r);
            break;
        }
        case CUTIE_OP_AREA_DOWN3: {
            Area3 a;
            long total = 0;
            for (int q = 0; q < 3; ++q) {
                const int32_t* j = i + 8 * q;
                AreaSeg& g = a.s[q];
                g.x = (const void*)p[2 * q]; g.y = (bf16_t*)p[2 * q + 1];
                g.B = j[0]; g.H = j[1]; g.W = j[2]; g.C = j[3]; g.ldx = j[4]; g.ldy = j[5]; g.r = j[6]; g.Cz = j[7];
                g.f32 = (op->flags >> q) & 1;
                if (g.r < 1 || (!g.f32 && (g.C & 7))) { cutie_set_error("area_down3: segment %d: r >= 1, C %% 8 for bf16 input", q); return -2; }
                const long np = (long)g.B * (g.H / g.r) * (g.W / g.r);
                g.n = g.f32 ? np : np * (g.C / 8);
                total += g.n;
            }
            a.rt = (op->flags >> 3) & 1;
            a.prio = PRIO;
            hipLaunchKernelGGL(area_down3_kernel, GRID1D(total, BS), dim3(BS), 0, s, a);
            break;
        }
        case CUTIE_OP_MASK_DOWN: {
            int K = i[0], r = i[3], h = i[1] / r, w = i[2] / r;
            long n = (long)K * h * w;
            float* m16 = (float*)p[2];
            if (!m16) { cutie_set_error("mask_down: m16 buffer required"); return -2; }
            hipLaunchKernelGGL(mask_down_pair_kernel, GRID1D((long)h * w * 64, BS), dim3(BS), 0, s, (const float*)p[0], m16, (uint4*)p[1], K, i[1], i[2], r,
                               i[4] > 8 ? i[4] / 8 : 1);
            break;
        }
        case CUTIE_OP_GAP: {
            if (i[2] > 256 || (i[2] & 7) || 256 % (i[2] / 8)) { cutie_set_error("gap: C must divide 256 octet-wise (C=%d)", i[2]); return -2; }
            if (!p[2]) { cutie_set_error("gap: scratch buffer required"); return -2; }
            int nchunk = (i[1] + 63) / 64;
            hipLaunchKernelGGL(gap_partial_kernel, dim3(nchunk, i[0]), dim3(256), 0, s, (const bf16_t*)p[0], (float*)p[2], i[1], i[2]);
            if (!(op->flags & 1))          // flags&1: the consumer (ECA_APPLY) finishes the reduction itself
                hipLaunchKernelGGL(gap_final_kernel, dim3(i[0]), dim3(256), 0, s, (const float*)p[2], (float*)p[1], nchunk, i[2], 1.f / (float)i[1]);
            break;
        }
        case CUTIE_OP_ECA_APPLY: {
            if (i[2] > 256 || (i[2] & 7)) { cutie_set_error("eca: C <= 256, C %% 8"); return -2; }
            if (p[6] && (i[2] != 256 || !p[8])) { cutie_set_error("eca: the fused 1x1 head needs C = 256 and an output"); return -2; }
            int nchunk = (i[1] + 63) / 64;
            hipLaunchKernelGGL(eca_apply_kernel, dim3((i[1] + 31) / 32, i[0]), dim3(256), 0, s, (const bf16_t*)p[0], (const float*)p[5], (const float*)p[2],
                               (const bf16_t*)p[3], (bf16_t*)p[4], (float*)p[1], i[1], i[2], nchunk, (op->flags & 1) ? (const long long*)p[5] : nullptr,
                               (const bf16_t*)p[6], (const float*)p[7], (float*)p[8], PRIO);
            break;
        }
        case CUTIE_OP_GRU: {
            long n = (long)i[0] * i[1];
            if (!(op->flags & 1) && (i[1] & 3) == 0 && (((uintptr_t)p[0] | (uintptr_t)p[1]) & 15) == 0 && ((uintptr_t)p[2] & 7) == 0)      // flags&1: one channel per thread (A/B switch)
                hipLaunchKernelGGL(gru4_kernel, GRID1D(n / 4, BS), dim3(BS), 0, s, (const float*)p[0], (float*)p[1], (bf16_t*)p[2], (long)i[0], i[1], PRIO);
            else
                hipLaunchKernelGGL(gru_kernel, GRID1D(n, BS), dim3(BS), 0, s, (const float*)p[0], (float*)p[1], (bf16_t*)p[2], (long)i[0], i[1]);
            break;
        }
        case CUTIE_OP_SEG_AGG:
            hipLaunchKernelGGL(seg_agg_kernel, GRID1D(i[1], BS), dim3(BS), 0, s, (const float*)p[0], (float*)p[1], i[0], i[1]);
            break;
        case CUTIE_OP_UP4_SOFTMAX: {
            long n = (long)16 * i[1] * i[2];
            if (op->flags & 1) {                              // p0 = raw logits [K,h,w] (K = P - 1): SEG_AGG fused
                if (i[0] > 16) { cutie_set_error("up4_softmax: the fused form holds P <= 16 planes in registers (P=%d)", i[0]); return -2; }
                const bool vec = !(op->flags & 2) && i[0] <= 8 && (((uintptr_t)p[1] | (uintptr_t)p[2]) & 15) == 0;      // four pixels per thread, 16-byte stores
                const int nclip = i[4] > 1 ? i[4] : 1;           // clips in lock step: grid.y (the four-pixel forms only)
                if (nclip > 1 && !vec) { cutie_set_error("up4_softmax: several clips per launch need the four-pixel form (P <= 8, aligned outputs)"); return -2; }
                if (op->flags & 4) {                              // + MASK_DOWN of the probabilities (p3 = m16, p4 = pair, i3 = channel pitch of pair)
                    if (!vec || !p[3] || !p[4] || (i[1] & 3) || (i[2] & 3) || i[3] < 8 || (i[3] & 7)) { cutie_set_error("up4_softmax: the mask-down form needs P <= 8, h, w multiples of 4, m16 and pair"); return -2; }
                    const int ncell = (i[1] / 4) * (i[2] / 4);
#define UP4_MD_(KC, SH) hipLaunchKernelGGL((up4_softmax_md_kernel<8, KC, SH>), dim3((ncell + 3) / 4, nclip), dim3(256), 0, s, (const float*)p[0], (float*)p[1], (float*)p[2], i[0] - 1, \
                                       i[1], i[2], (float*)p[3], (uint4*)p[4], i[3] / 8, PRIO)
#define UP4_MD(KC) { if (op->flags & 16) UP4_MD_(KC, false); else UP4_MD_(KC, true); }      /* flags&16: every lane aggregates its own source pixels (A/B switch) */
                    switch ((op->flags & 8) ? 0 : i[0] - 1) {          // object count as a compile-time constant (see up4_four); flags&8: run-time K
                        case 1: UP4_MD(1); break; case 2: UP4_MD(2); break; case 3: UP4_MD(3); break; case 4: UP4_MD(4); break;
                        case 5: UP4_MD(5); break; case 6: UP4_MD(6); break; case 7: UP4_MD(7); break; default: UP4_MD(0); break;
                    }
#undef UP4_MD
#undef UP4_MD_
                    break;
                }
                if (vec) {
                    const long n4 = (long)4 * i[1] * i[2];
#define UP4_F4(KC) hipLaunchKernelGGL((up4_softmax_fused4_kernel<8, KC>), dim3((unsigned)((n4 + BS - 1) / BS), nclip), dim3(BS), 0, s, (const float*)p[0], (float*)p[1], (float*)p[2], i[0] - 1, i[1], i[2], PRIO)
                    switch ((op->flags & 8) ? 0 : i[0] - 1) {
                        case 1: UP4_F4(1); break; case 2: UP4_F4(2); break; case 3: UP4_F4(3); break; case 4: UP4_F4(4); break;
                        case 5: UP4_F4(5); break; case 6: UP4_F4(6); break; case 7: UP4_F4(7); break; default: UP4_F4(0); break;
                    }
#undef UP4_F4
                    break;
                }
                if (i[0] <= 8) hipLaunchKernelGGL(up4_softmax_fused_kernel<8>, GRID1D(n, BS), dim3(BS), 0, s, (const float*)p[0], (float*)p[1], (float*)p[2], i[0] - 1, i[1], i[2]);
                else hipLaunchKernelGGL(up4_softmax_fused_kernel<16>, GRID1D(n, BS), dim3(BS), 0, s, (const float*)p[0], (float*)p[1], (float*)p[2], i[0] - 1, i[1], i[2]);
                break;
            }
            hipLaunchKernelGGL(up4_softmax_kernel, GRID1D(n, BS), dim3(BS), 0, s, (const float*)p[0], (float*)p[1], (float*)p[2], i[0], i[1], i[2]);
            break;
        }
        case CUTIE_OP_MASK_MERGE: {
            long n = (long)i[2] * i[3];
            hipLaunchKernelGGL(mask_merge_kernel, GRID1D(n, BS), dim3(BS), 0, s, (const void*)p[0], (const float*)p[1], (const int*)p[2], (float*)p[3],
                               i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8], op->flags & 1);
            break;
        }
        case CUTIE_OP_AGG_SOFTMAX:
            hipLaunchKernelGGL(agg_softmax_kernel, GRID1D(i[1], BS), dim3(BS), 0, s, (const float*)p[0], (float*)p[1], i[0], (long)i[1]);
            break;
        case CUTIE_OP_LINEAR: {
            if ((i[2] & 7) || (i[3] & 3) || i[2] < 256) { cutie_set_error("linear: Kd %% 8, ldx %% 4, Kd >= 256 required (Kd=%d)", i[2]); return -2; }
            const int add_rows = i[5] > 0 ? i[5] : 1;
            const bool ln = op->flags & 2;
            if (ln && (i[2] != 256 || !p[6] || !p[7] || (i[3] & 3))) { cutie_set_error("linear: fused LayerNorm needs Kd == 256, gamma and beta"); return -2; }
            if ((ln || i[6] > 0) && (i[2] & 127)) { cutie_set_error("linear: LayerNorm / add_cols need Kd %% 128 == 0"); return -2; }
            if (i[6] > 0 && (i[6] & 15)) { cutie_set_error("linear: add_cols must be a multiple of 16"); return -2; }
            if ((i[2] & 511) == 0 && i[2] >= 1024 && !ln)                // long K (FFN linear2, K = 2048): 16 waves share it
                hipLaunchKernelGGL(linear_mfma_kernel<16>, dim3((i[1] + 15) / 16, (i[0] + 15) / 16), dim3(1024), 0, s, (const float*)p[0], (const float*)p[1], (const bf16_t*)p[2],
                                   (const float*)p[3], (const float*)p[4], (float*)p[5], i[0], i[1], i[2], i[3], i[4], add_rows, op->flags & 1, i[6],
                                   (const float*)nullptr, (const float*)nullptr, (float*)nullptr, 1e-5f);
            else if ((i[2] & 127) == 0)
                hipLaunchKernelGGL(linear_mfma_kernel<4>, dim3((i[1] + 15) / 16, (i[0] + 15) / 16), dim3(256), 0, s, (const float*)p[0], (const float*)p[1], (const bf16_t*)p[2],
                                   (const float*)p[3], (const float*)p[4], (float*)p[5], i[0], i[1], i[2], i[3], i[4], add_rows, op->flags & 1, i[6],
                                   ln ? (const float*)p[6] : nullptr, (const float*)p[7], (float*)p[8], op->f[0] > 0.f ? op->f[0] : 1e-5f);
            else if (i[2] >= 512)
                hipLaunchKernelGGL(linear_small_kernel<64>, dim3((i[1] + 3) / 4, (i[0] + 15) / 16), dim3(256), 0, s, (const float*)p[0], (const float*)p[1], (const bf16_t*)p[2],
                                   (const float*)p[3], (const float*)p[4], (float*)p[5], i[0], i[1], i[2], i[3], i[4], add_rows, op->flags & 1);
            else
                hipLaunchKernelGGL(linear_small_kernel<32>, dim3((i[1] + 7) / 8, (i[0] + 15) / 16), dim3(256), 0, s, (const float*)p[0], (const float*)p[1], (const bf16_t*)p[2],
                                   (const float*)p[3], (const float*)p[4], (float*)p[5], i[0], i[1], i[2], i[3], i[4], add_rows, op->flags & 1);
            break;
        }
        case CUTIE_OP_LAYERNORM:
            hipLaunchKernelGGL(layernorm_kernel, dim3((i[0] + 3) / 4), dim3(256), 0, s, (const float*)p[0], (const float*)p[1], (const float*)p[2], (float*)p[3], i[0], i[1]);
            break;
        case CUTIE_OP_QUERY_INIT: {
            long n = (long)i[0] * i[1];
            hipLaunchKernelGGL(query_init_kernel, GRID1D(n, BS), dim3(BS), 0, s, (const float*)p[0], (float*)p[1], i[0], i[1]);
            break;
        }
        case CUTIE_OP_SUMMARIZE:
            if (i[3] != 16 || (i[2] & 63)) { cutie_set_error("summarize: Q must be 16, C %% 64"); return -2; }
            {
                if (!p[4]) { cutie_set_error("summarize: scratch buffer required"); return -2; }
                int nchunk = (i[1] + 127) / 128, n = i[3] * (i[2] + 1);
                const int ldf = i[4] > 0 ? i[4] : i[2], ldw = i[5] > 0 ? i[5] : i[3];
                if (op->flags & 1) hipLaunchKernelGGL(summarize_kernel<true>, dim3(i[2] / 64, i[0], nchunk), dim3(256), 0, s, (const void*)p[0], (const float*)p[1], (const float*)p[2], (float*)p[4], i[1], i[2], i[3], ldf, ldw);
                else hipLaunchKernelGGL(summarize_kernel<false>, dim3(i[2] / 64, i[0], nchunk), dim3(256), 0, s, (const void*)p[0], (const float*)p[1], (const float*)p[2], (float*)p[4], i[1], i[2], i[3], ldf, ldw);
                hipLaunchKernelGGL(summarize_final_kernel, dim3((n + 255) / 256, i[0]), dim3(256), 0, s, (const float*)p[4], (float*)p[3], nchunk, n);
            }
            break;
        case CUTIE_OP_ADD_PE: {
            long n8 = i[1] / 8;
            hipLaunchKernelGGL(add_pe_kernel, GRID1D((long)i[0] * n8, BS), dim3(BS), 0, s, (const uint4*)p[0], (const uint4*)p[1], (uint4*)p[2], i[0], n8);
            break;
        }
        case CUTIE_OP_MEMSET32:
            hipLaunchKernelGGL(memset32_kernel, GRID1D(i[0], BS), dim3(BS), 0, s, (uint32_t*)p[0], (long)i[0], (uint32_t)i[1]);
            break;
        case CUTIE_OP_COPY2D: {
            int roww = i[1] / 4;
            long n = (long)i[0] * roww;
            hipLaunchKernelGGL(copy2d_kernel, GRID1D(n, BS), dim3(BS), 0, s, (const uint32_t*)p[0], (uint32_t*)p[1], (long)i[0], roww, (long)i[2] / 4, (long)i[3] / 4);
            break;
        }
        case CUTIE_OP_BANK_WRITE: {
            BankWrite w;
            int most = 0;
            for (int k = 0; k < 6; ++k) {
                w.src[k] = (const uint32_t*)p[2 * k]; w.dst[k] = (uint32_t*)p[2 * k + 1]; w.words[k] = (p[2 * k] && p[2 * k + 1]) ? i[k] : 0;
                if (w.words[k] < 0) { cutie_set_error("bank_write: negative size"); return -2; }
                most = w.words[k] > most ? w.words[k] : most;
            }
            for (int k = 0; k < 2; ++k) {
                w.dst[6 + k] = (uint32_t*)p[12 + k]; w.words[6 + k] = p[12 + k] ? i[6 + k] : 0; w.pattern[k] = (uint32_t)i[8 + k];
                if (w.words[6 + k] < 0) { cutie_set_error("bank_write: negative size"); return -2; }
                most = w.words[6 + k] > most ? w.words[6 + k] : most;
            }
            if (most > 0) {
                int gx = (most + 1023) / 1024;
                gx = gx > 128 ? 128 : gx;                 // 8 segments x 128 blocks: every CU gets work, a block moves >= 4 KB per pass
                hipLaunchKernelGGL(bank_write_kernel, dim3(gx, 8), dim3(256), 0, s, w);
            }
            break;
        }
        case CUTIE_OP_AXPY:
            hipLaunchKernelGGL(axpy_kernel, GRID1D(i[0], BS), dim3(BS), 0, s, (const float*)p[0], (float*)p[1], (long)i[0], op->f[0]);
            break;
        case CUTIE_OP_USAGE_TICK:
        {
            const long nA = p[0] ? i[0] : 0, nB = p[1] ? i[1] : 0, nU = (p[2] && p[3]) ? i[2] : 0;
            const long n = nA > nB ? (nA > nU ? nA : nU) : (nB > nU ? nB : nU);
            if (n > 0)
                hipLaunchKernelGGL(tick_kernel, GRID1D(n, BS), dim3(BS), 0, s, (float*)p[0], nA, (float*)p[1], nB, (float*)p[2], (float*)p[3], nU, op->flags & 3);
        }
            break;
        case CUTIE_OP_CAST:
            hipLaunchKernelGGL(cast_kernel, GRID1D(i[0], BS), dim3(BS), 0, s, (const void*)p[0], (void*)p[1], (long)i[0], op->flags & 1);
            break;
        case CUTIE_OP_FLIP_W:
            if (i[0] < 1 || i[1] < 1) { cutie_set_error("flip_w: empty shape"); return -2; }
            hipLaunchKernelGGL(flip_w_kernel, GRID1D((long)i[0] * i[1], BS), dim3(BS), 0, s, (const float*)p[0], (float*)p[1], (long)i[0], i[1], i[2], i[3],
                               op->f[0], op->f[1]);
            break;
        case CUTIE_OP_RESIZE: {
            const long n = (long)i[0] * i[3] * i[4];
            if (i[1] < 1 || i[2] < 1 || i[3] < 1 || i[4] < 1) { cutie_set_error("resize: empty shape"); return -2; }
            hipLaunchKernelGGL(resize_kernel, GRID1D(n, BS), dim3(BS), 0, s, (const float*)p[0], (float*)p[1], i[0], i[1], i[2], i[3], i[4],
                               (long)i[5], i[6], op->flags & 1);
            break;
        }
        case CUTIE_OP_PROB_TO_ID: {
            const long n = (long)i[1] * i[2];
            const long plane = (long)i[3];
            if (i[0] < 1) { cutie_set_error("prob_to_id: P >= 1"); return -2; }
            if ((op->flags & 3) == 0)
                hipLaunchKernelGGL(prob_to_id_kernel<uint8_t>, GRID1D(n, BS), dim3(BS), 0, s, (const float*)p[0], (const int*)p[1], (uint8_t*)p[2], i[0], i[1], i[2], plane, i[4]);
            else if ((op->flags & 3) == 1)
                hipLaunchKernelGGL(prob_to_id_kernel<int32_t>, GRID1D(n, BS), dim3(BS), 0, s, (const float*)p[0], (const int*)p[1], (int32_t*)p[2], i[0], i[1], i[2], plane, i[4]);
            else
                hipLaunchKernelGGL(prob_to_id_kernel<long long>, GRID1D(n, BS), dim3(BS), 0, s, (const float*)p[0], (const int*)p[1], (long long*)p[2], i[0], i[1], i[2], plane, i[4]);
            break;
        }
        default:
            cutie_set_error("elementwise: unknown op kind %d", op->kind);
            return -3;
    }
    return (int)hipGetLastError();
}
