// STEM: the first three launches of the ResNet encoders in one -- IMG_PREP (normalise, pad, attach mask / others planes), the 7x7 /
// stride-2 / pad-3 convolution with folded BatchNorm (pixel_encoder.conv1, mask_encoder.conv1: resnet.py conv1 + bn1; big_modules.py
// 30-33, 95-100) and the 3x3 / stride-2 / pad-1 max pooling (+ ReLU: before the pool in the pixel encoder, after it in the mask
// encoder -- the same thing, ReLU and max commute).  Round 2 ran them as three launches through two HBM round trips of a stride-2 map
// (480p: 30.4 + 10.2 + 6.6 us per frame, the 7x7 conv at 2.6 % of the MFMA peak: Cin = 3 padded to 8, register-staged im2col);
// this kernel: 13.5 us (4 x 16 tiles with per-wave weight loads: 24 us; weights staged once per block: 17.7; 8 x 16 tiles, one round: 13.5).
//
// One block = ST_PH x 16 pooled pixels of one object (8 x 16: they need 17 x 33 conv pixels, which need a 39 x 71 input patch; 210 blocks
// at 480p: one round on 256 CUs).
//   1. the patch is built in LDS as [39][72] pixels x 8 bf16 channels (r, g, b, mask, others, 0, 0, 0), straight from the fp32 frame;
//      pixels outside the padded frame are the conv's zero padding, the 72nd column is zero (the 8th "tap" of a row, see below);
//   2. implicit GEMM on v_mfma_f32_16x16x32_bf16: a K step is half a kernel row = 4 taps x 8 channels, so a B fragment is ONE
//      16-byte LDS read of the pixel the tap lands on (lane group g <-> tap 4 (s & 1) + g; the 8th tap has zero weights); A = the
//      weights (staged once per block in LDS in fragment order, then 2 x 14 fragments per wave in registers); 8 waves = 4 (pixel groups) x 2 (channel halves);
//   3. bias (+ nothing else: ReLU moves behind the pool), bf16, into an LDS tile [561][64] that aliases the patch; conv pixels outside
//      the conv's output range are written as -inf (the pool's padding);
//   4. 3 x 3 max over the tile, ReLU, 32-byte stores.
// Results: identical rounding points as the three-launch form (bf16 after the conv, max of bf16 values); only the fp32 summation
// order inside the conv differs.
#include "common.h"
#include <math.h>

#ifndef ST_PH
#define ST_PH 8
#endif
#define ST_PW 16
#define ST_CH (2 * ST_PH + 1)                            // conv rows of a block
#define ST_CW (2 * ST_PW + 1)                            // conv columns (33)
#define ST_NCONV (ST_CH * ST_CW)
#define ST_MF ((ST_NCONV + 15) / 16)                     // M fragments
#define ST_IH (2 * ST_CH + 5)                            // input rows
#define ST_IW 72                                         // input columns: 71 + one zero column
#define ST_CLD 72                                        // bf16 pitch of the conv tile (64 + 8: rows 144 B apart)

typedef __attribute__((ext_vector_type(4))) unsigned int st_u4;

#define ST_MAXIMG 12
struct StemParams {
    const float* img; const float* masks; const bf16_t* W; const float* bias; bf16_t* y;
    int h0, w0, H, W_, pl, pt, K, Kpad, relu, gx, nb;
    float m0, m1, m2, s0, s1, s2;
    // ABI 4: several frames per launch (grid.y): frame f > 0 reads imgs[f - 1] (f = 0: img), masks + f * mstride floats and writes
    // y + f * K * (H/4) * (W/4) * 64 -- the look-ahead window of the image encoder / the clips of a lock-step group in ONE round-filling launch
    const float* imgs[ST_MAXIMG - 1]; long mstride;
};

#define ST_NT 512                                        // 8 waves: 4 pixel groups x 2 channel halves (two waves per SIMD hide each other's LDS latency)
__global__ __launch_bounds__(ST_NT) void stem_kernel(StemParams p) {
    // dynamic LDS: [conv tile 82944 B (the 44928-B patch aliases its start)][weights in fragment order 57344 B]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    static_assert(ST_IH * ST_IW * 16 <= ST_MF * 16 * ST_CLD * 2, "patch fits");
    st_u4* patch = reinterpret_cast<st_u4*>(lds);
    bf16_t* ct = reinterpret_cast<bf16_t*>(lds);
    st_u4* wl = reinterpret_cast<st_u4*>(lds + ST_MF * 16 * ST_CLD * 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const int wm = wave & 3, wn = wave >> 2;
    const int k = blockIdx.z;
    const int OH = p.H >> 1, OW = p.W_ >> 1, PH = p.H >> 2, PW = p.W_ >> 2;
    // (block-uniform) frame blockIdx.y of the launch: its image, masks and output (locals: writing to the by-value parameter struct sends it to scratch)
    const int fr = blockIdx.y;
    const float* __restrict__ const img_ = fr > 0 ? p.imgs[fr - 1] : p.img;
    const float* __restrict__ const masks_ = p.masks ? p.masks + (long)fr * p.mstride : nullptr;
    bf16_t* __restrict__ const y_ = p.y + (long)fr * p.K * PH * PW * 64;
    // XCD-aware order: hardware block b runs on XCD b % 8; every XCD gets one contiguous band of tiles, so the halo pixels that
    // neighbouring tiles share are fetched into ONE L2 (round-robin order: 21.5 MB fetched per launch for a 4.9 MB frame, rocprofv3 PMC)
    const int per = (p.nb + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (logical >= p.nb || (int)(blockIdx.x >> 3) >= per) return;
    const int by = logical / p.gx, bx = logical - by * p.gx;
    const int py0 = by * ST_PH, px0 = bx * ST_PW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;     // first conv pixel of the block
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;     // first input pixel

    // ---- weights -> LDS in fragment order, once per block (every wave loading its own 28 fragments from global memory pulled 224 KB
    // per block through one CU: the bytes-per-CU bound of profiles/r03_qchain.md).  Piece (nf, s, lane): channel row 16 nf + (lane & 15),
    // taps 4 (s & 1) + (lane >> 4) of kernel row s >> 1; the 8th tap of a row is zero ----
    st_u4 wpre[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int pc = tid + ST_NT * i;                  // 0 .. 3583 = 4 x 14 x 64
        const int l_ = pc & 63, fs = pc >> 6, nf = fs / 14, s_ = fs - nf * 14;
        const int kw = (s_ & 1) * 4 + (l_ >> 4);
        const st_u4 v = *reinterpret_cast<const st_u4*>(p.W + (long)(nf * 16 + (l_ & 15)) * p.Kpad + ((s_ >> 1) * 7 + min(kw, 6)) * 8);
        const st_u4 z = {0u, 0u, 0u, 0u};
        wpre[i] = kw < 7 ? v : z;
    }
    // ---- the input patch: 6 pixels per thread, all their loads issued before the first is used ----
    const long plane = (long)p.h0 * p.w0, HWp = (long)p.H * p.W_;
    constexpr int NPIX = (ST_IH * ST_IW + ST_NT - 1) / ST_NT;   // 6
    float pr_[NPIX], pg_[NPIX], pb_[NPIX], pm_[NPIX], po_[NPIX];
    bool pin_[NPIX], pim_[NPIX];
    long mpix_[NPIX];
#pragma unroll
    for (int t = 0; t < NPIX; ++t) {
        const int e = min(tid + ST_NT * t, ST_IH * ST_IW - 1);
        const int ry = e / ST_IW, rx = e - ry * ST_IW;
        const int iy = iy0 + ry, ix = ix0 + rx;
        pin_[t] = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W_ && rx < ST_IW - 1;
        const int sy = iy - p.pt, sx = ix - p.pl;
        pim_[t] = pin_[t] && (unsigned)sy < (unsigned)p.h0 && (unsigned)sx < (unsigned)p.w0;
        const long o = (long)min(max(sy, 0), p.h0 - 1) * p.w0 + min(max(sx, 0), p.w0 - 1);     // clamped: the loads are unconditional
        pr_[t] = img_[o]; pg_[t] = img_[plane + o]; pb_[t] = img_[2 * plane + o];
        pm_[t] = 0.f; po_[t] = 0.f;
        mpix_[t] = (long)min(max(iy, 0), p.H - 1) * p.W_ + min(max(ix, 0), p.W_ - 1);
    }
    if (masks_) {                                       // (block-uniform)
        // four objects per round, the loads of all six pixels in flight together (a `for (j < K) sum += masks[j]` per pixel compiled to
        // one dependent round trip per object and pixel: 6 (K + 1) of them in front of the first MFMA; tools/isa_waits.py).  Same
        // summation order per pixel: objects ascending.
        float msum[NPIX];
#pragma unroll
        for (int t = 0; t < NPIX; ++t) { msum[t] = 0.f; pm_[t] = masks_[(long)k * HWp + mpix_[t]]; }
        for (int j0 = 0; j0 < p.K; j0 += 4) {
            float mv[NPIX][4];
#pragma unroll
            for (int t = 0; t < NPIX; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) mv[t][u] = masks_[(long)min(j0 + u, p.K - 1) * HWp + mpix_[t]];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + u < p.K) {
#pragma unroll
                    for (int t = 0; t < NPIX; ++t) msum[t] += mv[t][u];
                }
        }
#pragma unroll
        for (int t = 0; t < NPIX; ++t) po_[t] = fminf(fmaxf(msum[t] - pm_[t], 0.f), 1.f);
    }
#pragma unroll
    for (int t = 0; t < NPIX; ++t) {
        const int e = tid + ST_NT * t;
        float r = pim_[t] ? pr_[t] : 0.f, gg = pim_[t] ? pg_[t] : 0.f, b = pim_[t] ? pb_[t] : 0.f;
        r = (r - p.m0) / p.s0; gg = (gg - p.m1) / p.s1; b = (b - p.m2) / p.s2;            // (the arithmetic of IMG_PREP, to the bit)
        const st_u4 v = {pack_bf2(r, gg), pack_bf2(b, pm_[t]), pack_bf2(po_[t], 0.f), 0u};
        const st_u4 z = {0u, 0u, 0u, 0u};
        if (e < ST_IH * ST_IW) patch[e] = pin_[t] ? v : z;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) wl[tid + ST_NT * i] = wpre[i];
    __syncthreads();
    st_u4 wa[2][14];                                     // this wave's 2 x 14 weight fragments, held in registers for the whole block
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int s = 0; s < 14; ++s) wa[n][s] = wl[((2 * wn + n) * 14 + s) * 64 + lane];
    // ---- implicit GEMM: 14 K steps; this wave: M fragments NB wm .. (9 or 8 of them), N fragments 2 wn, 2 wn + 1 ----
    constexpr int NB = (ST_MF + 3) / 4;
    const int nb = min(NB, ST_MF - wm * NB);
    int boff[NB];                                        // patch index of (conv pixel, tap g) at kernel row 0, first half
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int idx = min((wm * NB + b) * 16 + c, ST_NCONV - 1);
        const int cyl = idx / ST_CW, cxl = idx - cyl * ST_CW;
        boff[b] = 2 * cyl * ST_IW + 2 * cxl + g;
    }
    f32x4 acc[NB][2];
#pragma unroll
    for (int b = 0; b < NB; ++b) { acc[b][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[b][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int so = (s >> 1) * ST_IW + (s & 1) * 4;
        st_u4 bf[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) bf[b] = patch[boff[b] + so];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b < nb) {
                acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[0][s]), __builtin_bit_cast(bf16x8, bf[b]), acc[b][0], 0, 0, 0);
                acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[1][s]), __builtin_bit_cast(bf16x8, bf[b]), acc[b][1], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                     // every wave has read its last patch fragment: the conv tile may overwrite it
    // ---- bias, bf16, conv tile (lane: conv pixel c of the fragment, channels 4 g .. 4 g + 3 of the N fragment) ----
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int idx = (wm * NB + b) * 16 + c;
        if (b < nb && idx < ST_NCONV) {
            const int cyl = idx / ST_CW, cxl = idx - cyl * ST_CW;
            const bool valid = (unsigned)(cy0 + cyl) < (unsigned)OH && (unsigned)(cx0 + cxl) < (unsigned)OW;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int ch = (2 * wn + n) * 16 + 4 * g;
                const float4 bv = *reinterpret_cast<const float4*>(p.bias + ch);
                uint2 o = make_uint2(pack_bf2(acc[b][n][0] + bv.x, acc[b][n][1] + bv.y), pack_bf2(acc[b][n][2] + bv.z, acc[b][n][3] + bv.w));
                if (!valid) o = make_uint2(0xff80ff80u, 0xff80ff80u);       // -inf: the pool's padding
                *reinterpret_cast<uint2*>(ct + idx * ST_CLD + ch) = o;
            }
        }
    }
    __syncthreads();
    // ---- 3 x 3 / stride 2 max pool (+ ReLU): thread = (pooled pixel, 8 channels) ----
    {
        for (int task = tid; task < ST_PH * ST_PW * 8; task += ST_NT) {
        const int pp = task >> 3, q = task & 7, ppy = pp >> 4, ppx = pp & 15;
        const int py = py0 + ppy, px = px0 + ppx;
        float m[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const st_u4 a = *reinterpret_cast<const st_u4*>(ct + ((2 * ppy + dy) * ST_CW + 2 * ppx + dx) * ST_CLD + q * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    m[2 * i] = fmaxf(m[2 * i], __uint_as_float(a[i] << 16));
                    m[2 * i + 1] = fmaxf(m[2 * i + 1], __uint_as_float(a[i] & 0xffff0000u));
                }
            }
        if (p.relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], 0.f);
        }
        if (py < PH && px < PW)
            *reinterpret_cast<uint4*>(y_ + (((long)k * PH + py) * PW + px) * 64 + q * 8) =
                make_uint4(pack_bf2(m[0], m[1]), pack_bf2(m[2], m[3]), pack_bf2(m[4], m[5]), pack_bf2(m[6], m[7]));
        }
    }
}

int launch_stem(const cutie_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const uint64_t* q = op->p;
    if (!q[0] || !q[2] || !q[3] || !q[4]) { cutie_set_error("stem: image, packed weights, bias and output required"); return -2; }
    if ((i[2] & 15) || (i[3] & 15) || i[6] < 1 || i[7] < 392 || (i[7] & 7)) {
        cutie_set_error("stem: H, W multiples of 16, K >= 1, Kpad >= 392 (H=%d W=%d K=%d Kpad=%d)", i[2], i[3], i[6], i[7]);
        return -2;
    }
    StemParams p;
    p.img = (const float*)q[0]; p.masks = (const float*)q[1]; p.W = (const bf16_t*)q[2]; p.bias = (const float*)q[3]; p.y = (bf16_t*)q[4];
    p.h0 = i[0]; p.w0 = i[1]; p.H = i[2]; p.W_ = i[3]; p.pl = i[4]; p.pt = i[5]; p.K = q[1] ? i[6] : 1; p.Kpad = i[7]; p.relu = op->flags & 1;
    p.m0 = op->f[0]; p.m1 = op->f[1]; p.m2 = op->f[2]; p.s0 = op->f[3]; p.s1 = op->f[4]; p.s2 = op->f[5];
    const int PH = i[2] >> 2, PW = i[3] >> 2;
    p.gx = (PW + ST_PW - 1) / ST_PW;
    p.nb = p.gx * ((PH + ST_PH - 1) / ST_PH);
    constexpr size_t lds = ST_MF * 16 * ST_CLD * 2 + 4 * 14 * 64 * 16;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            cutie_set_error("stem: cannot raise the dynamic LDS limit to %d bytes", (int)lds);
            return -2;
        }
        attr_set = true;
    }
    int nimg = i[8] > 1 ? i[8] : 1;                      // frames per launch: image f > 0 at p[4 + f] (p5 .. p15), masks i9 floats apart
    if (nimg > ST_MAXIMG) { cutie_set_error("stem: at most %d frames per launch (%d)", ST_MAXIMG, nimg); return -2; }
    p.mstride = i[9];
    for (int f = 1; f < ST_MAXIMG; ++f) {
        p.imgs[f - 1] = f < nimg ? (const float*)q[4 + f] : nullptr;
        if (f < nimg && !q[4 + f]) { cutie_set_error("stem: frame %d of %d has no image", f, nimg); return -2; }
    }
    hipLaunchKernelGGL(stem_kernel, dim3(((p.nb + 7) / 8) * 8, nimg, p.K), dim3(ST_NT), lds, s, p);
    return (int)hipGetLastError();
}
