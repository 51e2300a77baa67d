// L2 -> LDS streaming micro-benchmark with LDS-DMA (buffer_load_dwordx4 ... lds), tools only: what per-CU fill rate does the
// tile pipeline of conv_dma.hip have available, as a function of waves per block, blocks per CU, ring depth, pieces per wave and
// the per-tile barrier?  No MFMA, no LDS reads: the upper bound of the operand stream.
//   hipcc --offload-arch=gfx950 -O3 tools/dmabench.hip -o gpurun_out/dmabench && gpurun_out/dmabench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define WAIT_VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | (7 << 4) | (15 << 8))

// NW waves, P pieces (1 KiB each) per wave and tile, NS ring stages, BAR: barrier per tile, ROWS: a piece = 8 rows x 128 B at a 256-B pitch
template <int NW, int P, int NS, bool BAR, bool ROWS, int SHARE = 0>
__global__ __launch_bounds__(NW * 64) void dma_stream(const char* src, unsigned window, int tiles, unsigned* out) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int STAGE = NW * P * 1024;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x7fffffff, 0x00020000);
    unsigned voff[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const unsigned piece = (unsigned)((wave * P + i) * (ROWS ? 2048 : 1024));
        voff[i] = piece + (ROWS ? (unsigned)((lane >> 3) * 256 + (((lane & 7) ^ (lane >> 3)) * 16)) : (unsigned)lane * 16u);
    }
    // SHARE 0: every block streams its own part of the window; 1: ALL blocks stream the same bytes in lockstep (the weight operand
    // of a GEMM); 2: half of the pieces shared, half private (weights + activations); 3: shared bytes, but every block starts at
    // its own tile of the shared sequence (staggered K loop)
    unsigned soff = (unsigned)(((unsigned long long)blockIdx.x * 40503u * 4096u) % window);      // this block's start inside the window
    unsigned soffS = SHARE == 3 ? (unsigned)((blockIdx.x * 5u) % 64u) * (unsigned)(NW * P * 1024 * (ROWS ? 2 : 1)) : 0u;
    const unsigned shared_span = 64u * (unsigned)(NW * P * 1024 * (ROWS ? 2 : 1));               // 64 tiles of shared data, cycled
    const unsigned tile_bytes = STAGE * (ROWS ? 2 : 1);
    int ld = 0;
#define ISSUE()                                                                                             \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < P; ++i)                                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds + ld + (wave * P + i) * 1024), 16, voff[i],         \
                (SHARE == 1 || SHARE == 3 || (SHARE == 2 && i < P / 2)) ? soffS : soff, 0, 0);                      \
        soff += tile_bytes; soff = soff >= window ? soff - window : soff;                                   \
        soffS += tile_bytes; soffS = soffS >= shared_span ? soffS - shared_span : soffS;                    \
        ld = ld == (NS - 1) * STAGE ? 0 : ld + STAGE;                                                       \
    }
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) ISSUE()
    for (int t = 0; t < tiles; ++t) {
        ISSUE()
        WAIT_VMCNT((NS - 1) * P);
        if (BAR) { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    }
    WAIT_VMCNT(0);
    __syncthreads();
    if (reinterpret_cast<unsigned*>(lds)[threadIdx.x] == 0x12345678u) out[0] = 1;
#endif
}

template <int NW, int P, int NS, bool BAR, bool ROWS, int SHARE = 0>
void run(const char* src, unsigned window, unsigned* out, int blocks_per_cu) {
    constexpr int STAGE = NW * P * 1024;
    const int lds = 160 * 1024 / blocks_per_cu / 1024 * 1024 - (blocks_per_cu > 1 ? 1024 : 0);
    if (lds < NS * STAGE) return;
    hipFuncSetAttribute(reinterpret_cast<const void*>(dma_stream<NW, P, NS, BAR, ROWS, SHARE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int nblocks = 256 * blocks_per_cu, tiles = 512 / P;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((dma_stream<NW, P, NS, BAR, ROWS, SHARE>), dim3(nblocks), dim3(NW * 64), lds, 0, src, window, tiles, out);
    hipEventRecord(a, 0);
    for (int q = 0; q < 5; ++q) hipLaunchKernelGGL((dma_stream<NW, P, NS, BAR, ROWS, SHARE>), dim3(nblocks), dim3(NW * 64), lds, 0, src, window, tiles, out);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double bytes = (double)nblocks * tiles * STAGE;
    printf("waves/blk %2d  blk/CU %d (%2d waves/CU)  pieces/wave %d  ring %d  barrier %d  rows %d  share %d : %7.1f us  %6.2f TB/s  %6.1f GB/s per CU  (%5.2f us per %d KiB tile)\n",
           NW, blocks_per_cu, NW * blocks_per_cu, P, NS, (int)BAR, (int)ROWS, SHARE, ms * 1e3, bytes / ms / 1e9, bytes / 256 / ms / 1e6, ms * 1e3 / tiles, STAGE / 1024);
}

int main() {
    for (unsigned mb : {2u, 24u}) {
        const unsigned window = mb << 20;
        char* src; unsigned* out;
        hipMalloc(&src, (size_t)window + (16u << 20)); hipMalloc(&out, 4);
        hipMemset(src, 1, (size_t)window + (16u << 20));
        printf("---- window %u MiB\n", mb);
        for (int bpc : {1, 2, 3}) {
            run<4, 8, 3, true, true, 0>(src, window, out, bpc);    // conv_dma 128x128, 4 waves: private data
            run<4, 8, 3, true, true, 1>(src, window, out, bpc);    // all blocks the same bytes in lockstep
            run<4, 8, 3, true, true, 2>(src, window, out, bpc);    // half shared (weights), half private (activations)
            run<4, 8, 3, true, true, 3>(src, window, out, bpc);    // shared bytes, staggered start
            run<4, 4, 3, true, true, 0>(src, window, out, bpc);    // 64x64
            run<4, 4, 3, true, true, 1>(src, window, out, bpc);
            run<4, 4, 3, true, true, 2>(src, window, out, bpc);
            run<4, 4, 3, true, true, 3>(src, window, out, bpc);
        }
        hipFree(src); hipFree(out);
    }
    return 0;
}
