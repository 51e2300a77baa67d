// L2 -> CU streaming micro-benchmark (tools only): per-CU fill bandwidth vs 16-B loads in flight per thread.
// hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o gpurun_out/membench && gpurun_out/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int U, bool LDS>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ src, unsigned* __restrict__ out, int chunks_per_block, int nchunk_total) {
    __shared__ u32x4 lds[LDS ? 256 * U : 1];
    const int tid = threadIdx.x;
    // every block walks its own window of the (L2-resident) buffer: wave-contiguous 1 KiB rows
    long base = ((long)blockIdx.x * 7919 * 256) % nchunk_total;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < chunks_per_block; it += U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long idx = (base + (long)(it + u) * 256 + tid) % nchunk_total;
            v[u] = src[idx];
        }
        if (LDS) {
#pragma unroll
            for (int u = 0; u < U; ++u) lds[u * 256 + tid] = v[u];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < U; ++u) { u32x4 t = lds[u * 256 + (tid ^ 1)]; acc.x ^= t.x; acc.y ^= t.y; acc.z ^= t.z; acc.w ^= t.w; }
            __syncthreads();
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int U, bool LDS>
void run(const u32x4* src, unsigned* out, int nblocks, int nchunk_total, const char* tag) {
    const int cpb = 96 * U;   // chunks (x256 threads x16 B) per block
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((stream_kernel<U, LDS>), dim3(nblocks), dim3(256), 0, 0, src, out, cpb, nchunk_total);
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((stream_kernel<U, LDS>), dim3(nblocks), dim3(256), 0, 0, src, out, cpb, nchunk_total);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    double bytes = (double)nblocks * cpb * 256 * 16;
    printf("%s U=%2d blocks=%4d  %7.1f us  %7.2f TB/s total  %6.1f GB/s per block\n", tag, U, nblocks, ms * 1e3, bytes / ms / 1e9, bytes / nblocks / ms / 1e6);
}

int main() {
    for (long mb : {2L, 64L}) {
        const long nchunk = mb * 1024 * 1024 / 16;
        u32x4* src; unsigned* out;
        hipMalloc(&src, nchunk * 16); hipMalloc(&out, 4);
        hipMemset(src, 1, nchunk * 16);
        printf("---- buffer %ld MiB\n", mb);
        for (int nb : {256, 512, 1024}) {
            run<2, false>(src, out, nb, (int)nchunk, "reg ");
            run<4, false>(src, out, nb, (int)nchunk, "reg ");
            run<8, false>(src, out, nb, (int)nchunk, "reg ");
            run<16, false>(src, out, nb, (int)nchunk, "reg ");
            run<6, true>(src, out, nb, (int)nchunk, "lds ");
            run<12, true>(src, out, nb, (int)nchunk, "lds ");
        }
        hipFree(src); hipFree(out);
    }
    return 0;
}
