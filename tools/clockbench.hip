// Shader clock estimate: s_memtime (core clock) vs wall_clock64 (100 MHz) around a dependent-FMA loop.  tools only.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void clk(float* out, long long* t, int iters, int mfma) {
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    long long c0 = clock64(), w0 = wall_clock64();
    float a = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    bf16x8 x; for (int i = 0; i < 8; ++i) x[i] = (__bf16)1.0f;
    for (int i = 0; i < iters; ++i) {
        if (mfma) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, acc, 0, 0, 0);
        else a = a * 1.0001f + 0.5f;
    }
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + acc[0];
}
int main() {
    float* out; long long* t; hipMalloc(&out, 1024 * 256 * 4 * 4); hipMalloc(&t, 16);
    for (int mfma = 0; mfma < 2; ++mfma)
        for (int blocks : {1, 256, 1024})
            for (int iters : {2000, 20000, 200000, 2000000}) {
                long long h[2];
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                hipLaunchKernelGGL(clk, dim3(blocks), dim3(256), 0, 0, out, t, iters, mfma);
                hipEventRecord(a, 0);
                hipLaunchKernelGGL(clk, dim3(blocks), dim3(256), 0, 0, out, t, iters, mfma);
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
                printf("mfma=%d blocks=%4d iters=%7d: core cycles %9lld  wall(100MHz) %8lld -> %.2f GHz ; %.1f cyc/iter ; event %.1f us\n", mfma, blocks, iters, h[0], h[1],
                       h[0] / (h[1] * 10.0) , (double)h[0] / iters, ms * 1e3);
            }
    return 0;
}
