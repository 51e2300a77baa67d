// Does hipExtLaunchKernel(..., flags = hipExtAnyOrderLaunch) drop the barrier between two launches of ONE stream on gfx950?
// Two independent spin kernels, device clock stamps of start and end; prints the overlap.   hipcc --offload-arch=gfx950 anyorder.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void spin_kernel(unsigned long long* stamp, int us) {
    unsigned long long t0 = wall_clock64();           // 100 MHz
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < (unsigned long long)us * 100) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}
__global__ void tiny_kernel(unsigned long long* stamp) { if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = wall_clock64(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    unsigned long long* d; CK(hipMalloc(&d, 64 * 8)); CK(hipMemset(d, 0, 64 * 8));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long h[64];
    for (int flags = 0; flags < 2; ++flags) {
        for (int rep = 0; rep < 3; ++rep) {
            unsigned long long *a = d, *b = d + 2, *c = d + 4; int us = 50;
            void* aa[] = {&a, &us}; void* ab[] = {&b, &us}; void* ac[] = {&c};
            CK(hipExtLaunchKernel((const void*)spin_kernel, dim3(32), dim3(64), aa, 0, s, nullptr, nullptr, 0));
            CK(hipExtLaunchKernel((const void*)spin_kernel, dim3(32), dim3(64), ab, 0, s, nullptr, nullptr, flags));
            CK(hipExtLaunchKernel((const void*)tiny_kernel, dim3(1), dim3(64), ac, 0, s, nullptr, nullptr, 0));
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost));
            printf("flags %d rep %d: A %.2f..%.2f us, B %.2f..%.2f us, C at %.2f us (after B's end: %.2f)\n", flags, rep, 0.0, (h[1] - h[0]) / 100.0,
                   ((long long)h[2] - (long long)h[0]) / 100.0, ((long long)h[3] - (long long)h[0]) / 100.0, ((long long)h[4] - (long long)h[0]) / 100.0,
                   ((long long)h[4] - (long long)h[3]) / 100.0);
        }
    }
    // a chain of 200 tiny dependent launches against 200 any-order ones: the per-launch boundary
    for (int flags = 0; flags < 2; ++flags) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        unsigned long long* c = d + 8; void* ac[] = {&c};
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 200; ++i) CK(hipExtLaunchKernel((const void*)tiny_kernel, dim3(1), dim3(64), ac, 0, s, nullptr, nullptr, flags));
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("flags %d: 200 tiny launches %.1f us (%.2f us each)\n", flags, ms * 1000, ms * 5);
    }
    return 0;
}
