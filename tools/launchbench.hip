// Host cost of a kernel launch (tools only): hipLaunchKernelGGL in a loop, small and large kernarg structs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { void* p[12]; int i[28]; };
__global__ void k_small(int* x) { if (x && threadIdx.x == 1000) x[0] = 1; }
__global__ void k_big(Big b) { if (b.p[0] && threadIdx.x == 1000) ((int*)b.p[0])[0] = b.i[3]; }
int main() {
    hipStream_t s; hipStreamCreate(&s);
    Big b = {};
    for (int rep = 0; rep < 2; ++rep)
        for (int which = 0; which < 2; ++which) {
            const int N = 20000;
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (which == 0) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, (int*)nullptr);
                else hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, b);
            }
            auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(s);
            auto t2 = std::chrono::steady_clock::now();
            printf("%s: host issue %.2f us/launch, with drain %.2f us/launch\n", which ? "big kernarg" : "small kernarg",
                   std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
        }
    return 0;
}
