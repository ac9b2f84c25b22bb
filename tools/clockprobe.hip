// clockprobe: effective shader clock and dependent-op latency under low occupancy (tuning aid)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long* out, int iters, double seed) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double a = seed + threadIdx.x;
    for (int i = 0; i < iters; i++) { a = a + 1.0; a = a > 3.0 ? a : a + 0.5; }   // dependent f64 add + cmp/select
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = r1 - r0; out[blockIdx.x * 4 + 2] = (unsigned long long)a; }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 4096 * 32);
    unsigned long long h[4096 * 4];
    for (int blocks : {1, 16, 256, 2048}) for (int threads : {64, 640}) {
        const int iters = 2000000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d, 1000, 0.0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, blocks * 32, hipMemcpyDeviceToHost);
        double clk = (double)h[0] / (double)h[1] * 100.0;   // MHz (s_memrealtime ticks at 100 MHz)
        printf("blocks %5d threads %4d: %.2f ms  shader clock %.0f MHz  cycles/iter %.2f  ns/iter %.2f\n", blocks, threads, ms, clk,
               (double)h[0] / iters, ms * 1e6 / iters);
    }
    return 0;
}
