// streamprobe: do kernels on two HIP streams overlap on this device? (tuning aid)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(double* out, int iters) {
    double a = threadIdx.x;
    for (int i = 0; i < iters; i++) a = a * 1.0000001 + 0.5;
    if (a == 12345.0) out[0] = a;
}
int main() {
    double* d; hipMalloc(&d, 64);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const int iters = 400000;
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 1000); hipDeviceSynchronize();
    double t0 = now();
    hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s1, d, iters); hipDeviceSynchronize();
    double t1 = now();
    hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s1, d, iters);
    hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s2, d, iters);
    hipDeviceSynchronize();
    double t2 = now();
    // many small launches on s2 against one long kernel on s1
    hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s1, d, iters);
    for (int k = 0; k < 300; k++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s2, d, iters / 400);
    hipDeviceSynchronize();
    double t3 = now();
    for (int k = 0; k < 300; k++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s2, d, iters / 400);
    hipDeviceSynchronize();
    double t4 = now();
    printf("one kernel %.2f ms | two streams %.2f ms | long + 300 small on other stream %.2f ms | 300 small alone %.2f ms\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    return 0;
}
