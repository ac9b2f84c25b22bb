// issueprobe: VALU issue rate of one wave / several waves per SIMD for FP64 and 32-bit ops at a given ILP (tuning aid)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP, int KIND>
__global__ void probe(unsigned long long* out, int iters, double seed) {
    double a[ILP];
    int q[ILP];
    for (int k = 0; k < ILP; k++) { a[k] = seed + threadIdx.x + k; q[k] = threadIdx.x + k; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int k = 0; k < ILP; k++) {
                if (KIND == 0) asm volatile("v_add_f64 %0, %0, 1.0" : "+v"(a[k]));
                if (KIND == 1) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[k]) : "v"(seed));
                if (KIND == 2) asm volatile("v_add_u32 %0, %0, 3" : "+v"(q[k]));
                if (KIND == 3) { asm volatile("v_add_f64 %0, %0, 1.0" : "+v"(a[k])); asm volatile("v_add_u32 %0, %0, 3" : "+v"(q[k])); }
                if (KIND == 4) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(q[k]));
                if (KIND == 5) q[k] = __builtin_amdgcn_update_dpp(q[k], q[k], 0x138, 0xf, 0xf, false) + 1;   // wave_shr:1
                if (KIND == 6) q[k] = __builtin_amdgcn_update_dpp(q[k], q[k], 0x111, 0xf, 0xf, false) + 1;   // row_shr:1
                if (KIND == 7) q[k] = __builtin_amdgcn_update_dpp(q[k], q[k], 0x142, 0xa, 0xf, false) + 1;   // row_bcast:15
                if (KIND == 8) q[k] = __builtin_amdgcn_readlane(q[k], 5) + q[k];
                if (KIND == 9) q[k] = max(max(q[k], 3), q[(k + 1) % ILP]);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0; int z = 0;
    for (int k = 0; k < ILP; k++) { s += a[k]; z += q[k]; }
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)s + z; out[2] = r1 - r0; }
}
template <int ILP, int KIND>
void run(unsigned long long* d, int threads, const char* name) {
    const int iters = 200000;
    unsigned long long h[3];
    hipLaunchKernelGGL((probe<ILP, KIND>), dim3(1), dim3(threads), 0, 0, d, 100, 0.0);
    hipLaunchKernelGGL((probe<ILP, KIND>), dim3(1), dim3(threads), 0, 0, d, iters, 0.0);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    const int per = ((KIND == 3 || (KIND >= 5 && KIND <= 8)) ? 2 : 1) * ILP * 8;
    printf("%-14s ILP %d threads %4d: %.2f memtime ticks, %.3f ns per instruction per wave (realtime)\n", name, ILP, threads, (double)h[0] / iters / per, (double)h[2] * 10.0 / iters / per);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    for (int threads : {64, 640, 1024}) {
        run<1, 0>(d, threads, "v_add_f64"); run<2, 0>(d, threads, "v_add_f64"); run<4, 0>(d, threads, "v_add_f64"); run<8, 0>(d, threads, "v_add_f64");
        run<1, 1>(d, threads, "v_max_f64"); run<4, 1>(d, threads, "v_max_f64");
        run<1, 2>(d, threads, "v_add_u32"); run<4, 2>(d, threads, "v_add_u32"); run<8, 2>(d, threads, "v_add_u32");
        run<4, 3>(d, threads, "f64+u32 mix"); run<8, 3>(d, threads, "f64+u32 mix");
        run<4, 4>(d, threads, "v_cndmask");
        run<1, 5>(d, threads, "wave_shr1+add"); run<4, 5>(d, threads, "wave_shr1+add");
        run<1, 6>(d, threads, "row_shr1+add"); run<4, 6>(d, threads, "row_shr1+add");
        run<1, 7>(d, threads, "row_bcast15+add"); run<4, 7>(d, threads, "row_bcast15+add");
        run<4, 8>(d, threads, "readlane+add");
        run<4, 9>(d, threads, "max3");
    }
    return 0;
}
