// Microbenchmarks that calibrate the FP64 peaks this design is priced against (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/microbench_f64.hip && /tmp/mb
// 1. v_mfma_f64_16x16x4_f64 issue rate (independent accumulators) and dependent (D -> B operand) latency
// 2. v_fma_f64 VALU rate, and VALU + MFMA interleaved (do they overlap?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(double *out, int iters, double a0, double b0) {
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double f0 = a, f1 = b, f2 = a + b, f3 = a - b, f4 = 1.0, f5 = 2.0, f6 = 3.0, f7 = 4.0;
    for (int i = 0; i < iters; i++) {
        if constexpr (MODE == 0) {  // 4 independent accumulators
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        } else if constexpr (MODE == 1) {  // one accumulator chain (C dependency)
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        } else if constexpr (MODE == 2) {  // D -> B operand dependency (mat-vec chain)
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c3[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c0[1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c1[2], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c2[3], c3, 0, 0, 0);
        } else if constexpr (MODE == 3) {  // VALU only: 32 independent-ish v_fma_f64
#pragma unroll
            for (int r = 0; r < 4; r++) {
                f0 = __builtin_fma(f0, a, b); f1 = __builtin_fma(f1, a, b); f2 = __builtin_fma(f2, a, b); f3 = __builtin_fma(f3, a, b);
                f4 = __builtin_fma(f4, a, b); f5 = __builtin_fma(f5, a, b); f6 = __builtin_fma(f6, a, b); f7 = __builtin_fma(f7, a, b);
            }
        } else if constexpr (MODE == 4) {  // 4 MFMA + 32 VALU fma interleaved
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            f0 = __builtin_fma(f0, a, b); f1 = __builtin_fma(f1, a, b); f2 = __builtin_fma(f2, a, b); f3 = __builtin_fma(f3, a, b);
            f4 = __builtin_fma(f4, a, b); f5 = __builtin_fma(f5, a, b); f6 = __builtin_fma(f6, a, b); f7 = __builtin_fma(f7, a, b);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            f0 = __builtin_fma(f0, a, b); f1 = __builtin_fma(f1, a, b); f2 = __builtin_fma(f2, a, b); f3 = __builtin_fma(f3, a, b);
            f4 = __builtin_fma(f4, a, b); f5 = __builtin_fma(f5, a, b); f6 = __builtin_fma(f6, a, b); f7 = __builtin_fma(f7, a, b);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            f0 = __builtin_fma(f0, a, b); f1 = __builtin_fma(f1, a, b); f2 = __builtin_fma(f2, a, b); f3 = __builtin_fma(f3, a, b);
            f4 = __builtin_fma(f4, a, b); f5 = __builtin_fma(f5, a, b); f6 = __builtin_fma(f6, a, b); f7 = __builtin_fma(f7, a, b);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            f0 = __builtin_fma(f0, a, b); f1 = __builtin_fma(f1, a, b); f2 = __builtin_fma(f2, a, b); f3 = __builtin_fma(f3, a, b);
            f4 = __builtin_fma(f4, a, b); f5 = __builtin_fma(f5, a, b); f6 = __builtin_fma(f6, a, b); f7 = __builtin_fma(f7, a, b);
        } else if constexpr (MODE == 5) {  // 4 MFMA + 8 VALU fma
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            f0 = __builtin_fma(f0, a, b); f1 = __builtin_fma(f1, a, b);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            f2 = __builtin_fma(f2, a, b); f3 = __builtin_fma(f3, a, b);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            f4 = __builtin_fma(f4, a, b); f5 = __builtin_fma(f5, a, b);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            f6 = __builtin_fma(f6, a, b); f7 = __builtin_fma(f7, a, b);
        }
    }
    d4 s = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

template <int MODE>
void run(const char *name, int wg, int threads, double mfma_per_iter, double fma_per_iter) {
    double *out;
    hipMalloc(&out, sizeof(double) * wg * threads);
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(threads), 0, 0, out, 100, 1.0, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(threads), 0, 0, out, iters, 1.0, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)wg * threads / 64;
    double flop = waves * iters * (mfma_per_iter * 2048.0 + fma_per_iter * 128.0);
    double wave_ns_per_iter = ms * 1e6 / iters;  // per wave (all waves run concurrently when wg<=256*k)
    printf("%-34s wg=%4d thr=%3d  %8.3f ms  %7.2f TFLOP/s  %7.1f ns/iter/wave (~%6.0f cyc @2.4GHz)\n", name, wg, threads, ms,
           flop / ms / 1e9, wave_ns_per_iter, wave_ns_per_iter * 2.4);
    hipFree(out);
}

int main() {
    run<0>("mfma_f64 4 indep acc", 256, 256, 4, 0);
    run<0>("mfma_f64 4 indep acc (2 wave/SIMD)", 256, 512, 4, 0);
    run<1>("mfma_f64 1 acc chain", 256, 256, 4, 0);
    run<2>("mfma_f64 D->B chain", 256, 256, 4, 0);
    run<3>("v_fma_f64 x32", 256, 256, 0, 32);
    run<3>("v_fma_f64 x32 (2 wave/SIMD)", 256, 512, 0, 32);
    run<4>("4 mfma + 32 v_fma_f64", 256, 256, 4, 32);
    run<5>("4 mfma + 8 v_fma_f64", 256, 256, 4, 8);
    return 0;
}
