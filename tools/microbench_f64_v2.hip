// FP64 MFMA pipeline calibration, v2: real shader cycles (s_memtime) and the clock actually held
// (s_memtime / s_memrealtime @100 MHz).  hipcc --offload-arch=gfx950 -O3 -o /tmp/mb2 tools/microbench_f64_v2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

#define MF(c, a, b) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(double *out, unsigned long long *stamps, int iters, const double *in) {
    const int lane = threadIdx.x;
    double a0 = in[lane], a1 = in[lane + 64], a2 = in[lane + 128], a3 = in[lane + 192];
    double b0 = in[lane + 256], b1 = in[lane + 320], b2 = in[lane + 384], b3 = in[lane + 448];
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double f[8];
    for (int i = 0; i < 8; i++) f[i] = in[lane + i];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if constexpr (MODE == 0) {  // 4 independent accumulators, same A/B
            MF(c0, a0, b0); MF(c1, a0, b0); MF(c2, a0, b0); MF(c3, a0, b0);
        } else if constexpr (MODE == 1) {  // one accumulator chain, same A/B
            MF(c0, a0, b0); MF(c0, a0, b0); MF(c0, a0, b0); MF(c0, a0, b0);
        } else if constexpr (MODE == 2) {  // one accumulator chain, different A/B registers each
            MF(c0, a0, b0); MF(c0, a1, b1); MF(c0, a2, b2); MF(c0, a3, b3);
        } else if constexpr (MODE == 3) {  // 4 independent accumulators, different A/B
            MF(c0, a0, b0); MF(c1, a1, b1); MF(c2, a2, b2); MF(c3, a3, b3);
        } else if constexpr (MODE == 4) {  // 2 accumulators alternating
            MF(c0, a0, b0); MF(c1, a1, b1); MF(c0, a2, b2); MF(c1, a3, b3);
        } else if constexpr (MODE == 5) {  // chain of 4 on c0 then chain of 4 on c1 (8 per iter)
            MF(c0, a0, b0); MF(c0, a1, b1); MF(c0, a2, b2); MF(c0, a3, b3);
            MF(c1, a0, b0); MF(c1, a1, b1); MF(c1, a2, b2); MF(c1, a3, b3);
        } else if constexpr (MODE == 6) {  // mat-vec chain: D feeds next B (fully dependent), acc chained too
            MF(c0, a0, c0[0]); MF(c0, a1, c0[1]); MF(c0, a2, c0[2]); MF(c0, a3, c0[3]);
        } else if constexpr (MODE == 7) {  // VALU only 32 fma
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int q = 0; q < 8; q++) f[q] = __builtin_fma(f[q], a0, b0);
        } else if constexpr (MODE == 8) {  // chain + 4 valu fma per mfma
            MF(c0, a0, b0);
#pragma unroll
            for (int q = 0; q < 4; q++) f[q] = __builtin_fma(f[q], a0, b0);
            MF(c0, a1, b1);
#pragma unroll
            for (int q = 4; q < 8; q++) f[q] = __builtin_fma(f[q], a0, b0);
            MF(c0, a2, b2);
#pragma unroll
            for (int q = 0; q < 4; q++) f[q] = __builtin_fma(f[q], a0, b0);
            MF(c0, a3, b3);
#pragma unroll
            for (int q = 4; q < 8; q++) f[q] = __builtin_fma(f[q], a0, b0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    d4 s = c0 + c1 + c2 + c3;
    double fs = 0;
    for (int q = 0; q < 8; q++) fs += f[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + fs;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int MODE>
void run(const char *name, double mfma_per_iter, double fma_per_iter, bool zeros) {
    const int wg = 256, threads = 256, iters = 20000;
    double *out, *in;
    unsigned long long *st;
    hipMalloc(&out, sizeof(double) * wg * threads);
    hipMalloc(&in, sizeof(double) * 1024);
    hipMalloc(&st, sizeof(unsigned long long) * 2 * wg);
    std::vector<double> h(1024);
    for (int i = 0; i < 1024; i++) h[i] = zeros ? 0.0 : (0.3 + 0.4 * ((i * 2654435761u) % 1000) / 1000.0) * ((i & 1) ? -1 : 1) * 1e-3;
    hipMemcpy(in, h.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(threads), 0, 0, out, st, 2000, in);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(threads), 0, 0, out, st, iters, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs(2 * wg);
    hipMemcpy(hs.data(), st, sizeof(unsigned long long) * 2 * wg, hipMemcpyDeviceToHost);
    std::vector<double> cyc, mhz;
    for (int i = 0; i < wg; i++) { cyc.push_back((double)hs[2 * i] / iters); mhz.push_back((double)hs[2 * i] / (double)hs[2 * i + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
    double waves = (double)wg * threads / 64;
    double flop = waves * iters * (mfma_per_iter * 2048.0 + fma_per_iter * 128.0);
    printf("%-44s %s %7.3f ms %6.2f TF/s | cycles/iter %7.1f (per mfma %6.1f) | clock %5.0f MHz\n", name, zeros ? "zeros" : "rand ", ms,
           flop / ms / 1e9, cyc[wg / 2], mfma_per_iter > 0 ? cyc[wg / 2] / mfma_per_iter : 0.0, mhz[wg / 2]);
    hipFree(out); hipFree(in); hipFree(st);
}

int main() {
    for (int z = 0; z < 2; z++) {
        run<0>("4 indep acc, same A/B", 4, 0, z);
        run<1>("1 acc chain, same A/B", 4, 0, z);
        run<2>("1 acc chain, different A/B", 4, 0, z);
        run<3>("4 indep acc, different A/B", 4, 0, z);
        run<4>("2 acc alternating", 4, 0, z);
        run<5>("chain4 on c0 then chain4 on c1", 8, 0, z);
        run<6>("mat-vec chain D->B + acc", 4, 0, z);
        run<7>("VALU 32 v_fma_f64", 0, 32, z);
        run<8>("chain + 4 v_fma_f64 per mfma", 4, 16, z);
    }
    return 0;
}
