// FP64 pipeline calibration v3 (long unrolled bodies, s_memtime cycles):
//  - v_mfma_f64_16x16x4 vs v_mfma_f64_4x4x4 (4 blocks) issue cost
//  - does VALU work hide behind an executing FP64 MFMA?  (f64 fma / f32 fma / int / accvgpr moves)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mb3 tools/microbench_f64_v3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF16(c, a, b) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
#define MF4(c, a, b) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)
#define SB __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(double *out, unsigned long long *stamps, int iters, const double *in) {
    const int lane = threadIdx.x;
    double a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = in[lane + 64 * i]; b[i] = in[lane + 7 + 64 * i]; }
    d4 c0 = {0, 0, 0, 0}, c1 = c0;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    double f[8];
    float g[8];
    int ii[8];
    for (int i = 0; i < 8; i++) { f[i] = in[lane + i]; g[i] = (float)in[lane + i]; ii[i] = lane + i; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if constexpr (MODE == 0) { MF16(c0, a[u], b[u]); }                       // 16x16x4 one chain
            else if constexpr (MODE == 1) { MF16(c0, a[u], b[u]); MF16(c1, a[u], b[7 - u]); }  // two chains alternating (count 2)
            else if constexpr (MODE == 2) { MF4(s0, a[u], b[u]); }                    // 4x4x4 one chain
            else if constexpr (MODE == 3) { MF4(s0, a[u], b[u]); MF4(s1, a[u], b[7 - u]); MF4(s2, a[7 - u], b[u]); MF4(s3, a[7 - u], b[7 - u]); }  // 4 chains
            else if constexpr (MODE == 4) { MF16(c0, a[u], b[u]); SB;                 // + 4 f64 fma per mfma
#pragma unroll
                for (int q = 0; q < 4; q++) f[q] = __builtin_fma(f[q], a[0], b[0]); SB; }
            else if constexpr (MODE == 5) { MF16(c0, a[u], b[u]); SB;                 // + 8 f64 fma per mfma
#pragma unroll
                for (int q = 0; q < 8; q++) f[q] = __builtin_fma(f[q], a[0], b[0]); SB; }
            else if constexpr (MODE == 6) { MF16(c0, a[u], b[u]); SB;                 // + 8 f32 fma per mfma
#pragma unroll
                for (int q = 0; q < 8; q++) g[q] = __builtin_fmaf(g[q], 1.0001f, 0.5f); SB; }
            else if constexpr (MODE == 7) { MF16(c0, a[u], b[u]); SB;                 // + 8 int ops per mfma
#pragma unroll
                for (int q = 0; q < 8; q++) ii[q] = ii[q] * 3 + 1; SB; }
            else if constexpr (MODE == 8) { MF16(c0, a[u], b[u]); SB;                 // + 16 f64 fma per mfma
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int q = 0; q < 8; q++) f[q] = __builtin_fma(f[q], a[0], b[0]); SB; }
            else if constexpr (MODE == 9) { MF16(c0, a[u], b[u]); SB;                 // + 8 f64 max/min per mfma
#pragma unroll
                for (int q = 0; q < 8; q++) f[q] = fmin(fmax(f[q], a[0]), b[0]) ; SB; }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    d4 s = c0 + c1;
    double fs = s0 + s1 + s2 + s3;
    for (int q = 0; q < 8; q++) fs += f[q] + g[q] + ii[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + fs;
    if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int mfma_per_u, double flop_per_mfma) {
    const int wg = 256, threads = 256, iters = 4000;
    double *out, *in;
    unsigned long long *st;
    hipMalloc(&out, sizeof(double) * wg * threads);
    hipMalloc(&in, sizeof(double) * 2048);
    hipMalloc(&st, sizeof(unsigned long long) * wg);
    std::vector<double> h(2048);
    for (int i = 0; i < 2048; i++) h[i] = (0.3 + 0.4 * ((i * 2654435761u) % 1000) / 1000.0) * ((i & 1) ? -1 : 1) * 1e-3;
    hipMemcpy(in, h.data(), 2048 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(threads), 0, 0, out, st, 200, in);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(threads), 0, 0, out, st, iters, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs(wg);
    hipMemcpy(hs.data(), st, sizeof(unsigned long long) * wg, hipMemcpyDeviceToHost);
    std::sort(hs.begin(), hs.end());
    double cyc_per_u = (double)hs[wg / 2] / iters / 8.0;
    double tf = (double)wg * threads / 64 * iters * 8.0 * mfma_per_u * flop_per_mfma / ms / 1e9;
    printf("%-46s %7.3f ms | cycles per unrolled step %7.1f | per mfma %6.1f | %6.2f TF/s (mfma flops)\n", name, ms, cyc_per_u,
           cyc_per_u / mfma_per_u, tf);
    hipFree(out); hipFree(in); hipFree(st);
}

int main() {
    run<0>("16x16x4  one acc chain", 1, 2048);
    run<1>("16x16x4  two chains alternating", 2, 2048);
    run<2>("4x4x4_4b one acc chain", 1, 512);
    run<3>("4x4x4_4b four chains", 4, 512);
    run<4>("16x16x4 + 4 v_fma_f64", 1, 2048);
    run<5>("16x16x4 + 8 v_fma_f64", 1, 2048);
    run<8>("16x16x4 + 16 v_fma_f64", 1, 2048);
    run<9>("16x16x4 + 8 (v_max_f64+v_min_f64)", 1, 2048);
    run<6>("16x16x4 + 8 v_fma_f32", 1, 2048);
    run<7>("16x16x4 + 8 int mul-add", 1, 2048);
    return 0;
}
