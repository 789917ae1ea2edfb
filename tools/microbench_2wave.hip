// Can a SECOND wavefront on a SIMD fill the issue slots the first one loses around its v_mfma_f64_4x4x4?
// (the headline kernel runs one wavefront per SIMD; its 786 MFMAs take 12.6 k of the iteration's clocks, the rest
// is ~930 vector instructions that never overlap the wavefront's OWN MFMAs - profiles/r03_microbench_issue.txt.)
// Each wavefront runs trips of one RUN of MR independent MFMAs followed by one RUN of VR v_fma_f64 + AR v_accvgpr_read + OR v_mov_b32 -
// the shape of the kernel's steady state (per run: 16.4 MFMAs, 11.4 FP64 vector instructions, 4.8 accumulator reads, 4.5 other).
// The same per-wavefront work is timed with ONE wavefront per SIMD (256-thread workgroups, one per CU) and with TWO (512-thread
// workgroups, one per CU: twice the work per SIMD).  `ratio` = t(two) / t(one): 2.0 = the SIMD serialises the two wavefronts completely
// (nothing to win from a second wavefront), 1.0 = the second wavefront is free.
// build: hipcc --offload-arch=gfx950 -O2 -o microbench_2wave microbench_2wave.hip ; run: ./microbench_2wave
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 8
template <int WPS, int MR, int VR, int AR, int OR, bool DEP>
__global__ __launch_bounds__(256 * WPS) void bench(double *out, long long *cyc, int iters) {
    extern __shared__ double lds[];  // (sized by the host so that ONE workgroup fits a CU)
    if (threadIdx.x == 0) lds[0] = 0.0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    double f0 = a, f1 = b, f2 = a + b, f3 = a - b;
    int m0 = threadIdx.x, m1 = 1, m2 = 2, m3 = 3, r0 = 0;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int k = 0; k < MR; k++) {
                if (DEP || k % 4 == 0) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
                else if (k % 4 == 1) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
                else if (k % 4 == 2) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int k = 0; k < VR; k++) {
                if (k % 4 == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f0) : "v"(b), "v"(a));
                if (k % 4 == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f1) : "v"(b), "v"(a));
                if (k % 4 == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f2) : "v"(b), "v"(a));
                if (k % 4 == 3) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f3) : "v"(b), "v"(a));
            }
#pragma unroll
            for (int k = 0; k < AR; k++) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(r0));
#pragma unroll
            for (int k = 0; k < OR; k++) {
                if (k % 4 == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(m0) : "v"(m1));
                if (k % 4 == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(m1) : "v"(m2));
                if (k % 4 == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(m2) : "v"(m3));
                if (k % 4 == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(m3) : "v"(m0));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = acc0 + acc1 + acc2 + acc3 + f0 + f1 + f2 + f3 + m0 + m1 + m2 + m3 + r0 + lds[0];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int WPS, int MR, int VR, int AR, int OR, bool DEP>
void one(double *out, long long *cyc, double *ns_per_run, double *ticks_per_run) {
    const int iters = 4000;
    const size_t lds = 100 * 1024;  // one workgroup per CU
    auto k = bench<WPS, MR, VR, AR, OR, DEP>;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(256), dim3(256 * WPS), lds, 0, out, cyc, 50);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    long long c = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(256 * WPS), lds, 0, out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost); }
    }
    *ns_per_run = best * 1e6 / ((double)iters * REP);
    *ticks_per_run = (double)c / ((double)iters * REP);
}

template <int MR, int VR, int AR, int OR, bool DEP>
void row(const char *name, double *out, long long *cyc) {
    double n1, t1, n2, t2;
    one<1, MR, VR, AR, OR, DEP>(out, cyc, &n1, &t1);
    one<2, MR, VR, AR, OR, DEP>(out, cyc, &n2, &t2);
    const double model = 16.0 * MR + (VR + AR + OR ? 8.0 : 0.0) + 4.0 * VR + 8.0 * AR + 4.0 * OR;  // profiles/r03_microbench_issue.txt's rule
    const double dp = 16.0 * MR + 4.0 * VR;                                                          // FP64 pipe time alone
    printf("%-44s MFMA %2d FP64 %2d accread %2d mov %2d | one wave/SIMD: %7.1f ns %7.1f ticks (rule %5.0f) | two: %7.1f ns %7.1f ticks | ratio %.3f | "
           "FP64-pipe floor for two %5.0f ticks: two waves run at %.2f of it\n",
           name, MR, VR, AR, OR, n1, t1, model, n2, t2, n2 / n1, 2 * dp, 2 * dp / t2);
}

int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 256 * 512 * sizeof(double));
    hipMalloc(&cyc, sizeof(long long));
    row<16, 0, 0, 0, false>("MFMA only (4 chains)", out, cyc);
    row<16, 0, 0, 0, true>("MFMA only, ONE dependent chain", out, cyc);
    row<16, 12, 0, 0, false>("MFMA run + FP64 vector run", out, cyc);
    row<16, 0, 0, 12, false>("MFMA run + v_mov_b32 run", out, cyc);
    row<16, 0, 12, 0, false>("MFMA run + v_accvgpr_read run", out, cyc);
    row<16, 11, 5, 5, false>("the headline kernel's mix per run", out, cyc);
    row<16, 8, 0, 0, false>("the mix an ideal rewrite would leave", out, cyc);
    row<4, 3, 1, 1, false>("the same mix in short runs (4 MFMAs)", out, cyc);
    row<1, 1, 0, 0, false>("alternating MFMA / FP64", out, cyc);
    row<0, 16, 0, 0, false>("FP64 vector only", out, cyc);
    return 0;
}
