// Does a wavefront issue other instructions in the shadow of its own v_mfma_f64_4x4x4 (4 passes = 16 clocks)?
// One wavefront per SIMD; loops of ONE independent MFMA followed by K instructions of one kind, clocks per loop trip by s_memtime.
// build: hipcc --offload-arch=gfx950 -O2 -o microbench_issue microbench_issue.hip ; run: ./microbench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int KIND, int K>
__global__ __launch_bounds__(256) void bench(double *out, long long *cyc, int iters) {
    __shared__ double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    double f0 = a, f1 = b, f2 = a + b, f3 = a - b;
    int m0 = threadIdx.x, m1 = 1, m2 = 2, m3 = 3;
    const double *lp = lds + (threadIdx.x & 63);
    double l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            // four independent accumulators: no MFMA waits for another one's result
            if (r % 4 == 0) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
            if (r % 4 == 1) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
            if (r % 4 == 2) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
            if (r % 4 == 3) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (KIND == 0) {  // FP64 vector instruction, independent chains
                    if (k % 4 == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f0) : "v"(b), "v"(a));
                    if (k % 4 == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f1) : "v"(b), "v"(a));
                    if (k % 4 == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f2) : "v"(b), "v"(a));
                    if (k % 4 == 3) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f3) : "v"(b), "v"(a));
                } else if (KIND == 1) {  // 32-bit move (what v_accvgpr_read / write and DPP moves cost)
                    if (k % 4 == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(m0) : "v"(m1));
                    if (k % 4 == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(m1) : "v"(m2));
                    if (k % 4 == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(m2) : "v"(m3));
                    if (k % 4 == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(m3) : "v"(m0));
                } else if (KIND == 2) {  // s_nop 0
                    asm volatile("s_nop 0");
                } else if (KIND == 3) {  // LDS read (ds_read_b64), results never waited for inside the loop
                    if (k % 4 == 0) asm volatile("ds_read_b64 %0, %1" : "=v"(l0) : "v"((int)(size_t)lp));
                    if (k % 4 == 1) asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(l1) : "v"((int)(size_t)lp));
                    if (k % 4 == 2) asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(l2) : "v"((int)(size_t)lp));
                    if (k % 4 == 3) asm volatile("ds_read_b64 %0, %1 offset:1536" : "=v"(l3) : "v"((int)(size_t)lp));
                } else if (KIND == 4) {  // scalar ALU
                    asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
                } else if (KIND == 5) {  // s_waitcnt that never waits
                    asm volatile("s_waitcnt lgkmcnt(15)");
                } else if (KIND == 6) {  // v_accvgpr_read
                    asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(m0));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc0 + acc1 + acc2 + acc3 + f0 + f1 + f2 + f3 + m0 + m1 + m2 + m3 + l0 + l1 + l2 + l3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int K>
void run(const char *name, double *out, long long *cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((bench<KIND, K>), dim3(256), dim3(256), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((bench<KIND, K>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-10s K=%d  %.2f ns per MFMA+K  (s_memtime ticks per trip %.2f)\n", name, K, ms * 1e6 / ((double)iters * REP), (double)c / ((double)iters * REP));
}

int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * sizeof(double));
    hipMalloc(&cyc, sizeof(long long));
#define ROW(KIND, NAME) run<KIND, 0>(NAME, out, cyc); run<KIND, 1>(NAME, out, cyc); run<KIND, 2>(NAME, out, cyc); run<KIND, 3>(NAME, out, cyc); run<KIND, 4>(NAME, out, cyc); run<KIND, 6>(NAME, out, cyc); run<KIND, 8>(NAME, out, cyc);
    ROW(0, "v_fma_f64")
    ROW(1, "v_mov_b32")
    ROW(2, "s_nop")
    ROW(3, "ds_read")
    ROW(4, "s_add")
    ROW(5, "s_waitcnt")
    ROW(6, "accread")
    return 0;
}
