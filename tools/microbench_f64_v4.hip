// FP64 matrix-pipe calibration v4: v_mfma_f64_4x4x4 with the A operand freshly read from LDS, as the kernels of this library
// issue it (s_memtime cycles per MFMA, one wavefront per SIMD unless stated).
//   0: operands in registers, 38 independent accumulators (the ceiling)
//   1: per k-slab 19 ds_read_b64 (next slab's A blocks) + 38 MFMAs (2 instance groups)     - hmpc_fused_kernel.inc, NG = 2
//   2: per k-slab 19 ds_read_b64 + 19 MFMAs (1 instance group)                             - NG = 1
//   3: as 1 with 8 wavefronts per workgroup (two per SIMD)
//   4: per pair of blocks one ds_read_b128, a ring of 4 pairs, 12 accumulators             - admm_mfma4.hpp / fista_r_kernel.inc
//   5: as 1, and the MFMA results in VGPRs moved through v_accvgpr (no: compiled with -amdgpu-mfma-vgpr-form) - flag run
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mb4 tools/microbench_f64_v4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define MF4(c, a, b) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)
#define SB __builtin_amdgcn_sched_barrier(0)
constexpr int NR = 19, SLABS = 16;

template <int MODE, int NT>
__global__ __launch_bounds__(NT, NT / 256) void k(double *out, unsigned long long *stamps, int iters, const double *in) {
    __shared__ double lds[SLABS * NR * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < SLABS * NR * 64; i += NT) lds[i] = in[i % 2048];
    __syncthreads();
    double acc[2][NR];
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int R = 0; R < NR; R++) acc[g][R] = 0.0;
    double b0 = in[lane], b1 = in[lane + 64];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int lo = lane * 8;
    for (int it = 0; it < iters; it++) {
        asm volatile("" : "+v"(lo));
        const char *ab = reinterpret_cast<const char *>(lds) + lo;
        if constexpr (MODE == 0) {
            double a[NR];
#pragma unroll
            for (int R = 0; R < NR; R++) a[R] = b0 + R;
#pragma unroll
            for (int s = 0; s < SLABS; s++) {
#pragma unroll
                for (int R = 0; R < NR; R++) { MF4(acc[0][R], a[R], b0); MF4(acc[1][R], a[R], b1); }
            }
        } else if constexpr (MODE == 1 || MODE == 2 || MODE == 3) {
            double an[NR];
#pragma unroll
            for (int R = 0; R < NR; R++) an[R] = *reinterpret_cast<const double *>(ab + R * 512);
#pragma unroll
            for (int s = 0; s < SLABS; s++) {
                double ac[NR];
#pragma unroll
                for (int R = 0; R < NR; R++) ac[R] = an[R];
                SB;
                if (s + 1 < SLABS) {
#pragma unroll
                    for (int R = 0; R < NR; R++) an[R] = *reinterpret_cast<const double *>(ab + ((s + 1) * NR + R) * 512);
                }
#pragma unroll
                for (int R = 0; R < NR; R++) {
                    MF4(acc[0][R], ac[R], b0);
                    if constexpr (MODE != 2) MF4(acc[1][R], ac[R], b1);
                }
                SB;
            }
        } else if constexpr (MODE == 4) {
            // 12 accumulators, blocks in pairs (ds_read_b128), ring 4 pairs ahead
            const char *pb = reinterpret_cast<const char *>(lds) + (lane & 15) * 16;
            constexpr int NP = SLABS * NR / 2;
            double2 r0 = *reinterpret_cast<const double2 *>(pb), r1 = *reinterpret_cast<const double2 *>(pb + 256),
                    r2 = *reinterpret_cast<const double2 *>(pb + 512), r3 = *reinterpret_cast<const double2 *>(pb + 768), cur;
#pragma unroll
            for (int q = 0; q < NP; q++) {
                const double2 nw = *reinterpret_cast<const double2 *>(pb + ((q + 4) % NP) * 256);
                cur = r0; r0 = r1; r1 = r2; r2 = r3; r3 = nw;
                MF4(acc[0][(2 * q) % 12], cur.x, b0);
                MF4(acc[0][(2 * q + 1) % 12], cur.y, b1);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double fs = 0;
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int R = 0; R < NR; R++) fs += acc[g][R];
    out[blockIdx.x * blockDim.x + threadIdx.x] = fs;
    if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

// Do FP64 vector instructions of ONE wavefront overlap with the FP64 MFMAs of ANOTHER wavefront on the same SIMD?
// 512 threads = two wavefronts per SIMD: wavefronts 0-3 (one per SIMD) run WHAT0, wavefronts 4-7 run WHAT1
// (0 = nothing, 1 = 4x4x4 MFMA chain x 38 accumulators, 2 = v_fma_f64 x 16 independent chains)
template <int WHAT0, int WHAT1>
__global__ __launch_bounds__(512, 2) void k2(double *out, unsigned long long *stamps, int iters, const double *in) {
    const int lane = threadIdx.x & 63, half = threadIdx.x >> 8;
    const int what = half ? WHAT1 : WHAT0;
    double acc[38], f[16];
#pragma unroll
    for (int R = 0; R < 38; R++) acc[R] = 0.0;
#pragma unroll
    for (int q = 0; q < 16; q++) f[q] = in[lane + q];
    const double b0 = in[lane], a0 = in[lane + 64];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (what == 1) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int R = 0; R < 38; R++) MF4(acc[R], a0, b0);
        }
    } else if (what == 2) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 76; r++)
#pragma unroll
                for (int q = 0; q < 16; q++) f[q] = __builtin_fma(f[q], a0, b0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double fs = 0;
#pragma unroll
    for (int R = 0; R < 38; R++) fs += acc[R];
#pragma unroll
    for (int q = 0; q < 16; q++) fs += f[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = fs;
    if ((threadIdx.x & 255) == 0) stamps[blockIdx.x * 2 + half] = t1 - t0;
}

template <int WHAT0, int WHAT1>
void run2(const char *name) {
    const int wg = 256, iters = 300;
    double *out, *in;
    unsigned long long *st;
    hipMalloc(&out, sizeof(double) * wg * 512);
    hipMalloc(&in, sizeof(double) * 2048);
    hipMalloc(&st, sizeof(unsigned long long) * wg * 2);
    std::vector<double> h(2048);
    for (int i = 0; i < 2048; i++) h[i] = (0.3 + 0.4 * ((i * 2654435761u) % 1000) / 1000.0) * ((i & 1) ? -1 : 1) * 1e-3;
    hipMemcpy(in, h.data(), 2048 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k2<WHAT0, WHAT1>), dim3(wg), dim3(512), 0, 0, out, st, 20, in);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k2<WHAT0, WHAT1>), dim3(wg), dim3(512), 0, 0, out, st, iters, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs(wg * 2);
    hipMemcpy(hs.data(), st, sizeof(unsigned long long) * wg * 2, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> c0, c1;
    for (int i = 0; i < wg; i++) { c0.push_back(hs[2 * i]); c1.push_back(hs[2 * i + 1]); }
    std::sort(c0.begin(), c0.end()); std::sort(c1.begin(), c1.end());
    printf("%-66s %8.3f ms | wave A: %7.1f cycles per 304 MFMA | wave B: %7.1f cycles per 1216 v_fma_f64\n", name, ms,
           (double)c0[wg / 2] / iters, (double)c1[wg / 2] / iters);
    hipFree(out); hipFree(in); hipFree(st);
}

template <int MODE, int NT>
void run(const char *name, int mfma_per_iter) {
    const int wg = 256, iters = 300;
    double *out, *in;
    unsigned long long *st;
    hipMalloc(&out, sizeof(double) * wg * NT);
    hipMalloc(&in, sizeof(double) * 2048);
    hipMalloc(&st, sizeof(unsigned long long) * wg);
    std::vector<double> h(2048);
    for (int i = 0; i < 2048; i++) h[i] = (0.3 + 0.4 * ((i * 2654435761u) % 1000) / 1000.0) * ((i & 1) ? -1 : 1) * 1e-3;
    hipMemcpy(in, h.data(), 2048 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<MODE, NT>), dim3(wg), dim3(NT), 0, 0, out, st, 20, in);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NT>), dim3(wg), dim3(NT), 0, 0, out, st, iters, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs(wg);
    hipMemcpy(hs.data(), st, sizeof(unsigned long long) * wg, hipMemcpyDeviceToHost);
    std::sort(hs.begin(), hs.end());
    const double cyc = (double)hs[wg / 2] / iters / mfma_per_iter;
    const double tf = (double)wg * NT / 64 * iters * mfma_per_iter * 512.0 / ms / 1e9;
    printf("%-66s %8.3f ms | cycles per MFMA (one wavefront) %6.1f | %6.2f TF/s\n", name, ms, cyc, tf);
    hipFree(out); hipFree(in); hipFree(st);
}

int main() {
    run<0, 256>("0 registers only, 38 accumulators", SLABS * NR * 2);
    run<1, 256>("1 19 ds_read_b64 + 38 MFMA per k-slab, 1 wave/SIMD", SLABS * NR * 2);
    run<2, 256>("2 19 ds_read_b64 + 19 MFMA per k-slab, 1 wave/SIMD", SLABS * NR);
    run<3, 512>("3 19 ds_read_b64 + 38 MFMA per k-slab, 2 waves/SIMD", SLABS * NR * 2);
    run<2, 512>("2' 19 ds_read_b64 + 19 MFMA per k-slab, 2 waves/SIMD", SLABS * NR);
    run<4, 256>("4 ds_read_b128 per block pair, ring of 4, 12 accumulators", SLABS * NR / 2 * 2);
    run2<1, 0>("two waves per SIMD: A = MFMA loop, B idle");
    run2<0, 2>("two waves per SIMD: A idle, B = v_fma_f64 loop");
    run2<1, 2>("two waves per SIMD: A = MFMA loop, B = v_fma_f64 loop");
    run2<1, 1>("two waves per SIMD: both MFMA loops");
    run2<2, 2>("two waves per SIMD: both v_fma_f64 loops");
    return 0;
}
