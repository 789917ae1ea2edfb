// How fast can every CU stream the SAME table out of L2 (the FUSED kernels' situation)?  Three ways, 256 workgroups x 4 wavefronts:
//   dma      : buffer_load_dwordx4 ... lds (LDS-DMA), each wavefront a quarter of a 36 KB chunk, barrier per chunk
//   vgpr     : global_load_dwordx4 into registers, each wavefront a quarter of the chunk (then it would ds_write it)
//   vgpr_all : every wavefront loads the whole chunk into registers (no LDS at all: 4x the requests, the L1 absorbs what it can)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/l2s tools/microbench_l2stream.hip && /tmp/l2s
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int CHB = 36864, NCH = 18, REPS = 400;  // 18 chunks = 663 KB, as ME of the C5 split solver
typedef double d2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 1) void k_dma(const double *tab, double *out) {
    __shared__ __attribute__((aligned(1024))) double s0[CHB / 8];
    __shared__ __attribute__((aligned(1024))) double s1[CHB / 8];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(tab), 0, NCH * CHB, 0x00020000);
    double acc = 0.0;
    for (int r = 0; r < REPS; r++)
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            char *lb = reinterpret_cast<char *>((c & 1) ? s1 : s0) + wave * 1024;
#pragma unroll
            for (int t = 0; t < CHB / 1024; t += 4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lb + t * 1024), 16, lane * 16,
                                                         c * CHB + (t + wave) * 1024, 0, 0);
            __syncthreads();
            acc += ((c & 1) ? s1 : s0)[threadIdx.x];
        }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <bool ALL>
__global__ __launch_bounds__(256, 1) void k_vgpr(const double *tab, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const d2_t *t2 = reinterpret_cast<const d2_t *>(tab);
    d2_t acc = {0.0, 0.0};
    for (int r = 0; r < REPS; r++) {
        const d2_t *base = t2;
        asm volatile("" : "+v"(base));
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            d2_t v[ALL ? 36 : 9];
#pragma unroll
            for (int t = 0; t < (ALL ? 36 : 9); t++) v[t] = base[(c * CHB + ((ALL ? t : 4 * t + wave)) * 1024) / 16 + lane];
#pragma unroll
            for (int t = 0; t < (ALL ? 36 : 9); t++) acc += v[t];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
}

int main() {
    const size_t n = (size_t)NCH * CHB / 8;
    std::vector<double> h(n, 1.0);
    double *tab, *out;
    hipMalloc(&tab, n * 8);
    hipMalloc(&out, 256 * 256 * 8);
    hipMemcpy(tab, h.data(), n * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch, double bytes_per_wg) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-9s %8.3f ms  %6.2f TB/s requested (256 workgroups x %.0f KB x %d passes)\n", name, ms, 256.0 * bytes_per_wg * REPS / ms / 1e9,
               bytes_per_wg / 1024, REPS);
    };
    run("dma", [&] { hipLaunchKernelGGL(k_dma, dim3(256), dim3(256), 0, 0, tab, out); }, (double)NCH * CHB);
    run("vgpr", [&] { hipLaunchKernelGGL(k_vgpr<false>, dim3(256), dim3(256), 0, 0, tab, out); }, (double)NCH * CHB);
    run("vgpr_all", [&] { hipLaunchKernelGGL(k_vgpr<true>, dim3(256), dim3(256), 0, 0, tab, out); }, 4.0 * NCH * CHB);
    return 0;
}
