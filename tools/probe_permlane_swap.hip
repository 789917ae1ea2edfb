#include <hip/hip_runtime.h>
__device__ inline double xor16_sum(double x) {
    unsigned lo = __double2loint(x), hi = __double2hiint(x);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    double p = __hiloint2double(b[0], a[0]), q = __hiloint2double(b[1], a[1]);
    return p + q;
}
__device__ inline double xor32_sum(double x) {
    unsigned lo = __double2loint(x), hi = __double2hiint(x);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    double p = __hiloint2double(b[0], a[0]), q = __hiloint2double(b[1], a[1]);
    return p + q;
}
__global__ void k(const double *in, double *o) {
    double x = in[threadIdx.x];
    double s = xor32_sum(xor16_sum(x));
    o[threadIdx.x] = s;
    double r = x + __shfl_xor(x, 16);
    r += __shfl_xor(r, 32);
    o[64 + threadIdx.x] = r;
}
int main() {
    double h[64], *d, *o, ho[128];
    for (int i = 0; i < 64; i++) h[i] = i * 1.25 + 0.5;
    hipMalloc(&d, 512); hipMalloc(&o, 1024);
    hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) if (ho[i] != ho[64 + i]) bad++;
    printf("bad=%d  %g %g %g %g\n", bad, ho[0], ho[64], ho[17], ho[64 + 17]);
    return bad;
}
