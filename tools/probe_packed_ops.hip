// Lane semantics of the "packed" helpers of admm_tvr_kernel.inc (quad permutes, bank broadcast): prints what each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ double rot_r(double v) { int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x93, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x93, 0xF, 0xF, true); return __hiloint2double(hi, lo); }
__device__ double rot_l(double v) { int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x39, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x39, 0xF, 0xF, true); return __hiloint2double(hi, lo); }
template <int j> __device__ double col_rep(double v) { int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), j * 0x55, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), j * 0x55, 0xF, 0xF, true); return __hiloint2double(hi, lo); }
template <int J> __device__ double bank_bcast(double v) { constexpr int pat = ((J << 2) << 5) | 0x13; int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pat), hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pat); return __hiloint2double(hi, lo); }
__global__ void k(double *o) {
    const int l = threadIdx.x;
    const double v = (double)l;
    o[0 * 64 + l] = rot_r(v); o[1 * 64 + l] = rot_l(v); o[2 * 64 + l] = col_rep<2>(v); o[3 * 64 + l] = bank_bcast<1>(v); o[4 * 64 + l] = bank_bcast<3>(v);
}
int main() {
    double *d; hipMalloc(&d, 5 * 64 * 8); k<<<1, 64>>>(d); double h[5 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *nm[5] = {"rot_r", "rot_l", "col_rep<2>", "bank_bcast<1>", "bank_bcast<3>"};
    for (int r = 0; r < 5; r++) { printf("%-14s", nm[r]); for (int l = 0; l < 24; l++) printf(" %2d", (int)h[r * 64 + l]); printf(" ... lane 40: %d\n", (int)h[r * 64 + 40]); }
    return 0;
}
