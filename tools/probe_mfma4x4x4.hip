// Probe the lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 (A, B, D operands; cbsz/abid broadcast).
// Prints, for every D lane, which A lanes and B lanes feed it and how they pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
template <int CBSZ, int ABID>
__global__ void k(const double *a, const double *b, double *d) {
    int l = threadIdx.x;
    double c = 0.0;
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c, CBSZ, ABID, 0);
    d[l] = c;
}
template <int CBSZ, int ABID>
std::vector<double> run(const std::vector<double> &a, const std::vector<double> &b) {
    double *da, *db, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
    hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, da, db, dd);
    std::vector<double> d(64);
    hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dd);
    return d;
}
static std::vector<int> bits(double v) { std::vector<int> r; unsigned long long u = (unsigned long long)v; for (int i = 0; i < 64; i++) if (u >> i & 1) r.push_back(i); return r; }
template <int CBSZ, int ABID>
void probe() {
    printf("=== cbsz=%d abid=%d\n", CBSZ, ABID);
    std::vector<double> ones(64, 1.0), p2(64), pr(64), qr(64);
    for (int i = 0; i < 64; i++) { p2[i] = std::ldexp(1.0, i); pr[i] = 3 + 2 * i; qr[i] = 1000 + 7 * i * i + i; }
    auto dB = run<CBSZ, ABID>(ones, p2);  // which B lanes feed each D lane
    auto dA = run<CBSZ, ABID>(p2, ones);  // which A lanes
    auto dP = run<CBSZ, ABID>(pr, qr);
    for (int l = 0; l < 64; l++) {
        auto sb = bits(dB[l]), sa = bits(dA[l]);
        printf("D lane %2d: A lanes {", l);
        for (int x : sa) printf("%d ", x);
        printf("} B lanes {");
        for (int x : sb) printf("%d ", x);
        printf("} pairing:");
        // brute-force pairing
        int perm[4] = {0, 1, 2, 3};
        bool found = false;
        if (sa.size() == 4 && sb.size() == 4) {
            do {
                double s = 0;
                for (int i = 0; i < 4; i++) s += pr[sa[i]] * qr[sb[perm[i]]];
                if (s == dP[l]) { for (int i = 0; i < 4; i++) printf(" (%d,%d)", sa[i], sb[perm[i]]); found = true; break; }
            } while (std::next_permutation(perm, perm + 4));
        }
        if (!found) printf(" ?");
        printf("\n");
    }
}
#include <algorithm>
int main() {
    probe<0, 0>();
    probe<2, 0>();
    probe<2, 1>();
    probe<2, 3>();
    return 0;
}
