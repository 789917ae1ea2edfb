// Host side of the FUSED HMPC variant (hmpc_fused.hpp): table layout, constants, launch, run-time specialisation.
#include "hmpc_fused.hpp"

#include <cmath>

#include "hmpc_fused_kernel.inc"
#include "rtc_common.hpp"

namespace spcies {
namespace hfused {

static const char *const kFusedSource =
#include "hmpc_fused_src.inc"
    ;

// ---- shapes instantiated at build time: (n, m, N, SYM, USE_SOC, MODE).  Everything else is compiled by hiprtc at create time.
#define SPCIES_HFUSED_SHAPES(X) X(12, 2, 15, true, false, 0) X(12, 2, 15, true, false, 1)

namespace {

struct Dims {  // mirrors Shape<> of the kernel file for run-time values
    int nm, dim, n_y, n_soc, n_s, n_box, n_sb, NA, NB, NC, NR, NRP, NXS, NUS, NE, NK, JC, NCH, CHB, o0, ncol;
    Dims(int n, int m, int N, bool use_soc, int mode, int ny = 0) {
        nm = n + m;
        dim = (N - 1) * nm + m + 3 * nm;
        n_y = ny ? ny : nm;
        n_soc = use_soc ? 2 * n_y : n_y;
        n_box = (mode == 1 && ny) ? N * ny : dim - 3 * nm;
        n_sb = (mode == 0 && ny) ? N * ny : 0;
        n_s = mode == 0 ? n_sb + 3 * n_soc : n_box + 3 * n_soc;
        NA = ((mode == 0 ? dim : n_box) + 15) / 16;
        NB = (n_sb + 15) / 16;
        NC = 3 * ((n_soc + 15) / 16);
        NR = NA + NB + NC + ((mode == 1 && ny) ? 1 : 0);  // (+ the u row register of the coupled solver without the splitting)
        NXS = (n + 3) / 4;
        NUS = (m + 3) / 4;
        NE = mode == 0 ? NXS : 2 * NXS + NUS;
        NK = 4 * NR;
        ncol = 16 * NR + 4 * (NE + 1);  // columns of the extended matrix: state | inputs | 1
        int chunk = SPCIES_HFUSED_CHUNK;
        if (const char *ev = getenv("SPCIES_HFUSED_CHUNK")) chunk = atoi(ev);  // (with SPCIES_HFUSED_RTC=1 and the same -D in SPCIES_HFUSED_FLAGS)
        NRP = (NR + 1) / 2 * 2;
        JC = (chunk / (NRP * 512)) > 0 ? (chunk / (NRP * 512)) : 1;
        const int lds_tables = 2 * 16 * (NA + NB) * 8 + 2 * 16 * (NC / 3) * 8;  // (NC from the line above: cones only)
        while (JC > 1 && 3 * ((JC * NRP * 512 + 4095) / 4096 * 4096) + lds_tables > 160 * 1024) JC--;  // (as Shape::pick_jc)
        NCH = ((NK + JC - 1) / JC + 2) / 3 * 3;
        CHB = (JC * NRP * 512 + 4095) / 4096 * 4096;
        o0 = (N - 1) * nm + m;
    }
};

int builtin_index(int n, int m, int N, bool sym, bool use_soc, int mode) {
    int idx = 0;
#define X(nn, mm, NN, SS, UU, MM)                                                               \
    if (n == nn && m == mm && N == NN && sym == SS && use_soc == UU && mode == MM) return idx; \
    idx++;
    SPCIES_HFUSED_SHAPES(X)
#undef X
    return -1;
}

template <int n, int m, int N, bool SYM, bool USE_SOC, int MODE>
int launch_builtin(const Args &a, const double *ME, const double *PRO, const double *C, const double *x0, const double *xr, const double *ur, double *u,
                   int *k, int *e, double *const *f, bool want_sol, unsigned grid, hipStream_t st) {
    if (want_sol)
        hipLaunchKernelGGL((hmpc_fused_kernel<n, m, N, SYM, USE_SOC, MODE, true>), dim3(grid), dim3(kNWV * 64), 0, st, a, ME, PRO, C, x0, xr, ur, u, k, e,
                           f[0], f[1], f[2], f[3], f[4], f[5]);
    else
        hipLaunchKernelGGL((hmpc_fused_kernel<n, m, N, SYM, USE_SOC, MODE, false>), dim3(grid), dim3(kNWV * 64), 0, st, a, ME, PRO, C, x0, xr, ur, u, k, e,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

// Common tail of the two plan builders: Mx = the extended matrix [16 NR][D.ncol] in internal row / column order
// (columns: state | per-instance inputs in k-slabs of four | the constant)
int finish_plan(Plan &p, const Dims &D, int n, int m, int N, int use_soc, int symmetric, int mode, const std::vector<double> &Mx,
                std::vector<double> &flat, int ny = 0) {
    const int ncol = D.ncol, NP = 16 * D.NR;
    for (double x : Mx)
        if (!std::isfinite(x)) { p.why = "non-finite M1 / M2"; return 0; }
    // the table in issue order: chunk | k-slab in chunk | pair of row registers | lane (k = l >> 4, b = (l >> 2) & 3, i = l & 3) |
    // register of the pair
    std::vector<double> tab((size_t)D.NCH * (D.CHB / 8), 0.0);
    for (int c = 0; c < D.NCH; c++)
        for (int jj = 0; jj < D.JC; jj++) {
            const int J = c * D.JC + jj;
            if (J >= D.NK) continue;  // (zero blocks pad the last chunks)
            for (int R = 0; R < D.NR; R++)
                for (int l = 0; l < 64; l++) {
                    const int k = l >> 4, b = (l >> 2) & 3, i = l & 3;
                    tab[(size_t)c * (D.CHB / 8) + ((size_t)(jj * (D.NRP / 2) + R / 2) * 64 + l) * 2 + (R & 1)] =
                        Mx[(size_t)(16 * R + 4 * b + i) * ncol + 4 * J + k];
                }
        }
    // the prologue table: k-slabs of the inputs and the constant slab (column NP + 4 NE, in k = 0)
    std::vector<double> ptab((size_t)(D.NE + 1) * (D.NRP / 2) * 128, 0.0);
    for (int J = 0; J <= D.NE; J++)
        for (int R = 0; R < D.NR; R++)
            for (int l = 0; l < 64; l++) {
                const int k = l >> 4, b = (l >> 2) & 3, i = l & 3;
                if (J == D.NE && k > 0) continue;
                ptab[((size_t)(J * (D.NRP / 2) + R / 2) * 64 + l) * 2 + (R & 1)] = Mx[(size_t)(16 * R + 4 * b + i) * ncol + NP + 4 * J + k];
            }
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_ME, tab.size() * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_ME, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_PRO, ptab.size() * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_PRO, ptab.data(), ptab.size() * sizeof(double), hipMemcpyHostToDevice));
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_C, flat.size() * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_C, flat.data(), flat.size() * sizeof(double), hipMemcpyHostToDevice));
    p.n = n; p.m = m; p.N = N; p.use_soc = use_soc; p.symmetric = symmetric; p.mode = mode; p.ny = ny;
    p.NR = D.NR; p.NK = D.NK; p.NCH = D.NCH; p.CHB = D.CHB;
    hipDeviceProp_t prop;
    int dev = 0;
    SPCIES_HIP_CHECK(hipGetDevice(&dev));
    SPCIES_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    p.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    p.builtin = ny ? -1 : builtin_index(n, m, N, symmetric != 0, use_soc != 0, mode);
    const char *force = getenv("SPCIES_HFUSED_RTC");  // kernel experiments: re-specialise a built-in shape (with SPCIES_HFUSED_FLAGS)
    if (force && force[0] == '1') p.builtin = -1;
    if (p.builtin < 0) {  // not among the build-time shapes: specialise now (SPCIES_HIP_RTC=0 turns it off)
        const char *ev = getenv("SPCIES_HIP_RTC");
        if (ev && ev[0] == '0') { p.why = "shape not instantiated at build time and SPCIES_HIP_RTC=0"; return 0; }
        // (as the build-time instantiations; the unroll threshold: clang otherwise leaves the chunk loop of the larger shapes rolled and
        // indexes the state arrays at run time - scratch memory)
        std::vector<std::string> names, extra = {"-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-pragma-unroll-threshold=1000000"};
        for (int s = 0; s < 2; s++) {
            char nm[160];
            snprintf(nm, sizeof(nm), "spcies::hfused::hmpc_fused_kernel<%d, %d, %d, %s, %s, %d, %s, %d>", n, m, N, symmetric ? "true" : "false",
                     use_soc ? "true" : "false", mode, s ? "true" : "false", ny);
            names.push_back(nm);
        }
        if (const char *fl = getenv("SPCIES_HFUSED_FLAGS")) {
            std::string tok;
            for (const char *c = fl;; c++) {
                if (*c == ' ' || *c == '\0') {
                    if (!tok.empty()) extra.push_back(tok);
                    tok.clear();
                    if (!*c) break;
                } else {
                    tok.push_back(*c);
                }
            }
        }
        hipModule_t mod = nullptr;
        hipFunction_t fns[2] = {nullptr, nullptr};
        if (rtc::compile_module(kFusedSource, "spcies_hmpc_fused.hip", names, extra, &mod, fns) != 0) {
            p.why = g_last_error;
            p.build_failed = true;
            return 0;  // not an error: AUTO falls back, the reason is reported when FUSED is asked for
        }
        p.module = mod;
        p.fn[0] = fns[0];
        p.fn[1] = fns[1];
    }
    p.ok = true;
    p.why.clear();
    return 0;
}

}  // namespace

void plan_free(Plan &p) {
    if (p.d_ZR) hipFree(p.d_ZR);
    if (p.d_T) hipFree(p.d_T);
    p.d_ZR = p.d_T = nullptr;
    p.cap_T = 0;
    if (p.d_ME) hipFree(p.d_ME);
    if (p.d_PRO) hipFree(p.d_PRO);
    if (p.d_C) hipFree(p.d_C);
    if (p.module) rtc::unload_module((hipModule_t)p.module);
    p.d_ME = p.d_PRO = p.d_C = nullptr;
    p.module = nullptr;
    p.ok = false;
}

static int put(std::vector<double> &flat, const std::vector<double> &v) {
    const int off = (int)flat.size();
    flat.insert(flat.end(), v.begin(), v.end());
    while (flat.size() % 8) flat.push_back(0.0);
    return off;
}

int plan_build_split(Plan &p, const SplitHost &h) {
    const int n = h.n, m = h.m, N = h.N;
    const int ny = h.coupled ? h.n_y : 0;
    const Dims D(n, m, N, h.use_soc != 0, 0, ny);
    if (D.dim != h.dim || D.n_s != h.n_s || D.n_soc != h.n_soc) { p.why = "unexpected HMPC dimensions"; return 0; }
    if (D.NR > 24) { p.why = "FUSED: more than 24 row registers (dim + padded slacks and cones > 384 rows)"; return 0; }
    const int np = h.dim + h.n_s, nc = h.n_eq + h.n_s;
    // internal row -> row of (z, s), or -1 for a pad
    const int NP = 16 * D.NR;
    std::vector<int> orig(NP, -1);
    for (int r = 0; r < h.dim; r++) orig[r] = r;
    for (int j = 0; j < D.n_sb; j++) orig[16 * D.NA + j] = h.dim + j;  // coupled constraints: the box slacks of the outputs
    for (int t = 0; t < D.n_soc; t++)  // component i of cone t: register NA + NB + 3 (t / 16) + i, row t % 16
        for (int i = 0; i < 3; i++) orig[16 * (D.NA + D.NB + 3 * (t / 16) + i) + t % 16] = h.dim + D.n_sb + 3 * t + i;
    // extended matrix: hat = -M1 q_hat + M2 bh,  M2 bh = c_const + (-M2[:, :n] A) x0  (:97-104, :174-190)
    const int ncol = D.ncol;
    std::vector<double> Mx((size_t)NP * ncol, 0.0);
    for (int ri = 0; ri < NP; ri++) {
        const int ro = orig[ri];
        if (ro < 0) continue;
        double *row = &Mx[(size_t)ri * ncol];
        for (int ci = 0; ci < NP; ci++)
            if (orig[ci] >= 0) row[ci] = -h.M1[(size_t)ro * np + orig[ci]];
        for (int c = 0; c < n; c++) {
            double acc = 0.0;
            for (int j = 0; j < n; j++) acc -= h.M2[(size_t)ro * nc + j] * h.A[j * n + c];
            row[NP + c] = acc;
        }
        double cc = 0.0;
        for (int j = n; j < nc; j++) cc += h.M2[(size_t)ro * nc + j] * h.bh_nat[j];
        row[NP + 4 * D.NE] = cc;
    }
    // constants: QQ, Te, Se, bounds per internal z row, cone shifts per internal cone row
    std::vector<double> flat;
    std::vector<double> lbv(16 * (D.NA + D.NB), 0.0), ubv(16 * (D.NA + D.NB), 0.0), d1(16 * (D.NC / 3), 0.0), d2(16 * (D.NC / 3), 0.0);  // cone shifts, one per cone (sets of sixteen)
    for (int r = 0; r < h.dim; r++) {  // (coupled constraints: z is free)
        lbv[r] = (!ny && r < D.n_box) ? h.LB[r] : -1e300;
        ubv[r] = (!ny && r < D.n_box) ? h.UB[r] : 1e300;
    }
    for (int j = 0; j < D.n_sb; j++) { lbv[16 * D.NA + j] = h.LBy[j % ny]; ubv[16 * D.NA + j] = h.UBy[j % ny]; }
    if (!h.use_soc)
        for (int t = 0; t < D.n_soc; t++) { d1[t] = h.LBy[t]; d2[t] = h.UBy[t]; }
    p.oQQ = put(flat, std::vector<double>(h.QQ, h.QQ + n * n));
    p.oTe = put(flat, std::vector<double>(h.Te, h.Te + n * n));
    p.oSe = put(flat, std::vector<double>(h.Se, h.Se + m * m));
    p.oLB = put(flat, lbv);
    p.oUB = put(flat, ubv);
    p.oD1 = put(flat, d1);
    p.oD2 = put(flat, d2);
    p.oZcol = p.oZcoef = p.oZd = 0;
    return finish_plan(p, D, n, m, N, h.use_soc, h.symmetric, 0, Mx, flat, ny);
}

int plan_build_nosplit(Plan &p, const NosplitHost &h) {
    const int n = h.n, m = h.m, N = h.N;
    // coupled output constraints (COUPLED_CONSTRAINTS): the box-type slack rows are the N n_y output slacks, not one per decision variable
    int ny = 0;
    {
        const int nm = n + m, dim0 = (N - 1) * nm + m + 3 * nm;
        if (h.n_box != dim0 - 3 * nm) {
            if (h.n_box <= 0 || h.n_box % N != 0) { p.why = "unexpected HMPC dimensions"; return 0; }
            ny = h.n_box / N;
        }
    }
    const Dims D(n, m, N, h.use_soc != 0, 1, ny);
    if (D.dim != h.dim || D.n_s != h.n_s || D.n_soc != h.n_soc || D.n_box != h.n_box) { p.why = "unexpected HMPC dimensions"; return 0; }
    if (D.NR > 24) { p.why = "FUSED: more than 24 row registers"; return 0; }
    if (ny && m > 16) { p.why = "FUSED, coupled constraints: m > 16"; return 0; }
    const int dim = h.dim, n_s = h.n_s, NP = 16 * D.NR;
    // dense C [n_s][dim] from its CSR form
    std::vector<double> Cd((size_t)n_s * dim, 0.0);
    for (int i = 0; i < n_s; i++)
        for (int q = h.C_row[i]; q < h.C_row[i + 1]; q++) Cd[(size_t)i * dim + h.C_col[q]] += h.C_val[q];
    // G1 = M1 C'  [dim][n_s];  CG = C G1  [n_s][n_s]  (code_HMPC_ADMM_C.c:123-157, 161-171 folded)
    std::vector<double> G1((size_t)dim * n_s, 0.0), CG((size_t)n_s * n_s, 0.0);
    for (int i = 0; i < dim; i++)
        for (int j = 0; j < dim; j++) {
            const double mij = h.M1[(size_t)i * dim + j];
            if (mij == 0.0) continue;
            for (int r = 0; r < n_s; r++) {
                const double c = Cd[(size_t)r * dim + j];
                if (c != 0.0) G1[(size_t)i * n_s + r] += mij * c;
            }
        }
    for (int i = 0; i < n_s; i++)
        for (int j = 0; j < dim; j++) {
            const double c = Cd[(size_t)i * dim + j];
            if (c == 0.0) continue;
            for (int r = 0; r < n_s; r++) CG[(size_t)i * n_s + r] += c * G1[(size_t)j * n_s + r];
        }
    // CIz = M2 b + M1 q = Lx0 x0 + Lxr xr + Lur ur   (b = -A x0, q rows xe / xc / ue: :83-105)
    const int o0 = D.o0, xe = o0, xc = o0 + 2 * n, ue = o0 + 3 * n;
    const int nin = 2 * n + m;
    std::vector<double> L((size_t)dim * nin, 0.0);  // columns: x0 (n), xr (n), ur (m)
    for (int i = 0; i < dim; i++) {
        for (int c = 0; c < n; c++) {
            double acc = 0.0;
            for (int j = 0; j < n; j++) {
                acc -= h.M2[(size_t)i * n + j] * h.A[j * n + c];
                acc -= (h.M1[(size_t)i * dim + xe + j] + h.M1[(size_t)i * dim + xc + j]) * h.QQ[j * n + c];
            }
            L[(size_t)i * nin + c] = acc;
            double accr = 0.0;
            for (int j = 0; j < n; j++) accr -= h.M1[(size_t)i * dim + xe + j] * h.Te[j * n + c];
            L[(size_t)i * nin + n + c] = accr;
        }
        for (int c = 0; c < m; c++) {
            double acc = 0.0;
            for (int j = 0; j < m; j++) acc -= h.M1[(size_t)i * dim + ue + j] * h.Se[j * m + c];
            L[(size_t)i * nin + 2 * n + c] = acc;
        }
    }
    std::vector<double> dv(h.d, h.d + (h.d ? n_s : 0));
    if (dv.empty()) dv.assign(n_s, 0.0);
    // internal row -> row of s, or -1
    std::vector<int> orig(NP, -1);
    for (int r = 0; r < D.n_box; r++) orig[r] = r;
    for (int t = 0; t < D.n_soc; t++)
        for (int i = 0; i < 3; i++) orig[16 * (D.NA + 3 * (t / 16) + i) + t % 16] = D.n_box + 3 * t + i;
    const int ncol = D.ncol;
    std::vector<double> Mx((size_t)NP * ncol, 0.0);
    const int cx0 = NP, cxr = NP + 4 * D.NXS, cur = NP + 8 * D.NXS, cone = NP + 4 * D.NE;
    for (int ri = 0; ri < NP; ri++) {
        const int ro = orig[ri];
        if (ro < 0) continue;
        double *row = &Mx[(size_t)ri * ncol];
        double cst = -dv[ro];
        for (int ci = 0; ci < NP; ci++)
            if (orig[ci] >= 0) {
                row[ci] = CG[(size_t)ro * n_s + orig[ci]];
                cst -= h.rho * CG[(size_t)ro * n_s + orig[ci]] * dv[orig[ci]];  // operand is rho s + lambda: -rho d goes to the constant
            }
        for (int j = 0; j < dim; j++) {
            const double c = Cd[(size_t)ro * dim + j];
            if (c == 0.0) continue;
            for (int q = 0; q < n; q++) {
                row[cx0 + q] += c * L[(size_t)j * nin + q];
                row[cxr + q] += c * L[(size_t)j * nin + n + q];
            }
            for (int q = 0; q < m; q++) row[cur + q] += c * L[(size_t)j * nin + 2 * n + q];
        }
        row[cone] = cst;
    }
    if (ny) {  // u = z[0 .. m) = (M2 b + M1 q)[0 .. m) + G1[0 .. m) (rho (s - d) + lambda): the last row register
        for (int jr = 0; jr < m; jr++) {
            double *row = &Mx[(size_t)(16 * (D.NR - 1) + jr) * ncol];
            double cst = 0.0;
            for (int ci = 0; ci < NP; ci++)
                if (orig[ci] >= 0) {
                    row[ci] = G1[(size_t)jr * n_s + orig[ci]];
                    cst -= h.rho * G1[(size_t)jr * n_s + orig[ci]] * dv[orig[ci]];
                }
            for (int q = 0; q < n; q++) {
                row[cx0 + q] = L[(size_t)jr * nin + q];
                row[cxr + q] = L[(size_t)jr * nin + n + q];
            }
            for (int q = 0; q < m; q++) row[cur + q] = L[(size_t)jr * nin + 2 * n + q];
            row[cone] = cst;
        }
    }
    // z of the last product from C z - d: one slack row per decision variable whose row of C holds that entry alone
    std::vector<double> zcol(NP, -1.0), zcoef(NP, 0.0), zd(NP, 0.0);
    std::vector<int> inv(n_s, -1);
    for (int ri = 0; ri < NP; ri++)
        if (orig[ri] >= 0) inv[orig[ri]] = ri;
    for (int c = 0; c < dim && !ny; c++) {  // (coupled constraints: u comes from its own row register, the z record from the GEMM variant)
        int found = -1;
        for (int i = 0; i < n_s && found < 0; i++) {
            if (Cd[(size_t)i * dim + c] == 0.0 || zcol[inv[i]] >= 0) continue;
            int nnz = 0;
            for (int j = 0; j < dim; j++) nnz += Cd[(size_t)i * dim + j] != 0.0;
            if (nnz == 1) found = i;
        }
        if (found < 0) { p.why = "FUSED: a decision variable has no slack row of its own (coupled constraints)"; return 0; }
        zcol[inv[found]] = c;
        zcoef[inv[found]] = 1.0 / Cd[(size_t)found * dim + c];
        zd[inv[found]] = dv[found];
    }
    std::vector<double> flat;
    std::vector<double> lbv(16 * D.NA, 0.0), ubv(16 * D.NA, 0.0), d1(16 * (D.NC / 3), 0.0), d2(16 * (D.NC / 3), 0.0);  // cone shifts, one per cone (sets of sixteen)
    for (int r = 0; r < D.n_box; r++) { lbv[r] = h.LB[r]; ubv[r] = h.UB[r]; }
    if (!h.use_soc)
        for (int t = 0; t < D.n_soc; t++) { d1[t] = h.LBy[t]; d2[t] = h.UBy[t]; }
    p.oQQ = p.oTe = p.oSe = 0;
    p.oLB = put(flat, lbv);
    p.oUB = put(flat, ubv);
    p.oD1 = put(flat, d1);
    p.oD2 = put(flat, d2);
    p.oZcol = put(flat, zcol);
    p.oZcoef = put(flat, zcoef);
    p.oZd = put(flat, zd);
    if (ny) {  // the z record of the coupled form: z = L [x0; xr; ur] + G1 (rho s + lambda) - rho G1 d, one row of d_ZR per decision variable
        const int w = nin + n_s + 1;
        std::vector<double> ZR((size_t)dim * w, 0.0);
        for (int i = 0; i < dim; i++) {
            double cst = 0.0;
            for (int q = 0; q < nin; q++) ZR[(size_t)i * w + q] = L[(size_t)i * nin + q];
            for (int r = 0; r < n_s; r++) {
                ZR[(size_t)i * w + nin + r] = G1[(size_t)i * n_s + r];
                cst -= h.rho * G1[(size_t)i * n_s + r] * dv[r];
            }
            ZR[(size_t)i * w + nin + n_s] = cst;
        }
        for (double x : ZR)
            if (!std::isfinite(x)) { p.why = "non-finite M1 / M2"; return 0; }
        if (p.d_ZR) hipFree(p.d_ZR);
        SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_ZR, ZR.size() * sizeof(double)));
        SPCIES_HIP_CHECK(hipMemcpy(p.d_ZR, ZR.data(), ZR.size() * sizeof(double), hipMemcpyHostToDevice));
        p.z_dim = dim; p.z_ns = n_s; p.z_rho = h.rho;
    }
    return finish_plan(p, D, n, m, N, h.use_soc, h.symmetric, 1, Mx, flat, ny);
}

// z of the iteration every instance stopped at, HMPC without the splitting and with coupled constraints (code_HMPC_ADMM_C.c:123-157):
// z = (M2 b + M1 q) + (M1 C') (rho (s - d) + lambda) with the s, lambda the last product consumed - the operand T = rho s + lambda the FUSED
// kernel left in `T` at the exit.  One workgroup per instance (grid-stride), the inputs and T staged in LDS, one row of z per thread and
// pass; once per solve, off the iteration: 2 dim (2 n + m + n_s) flop per instance.
__global__ __launch_bounds__(128) void z_from_operand_kernel(const double *__restrict__ ZR, int dim, int n_s, int n, int m, const double *__restrict__ x0g,
                                                             const double *__restrict__ xrg, const double *__restrict__ urg, int ref_stride, long B,
                                                             const double *__restrict__ T, double *__restrict__ z_out) {
    extern __shared__ double sh[];  // [x0; xr; ur] (2 n + m) | T (n_s)
    const int nin = 2 * n + m, w = nin + n_s + 1;
    for (long inst = blockIdx.x; inst < B; inst += gridDim.x) {
        const double *xr = ref_stride ? xrg + inst * n : xrg, *ur = ref_stride ? urg + inst * m : urg;
        for (int i = threadIdx.x; i < nin + n_s; i += blockDim.x)
            sh[i] = i < n ? x0g[inst * n + i] : (i < 2 * n ? xr[i - n] : (i < nin ? ur[i - 2 * n] : T[inst * (long)n_s + (i - nin)]));
        __syncthreads();
        for (int j = threadIdx.x; j < dim; j += blockDim.x) {
            const double *row = ZR + (size_t)j * w;
            double a = row[nin + n_s];
            for (int i = 0; i < nin + n_s; i++) a += row[i] * sh[i];
            z_out[inst * (long)dim + j] = a;
        }
        __syncthreads();
    }
}

int launch(Plan &p, int k_max, double tol_p, double tol_d, double rho, double rho_i, double sigma, double sigma_i, double alpha,
           const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u, int *k, int *e,
           double *const *f, hipStream_t st) {
    if (!p.ok) return fail(SPCIES_HIP_ENOSUP, "FUSED variant unavailable: %s", p.why.c_str());
    Args a{};
    a.B = B; a.ref_stride = ref_stride; a.k_max = k_max; a.tol_p = tol_p; a.tol_d = tol_d; a.rho = rho; a.rho_i = rho_i;
    a.sigma = sigma; a.sigma_i = sigma_i; a.alpha = alpha;
    a.oQQ = p.oQQ; a.oTe = p.oTe; a.oSe = p.oSe; a.oLB = p.oLB; a.oUB = p.oUB; a.oD1 = p.oD1; a.oD2 = p.oD2;
    a.oZcol = p.oZcol; a.oZcoef = p.oZcoef; a.oZd = p.oZd;
    const int nf = p.mode == 0 ? 6 : 3;
    double *ff[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool want_sol = false;
    for (int i = 0; i < nf; i++) { ff[i] = f[i]; want_sol |= f[i] != nullptr; }
    double *z_want = nullptr;
    if (p.mode == 1 && p.ny > 0 && f[0]) {
        // coupled constraints: no slack row carries a decision variable, so the kernel cannot read z off its accumulators.  It leaves the
        // operand of every instance's last product in d_T (field slot 3, unused by this solver's record) and z_from_operand_kernel forms z
        if (!p.d_ZR) return fail(SPCIES_HIP_ENOSUP, "FUSED, HMPC without the splitting and with coupled constraints: no z-record table");
        if (rho != p.z_rho) return fail(SPCIES_HIP_ENOSUP, "FUSED: rho changed after the z-record table was folded");
        if (B > p.cap_T) {  // grown on demand (a device synchronisation: make one record call before capturing into a hipGraph)
            if (p.d_T) hipFree(p.d_T);
            p.d_T = nullptr;
            p.cap_T = 0;
            SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_T, (size_t)B * p.z_ns * sizeof(double)));
            p.cap_T = B;
        }
        z_want = f[0];
        ff[0] = nullptr;
        ff[3] = p.d_T;
    }
    const long groups = (B + 31) / 32;
    const unsigned grid = (unsigned)std::min<long>(groups, p.num_cu);
    const double *ME = p.d_ME, *PRO = p.d_PRO, *C = p.d_C;
    if (p.builtin >= 0) {  // (the build-time shapes have box constraints: no z_want here)
        int idx = 0;
#define X(nn, mm, NN, SS, UU, MM)                                                                                              \
    if (p.builtin == idx) return launch_builtin<nn, mm, NN, SS, UU, MM>(a, ME, PRO, C, x0, xr, ur, u, k, e, ff, want_sol, grid, st); \
    idx++;
        SPCIES_HFUSED_SHAPES(X)
#undef X
        return fail(SPCIES_HIP_ENOSUP, "FUSED: bad build-time shape index");
    }
    double *f0 = ff[0], *f1 = ff[1], *f2 = ff[2], *f3 = ff[3], *f4 = ff[4], *f5 = ff[5];
    void *params[] = {&a, &ME, &PRO, &C, &x0, &xr, &ur, &u, &k, &e, &f0, &f1, &f2, &f3, &f4, &f5};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn[want_sol ? 1 : 0], grid, 1, 1, kNWV * 64, 1, 1, 0, st, params, nullptr));
    if (z_want) {
        const int nin = 2 * p.n + p.m;
        const unsigned zgrid = (unsigned)std::min<long>(B, 8L * p.num_cu);
        hipLaunchKernelGGL(z_from_operand_kernel, dim3(zgrid), dim3(128), (size_t)(nin + p.z_ns) * sizeof(double), st, p.d_ZR, p.z_dim, p.z_ns, p.n, p.m, x0,
                           xr, ur, ref_stride, B, p.d_T, z_want);
        SPCIES_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

}  // namespace hfused
}  // namespace spcies
