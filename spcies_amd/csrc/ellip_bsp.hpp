// Variant BSP of the ellipMPC ADMM solver (formulations/+ellipMPC/code_ellipMPC_ADMM_C.c:20-523): the generator of
// soc_bsp.hpp applied to a banded solver.  The z-update of the lax-type iteration is the equality-constrained QP
//     z = -H^-1 (q_hat + G' mu),   W mu = -(G H^-1 q_hat + b),   W = G H^-1 G',
// with H^-1 = blkdiag(Hi_0, Hi_1 .. Hi_{N-1}, Hi_N) (diagonal but for the dense terminal block inv(T + rho P)) and G the
// dynamics [B -I; A B -I; ...] - which the reference solves through its banded Cholesky factors Alpha / Beta
// (:355-486).  Here W is factorised L D L' on the host and the iteration becomes the same kind of program as the soc
// solver's: rhs = (-G H^-1) q_hat - b column by column, block forward / backward substitution, z = (-H^-1) q_hat +
// (-H^-1 G') mu row by row with the box update of each slab in w-form (w = v + lambda / rho), and the terminal block in
// the reference's P-coordinates (:146-156, 318-386): q_hat_N = qT + P_half lambda_N - rho P v_N,
// v_N = ellipsoid projection of z_N + (1/rho) Pinv_half lambda_N, lambda_N += P_half rho (z_N - v_N) - four dense block
// products of the terminal slabs and one cross-lane norm.  Same result as the banded path up to rounding (1e-10 bar).
#pragma once
#include "soc_bsp.hpp"

namespace spcies {
namespace bsp {

struct EArgs {  // mirrored in the generated source
    int n, m, N, dim, k_max, ref_stride;
    double tol, rho, rho_i, r2, r;
    long B;
};

// dense L D L' of a symmetric positive definite matrix (no pivoting)
inline bool dense_ldl(const Dense &W, int n, Dense &L, std::vector<double> &Dinv) {
    L.assign((size_t)n * n, 0.0);
    Dinv.assign(n, 0.0);
    std::vector<double> D(n, 0.0);
    for (int j = 0; j < n; j++) {
        double d = W[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k] * D[k];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        D[j] = d;
        Dinv[j] = 1.0 / d;
        L[(size_t)j * n + j] = 1.0;
        for (int i = j + 1; i < n; i++) {
            double s = W[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k] * D[k];
            L[(size_t)i * n + j] = s / d;
        }
    }
    return true;
}

inline int build_ellip(Plan &p, const AdmmHost &a, int pf_request = 0) {
    // equ mode (equMPC ADMM, code_equMPC_ADMM_C.c): no terminal variable - the last block row of G reads A x_{N-1} + B u_{N-1} = xr - so
    // the program has no terminal slabs (TS = 0) and the reference enters the right-hand side of its last n rows
    const bool equ = !a.terminal;
    const int n = a.n, m = a.m, nm = n + m, N = a.N, dz = m + (N - 1) * nm, dim = equ ? dz : N * nm, nr = N * n;
    const int ZS = (dz + 3) / 4, TS = equ ? 0 : (n + 3) / 4, NP = ZS + TS, NR = (nr + 3) / 4, PR_ = 4 * NP, RR = 4 * NR;
    p.ok = false;
    p.src.clear();
    p.ZS = ZS; p.SS = TS; p.NR = NR;
    // lax mode: the same program for the laxMPC ADMM solver (scalar or vector rho, constant or stage-wise bounds): the terminal
    // block is a box like the others, its dense weight inv(T + rho I) stays inside the terminal slabs
    const bool lax = !a.ellip;
    if (equ && (a.ellip || N < 2)) { p.why = "no terminal block"; return 0; }
    std::vector<double> rho_row(dim, a.rho), lb_row(dim, 0.0), ub_row(dim, 0.0);
    for (int r = 0; r < dim; r++) {
        const int e = (r < m) ? -1 : (r < dz ? (r - m) % nm : -2);
        if (a.ellip || a.gen) {
            lb_row[r] = r < m ? a.LBu0[r] : (r < dz ? a.LBz[r - m] : (lax ? a.LBN[r - dz] : 0.0));
            ub_row[r] = r < m ? a.UBu0[r] : (r < dz ? a.UBz[r - m] : (lax ? a.UBN[r - dz] : 0.0));
            if (a.gen) rho_row[r] = r < m ? a.rho_0[r] : (r < dz ? a.rho_v[r - m] : a.rho_N[r - dz]);
        } else {
            const int j = (e == -1) ? n + r : (e == -2 ? r - dz : e);
            lb_row[r] = a.LB[j];
            ub_row[r] = a.UB[j];
        }
    }
    if (m > 4) { p.why = "m > 4 (u rows outside slab 0)"; return 0; }
    if (NR + ZS + 4 * TS > 190) { p.why = "state does not fit the register file"; return 0; }
    auto ip = [&](int j) { return j < dz ? j : 4 * ZS + (j - dz); };
    // ---- H^-1, G (natural order), W = G H^-1 G'
    Dense Hinv((size_t)dim * dim, 0.0), G((size_t)nr * dim, 0.0);
    for (int j = 0; j < m; j++) Hinv[(size_t)j * dim + j] = a.Hi_0[j];
    for (int l = 0; l < N - 1; l++)
        for (int j = 0; j < nm; j++) {
            const int r = m + l * nm + j;
            Hinv[(size_t)r * dim + r] = a.Hi[(size_t)l * nm + j];
        }
    if (!equ)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Hinv[(size_t)(dz + i) * dim + dz + j] = a.Hi_N[(size_t)i * n + j];
    for (int i = 0; i < n; i++) {  // block 0: B u0 - x1 = -A x0
        for (int j = 0; j < m; j++) G[(size_t)i * dim + j] = a.AB[(size_t)i * nm + n + j];
        G[(size_t)i * dim + m + i] = -1.0;
    }
    for (int l = 1; l < N; l++)  // block l: A x_l + B u_l - x_{l+1} = 0
        for (int i = 0; i < n; i++) {
            const int row = l * n + i, c0 = m + (l - 1) * nm;
            for (int j = 0; j < nm; j++) G[(size_t)row * dim + c0 + j] = a.AB[(size_t)i * nm + j];
            if (l < N - 1 || !equ) G[(size_t)row * dim + (l < N - 1 ? m + l * nm + i : dz + i)] = -1.0;  // (equ: x_N = xr sits in b)
        }
    Dense GH((size_t)nr * dim, 0.0), W((size_t)nr * nr, 0.0);
    for (int i = 0; i < nr; i++)
        for (int k = 0; k < dim; k++) {
            const double g = G[(size_t)i * dim + k];
            if (g == 0.0) continue;
            for (int j = 0; j < dim; j++) {
                const double h = Hinv[(size_t)k * dim + j];
                if (h != 0.0) GH[(size_t)i * dim + j] += g * h;
            }
        }
    // (ellipMPC with a vector rho: the terminal block of H^-1, inv(T + diag(rho_N) P), is not symmetric, so neither is W; the reference
    // factorises it with MATLAB's chol, which reads the UPPER triangle - compute_ellipMPC_ADMM_ingredients.m:97-99)
    for (int i = 0; i < nr; i++)
        for (int j = i; j < nr; j++) {
            double s = 0.0;
            for (int k = 0; k < dim; k++) s += GH[(size_t)i * dim + k] * G[(size_t)j * dim + k];
            W[(size_t)i * nr + j] = W[(size_t)j * nr + i] = s;
        }
    Dense HGt((size_t)dim * nr, 0.0);  // H^-1 G'
    for (int j = 0; j < dim; j++)
        for (int k = 0; k < dim; k++) {
            const double h = Hinv[(size_t)j * dim + k];
            if (h == 0.0) continue;
            for (int i = 0; i < nr; i++) {
                const double g = G[(size_t)i * dim + k];
                if (g != 0.0) HGt[(size_t)j * nr + i] += h * g;
            }
        }
    Dense Ln;
    std::vector<double> Dn;
    if (!dense_ldl(W, nr, Ln, Dn)) { p.why = "W = G H^-1 G' is not positive definite"; return 0; }
    // ---- internal layout
    Dense Gm((size_t)RR * PR_, 0.0), HG((size_t)PR_ * RR, 0.0), H((size_t)PR_ * PR_, 0.0), L((size_t)RR * RR, 0.0), Dinv(RR, 1.0);
    for (int i = 0; i < nr; i++)
        for (int j = 0; j < dim; j++) {
            Gm[(size_t)i * PR_ + ip(j)] = -GH[(size_t)i * dim + j];
            HG[(size_t)ip(j) * RR + i] = -HGt[(size_t)j * nr + i];
        }
    for (int i = 0; i < dim; i++)
        for (int j = 0; j < dim; j++) H[(size_t)ip(i) * PR_ + ip(j)] = -Hinv[(size_t)i * dim + j];
    for (int i = 0; i < RR; i++) L[(size_t)i * RR + i] = 1.0;
    for (int i = 0; i < nr; i++) {
        for (int j = 0; j < i; j++) L[(size_t)i * RR + j] = Ln[(size_t)i * nr + j];
        Dinv[i] = Dn[i];
    }
    // ---- unit-box coordinates for the z slabs (round 5; soc_bsp.hpp has the algebra): scalar rho, the grouped program with the bound patterns in
    // registers, every row of the slab with a finite box around 0.  kappa = q / (rho D) - lb / D per instance and (q pattern, bound pattern);
    // c' = clamp01(w^ - kappa), s = w^ - 2 c'; rho D folded into the columns of -G H^-1 and -H^-1, 1 / D into the rows of -H^-1 and -H^-1 G',
    // -lb / D seeds the row's accumulator.  The terminal slabs (P-coordinates / the lax box) keep their form.  SPCIES_BSP_UNIT=0: off.
    std::vector<char> unit(ZS, 0);
    {
        // (measured at the C2 shape, 40 launches each, median: the lax program 7.40 -> 7.20 ms; the ellipMPC program 7.62 -> 7.69 - its terminal
        // P-coordinate phase reschedules worse - so the ellipsoid mode keeps the plain form unless SPCIES_BSP_UNIT=1 asks)
        const char *uev = getenv("SPCIES_BSP_UNIT");
        bool on = !a.gen && ((lax || equ) ? !(uev && uev[0] == '0') : (uev && uev[0] == '1')) &&
                  !(getenv("SPCIES_BSP_GROUP") && getenv("SPCIES_BSP_GROUP")[0] == '0') && !getenv("SPCIES_BSP_BND_REGS");
        std::map<std::vector<double>, int> pat;
        for (int J = 0; J < ZS; J++) {
            std::vector<double> key;
            for (int r = 0; r < 4; r++) {
                const int row = 4 * J + r;
                key.push_back(row < dz ? lb_row[row] : 0.0);
                key.push_back(row < dz ? ub_row[row] : 0.0);
            }
            pat.emplace(key, (int)pat.size());
        }
        if (pat.size() > 24) on = false;  // (the patterns must live in registers)
        for (int J = 0; J < ZS && on; J++) {
            bool ok = true;
            for (int r = 4 * J; r < 4 * J + 4; r++) {
                if (r >= dz) { ok = false; break; }
                const double lo = lb_row[r], hi = ub_row[r];
                if (!(std::isfinite(lo) && std::isfinite(hi)) || !(lo <= 0.0 && hi >= 0.0) || !(hi - lo > 1e-9) || hi - lo > 1e5) { ok = false; break; }
            }
            unit[J] = ok;
        }
        for (int J = 0; J < ZS; J++)
            if (unit[J])
                for (int r = 4 * J; r < 4 * J + 4; r++) {  // (z rows keep their index in the internal layout)
                    const double D = ub_row[r] - lb_row[r], sD = a.rho * D;
                    for (int i = 0; i < RR; i++) Gm[(size_t)i * PR_ + ip(r)] *= sD;
                    for (int i = 0; i < PR_; i++) H[(size_t)i * PR_ + ip(r)] *= sD;
                }
        for (int J = 0; J < ZS; J++)
            if (unit[J])
                for (int r = 4 * J; r < 4 * J + 4; r++) {
                    const double D = ub_row[r] - lb_row[r];
                    for (int j = 0; j < PR_; j++) H[(size_t)ip(r) * PR_ + j] /= D;
                    for (int j = 0; j < RR; j++) HG[(size_t)ip(r) * RR + j] /= D;
                }
    }
    const BlockList bG = blocks_of(Gm, RR, PR_), bHG = blocks_of(HG, PR_, RR), bH = blocks_of(H, PR_, PR_), bL = blocks_of(L, RR, RR);
    // ring depth: measured at the C2 shape (ms, scratch B per lane): 8: 10.5 / 0, 12: 9.94 / 116, 16: 9.77 / 148, 20: 9.58 / 156, 24: 9.29 / 280
    // (the laxMPC programs do not care: 12.5 ms and no scratch at any depth)
    // round 3: the ring holds block PAIRS read by ds_read_b128 (PF counts pairs; SPCIES_BSP_PAIRS=0: single blocks, PF counts blocks) -
    // one ds_read_b64 per product kept the LDS pipe busier than the matrix pipe (soc_bsp.hpp)
    const bool pairs = !(getenv("SPCIES_BSP_PAIRS") && getenv("SPCIES_BSP_PAIRS")[0] == '0');
    int SEG_EVERY = 4, PF = pairs ? 12 : 24;
    if (const char *ev = getenv("SPCIES_BSP_SEG")) SEG_EVERY = std::max(1, atoi(ev));
    if (const char *ev = getenv("SPCIES_BSP_PF")) PF = std::min(64, std::max(2, atoi(ev)));
    if (pf_request > 0) PF = pairs ? std::max(2, pf_request / 2) : pf_request;
    std::vector<double> &tab = p.table;
    tab.clear();
    std::string body;
    char line[512], a1[64], a2[64];
    int n_mfma = 0;
    auto emit_block = [&](const double *blk) {
        const int t = (int)(tab.size() / 16);
        for (int k = 0; k < 4; k++)
            for (int i = 0; i < 4; i++) tab.push_back(blk[i * 4 + k]);
        return t;
    };
    auto block_of = [&](const Dense &M, int cols, int Ib, int Jb, double *out) {
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 4; k++) out[i * 4 + k] = M[(size_t)(4 * Ib + i) * cols + 4 * Jb + k];
    };
    auto mul44 = [&](const double *x, const double *y, double *o, double sign) {
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 4; k++) {
                double s = 0.0;
                for (int q = 0; q < 4; q++) s += x[i * 4 + q] * y[q * 4 + k];
                o[i * 4 + k] = sign * s;
            }
    };
    auto inv_unit_lower = [&](const double *x, double *o) {
        for (int col = 0; col < 4; col++)
            for (int i = 0; i < 4; i++) {
                double s = (i == col) ? 1.0 : 0.0;
                for (int q = 0; q < i; q++) s -= x[i * 4 + q] * o[q * 4 + col];
                o[i * 4 + col] = s;
            }
    };
    auto MF = [&](const char *acc, int t, const char *x) {
        if (pairs) {
            if (t % 2) snprintf(line, sizeof(line), "            MF(%s, a%d.y, %s); @%d@\n", acc, (t / 2) % PF, x, t / 2);
            else snprintf(line, sizeof(line), "            MF(%s, a%d.x, %s);\n", acc, (t / 2) % PF, x);
        } else {
            snprintf(line, sizeof(line), "            MF(%s, a%d, %s); @%d@\n", acc, t % PF, x, t);
        }
        body += line;
        n_mfma++;
    };
    // a dense n x n matrix (scaled) applied to TS slabs: out[k] += sum_j M[k][j] in[j]
    // (colscale: vector rho - P diag(rho_N), Pinv_half diag(rho_i_N), code_ellipMPC_ADMM_C.c:152, 328)
    auto dense_tail = [&](const std::vector<double> &M, double scale, const char *out, const char *in,
                          const std::vector<double> *colscale = nullptr) {
        Dense Mp((size_t)4 * TS * 4 * TS, 0.0);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Mp[(size_t)i * 4 * TS + j] = scale * M[(size_t)i * n + j] * (colscale ? (*colscale)[j] : 1.0);
        for (int k = 0; k < TS; k++)
            for (int j = 0; j < TS; j++) {
                double blk[16];
                block_of(Mp, 4 * TS, k, j, blk);
                bool nz = false;
                for (double x : blk) nz |= x != 0.0;
                if (!nz) continue;
                snprintf(a1, sizeof(a1), "%s[%d]", out, k);
                snprintf(a2, sizeof(a2), "%s[%d]", in, j);
                MF(a1, emit_block(blk), a2);
            }
    };
    auto qhat_expr = [&](int J, char *out, size_t cap) {
        if (J < ZS) snprintf(out, cap, "QHZ(%d)", J);
        else snprintf(out, cap, "qt[%d]", J - ZS);
    };
    // ---- terminal q_hat in P-coordinates (:146-156)
    if (!equ) body += "            // q_hat_N = qT + P_half lambda_N - rho P v_N  (lax: qT + lambda_N - rho v_N)\n            double qt[TS_];\n";
    for (int k = 0; k < TS; k++) {
        if (lax) snprintf(line, sizeof(line), "            qt[%d] = qTv[%d] + RHOT(%d) * (wN[%d] - 2.0 * fmin(fmax(wN[%d], LBT(%d)), UBT(%d)));\n", k, k, k, k, k, k, k);
        else snprintf(line, sizeof(line), "            qt[%d] = qTv[%d];\n", k, k);
        body += line;
    }
    if (!lax) {
        dense_tail(a.P_half, 1.0, "qt", "lamN");
        if (a.gen) dense_tail(a.P, -1.0, "qt", "vN", &a.rho_N);
        else dense_tail(a.P, -a.rho, "qt", "vN");
    }
    body += "            SEG;\n            // rhs = (-G H^-1) q_hat - b\n";
    const int bh_slabs = (n + 3) / 4;  // b = -A x0 in the first n rows
    const int bt_first = equ ? ((N - 1) * n) / 4 : NR;  // equ: b = xr in the last n rows (code_equMPC_ADMM_C.c:337-352)
    for (int Ib = 0; Ib < NR; Ib++) {
        if (Ib < bh_slabs && Ib >= bt_first) snprintf(line, sizeof(line), "            rh[%d] = -bh[%d] - bt[%d];\n", Ib, Ib, Ib - bt_first);
        else if (Ib < bh_slabs) snprintf(line, sizeof(line), "            rh[%d] = -bh[%d];\n", Ib, Ib);
        else if (Ib >= bt_first) snprintf(line, sizeof(line), "            rh[%d] = -bt[%d];\n", Ib, Ib - bt_first);
        else snprintf(line, sizeof(line), "            rh[%d] = 0.0;\n", Ib);
        body += line;
    }
    // (round 3: q_hat of SEG_EVERY slabs in ONE run of vector instructions in front of their products - a fence keeps the compiler from
    // putting each slab's four instructions right in front of its own MFMAs: a switch from the matrix instruction to a vector instruction costs
    // 8 clocks, profiles/r03_microbench_issue.txt)
    const bool group_qhat = !(getenv("SPCIES_BSP_GROUP") && getenv("SPCIES_BSP_GROUP")[0] == '0');
    // the distinct (LB, UB, rho) slab patterns: up to 24 live in registers; beyond (vector rho / stage-wise bounds) every slab reads its rows
    // from LDS - and those reads are requested ONE GROUP AHEAD, behind the previous group's products (a read requested in front of the run of
    // vector instructions that needs it is waited for: nothing hides it at one wavefront per SIMD)
    bool in_regs_early;
    {
        std::map<std::vector<double>, int> pat;
        for (int J = 0; J < ZS; J++) {
            std::vector<double> key;
            for (int r = 0; r < 4; r++) {
                const int row = 4 * J + r;
                key.push_back(row < dz ? lb_row[row] : 0.0);
                key.push_back(row < dz ? ub_row[row] : 0.0);
                key.push_back(row < dz ? rho_row[row] : 1.0);
            }
            pat.emplace(key, (int)pat.size());
        }
        in_regs_early = pat.size() <= 24;
        if (const char *ev = getenv("SPCIES_BSP_BND_REGS")) in_regs_early = atoi(ev) != 0 && pat.size() <= 24;
    }
    const bool prefetch_bounds = group_qhat && !in_regs_early && !(getenv("SPCIES_BSP_BND_PREFETCH") && getenv("SPCIES_BSP_BND_PREFETCH")[0] == '0');
    auto bound_loads = [&](int J0, int J1, const char *tag) {  // named copies of the group's bound / rho rows
        std::string o;
        for (int J = J0; J < J1; J++) {
            snprintf(line, sizeof(line), "              const double lb%s_%d = LBL(%d), ub%s_%d = UBL(%d), rh%s_%d = RHOL(%d);\n", tag, J, J, tag, J, J, tag, J, J);
            o += line;
        }
        return o;
    };
    if (prefetch_bounds) body += bound_loads(0, std::min(ZS, SEG_EVERY), "g");
    for (int J0 = 0; J0 < NP; J0 += SEG_EVERY) {
        const int J1 = std::min(NP, J0 + SEG_EVERY);
        if (!prefetch_bounds) body += "            {\n";
        for (int J = J0; J < J1 && group_qhat; J++) {
            if (bG.by_col[J].empty()) continue;
            char e[160];
            if (J < ZS && unit[J]) snprintf(e, sizeof(e), "QHZU(%d)", J);
            else if (prefetch_bounds && J < ZS) snprintf(e, sizeof(e), "(qv[QI_%d] + rhg_%d * (w[%d] - 2.0 * fmin(fmax(w[%d], lbg_%d), ubg_%d)))", J, J, J, J, J, J);
            else qhat_expr(J, e, sizeof(e));
            snprintf(line, sizeof(line), "              const double qh_%d = %s;\n", J, e);
            body += line;
        }
        if (group_qhat) body += "              SEG;\n";
        if (prefetch_bounds && J1 < ZS) body += bound_loads(J1, std::min(ZS, J1 + SEG_EVERY), "g");
        for (int J = J0; J < J1; J++) {
            if (bG.by_col[J].empty()) continue;
            char e[64], qn[32];
            qhat_expr(J, e, sizeof(e));
            snprintf(qn, sizeof(qn), "qh_%d", J);
            if (!group_qhat) {
                snprintf(line, sizeof(line), "              const double qh_%d = %s;\n", J, e);
                body += line;
            }
            for (int Ib : bG.by_col[J]) {
                double blk[16];
                block_of(Gm, PR_, Ib, J, blk);
                snprintf(a1, sizeof(a1), "rh[%d]", Ib);
                MF(a1, emit_block(blk), qn);
            }
        }
        body += prefetch_bounds ? "            SEG;\n" : "            }\n            SEG;\n";
    }
    // (the bound rows are read again in the update phase: laundering their index keeps the compiler from holding all of them
    // in registers across the solve)
    body += "            SEG;\n            LAUNDER;\n            // W mu = rhs: forward substitution by blocks\n";
    std::vector<std::vector<double>> Linv(NR, std::vector<double>(16));
    // round 3: both substitutions in column order (soc_bsp.hpp: x_J = Linv_JJ rh_J, then rh_I -= L_IJ x_J for the rows below; the
    // dependent chain D_J -> U_{J+1,J} -> D_{J+1} with the other updates between its links) - SPCIES_BSP_RL=0: one accumulator chain per row
    const bool rl = !(getenv("SPCIES_BSP_RL") && getenv("SPCIES_BSP_RL")[0] == '0');
    struct Pend { int I, J; };
    auto run_columns = [&](bool forward) {
        std::vector<Pend> queue;
        auto pop_front = [&]() { Pend q = queue.front(); queue.erase(queue.begin()); return q; };
        auto emit_update = [&](const Pend &q) {
            double lj[16], o[16];
            if (forward) {
                block_of(L, RR, q.I, q.J, lj);
                for (int e = 0; e < 16; e++) o[e] = -lj[e];
            } else {
                block_of(L, RR, q.J, q.I, lj);
                for (int i = 0; i < 4; i++)
                    for (int k = 0; k < 4; k++) o[i * 4 + k] = -lj[k * 4 + i] / Dinv[4 * q.I + i];
            }
            snprintf(a1, sizeof(a1), "rh[%d]", q.I);
            snprintf(a2, sizeof(a2), "rh[%d]", q.J);
            MF(a1, emit_block(o), a2);
        };
        int step = 0;
        for (int J = forward ? 0 : NR - 1; forward ? J < NR : J >= 0; J += forward ? 1 : -1, step++) {
            const int nxt = forward ? J + 1 : J - 1;
            double d[16];
            if (forward) {
                for (int e = 0; e < 16; e++) d[e] = Linv[J][e];
            } else {
                for (int i = 0; i < 4; i++)
                    for (int k = 0; k < 4; k++) d[i * 4 + k] = Linv[J][k * 4 + i] * Dinv[4 * J + k];
            }
            body += "            { double xx = 0.0;\n";
            snprintf(a2, sizeof(a2), "rh[%d]", J);
            MF("xx", emit_block(d), a2);
            snprintf(line, sizeof(line), "              rh[%d] = xx; }\n", J);
            body += line;
            bool crit = false, any = false;
            while (!queue.empty() && queue.front().I == nxt) { emit_update(pop_front()); any = true; }
            if (!any && !queue.empty()) emit_update(pop_front());
            if (forward) { for (int I : bL.by_col[J]) if (I > J) { if (I == nxt) crit = true; else queue.push_back(Pend{I, J}); } }
            else { for (int I : bL.by_row[J]) if (I < J) { if (I == nxt) crit = true; else queue.push_back(Pend{I, J}); } }
            std::stable_sort(queue.begin(), queue.end(), [&](const Pend &x, const Pend &y) { return forward ? x.I < y.I : x.I > y.I; });
            if (crit) emit_update(Pend{nxt, J});
            bool one = false;
            while ((queue.size() > 1 || (!one && !queue.empty())) && queue.front().I != nxt) { emit_update(pop_front()); one = true; }
            if (step % SEG_EVERY == SEG_EVERY - 1) body += "            SEG;\n";
        }
        while (!queue.empty()) emit_update(pop_front());
    };
    if (rl) {
        for (int Ib = 0; Ib < NR; Ib++) {
            double d[16];
            block_of(L, RR, Ib, Ib, d);
            inv_unit_lower(d, Linv[Ib].data());
        }
        run_columns(true);
        body += "            SEG;\n            // D^-1 and the backward substitution by blocks (U = L')\n";
        run_columns(false);
    }
    for (int Ib = 0; Ib < NR && !rl; Ib++) {
        double d[16];
        block_of(L, RR, Ib, Ib, d);
        inv_unit_lower(d, Linv[Ib].data());
        body += "            { double acc = 0.0;\n";
        snprintf(a2, sizeof(a2), "rh[%d]", Ib);
        MF("acc", emit_block(Linv[Ib].data()), a2);
        for (int J : bL.by_row[Ib]) {
            if (J >= Ib) continue;
            double b[16], o[16];
            block_of(L, RR, Ib, J, b);
            mul44(Linv[Ib].data(), b, o, -1.0);
            snprintf(a2, sizeof(a2), "rh[%d]", J);
            MF("acc", emit_block(o), a2);
        }
        snprintf(line, sizeof(line), "              rh[%d] = acc; }\n", Ib);
        body += line;
        if (Ib % SEG_EVERY == SEG_EVERY - 1) body += "            SEG;\n";
    }
    if (!rl) body += "            SEG;\n            // D^-1 and the backward substitution by blocks (U = L')\n";
    for (int Ib = NR - 1; Ib >= 0 && !rl; Ib--) {
        double ui[16], d[16];
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 4; k++) ui[i * 4 + k] = Linv[Ib][k * 4 + i];
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 4; k++) d[i * 4 + k] = ui[i * 4 + k] * Dinv[4 * Ib + k];
        body += "            { double acc = 0.0;\n";
        snprintf(a2, sizeof(a2), "rh[%d]", Ib);
        MF("acc", emit_block(d), a2);
        for (int J : bL.by_col[Ib]) {
            if (J <= Ib) continue;
            double lj[16], u[16], o[16];
            block_of(L, RR, J, Ib, lj);
            for (int i = 0; i < 4; i++)
                for (int k = 0; k < 4; k++) u[i * 4 + k] = lj[k * 4 + i];
            mul44(ui, u, o, -1.0);
            snprintf(a2, sizeof(a2), "rh[%d]", J);
            MF("acc", emit_block(o), a2);
        }
        snprintf(line, sizeof(line), "              rh[%d] = acc; }\n", Ib);
        body += line;
        if (Ib % SEG_EVERY == 0) body += "            SEG;\n";
    }
    // ---- z = (-H^-1) q_hat + (-H^-1 G') mu.  -H^-1 is diagonal on the z slabs (a block never reaches another slab) and dense
    // only inside the terminal slabs, whose q_hat sits in qt[] for the whole iteration: no q_hat has to be saved.
    for (int Ib = 0; Ib < ZS; Ib++)
        for (int J : bH.by_row[Ib])
            if (J != Ib) { p.why = "H^-1 couples two z slabs"; return 0; }
    body += "            LAUNDER;\n            // z = (-H^-1) q_hat + (-H^-1 G') mu; box update of the z slabs\n";
    // (in the update phase q_hat of a z slab is recomputed from a laundered copy of w: the compiler would otherwise keep the
    // value it formed for the right-hand side alive across both solves - one register pair per slab)
    auto prim_row = [&](int Ib, const char *acc) {
        for (int J : bH.by_row[Ib]) {
            double blk[16];
            block_of(H, PR_, Ib, J, blk);
            char e[64];
            if (J < ZS && group_qhat) snprintf(e, sizeof(e), "qp_%d", J);  // (formed with its group's, in front of the group's products)
            else if (J < ZS) snprintf(e, sizeof(e), "QHZP(%d)", J);
            else qhat_expr(J, e, sizeof(e));
            snprintf(line, sizeof(line), "            { const double qh = %s;\n  ", e);
            body += line;
            MF(acc, emit_block(blk), "qh");
            body += "            }\n";
        }
        for (int J : bHG.by_row[Ib]) {
            double blk[16];
            block_of(HG, RR, Ib, J, blk);
            snprintf(a2, sizeof(a2), "rh[%d]", J);
            MF(acc, emit_block(blk), a2);
        }
    };
    for (int Ib = 0; Ib < ZS && !group_qhat; Ib++) {
        body += "            { double ph = 0.0;\n";
        prim_row(Ib, "ph");
        // (the index of the bound rows is laundered right before the update: the compiler otherwise issues the bound reads of
        // all slabs at the top of the phase and holds 2 x ZS values in registers - 1 KB of scratch at the C2 shape)
        snprintf(line, sizeof(line), "              LAUNDER; ZUPD(%d, ph); }\n", Ib);
        body += line;
        if (Ib % SEG_EVERY == SEG_EVERY - 1) body += "            SEG;\n";
    }
    // (round 3: by groups of SEG_EVERY rows - the q_hat of the group in one run of vector instructions, the group's products, the group's box
    // updates in one run)
    if (prefetch_bounds && group_qhat) body += bound_loads(0, std::min(ZS, SEG_EVERY), "p");
    for (int I0 = 0; I0 < ZS && group_qhat; I0 += SEG_EVERY) {
        const int I1 = std::min(ZS, I0 + SEG_EVERY);
        if (!prefetch_bounds) body += "            {\n";
        for (int Ib = I0; Ib < I1; Ib++) {
            if (prefetch_bounds)
                snprintf(line, sizeof(line), "              double ph_%d = 0.0; double wl_%d = w[%d]; asm volatile(\"\" : \"+v\"(wl_%d)); const double qp_%d = qv[QI_%d] + rhp_%d * "
                         "(wl_%d - 2.0 * fmin(fmax(wl_%d, lbp_%d), ubp_%d));\n", Ib, Ib, Ib, Ib, Ib, Ib, Ib, Ib, Ib, Ib, Ib);
            else if (unit[Ib])
                snprintf(line, sizeof(line), "              double ph_%d = na3v[BI_%d]; double wl_%d = w[%d]; asm volatile(\"\" : \"+v\"(wl_%d)); const double cp_%d = "
                         "fmin(fmax(wl_%d - kapv[KI_%d], 0.0), 1.0); const double qp_%d = __builtin_fma(-2.0, cp_%d, wl_%d);\n", Ib, Ib, Ib, Ib, Ib, Ib, Ib, Ib, Ib, Ib, Ib);
            else snprintf(line, sizeof(line), "              double ph_%d = 0.0; const double qp_%d = QHZP(%d);\n", Ib, Ib, Ib);
            body += line;
        }
        body += "              SEG;\n";
        if (prefetch_bounds && I1 < ZS) body += bound_loads(I1, std::min(ZS, I1 + SEG_EVERY), "p");
        for (int Ib = I0; Ib < I1; Ib++) {
            snprintf(a1, sizeof(a1), "ph_%d", Ib);
            prim_row(Ib, std::string(a1).c_str());
        }
        body += "              SEG;\n";
        for (int Ib = I0; Ib < I1; Ib++) {
            if (prefetch_bounds) snprintf(line, sizeof(line), "              ZUPD2(%d, ph_%d, lbp_%d, ubp_%d);\n", Ib, Ib, Ib, Ib);
            else if (unit[Ib]) snprintf(line, sizeof(line), "              ZUPDU(%d, ph_%d, cp_%d, wl_%d);\n", Ib, Ib, Ib, Ib);
            else snprintf(line, sizeof(line), "              LAUNDER; ZUPD(%d, ph_%d);\n", Ib, Ib);
            body += line;
        }
        body += prefetch_bounds ? "            SEG;\n" : "            }\n            SEG;\n";
    }
    // ---- terminal block (:318-386)
    if (!equ) body += "            SEG;\n            { double zN[TS_], vn[TS_], dd[TS_], pv[TS_], tt[TS_];\n";
    for (int k = 0; k < TS; k++) {
        snprintf(line, sizeof(line), "              zN[%d] = 0.0;\n", k);
        body += line;
        snprintf(a1, sizeof(a1), "zN[%d]", k);
        prim_row(ZS + k, a1);
    }
    if (equ) {
    } else if (lax) {
        body += "              _Pragma(\"unroll\") for (int k_ = 0; k_ < TS_; k_++) ZUPDT(k_, zN[k_]);\n";
        body += "              (void)vn; (void)dd; (void)pv; (void)tt; }\n";
    } else {
        body += "              _Pragma(\"unroll\") for (int k_ = 0; k_ < TS_; k_++) vn[k_] = zN[k_];\n";
        if (a.gen) dense_tail(a.Pinv_half, 1.0, "vn", "lamN", &a.rho_i_N);
        else dense_tail(a.Pinv_half, a.rho_i, "vn", "lamN");
        body += "              _Pragma(\"unroll\") for (int k_ = 0; k_ < TS_; k_++) { dd[k_] = vn[k_] - CE(k_); pv[k_] = 0.0; }\n";
        dense_tail(a.P, 1.0, "pv", "dd");
        body += "              EUPD_A;\n";
        dense_tail(a.P_half, 1.0, "lamN", "tt");
        body += "              EUPD_B; }\n";
    }
    if (pairs) {  // pair p at doubles [32 p, 32 p + 32): element e of block 2 p + h at 32 p + 2 e + h
        if ((tab.size() / 16) % 2) tab.resize(tab.size() + 16, 0.0);
        std::vector<double> t2(tab.size());
        for (size_t b = 0; b < tab.size() / 16; b++)
            for (int e = 0; e < 16; e++) t2[(b / 2) * 32 + 2 * e + (b % 2)] = tab[b * 16 + e];
        tab.swap(t2);
    }
    p.n_blocks = (int)(tab.size() / 16);
    p.n_mfma = n_mfma;
    if (pairs) {
        const int np = p.n_blocks / 2, n_pad = (np + PF - 1) / PF * PF;
        std::string out;
        out.reserve(body.size() + (size_t)np * 40);
        auto refill = [&](int t) {
            const int nx = (t + PF) % n_pad;
            if (nx >= np) return;
            snprintf(line, sizeof(line), "a%d = PBLK(blk%d, %d);", t % PF, nx / 256, nx % 256);
            out += line;
        };
        bool last_seen = false;
        for (size_t i = 0; i < body.size();) {
            if (body[i] == '@') {
                const size_t j = body.find('@', i + 1);
                const int t = atoi(body.substr(i + 1, j - i - 1).c_str());
                last_seen |= t == np - 1;
                refill(t);
                i = j + 1;
            } else {
                out.push_back(body[i++]);
            }
        }
        out += "            ";
        if (!last_seen) refill(np - 1);  // (an odd number of products: the last pair's second half is a pad)
        for (int t = np; t < n_pad; t++) refill(t);
        out += "\n";
        body.swap(out);
    } else
    {
        const int nb = p.n_blocks, n_pad = (nb + PF - 1) / PF * PF;
        std::string out;
        out.reserve(body.size() + (size_t)nb * 40);
        auto refill = [&](int t) {
            const int nx = (t + PF) % n_pad;
            if (nx >= nb) return;
            snprintf(line, sizeof(line), "a%d = BLK(blk%d, %d);", t % PF, nx / 512, nx % 512);
            out += line;
        };
        for (size_t i = 0; i < body.size();) {
            if (body[i] == '@') {
                const size_t j = body.find('@', i + 1);
                refill(atoi(body.substr(i + 1, j - i - 1).c_str()));
                i = j + 1;
            } else {
                out.push_back(body[i++]);
            }
        }
        out += "            ";
        for (int t = nb; t < n_pad; t++) refill(t);
        out += "\n";
        body.swap(out);
    }
    // ---- row tables behind the blocks: LB, UB of the z slabs (pads pinned to 0), c of the terminal slabs
    const int rc_lb = (int)tab.size();
    for (int r = 0; r < 4 * ZS; r++) tab.push_back(r < dz ? lb_row[r] : 0.0);
    for (int r = 0; r < 4 * ZS; r++) tab.push_back(r < dz ? ub_row[r] : 0.0);
    for (int r = 0; r < 4 * TS; r++) tab.push_back(r < n ? (lax ? lb_row[dz + r] : a.c_ell[r]) : 0.0);
    for (int r = 0; r < 4 * TS; r++) tab.push_back(r < n ? ub_row[dz + r] : 0.0);
    for (int r = 0; r < 4 * ZS; r++) tab.push_back(r < dz ? rho_row[r] : 1.0);
    for (int r = 0; r < 4 * TS; r++) tab.push_back(r < n ? rho_row[dz + r] : 1.0);
    for (double &x : tab) {
        if (x > 1e300) x = 1e300;  // (+-inf bounds)
        if (x < -1e300) x = -1e300;
        if (!std::isfinite(x)) { p.why = "non-finite block"; return 0; }
    }
    if (tab.size() * sizeof(double) > 160 * 1024 - 1024 || p.n_blocks > 1536) { p.why = "block table exceeds the LDS"; return 0; }
    if (p.n_blocks <= (pairs ? 2 * PF : PF)) {  // a very small controller: a shorter ring
        if (p.n_blocks >= 6 && pf_request != p.n_blocks / 2) return build_ellip(p, a, p.n_blocks / 2);
        p.why = "fewer blocks than the prefetch ring";
        return 0;
    }
    // ---- q: slabs with the same row pattern share one register
    std::map<std::vector<int>, int> sig_index;
    std::vector<int> qi(ZS), qrow;
    auto row_type = [&](int r) {
        if (r >= dz) return 0;
        if (r < m) return 1 + r;
        const int e = (r - m) % nm;
        return e < n ? 1000 + e : 1 + (e - n);
    };
    for (int J = 0; J < ZS; J++) {
        std::vector<int> sig = {row_type(4 * J), row_type(4 * J + 1), row_type(4 * J + 2), row_type(4 * J + 3)};
        auto it = sig_index.find(sig);
        if (it == sig_index.end()) {
            it = sig_index.emplace(sig, (int)qrow.size()).first;
            qrow.push_back(4 * J);
        }
        qi[J] = it->second;
    }
    std::string s;
    auto def = [&](const char *name, long v) { snprintf(line, sizeof(line), "#define %s %ld\n", name, v); s += line; };
    def("LAX_", lax ? 1 : 0);
    def("EQU_", equ ? 1 : 0);
    def("RHO_SCALAR_", a.gen ? 0 : 1);
    def("ZS_", ZS); def("TS_", TS); def("TSA_", std::max(TS, 1)); def("NR_", NR); def("NQ_", (long)qrow.size()); def("NBH_", bh_slabs);
    def("NBT_", std::max(NR - bt_first, 1)); def("BT_ROW0_", equ ? (long)(N - 1) * n : (long)4 * NR); def("BT_FIRST_", bt_first);
    def("TAB_DOUBLES_", (long)tab.size()); def("RC_", rc_lb); def("DIM_", dim); def("DZ_", dz); def("NN_", n);
    s += "#define RING_INIT";
    for (int i = 0; i < PF; i++) {
        if (pairs) snprintf(line, sizeof(line), " double2 a%d = PBLK(blk%d, %d);", i, (i % (p.n_blocks / 2)) / 256, (i % (p.n_blocks / 2)) % 256);
        else snprintf(line, sizeof(line), " double a%d = BLK(blk%d, %d);", i, (i % p.n_blocks) / 512, (i % p.n_blocks) % 512);
        s += line;
    }
    s += "\n";
    {
        const int nb = p.n_blocks;
        def("NB0_", std::max(1, std::min(nb, 512))); def("NB1_", std::max(1, std::min(nb - 512, 512))); def("NB2_", std::max(1, nb - 1024));
    }
    for (int J = 0; J < ZS; J++) { snprintf(line, sizeof(line), "#define QI_%d %d\n", J, qi[J]); s += line; }
    // bound rows: slabs with the same (LB, UB) pattern share a register pair (6-8 patterns with stage-invariant bounds); with
    // more than 24 patterns the rows are read from LDS at every use
    {
        const int lb0 = rc_lb, ub0 = rc_lb + 4 * ZS;
        std::map<std::vector<double>, int> pat;
        std::vector<int> bi(ZS), brow;
        for (int J = 0; J < ZS; J++) {
            std::vector<double> key;
            for (int r = 0; r < 4; r++) {
                key.push_back(tab[lb0 + 4 * J + r]);
                key.push_back(tab[ub0 + 4 * J + r]);
                key.push_back(tab[lb0 + 8 * ZS + 8 * TS + 4 * J + r]);  // rho of the row
            }
            auto it = pat.find(key);
            if (it == pat.end()) { it = pat.emplace(key, (int)brow.size()).first; brow.push_back(J); }
            bi[J] = it->second;
        }
        bool in_regs = true && brow.size() <= 24;  // (measured: registers win for the ellipMPC ADMM program, LDS reads for soc)
        if (const char *ev = getenv("SPCIES_BSP_BND_REGS")) in_regs = atoi(ev) != 0 && brow.size() <= 24;
        def("BND_IN_REGS_", in_regs ? 1 : 0);
        def("NBND_", (long)brow.size());
        for (int J = 0; J < ZS; J++) { snprintf(line, sizeof(line), "#define BI_%d %d\n", J, bi[J]); s += line; }
        s += "static __device__ const int BROW_[NBND_] = {";
        for (size_t i = 0; i < brow.size(); i++) { snprintf(line, sizeof(line), "%s%d", i ? ", " : "", brow[i]); s += line; }
        s += "};\n";
        // (unit-box slabs) one kappa register per (q pattern, bound pattern)
        if (!in_regs) std::fill(unit.begin(), unit.end(), 0);  // (cannot happen: the unit decision asked for <= 24 patterns)
        std::map<std::pair<int, int>, int> combo;
        std::vector<int> kq, kb, ki(ZS, -1);
        for (int J = 0; J < ZS; J++) {
            if (!unit[J]) continue;
            auto key = std::make_pair(qi[J], bi[J]);
            auto it = combo.find(key);
            if (it == combo.end()) { it = combo.emplace(key, (int)kq.size()).first; kq.push_back(qi[J]); kb.push_back(bi[J]); }
            ki[J] = it->second;
        }
        def("NKAP_", (long)kq.size());
        for (int J = 0; J < ZS; J++) { snprintf(line, sizeof(line), "#define KI_%d %d\n", J, ki[J] < 0 ? 0 : ki[J]); s += line; }
        if (!kq.empty()) {
            s += "static __device__ const int KAPQ_[NKAP_] = {";
            for (size_t i = 0; i < kq.size(); i++) { snprintf(line, sizeof(line), "%s%d", i ? ", " : "", kq[i]); s += line; }
            s += "};\nstatic __device__ const int KAPB_[NKAP_] = {";
            for (size_t i = 0; i < kb.size(); i++) { snprintf(line, sizeof(line), "%s%d", i ? ", " : "", kb[i]); s += line; }
            s += "};\n";
        }
        s += "static __device__ const int KIA_[ZS_] = {";
        for (int J = 0; J < ZS; J++) { snprintf(line, sizeof(line), "%s%d", J ? ", " : "", ki[J]); s += line; }
        s += "};\nstatic __device__ const int BIA_[ZS_] = {";
        for (int J = 0; J < ZS; J++) { snprintf(line, sizeof(line), "%s%d", J ? ", " : "", bi[J]); s += line; }
        s += "};\n";
    }
    s += "static __device__ const int QROW_[NQ_] = {";
    for (size_t i = 0; i < qrow.size(); i++) { snprintf(line, sizeof(line), "%s%d", i ? ", " : "", qrow[i]); s += line; }
    s += "};\n";
    s += R"SRC(
struct EArgs {
    int n, m, N, dim, k_max, ref_stride;
    double tol, rho, rho_i, r2, r;
    long B;
};
// cross-row reductions without the LDS crossbar (gfx950 v_permlane16_swap / v_permlane32_swap: a handful of vector instructions instead of
// ds_bpermute round trips whose latency nothing hides at one wavefront per SIMD; bit-identical sums, tools/probe_permlane_swap.hip)
__device__ __forceinline__ double xsum16_(double x) {  // x + x[lane ^ 16]
    const unsigned lo = __double2loint(x), hi = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double xsum32_(double x) {  // x + x[lane ^ 32]
    const unsigned lo = __double2loint(x), hi = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double row0_(double x) {  // x[lane % 16]: the first 16-lane row in every row
    const unsigned lo = __double2loint(x), hi = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const auto c = __builtin_amdgcn_permlane32_swap(a[0], a[0], false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(b[0], b[0], false, false);
    return __hiloint2double(d[0], c[0]);
}
template <bool WANT_SOL>
__device__ __forceinline__ void ellip_bsp_body(const EArgs &p, const double *__restrict__ table_g, const double *__restrict__ cst,
                                               const double *__restrict__ x0g, const double *__restrict__ xrg,
                                               const double *__restrict__ urg, double *__restrict__ u_out, int *__restrict__ k_out,
                                               int *__restrict__ e_out, double *__restrict__ f0, double *__restrict__ f1,
                                               double *__restrict__ f2) {
    __shared__ __attribute__((aligned(16))) double ldsr[3 * 4 * ZS_ + 3 * 4 * TS_];
    __shared__ __attribute__((aligned(16))) double blk0[NB0_ * 16];
    __shared__ __attribute__((aligned(16))) double blk1[NB1_ * 16];
    __shared__ __attribute__((aligned(16))) double blk2[NB2_ * 16];
    for (int i = threadIdx.x; i < NB0_ * 16; i += 256) blk0[i] = table_g[i];
    for (int i = threadIdx.x; i < NB1_ * 16; i += 256) blk1[i] = table_g[512 * 16 + i];
    for (int i = threadIdx.x; i < NB2_ * 16; i += 256) blk2[i] = table_g[1024 * 16 + i];
    for (int i = threadIdx.x; i < 3 * 4 * ZS_ + 3 * 4 * TS_; i += 256) ldsr[i] = table_g[RC_ + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int n = p.n, m = p.m, nm = n + m, dim = p.dim;
    double *dump = const_cast<double *>(table_g) + TAB_DOUBLES_;
    const double rho = p.rho, tol = p.tol;
    const double *cA = cst, *cQ = cA + n * n, *cR = cQ + n * n, *cT = cR + m * m;  // A, diag(Q), diag(R) (negated weights), T
    int ao = g * 4 + (lane & 3);
    const long n_tiles = (p.B + 15) / 16;
#define BLK(arr, t) arr[(t) * 16 + ao]
#define PBLK(arr, pp) (*reinterpret_cast<const double2 *>(&arr[(pp) * 32 + 2 * ao]))
#define MF(acc, a, x) acc = __builtin_amdgcn_mfma_f64_4x4x4f64((a), (x), (acc), 0, 0, 0)
#define SEG __builtin_amdgcn_sched_barrier(0)
#define LAUNDER asm volatile("" : "+v"(go))
    for (long tile = (long)blockIdx.x * 4 + wave; tile < n_tiles; tile += (long)gridDim.x * 4) {
        const long inst = tile * 16 + c;
        const bool valid = inst < p.B;
        const long ii = valid ? inst : 0;
        const double *x0 = x0g + ii * n;
        const double *xr = p.ref_stride ? xrg + ii * n : xrg;
        const double *ur = p.ref_stride ? urg + ii * m : urg;
        // (gs: the lane's row group laundered per group of instances - the set-up's row addresses into A, Q, R, T depend on the lane only and
        // would otherwise be formed once in front of this loop and parked in scratch memory, soc_bsp.hpp)
        int gs = g;
        asm volatile("" : "+v"(gs));
        double bh[NBH_];
#pragma unroll
        for (int I = 0; I < NBH_; I++) {
            const int row = 4 * I + gs;
            double v = 0.0;
            if (row < n)
                for (int i = 0; i < n; i++) v -= cA[row * n + i] * x0[i];
            bh[I] = v;
        }
#if EQU_
        double bt[NBT_];  // equMPC: b = xr in the last n rows
#pragma unroll
        for (int I = 0; I < NBT_; I++) {
            const int row = 4 * (BT_FIRST_ + I) + gs - BT_ROW0_;
            bt[I] = (row >= 0 && row < n) ? xr[row] : 0.0;
        }
#endif
        double qv[NQ_], qTv[TSA_];
#pragma unroll
        for (int u = 0; u < NQ_; u++) {
            const int j = QROW_[u] + gs;
            double v = 0.0;
            if (j < m) {
                for (int i = 0; i < m; i++) v += cR[j * m + i] * ur[i];
            } else if (j < DZ_) {
                const int e = (j - m) % nm;
                if (e < n) {
                    for (int i = 0; i < n; i++) v += cQ[e * n + i] * xr[i];
                } else {
                    for (int i = 0; i < m; i++) v += cR[(e - n) * m + i] * ur[i];
                }
            }
            qv[u] = v;
        }
#pragma unroll
        for (int k = 0; k < TS_; k++) {
            const int e = 4 * k + gs;
            double v = 0.0;
            if (e < n)
                for (int i = 0; i < n; i++) v += cT[e * n + i] * xr[i];
            qTv[k] = v;
        }
        // state: w = v + lambda / rho per z slab (v = clamp(w), lambda = rho (w - v)); the terminal slabs keep v_N, lambda_N
        double w[ZS_], vN[TSA_], lamN[TSA_], wN[TSA_], rh[NR_];  // (lax: wN; ellip: vN, lamN - the unused ones fold away)
#pragma unroll
        for (int I = 0; I < ZS_; I++) w[I] = 0.0;
#pragma unroll
        for (int I = 0; I < TS_; I++) { vN[I] = 0.0; lamN[I] = 0.0; wN[I] = 0.0; }
        int go = g;
#define LBL(I) ldsr[4 * (I) + go]
#define UBL(I) ldsr[4 * ZS_ + 4 * (I) + go]
#if RHO_SCALAR_
#define RHOL(I) rho
#else
#define RHOL(I) ldsr[8 * ZS_ + 8 * TS_ + 4 * (I) + go]
#endif
#if BND_IN_REGS_
        double lbv[NBND_], ubv[NBND_];
#pragma unroll
        for (int u = 0; u < NBND_; u++) {
            lbv[u] = ldsr[4 * BROW_[u] + g];
            ubv[u] = ldsr[4 * ZS_ + 4 * BROW_[u] + g];
        }
#define LBR(I) lbv[BI_##I]
#define UBR(I) ubv[BI_##I]
#if RHO_SCALAR_
#define RHOR(I) rho
#else
        double rhov[NBND_];
#pragma unroll
        for (int u = 0; u < NBND_; u++) rhov[u] = ldsr[8 * ZS_ + 8 * TS_ + 4 * BROW_[u] + g];
#define RHOR(I) rhov[BI_##I]
#endif
#else
#define LBR(I) LBL(I)
#define UBR(I) UBL(I)
#define RHOR(I) RHOL(I)
#endif
#if NKAP_ > 0  // unit-box slabs (scalar rho, patterns in registers): kappa, -lb / D, D; w = 0 is w^ = kappa - lb / D
        double kapv[NKAP_], na3v[NBND_], dv[NBND_];
#pragma unroll
        for (int u = 0; u < NBND_; u++) {
            const double D_ = ubv[u] - lbv[u];
            dv[u] = D_;
            na3v[u] = (D_ > 0.0 && D_ < 1e6) ? -lbv[u] / D_ : 0.0;
        }
#pragma unroll
        for (int u = 0; u < NKAP_; u++) kapv[u] = qv[KAPQ_[u]] / (rho * (ubv[KAPB_[u]] - lbv[KAPB_[u]])) + na3v[KAPB_[u]];
#pragma unroll
        for (int I = 0; I < ZS_; I++)
            if (KIA_[I] >= 0) w[I] = kapv[KIA_[I]] + na3v[BIA_[I]];
#define QHZU(J) ({ const double c_ = fmin(fmax(w[J] - kapv[KI_##J], 0.0), 1.0); __builtin_fma(-2.0, c_, w[J]); })
        // v (= clamp(w)) and w in the caller's coordinates from the state (exit, record: cold)
#define VOF_(I) (KIA_[I] >= 0 ? __builtin_fma(UBL(I) - LBL(I), fmin(fmax(w[I] - kapv[KIA_[I] >= 0 ? KIA_[I] : 0], 0.0), 1.0), LBL(I)) : fmin(fmax(w[I], LBL(I)), UBL(I)))
#define WOF_(I) (KIA_[I] >= 0 ? __builtin_fma(UBL(I) - LBL(I), w[I] - kapv[KIA_[I] >= 0 ? KIA_[I] : 0], LBL(I)) : w[I])
#else
#define VOF_(I) fmin(fmax(w[I], LBL(I)), UBL(I))
#define WOF_(I) w[I]
#endif
#define CE(k) ldsr[8 * ZS_ + 4 * (k) + go]
#define LBT(k) ldsr[8 * ZS_ + 4 * (k) + go]
#define UBT(k) ldsr[8 * ZS_ + 4 * TS_ + 4 * (k) + go]
#if RHO_SCALAR_
#define RHOT(k) rho
#else
#define RHOT(k) ldsr[12 * ZS_ + 8 * TS_ + 4 * (k) + go]
#endif
#define QHZ(J) (qv[QI_##J] + RHOR(J) * (w[J] - 2.0 * fmin(fmax(w[J], LBR(J)), UBR(J))))
#define QHZP(J) ({ double wl_ = w[J]; asm volatile("" : "+v"(wl_)); qv[QI_##J] + RHOR(J) * (wl_ - 2.0 * fmin(fmax(wl_, LBR(J)), UBR(J))); })
        bool active = valid, res = false;
        int kk = 0;
        RING_INIT
        // z slab I: v = clamp(z + lambda / rho), lambda += rho (z - v), residuals (:490-568)
#define ZUPD(I, zh) ZUPD2(I, zh, LBR(I), UBR(I))
#define ZUPD2(I, zh, lbx_, ubx_)                                                                 \
    do {                                                                                         \
        const double lb_ = (lbx_), ub_ = (ubx_);                                                 \
        const double wo_ = w[I], vo_ = fmin(fmax(wo_, lb_), ub_);                                \
        const double wn_ = (zh) + (wo_ - vo_), v_ = fmin(fmax(wn_, lb_), ub_);                   \
        w[I] = wn_;                                                                              \
        res |= (fabs(vo_ - v_) > tol) | (fabs((zh) - v_) > tol);                                 \
        if (WANT_SOL) *((4 * (I) + 3 < DZ_ || 4 * (I) + g < DZ_) ? zp + 4 * (I) : dump) = (zh); \
    } while (0)
        // the same in unit-box coordinates: zh is (z_hat - lb) / D, cp the clamp of the old state, wl the old state; residuals x D
#define ZUPDU(I, zh, cp, wl)                                                                     \
    do {                                                                                         \
        const double wn_ = (zh) + ((wl) - (cp)), v_ = fmin(fmax(wn_ - kapv[KI_##I], 0.0), 1.0), D_ = dv[BI_##I]; \
        w[I] = wn_;                                                                              \
        res |= (fabs((cp) - v_) * D_ > tol) | (fabs((zh) - v_) * D_ > tol);                      \
        if (WANT_SOL) *(zp + 4 * (I)) = __builtin_fma(D_, (zh), lbv[BI_##I]);                    \
    } while (0)
        // lax: the terminal slabs are boxes like the others
#define ZUPDT(k, zh)                                                                             \
    do {                                                                                         \
        const double lb_ = LBT(k), ub_ = UBT(k);                                                 \
        const double wo_ = wN[k], vo_ = fmin(fmax(wo_, lb_), ub_);                               \
        const double wn_ = (zh) + (wo_ - vo_), v_ = fmin(fmax(wn_, lb_), ub_);                   \
        wN[k] = wn_;                                                                             \
        res |= (fabs(vo_ - v_) > tol) | (fabs((zh) - v_) > tol);                                 \
        if (WANT_SOL) *((4 * (k) + 3 < NN_ || 4 * (k) + g < NN_) ? zp + DZ_ + 4 * (k) : dump) = (zh); \
    } while (0)
        // terminal block: the P-projection onto the ellipsoid (:318-352) ...
#define EUPD_A                                                                                   \
    do {                                                                                         \
        double vpv_ = 0.0;                                                                       \
        _Pragma("unroll") for (int k_ = 0; k_ < TS_; k_++) vpv_ += dd[k_] * pv[k_];              \
        vpv_ = xsum16_(vpv_);                                                                          \
        vpv_ = xsum32_(vpv_);                                                                          \
        if (vpv_ > p.r2) {                                                                       \
            const double sc_ = p.r / sqrt(vpv_);                                                 \
            _Pragma("unroll") for (int k_ = 0; k_ < TS_; k_++) vn[k_] = sc_ * dd[k_] + CE(k_);   \
        }                                                                                        \
        _Pragma("unroll") for (int k_ = 0; k_ < TS_; k_++) tt[k_] = RHOT(k_) * (zN[k_] - vn[k_]); \
    } while (0)
        // ... and, after lambda_N += P_half tt, the residuals and the new v_N (:374-386)
#define EUPD_B                                                                                   \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < TS_; k_++) {                                     \
            res |= (fabs(vN[k_] - vn[k_]) > tol) | (fabs(zN[k_] - vn[k_]) > tol);                \
            vN[k_] = vn[k_];                                                                     \
            if (WANT_SOL) *((4 * k_ + 3 < NN_ || 4 * k_ + g < NN_) ? zp + DZ_ + 4 * k_ : dump) = zN[k_]; \
        }                                                                                        \
    } while (0)
        while (true) {
            kk += 1;
            res = false;
            asm volatile("" : "+v"(ao), "+v"(go));
            double *zp = (WANT_SOL && active) ? f0 + inst * dim + g : dump;
)SRC";
    s += body;
    s += R"SRC(
            SEG;
            unsigned long long bal = __ballot(res);
            bal |= bal >> 32;
            bal |= bal >> 16;
            const bool res_inst = (bal >> c) & 1ull;
            const bool done_now = active && (!res_inst || kk >= p.k_max);
            if (__any(done_now)) {
                if (done_now) {
                    if (g == 0) {
                        k_out[inst] = kk;
                        e_out[inst] = res_inst ? -1 : 1;
                    }
                    if (g < m) u_out[inst * m + g] = VOF_(0);  // u = v_0
                    if (WANT_SOL) {
                        double *vp = f1 + inst * dim + g, *lp = f2 + inst * dim + g;
#pragma unroll
                        for (int I = 0; I < ZS_; I++) {
                            const bool in_ = 4 * I + 3 < DZ_ || 4 * I + g < DZ_;
                            const double v_ = VOF_(I);
                            *(in_ ? vp + 4 * I : dump) = v_;
                            *(in_ ? lp + 4 * I : dump) = RHOL(I) * (WOF_(I) - v_);
                        }
#pragma unroll
                        for (int k = 0; k < TS_; k++) {
                            const bool in_ = 4 * k + 3 < NN_ || 4 * k + g < NN_;
#if LAX_
                            const double vt_ = fmin(fmax(wN[k], LBT(k)), UBT(k));
                            *(in_ ? vp + DZ_ + 4 * k : dump) = vt_;
                            *(in_ ? lp + DZ_ + 4 * k : dump) = RHOT(k) * (wN[k] - vt_);
#else
                            *(in_ ? vp + DZ_ + 4 * k : dump) = vN[k];
                            *(in_ ? lp + DZ_ + 4 * k : dump) = lamN[k];
#endif
                        }
                    }
                    active = false;
                }
            }
            if (!__any(active)) break;
        }
    }
}
extern "C" __global__ __launch_bounds__(256, 1) void ellip_bsp_kernel(EArgs p, const double *table_g, const double *cst, const double *x0g,
                                                                     const double *xrg, const double *urg, double *u_out, int *k_out,
                                                                     int *e_out) {
    ellip_bsp_body<false>(p, table_g, cst, x0g, xrg, urg, u_out, k_out, e_out, nullptr, nullptr, nullptr);
}
extern "C" __global__ __launch_bounds__(256, 1) void ellip_bsp_kernel_sol(EArgs p, const double *table_g, const double *cst,
                                                                         const double *x0g, const double *xrg, const double *urg,
                                                                         double *u_out, int *k_out, int *e_out, double *f0, double *f1,
                                                                         double *f2) {
    ellip_bsp_body<true>(p, table_g, cst, x0g, xrg, urg, u_out, k_out, e_out, f0, f1, f2);
}
)SRC";
    p.src = s;
    if (const char *path = getenv("SPCIES_BSP_DUMP")) {
        if (FILE *f = fopen(path, "w")) {
            fputs(s.c_str(), f);
            fclose(f);
        }
    }
    p.why = "not compiled yet";
    return 0;
}

inline int finish_ellip(Plan &p, const AdmmHost &a) {
    if (p.src.empty() || p.ok) return 0;
    if (p.d_table) hipFree(p.d_table);
    if (p.d_consts) hipFree(p.d_consts);
    p.d_table = p.d_consts = nullptr;
    int scratch = 0;
    int rc = compile_program(p, &scratch, "ellip_bsp_kernel", "ellip_bsp_kernel_sol");
    if (rc) return rc;
    if (!getenv("SPCIES_BSP_PF"))
        for (int pf : {12, 8, 4}) {  // (a compiler that spills far more than the one this was tuned with gets a shorter ring)
            if (scratch <= 640) break;
            rc = build_ellip(p, a, pf);
            if (rc) return rc;
            if (p.src.empty()) return fail(SPCIES_HIP_ENOSUP, "BSP program: %s", p.why.c_str());
            rc = compile_program(p, &scratch, "ellip_bsp_kernel", "ellip_bsp_kernel_sol");
            if (rc) return rc;
        }
    if (getenv("SPCIES_BSP_VERBOSE"))
        fprintf(stderr, "[spcies bsp] ellipMPC ADMM: %d blocks, %d MFMAs per iteration, table %zu B, scratch %d B per lane\n", p.n_blocks,
                p.n_mfma, p.table.size() * sizeof(double), scratch);
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_table, (p.table.size() + 4 * (size_t)(p.ZS + p.SS) + 8) * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_table, p.table.data(), p.table.size() * sizeof(double), hipMemcpyHostToDevice));
    const int n = a.n, m = a.m, nm = n + m;
    std::vector<double> cst((size_t)n * n * 2 + (size_t)m * m + (size_t)n * n, 0.0);
    double *cA = cst.data(), *cQ = cA + n * n, *cR = cQ + n * n, *cT = cR + m * m;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) {
            cA[i * n + j] = a.AB[(size_t)i * nm + j];
            cT[i * n + j] = a.terminal ? a.T[(size_t)i * n + j] : 0.0;
        }
        cQ[i * n + i] = a.Q[i];
    }
    for (int j = 0; j < m; j++) cR[j * m + j] = a.R[j];
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_consts, cst.size() * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_consts, cst.data(), cst.size() * sizeof(double), hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    int dev = 0;
    SPCIES_HIP_CHECK(hipGetDevice(&dev));
    SPCIES_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    p.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    p.ok = true;
    p.why.clear();
    return 0;
}

// z, v, lambda: all or none
inline int launch_ellip(Plan &p, const AdmmHost &a, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u,
                        int *k, int *e, double *z, double *v, double *lam, hipStream_t st) {
    if (!p.ok) return fail(SPCIES_HIP_ENOSUP, "BSP variant not available: %s", p.why.c_str());
    const bool any = z || v || lam;
    if (any && !(z && v && lam)) return fail(SPCIES_HIP_EINVAL, "BSP variant: pass all of z, v, lambda or none");
    EArgs ar{a.n, a.m, a.N, a.N * (a.n + a.m) - (a.terminal ? 0 : a.n), a.k_max, ref_stride, a.tol, a.rho, a.rho_i, a.r_ell * a.r_ell, a.r_ell, B};
    const long n_tiles = (B + 15) / 16;
    long wgs = (n_tiles + 3) / 4;
    if (wgs > p.num_cu) wgs = p.num_cu;
    const double *table = p.d_table, *cst = p.d_consts;
    void *params[] = {&ar, &table, &cst, &x0, &xr, &ur, &u, &k, &e, &z, &v, &lam};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel(p.fn[any ? 1 : 0], (unsigned)wgs, 1, 1, 256, 1, 1, 0, st, params, nullptr));
    return 0;
}

}  // namespace bsp
}  // namespace spcies
