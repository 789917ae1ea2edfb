// Variant BSP ("block-sparse program") of the ellipMPC ADMM soc solver (code_ellipMPC_ADMM_soc_C.c:84-296): the sparse
// KKT machinery on the FP64 matrix pipe.  The controller's sparse matrices - the L D L' factor of W, -Gh Hh^-1,
// -Hh^-1 Gh', -Hh^-1 - are block-sparse with dense 4x4 blocks (banded KKT structure), so one ADMM iteration is a FIXED
// sequence of v_mfma_f64_4x4x4 block products on 16 instances per wavefront (the MFMA4 lane layout, admm_mfma4.hpp:
// a register holds rows 4s+g of slab s for instance c = lane % 16, g = lane / 16; D layout == B layout).  Spcies is a
// code generator - its C platform prints one solver per controller - and so is this variant: the host emits the program
// as straight-line HIP (every register index and table offset a literal), hiprtc compiles it at create time, the blocks
// sit in LDS in issue order, the solver state (z, s, lambda, mu, the right-hand side) lives in registers for the whole
// solve.  C5 soc: 1 012 MFMAs per iteration for 16 instances, against TILE's ~900 dependent LDS steps for 2.
//
// Internal row layout: z in slabs 0 .. ZS-1 (dim rows, zero padded), s in the next SS slabs (so no slab mixes z and s
// rows); the rows of the right-hand side in their natural order.  Triangular solves by blocks with the diagonal blocks
// inverted on the host:  x_I = Linv_II rhs_I - sum_{J<I} (Linv_II L_IJ) x_J ;  y_I = (Uinv_II Dinv_I) x_I - sum_{J>I}
// (Uinv_II U_IJ) y_J  (U = L').  Sums run in another order than the reference's loops: parity 1e-10, not bit-exact.
#pragma once
#include <algorithm>
#include <cstdarg>
#include <map>
#include <numeric>

#include "bsp_sched.hpp"
#include "mfma4_rtc.hpp"
#include "soc_stream.hpp"

namespace spcies {
namespace bsp {

struct Args {  // mirrored in the generated source
    int n, m, N, dim, n_s, n_eq, k_max, ref_stride, r_stride, pad;
    double tol_p, tol_d, rho, rho_i, sigma, sigma_i;
    long B;
};

struct Plan {
    bool ok = false;
    std::string why = "not built";
    bool build_failed = false;  // the variant applies to this controller but its run-time specialisation failed (hiprtc missing, compile error): what SPCIES_HIP_STRICT reacts to
    std::string src;             // generated kernel source
    std::vector<double> table;   // blocks in issue order, then LB / UB rows (internal layout)
    int n_blocks = 0, ZS = 0, SS = 0, NR = 0, n_mfma = 0;
    bool legacy_order = false;   // ask build_soc for the round-2 form (finish_soc: the scheduled program spilled in this compiler)
    bool scheduled = false;      // the iteration was ordered by bsp_sched.hpp: compiled with LLVM's machine scheduler off
    double *d_table = nullptr, *d_consts = nullptr;  // d_consts: A | Q | R | T | PhiP (dense, for the per-instance setup)
    hipModule_t module = nullptr;
    hipFunction_t fn[2] = {nullptr, nullptr};
    Args args{};
    int num_cu = 256;
};
inline void plan_free(Plan &p) {
    if (p.module) rtc::unload_module(p.module);
    if (p.d_table) hipFree(p.d_table);
    if (p.d_consts) hipFree(p.d_consts);
    p.module = nullptr;
    p.d_table = p.d_consts = nullptr;
    p.ok = false;
}

typedef std::vector<double> Dense;  // row-major

struct BlockList {  // non-zero 4x4 blocks of a matrix, row-wise and column-wise
    int R = 0, C = 0;
    std::vector<std::vector<int>> by_row, by_col;
};
inline BlockList blocks_of(const Dense &M, int rows, int cols) {
    BlockList b;
    b.R = rows / 4;
    b.C = cols / 4;
    b.by_row.resize(b.R);
    b.by_col.resize(b.C);
    for (int I = 0; I < b.R; I++)
        for (int J = 0; J < b.C; J++) {
            bool nz = false;
            for (int i = 0; i < 4 && !nz; i++)
                for (int k = 0; k < 4; k++)
                    if (M[(size_t)(4 * I + i) * cols + 4 * J + k] != 0.0) { nz = true; break; }
            if (nz) {
                b.by_row[I].push_back(J);
                b.by_col[J].push_back(I);
            }
        }
    return b;
}

// Builds the program (source + table) of one soc controller.  Host only: no device call.
inline int build_soc(Plan &p, const SocDev &c, const double *F, const int *I, int pf_request = 0) {
    const int n = c.n, m = c.m, nm = n + m, N = c.N, dim = c.dim, n_s = c.n_s, n_eq = c.n_eq;
    const int ZS = (dim + 3) / 4, SS = (n_s + 3) / 4, NP = ZS + SS, nr = n_eq + n_s, NR = (nr + 3) / 4;
    const int PR_ = 4 * NP, RR = 4 * NR;
    auto ip = [&](int j) { return j < dim ? j : 4 * ZS + (j - dim); };  // internal row of primal row j
    p.ok = false;
    p.ZS = ZS; p.SS = SS; p.NR = NR;
    // ---- dense forms in the internal layout
    Dense G((size_t)RR * PR_, 0.0), HG((size_t)PR_ * RR, 0.0), H((size_t)PR_ * PR_, 0.0), L((size_t)RR * RR, 0.0), Dinv(RR, 1.0);
    {
        const double *Gv = F + c.GhHhi_val, *HGv = F + c.HhiGh_val, *Hv = F + c.Hhi_val, *Lv = F + c.L_val, *Dv = F + c.Dinv;
        const int *Gc = I + c.GhHhi_col, *Gr = I + c.GhHhi_row, *HGc = I + c.HhiGh_col, *HGr = I + c.HhiGh_row, *Hc = I + c.Hhi_col,
                  *Hr = I + c.Hhi_row, *Lc = I + c.L_col, *Lr = I + c.L_row;
        for (int i = 0; i < nr; i++)
            for (int j = Gr[i]; j < Gr[i + 1]; j++) G[(size_t)i * PR_ + ip(Gc[j])] = Gv[j];
        for (int i = 0; i < dim + n_s; i++) {
            for (int j = HGr[i]; j < HGr[i + 1]; j++) HG[(size_t)ip(i) * RR + HGc[j]] = HGv[j];
            for (int j = Hr[i]; j < Hr[i + 1]; j++) H[(size_t)ip(i) * PR_ + ip(Hc[j])] = Hv[j];
        }
        for (int i = 0; i < RR; i++) L[(size_t)i * RR + i] = 1.0;
        for (int j = 0; j < nr; j++) {
            for (int q = Lc[j]; q < Lc[j + 1]; q++) L[(size_t)Lr[q] * RR + j] = Lv[q];
            Dinv[j] = Dv[j];
        }
    }
    // ---- table + program text
    // scheduling fences every SEG_EVERY block rows / columns: without them the compiler hoists the LDS reads of whole phases
    // ring depth: measured at C5 with the split tail (ms, scratch B per lane): 4: 11.51 / 236, 8: 11.13 / 300, 12: 10.71 / 356, 16: 10.36 / 412,
    // 20: 10.12 / 504, 24: 10.49 / 540, 32: 10.60 / 684 - the deep ring pays for the spills it causes (they sit in the cold copy of the tail)
    int SEG_EVERY = 4, PF = 20;
    const bool use_sched = !p.legacy_order && !(getenv("SPCIES_BSP_SCHED") && getenv("SPCIES_BSP_SCHED")[0] == '0');
    // scheduled program: the ring holds block PAIRS read by ds_read_b128 (PF counts pairs; SPCIES_BSP_PAIRS=0: single blocks) - a
    // ds_read_b64 per block keeps the LDS pipe busier than the matrix pipe (measured at C5: 9.18 -> 8.42 ms)
    const bool pairs = use_sched && !(getenv("SPCIES_BSP_PAIRS") && getenv("SPCIES_BSP_PAIRS")[0] == '0');
    if (use_sched) PF = pairs ? 6 : 12;  // (measured at C5: pairs 4: 7.95 ms, 6: 7.8, 8: 8.1; single blocks 12: 9.14-9.24, 20: 9.29-9.8, 32: 11.1)
    if (const char *ev = getenv("SPCIES_BSP_PF")) PF = std::min(64, std::max(2, atoi(ev)));
    if (pf_request > 0) PF = pf_request;
    p.src.clear();
    if (const char *ev = getenv("SPCIES_BSP_SEG")) SEG_EVERY = std::max(1, atoi(ev));
    std::vector<double> &tab = p.table;
    tab.clear();
    std::string body;
    char line[512];
    int n_mfma = 0;
    auto emit_block = [&](const double *blk /* [i][k] row-major 4x4 */) {
        const int t = (int)(tab.size() / 16);
        for (int k = 0; k < 4; k++)
            for (int i = 0; i < 4; i++) tab.push_back(blk[i * 4 + k]);  // element (i, k) at k*4 + i  (A operand: i = lane%4, k = lane/16)
        return t;
    };
    auto block_of = [&](const Dense &M, int cols, int Ib, int Jb, double *out) {
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 4; k++) out[i * 4 + k] = M[(size_t)(4 * Ib + i) * cols + 4 * Jb + k];
    };
    auto mul44 = [&](const double *a, const double *b, double *o, double sign) {
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 4; k++) {
                double s = 0.0;
                for (int q = 0; q < 4; q++) s += a[i * 4 + q] * b[q * 4 + k];
                o[i * 4 + k] = sign * s;
            }
    };
    auto inv_unit_lower = [&](const double *a, double *o) {  // a = unit lower 4x4; forward substitution on the identity
        for (int col = 0; col < 4; col++)
            for (int i = 0; i < 4; i++) {
                double s = (i == col) ? 1.0 : 0.0;
                for (int q = 0; q < i; q++) s -= a[i * 4 + q] * o[q * 4 + col];
                o[i * 4 + col] = s;
            }
    };
    auto MF = [&](const char *acc, int t, const char *x) {
        // the A operand comes from a ring of PF named registers refilled PF blocks ahead of their use (the LDS round trip is
        // ~8 MFMAs long and the compiler does not hoist these reads by itself); @t@ is resolved once the block count is known
        snprintf(line, sizeof(line), "            MF(%s, a%d, %s); @%d@\n", acc, t % PF, x, t);
        body += line;
        n_mfma++;
    };
    char a1[64], a2[64];
    // q_hat of slab J from the current state (z slabs: q + lambda - sigma z; s slabs: mu - rho s)
    auto qhat_expr = [&](int J, char *out, size_t cap) {
        if (J < ZS) snprintf(out, cap, "QHZ(%d)", J);
        else snprintf(out, cap, "(mu[%d] - rho * sc[%d])", J - ZS, J - ZS);
    };
    // bound rows of the z slabs (rows past dim - n - 1 are free; pads are pinned to 0 by 0 <= z <= 0) and their distinct patterns: with
    // stage-invariant bounds a handful of (LB, UB) slab patterns repeat along the horizon - the scheduled program keeps them in registers
    // (the LDS pipe is the busiest unit of the iteration: every product's A operand comes through it)
    std::vector<double> lb_rows(4 * ZS), ub_rows(4 * ZS);
    for (int r = 0; r < 4 * ZS; r++) {
        lb_rows[r] = r < dim - n - 1 ? F[c.LB + r] : (r < dim ? -1e300 : 0.0);
        ub_rows[r] = r < dim - n - 1 ? F[c.UB + r] : (r < dim ? 1e300 : 0.0);
    }
    std::vector<int> bnd_of(ZS), bnd_slab;
    {
        std::map<std::vector<double>, int> pat;
        for (int J = 0; J < ZS; J++) {
            std::vector<double> key(lb_rows.begin() + 4 * J, lb_rows.begin() + 4 * J + 4);
            key.insert(key.end(), ub_rows.begin() + 4 * J, ub_rows.begin() + 4 * J + 4);
            auto it = pat.find(key);
            if (it == pat.end()) { it = pat.emplace(key, (int)bnd_slab.size()).first; bnd_slab.push_back(J); }
            bnd_of[J] = it->second;
        }
    }
    // ---- q: slabs with the same row pattern share one register
    std::map<std::vector<int>, int> sig_index;
    std::vector<int> qi(ZS), qrow;
    auto row_type = [&](int r) {  // 0: zero; 1 + j: (R ur)_j; 1000 + j: (Q xr)_j; 2000 + j: (T xr)_j
        if (r >= dim) return 0;
        if (r < m) return 1 + r;
        if (r < m + (N - 1) * nm) { const int e = (r - m) % nm; return e < n ? 1000 + e : 1 + (e - n); }
        if (r < m + (N - 1) * nm + n) return 2000 + (r - m - (N - 1) * nm);
        return 0;
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
    int bnd_regs_max = 16;
    if (const char *ev = getenv("SPCIES_BSP_BND_REGS")) bnd_regs_max = atoi(ev);
    const bool bnd_in_regs = use_sched && (int)bnd_slab.size() <= bnd_regs_max;
    // ---- unit-box coordinates for the z slabs whose four rows have a finite box around 0 (admm_mfma4u.hpp's algebra, round 5): per row
    // D = ub - lb, w' = (w - lb) / D, state w^ = w' + kappa, kappa = q / (sigma D) - lb / D:  c' = clamp01(w^ - kappa) (the hardware's [0, 1]
    // output modifier), s = w^ - 2 c', q_hat = sigma D s - sigma D folded into the columns of G and H, 1 / D into the rows of H and HG, -lb / D
    // seeds the row's accumulator - and w^+ = z_hat' + (w^ - c'): 6 vector instructions per slab and iteration instead of 10.  The other z
    // slabs (free terminal rows, pads, mixed slabs) and the cone slabs keep the plain form.  Scheduled programs only; SPCIES_BSP_UNIT=0: off.
    std::vector<char> unit(ZS, 0);
    const bool unit_on = !p.legacy_order && !(getenv("SPCIES_BSP_SCHED") && getenv("SPCIES_BSP_SCHED")[0] == '0') &&
                         !(getenv("SPCIES_BSP_UNIT") && getenv("SPCIES_BSP_UNIT")[0] == '0');
    int n_unit = 0;
    for (int J = 0; J < ZS && unit_on && bnd_in_regs; J++) {
        bool ok = true;
        for (int r = 4 * J; r < 4 * J + 4; r++) {
            if (r >= dim - n - 1) { ok = false; break; }
            const double lo = F[c.LB + r], hi = F[c.UB + r];
            if (!(std::isfinite(lo) && std::isfinite(hi)) || !(lo <= 0.0 && hi >= 0.0) || !(hi - lo > 1e-9) || hi - lo > 1e5) { ok = false; break; }
        }
        unit[J] = ok;
        n_unit += ok;
    }
    for (int J = 0; J < ZS; J++)
        if (unit[J])
            for (int r = 4 * J; r < 4 * J + 4; r++) {
                const double D = F[c.UB + r] - F[c.LB + r], sD = c.sigma * D;
                for (int i = 0; i < RR; i++) G[(size_t)i * PR_ + r] *= sD;
                for (int i = 0; i < PR_; i++) H[(size_t)i * PR_ + r] *= sD;
            }
    for (int J = 0; J < ZS; J++)
        if (unit[J])
            for (int r = 4 * J; r < 4 * J + 4; r++) {
                const double D = F[c.UB + r] - F[c.LB + r];
                for (int j = 0; j < PR_; j++) H[(size_t)r * PR_ + j] /= D;
                for (int j = 0; j < RR; j++) HG[(size_t)r * RR + j] /= D;
            }
    const BlockList bG = blocks_of(G, RR, PR_), bHG = blocks_of(HG, PR_, RR), bH = blocks_of(H, PR_, PR_), bL = blocks_of(L, RR, RR);
    // (unit-box) kappa is per instance and per (q pattern, bound pattern) of a slab: one register per distinct combination
    std::vector<int> kap_of(ZS, -1), kap_q, kap_b;
    {
        std::map<std::pair<int, int>, int> combo;
        for (int J = 0; J < ZS; J++) {
            if (!unit[J]) continue;
            auto key = std::make_pair(qi[J], bnd_of[J]);
            auto it = combo.find(key);
            if (it == combo.end()) { it = combo.emplace(key, (int)kap_q.size()).first; kap_q.push_back(qi[J]); kap_b.push_back(bnd_of[J]); }
            kap_of[J] = it->second;
        }
    }
    std::vector<double> kreg_blocks;  // (scheduled program) the register-resident blocks, 16 doubles each in A-operand order
    std::vector<int> bh_slabs;  // slabs of the right-hand side that hold a row of bh: -A x0, r, -PhiP xr (:97-131)
    std::map<int, int> saved;   // (row-order program) slab -> index into sv[]
    const bool early_bounds = !getenv("SPCIES_BSP_LATE_BOUNDS");
    if (SS > 8) { p.why = "more than 32 cone rows"; return 0; }
    // The iteration as a list of micro-operations ordered by bsp_sched.hpp (SPCIES_BSP_SCHED=0: the program in the generator's
    // own order, left to LLVM's scheduler - the round-2 form)
    p.scheduled = use_sched;
    if (use_sched) {
        sched::Program P;
        auto F_ = [](const char *fmt, ...) -> std::string {
            char buf[1024];
            va_list ap;
            va_start(ap, fmt);
            vsnprintf(buf, sizeof(buf), fmt, ap);
            va_end(ap);
            return std::string(buf);
        };
        auto R = [&](const char *base, int i) { return F_("%s_%d", base, i); };
        // ---- right-hand side: rh_I = -bh (the slabs that hold a row of bh) + G q_hat, column by column
        std::vector<char> started(NR, 0);
        for (int Ib = 0; Ib < NR; Ib++) {
            bool any = false;
            for (int r = 4 * Ib; r < 4 * Ib + 4; r++) any |= (r < n) || (r == n_eq - 1) || (r > n_eq && r <= n_eq + n);
            if (any) {
                P.stmt(sched::K_VALU, F_("double rh_%d = -bh[%d];", Ib, (int)bh_slabs.size()), "", {}, {R("rh", Ib)}, 1, false);
                bh_slabs.push_back(Ib);
                started[Ib] = 1;
            }
        }
        // q_hat of primal slab J from the current state, as micro-operations; `ph`: "g" (right-hand side) or "p" (primal phase)
        // the (LB, UB) rows of slab J when they are not register patterns: LDS reads, requested ONE GROUP AHEAD of the run of vector
        // instructions that needs them (behind the previous group's products: a read requested in front of its use is waited for)
        std::vector<char> bl_done[2] = {std::vector<char>(ZS, 0), std::vector<char>(ZS, 0)};
        auto bound_loads = [&](int J, const char *ph) {
            const int pi = ph[0] == 'p';
            if (bnd_in_regs || J >= ZS || bl_done[pi][J]) return;
            bl_done[pi][J] = 1;
            const std::string lb = F_("lb%s_%d", ph, J), ub = F_("ub%s_%d", ph, J), gov = pi ? "go_p" : "go_g";
            P.stmt(sched::K_LDS, F_("const double %s = LBR(%d);", lb.c_str(), J), "", {gov}, {lb}, 1);
            P.stmt(sched::K_LDS, F_("const double %s = UBR(%d);", ub.c_str(), J), "", {gov}, {ub}, 1);
        };
        auto qhat_ops = [&](int J, const char *ph) {
            if (J < ZS) {
                const bool prim = ph[0] == 'p';
                const std::string lb = F_("lb%s_%d", ph, J), ub = F_("ub%s_%d", ph, J), c1 = F_("c1%s_%d", ph, J), cc = F_("c%s_%d", ph, J),
                                  tt = F_("t%s_%d", ph, J), q = F_("q%s_%d", ph, J), w = R("w", J), gov = prim ? "go_p" : "go_g";
                // (the primal phase forms the same q_hat from the same state and the same bounds: without the laundered copy of w
                // and the laundered row index LLVM keeps the first phase's clamp and bounds alive across both solves - two values per slab)
                std::string wx = F_("w[%d]", J);
                if (prim) {
                    wx = F_("wp_%d", J);
                    P.stmt(sched::K_VALU, F_("double wp_%d = w[%d]; asm volatile(\"\" : \"+v\"(wp_%d));", J, J, J), "", {w}, {wx}, 0, false);
                }
                const std::string wv = prim ? wx : w;
                if (unit[J]) {  // c' = clamp01(w^ - kappa) (v_add_f64 ... clamp), s = w^ - 2 c': the operand of the column-scaled blocks
                    P.stmt(sched::K_VALU, F_("const double %s = fmin(fmax(%s - kapv[%d], 0.0), 1.0);", cc.c_str(), wx.c_str(), kap_of[J]), "", {wv}, {cc}, 1);
                    P.stmt(sched::K_VALU, F_("const double %s = __builtin_fma(-2.0, %s, %s);", q.c_str(), cc.c_str(), wx.c_str()), "", {cc, wv}, {q}, 1);
                    return;
                }
                if (bnd_in_regs) {  // (plain aliases: no instruction)
                    P.stmt(sched::K_VALU, F_("const double %s = lbv[%d], %s = ubv[%d];", lb.c_str(), bnd_of[J], ub.c_str(), bnd_of[J]), "", {}, {lb, ub}, 0, false);
                } else {
                    bound_loads(J, ph);
                }
                (void)gov;
                P.stmt(sched::K_VALU, F_("const double %s = fmax(%s, %s);", c1.c_str(), wx.c_str(), lb.c_str()), "", {wv, lb}, {c1}, 1);
                P.stmt(sched::K_VALU, F_("const double %s = fmin(%s, %s);", cc.c_str(), c1.c_str(), ub.c_str()), "", {c1, ub}, {cc}, 1);
                P.stmt(sched::K_VALU, F_("const double %s = __builtin_fma(-2.0, %s, %s);", tt.c_str(), cc.c_str(), wx.c_str()), "", {cc, wv}, {tt}, 1);
                P.stmt(sched::K_VALU, F_("const double %s = __builtin_fma(sigma, %s, qv[QI_%d]);", q.c_str(), tt.c_str(), J), "", {tt}, {q}, 1);
            } else {
                const int k = J - ZS;
                P.stmt(sched::K_VALU, F_("const double q%s_%d = __builtin_fma(-rho, sc[%d], mu[%d]);", ph, J, k, k), "", {R("sc", k), R("mu", k)},
                       {F_("q%s_%d", ph, J)}, 1);
            }
        };
        auto product = [&](const std::string &acc, std::vector<char> *st, int idx, const double *blk, const std::string &x) {
            const bool first = st && !(*st)[idx];
            if (st) (*st)[idx] = 1;
            P.mfma(acc, first, blk, x);
        };
        // program order by groups of GRP slabs: the q_hat of the group (one run of vector instructions), then the group's products - a vector
        // instruction alone between two MFMAs costs 12 clocks, in a run 4 (profiles/r03_microbench_issue.txt)
        int GRP = 6;  // (measured at C5, program order issued as it is: 4: 7.82 ms, 6: 7.76, 8: 7.78)
        if (const char *ev = getenv("SPCIES_BSP_GRP")) GRP = std::max(1, atoi(ev));
        for (int J0 = 0; J0 < NP; J0 += GRP) {
            for (int J = J0; J < std::min(NP, J0 + GRP); J++)
                if (!bG.by_col[J].empty()) qhat_ops(J, "g");
            for (int J = J0 + GRP; J < std::min(ZS, J0 + 2 * GRP); J++)
                if (!bG.by_col[J].empty()) bound_loads(J, "g");
            for (int J = J0; J < std::min(NP, J0 + GRP); J++)
                for (int Ib : bG.by_col[J]) {
                    double blk[16];
                    block_of(G, PR_, Ib, J, blk);
                    product(R("rh", Ib), &started, Ib, blk, R("qg", J));
                }
        }
        for (int Ib = 0; Ib < NR; Ib++)
            if (!started[Ib]) P.stmt(sched::K_VALU, F_("double rh_%d = 0.0;", Ib), "", {}, {R("rh", Ib)}, 1, false);
        // ---- W mu = rhs in column order: xf_J = Linv_JJ rh_J, rh_I -= L_IJ xf_J (I > J); then xb_J = (Uinv_JJ Dinv_J) xf_J,
        // xf_I -= D_I U_IJ xb_J (I < J).  Program order: the dependent chain D_J -> U_{J+1,J} -> D_{J+1} with the other updates
        // between its links; the scheduler refines it
        std::vector<std::vector<double>> Linv(NR, std::vector<double>(16));
        for (int Ib = 0; Ib < NR; Ib++) {
            double d[16];
            block_of(L, RR, Ib, Ib, d);
            inv_unit_lower(d, Linv[Ib].data());
        }
        struct Pend { int I, J; };
        auto run_columns = [&](bool forward) {
            std::vector<Pend> queue;
            auto pop_front = [&]() { Pend q = queue.front(); queue.erase(queue.begin()); return q; };
            auto update = [&](const Pend &q) {
                double lj[16], o[16];
                if (forward) {
                    block_of(L, RR, q.I, q.J, lj);
                    for (int e = 0; e < 16; e++) o[e] = -lj[e];
                    product(R("rh", q.I), nullptr, 0, o, R("xf", q.J));
                } else {
                    block_of(L, RR, q.J, q.I, lj);
                    for (int i = 0; i < 4; i++)
                        for (int k = 0; k < 4; k++) o[i * 4 + k] = -lj[k * 4 + i] / Dinv[4 * q.I + i];
                    product(R("xf", q.I), nullptr, 0, o, R("xb", q.J));
                }
            };
            for (int J = forward ? 0 : NR - 1; forward ? J < NR : J >= 0; J += forward ? 1 : -1) {
                const int nxt = forward ? J + 1 : J - 1;
                double d[16];
                if (forward) {
                    for (int e = 0; e < 16; e++) d[e] = Linv[J][e];
                    P.mfma(R("xf", J), true, d, R("rh", J));
                } else {
                    for (int i = 0; i < 4; i++)
                        for (int k = 0; k < 4; k++) d[i * 4 + k] = Linv[J][k * 4 + i] * Dinv[4 * J + k];
                    P.mfma(R("xb", J), true, d, R("xf", J));
                }
                bool any = false, crit = false;
                while (!queue.empty() && queue.front().I == nxt) { update(pop_front()); any = true; }
                if (!any && !queue.empty()) update(pop_front());
                if (forward) { for (int I : bL.by_col[J]) if (I > J) { if (I == nxt) crit = true; else queue.push_back(Pend{I, J}); } }
                else { for (int I : bL.by_row[J]) if (I < J) { if (I == nxt) crit = true; else queue.push_back(Pend{I, J}); } }
                std::stable_sort(queue.begin(), queue.end(), [&](const Pend &x, const Pend &y) { return forward ? x.I < y.I : x.I > y.I; });
                if (crit) update(Pend{nxt, J});
                bool one = false;
                while ((queue.size() > 1 || (!one && !queue.empty())) && queue.front().I != nxt) { update(pop_front()); one = true; }
            }
            while (!queue.empty()) update(pop_front());
        };
        run_columns(true);
        run_columns(false);
        P.stmt(sched::K_VALU, "asm volatile(\"\" : \"+v\"(go));", "", {}, {"go_g", "go_p"}, 0, false);
        // ---- primal_hat = (-Hh^-1) q_hat + (-Hh^-1 Gh') mu row by row; each row is consumed by the update of its slab.  q_hat of a
        // slab is a value of its own (formed before the slab's state is overwritten: the dependences say so), so a row that needs
        // the q_hat of another slab (the dense terminal weight) simply reads it
        std::vector<char> qp_done(NP, 0);
        auto need_qp = [&](int J) {
            if (!qp_done[J]) qhat_ops(J, "p");
            qp_done[J] = 1;
        };
        auto prim_row = [&](int Ib, const std::string &acc) {
            bool first = true;
            for (int J : bH.by_row[Ib]) need_qp(J);
            if (Ib < ZS && unit[Ib]) {  // z_hat' = (z_hat - lb) / D: the row's accumulator starts at -lb / D (no instruction: the first product's C operand)
                P.stmt(sched::K_VALU, F_("double %s = na3v[%d];", acc.c_str(), bnd_of[Ib]), "", {}, {acc}, 0, false);
                first = false;
            }
            for (int J : bH.by_row[Ib]) {
                double blk[16];
                block_of(H, PR_, Ib, J, blk);
                P.mfma(acc, first, blk, R("qp", J));
                first = false;
            }
            for (int J : bHG.by_row[Ib]) {
                double blk[16];
                block_of(HG, RR, Ib, J, blk);
                P.mfma(acc, first, blk, R("xb", J));
                first = false;
            }
            if (first) P.stmt(sched::K_VALU, "double " + acc + " = 0.0;", "", {}, {acc}, 1, false);
        };
        auto row_pre = [&](int Ib) {
            need_qp(Ib);  // (its clamp, bounds and w - clamp(w) are the update's too)
            for (int J : bH.by_row[Ib]) need_qp(J);
            P.stmt(sched::K_VALU, F_("const double dp_%d = wp_%d - cp_%d;", Ib, Ib, Ib), "", {R("wp", Ib), R("cp", Ib)}, {R("dp", Ib)}, 1);
        };
        auto row_post = [&](int Ib) {
            const std::string ph = R("ph", Ib), w = R("w", Ib), cc = R("cp", Ib), lb = R("lbp", Ib), ub = R("ubp", Ib), dd = R("dp", Ib);
            const std::string store = F_(" if (WANT_SOL) *((4 * %d + 3 < DIM_ || 4 * %d + g < DIM_) ? zhp + 4 * %d : dump) = ph_%d;", Ib, Ib, Ib, Ib);
            const std::string light = F_("w[%d] = ph_%d + dp_%d;", Ib, Ib, Ib) + store;
            const std::string full0 =
                F_("{ const double wn_ = ph_%d + dp_%d, z_ = fmin(fmax(wn_, lbp_%d), ubp_%d); w[%d] = wn_; res |= (fabs(cp_%d - z_) > tol_d) | "
                   "(fabs(z_ - ph_%d) > tol_p); }", Ib, Ib, Ib, Ib, Ib, Ib, Ib) + store;
            // (the checks of the other slabs: running maxima pinned per slab - an or-chain is sunk to the end of the iteration by LLVM with
            // every operand kept alive until then; fmax drops a NaN exactly like the reference's comparison)
            const std::string full =
                F_("{ const double wn_ = ph_%d + dp_%d, z_ = fmin(fmax(wn_, lbp_%d), ubp_%d); w[%d] = wn_; rd_ = fmax(rd_, fabs(cp_%d - z_)); "
                   "rp_ = fmax(rp_, fabs(z_ - ph_%d)); asm volatile(\"\" : \"+v\"(rd_), \"+v\"(rp_)); }", Ib, Ib, Ib, Ib, Ib, Ib, Ib) + store;
            if (unit[Ib]) {  // residuals in the caller's coordinates (x D); the record's z_hat = lb + D z_hat'
                const std::string ustore = F_(" if (WANT_SOL) *(zhp + 4 * %d) = __builtin_fma(dv[%d], ph_%d, lbv[%d]);", Ib, bnd_of[Ib], Ib, bnd_of[Ib]);
                const std::string ulight = F_("w[%d] = ph_%d + dp_%d;", Ib, Ib, Ib) + ustore;
                const std::string ufull0 =
                    F_("{ const double wn_ = ph_%d + dp_%d, z_ = fmin(fmax(wn_ - kapv[%d], 0.0), 1.0), D_ = dv[%d]; w[%d] = wn_; "
                       "res |= (fabs(cp_%d - z_) * D_ > tol_d) | (fabs(z_ - ph_%d) * D_ > tol_p); }", Ib, Ib, kap_of[Ib], bnd_of[Ib], Ib, Ib, Ib) + ustore;
                const std::string ufull =
                    F_("{ const double wn_ = ph_%d + dp_%d, z_ = fmin(fmax(wn_ - kapv[%d], 0.0), 1.0), D_ = dv[%d]; w[%d] = wn_; "
                       "rd_ = fmax(rd_, fabs(cp_%d - z_) * D_); rp_ = fmax(rp_, fabs(z_ - ph_%d) * D_); asm volatile(\"\" : \"+v\"(rd_), \"+v\"(rp_)); }",
                       Ib, Ib, kap_of[Ib], bnd_of[Ib], Ib, Ib, Ib) + ustore;
                if (Ib == 0) {
                    P.stmt(sched::K_VALU, ufull0, "", {ph, dd, cc}, {w, "res"}, 8);
                    P.mark = P.stmt(sched::K_MARK, "", "", {"res"}, {"branch"}, 2, false);
                } else {
                    P.stmt(sched::K_VALU, ulight, ufull, {ph, dd, cc, "branch"}, {w}, 1);
                }
                return;
            }
            if (Ib == 0) {
                P.stmt(sched::K_VALU, full0, "", {ph, dd, cc, lb, ub}, {w, "res"}, 8);
                // one wave-uniform branch: once the first slab's check has put every instance of the wavefront above its tolerance no
                // later check of this iteration can change the outcome - the rest of the iteration exists with and without the checks
                P.mark = P.stmt(sched::K_MARK, "", "", {"res"}, {"branch"}, 2, false);
            } else {
                P.stmt(sched::K_VALU, light, full, {ph, dd, cc, lb, ub, "branch"}, {w}, 2);
            }
        };
        // (slab 0 alone: the branch follows its check; then groups of GRP rows: their q_hat and w - clamp(w), their products, their updates)
        for (int I0 = 0; I0 < ZS; I0 = (I0 == 0 ? 1 : I0 + GRP)) {
            const int I1 = I0 == 0 ? 1 : std::min(ZS, I0 + GRP);
            for (int Ib = I0; Ib < I1; Ib++) row_pre(Ib);
            for (int Ib = I1; Ib < std::min(ZS, I1 + GRP); Ib++) bound_loads(Ib, "p");
            for (int Ib = I0; Ib < I1; Ib++) prim_row(Ib, R("ph", Ib));
            for (int Ib = I0; Ib < I1; Ib++) row_post(Ib);
        }
        {
            std::string args;
            sched::Op o;
            o.kind = sched::K_VALU;
            for (int k = 0; k < SS; k++) {
                prim_row(ZS + k, R("sh", k));
                args += (k ? ", sh_" : "sh_") + std::to_string(k);
            }
            o.text[0] = "{ double sh_[SS_] = {" + args + "}; SUPD_L(sh_); }";
            o.text[1] = "{ double sh_[SS_] = {" + args + "}; SUPD(sh_); }";
            for (int k = 0; k < SS; k++) {
                for (const std::string &v : {R("sh", k), R("mu", k), R("sc", k)}) o.reads.push_back(P.id(v));
                for (const std::string &v : {R("mu", k), R("sc", k)}) o.writes.push_back(P.id(v));
            }
            o.reads.push_back(P.id("branch"));
            o.cost = 10 * SS + 12;
            o.order = (int)P.ops.size();
            if (SS > 0) P.ops.push_back(o);
        }
        // (measured at C5 with pairs and the bounds in registers: 16: 7.77-7.8 ms, 20: 7.9, 24: 7.8, 28: 7.9-8.2, 32: 8.1-8.2 (15 scratch instructions in
        // the iteration), 40: 7.8-7.9, 64: 8.4; program order: 10.4; LLVM's own schedule of the round-2 program: 9.7)
        int window = 24;
        if (const char *ev = getenv("SPCIES_BSP_WINDOW")) window = std::max(1, atoi(ev));
        // The program order above - vector instructions in runs per group of slabs, the substitutions' dependent chain spaced by the other
        // updates - is issued as it is.  SPCIES_BSP_REORDER=1 list-schedules it (bsp_sched.hpp, `window` operations of look-ahead): measured at C5
        // 7.8 ms either way (grouped order 7.76; scheduled ungrouped order 7.8; grouped AND scheduled with window 24: 9.8, spills in the loop)
        std::vector<int> order = sched::schedule(P, window);
        if (!(getenv("SPCIES_BSP_REORDER") && getenv("SPCIES_BSP_REORDER")[0] == '1')) std::iota(order.begin(), order.end(), 0);
        // ---- blocks that appear many times in the stream (stage-invariant dynamics) stay in registers: every product's A operand
        // otherwise comes through the LDS pipe, the busiest unit of the iteration (SPCIES_BSP_KREG: how many; 0 = none)
        int kreg_max = 0;  // (measured at C5: 8 blocks 7.8 ms like none, 12: 8.0, 16: 8.3 - the registers they take cost more than the reads they save)
        if (const char *ev = getenv("SPCIES_BSP_KREG")) kreg_max = std::max(0, std::min(64, atoi(ev)));
        std::map<std::vector<double>, int> kreg_of;
        {
            std::map<std::vector<double>, int> freq;
            for (const sched::Op &o : P.ops)
                if (o.kind == sched::K_MFMA) freq[std::vector<double>(o.blk, o.blk + 16)]++;
            std::vector<std::pair<int, std::vector<double>>> byf;
            for (auto &kv : freq) byf.push_back({kv.second, kv.first});
            std::stable_sort(byf.begin(), byf.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
            for (size_t k = 0; k < byf.size() && (int)k < kreg_max && byf[k].first >= 3; k++) {
                kreg_of[byf[k].second] = (int)k;
                for (int kk = 0; kk < 4; kk++)
                    for (int i = 0; i < 4; i++) kreg_blocks.push_back(byf[k].second[i * 4 + kk]);
            }
        }
        // ---- emission: the blocks enter the table in issue order
        std::string head, light, full;
        bool after_mark = false;
        for (int idx : order) {
            const sched::Op &o = P.ops[idx];
            std::string l0, l1;
            if (o.kind == sched::K_MFMA) {
                const auto kr = kreg_of.find(std::vector<double>(o.blk, o.blk + 16));
                if (kr != kreg_of.end()) {
                    l0 = F_("            %s%s = __builtin_amdgcn_mfma_f64_4x4x4f64(kb[%d], %s, %s, 0, 0, 0);\n", o.acc_first ? "double " : "",
                            o.acc.c_str(), kr->second, o.x.c_str(), o.acc_first ? "0.0" : o.acc.c_str());
                    n_mfma++;
                    if (!after_mark) head += l0;
                    else { light += l0; full += l0; }
                    continue;
                }
                const int t = emit_block(o.blk);
                if (pairs) {  // ring of block PAIRS (one ds_read_b128 per two products)
                    const int P2 = t / 2;
                    l0 = F_("            %s%s = __builtin_amdgcn_mfma_f64_4x4x4f64(a%d.%c, %s, %s, 0, 0, 0);", o.acc_first ? "double " : "",
                            o.acc.c_str(), P2 % PF, t % 2 ? 'y' : 'x', o.x.c_str(), o.acc_first ? "0.0" : o.acc.c_str());
                    l0 += (t % 2) ? F_(" @%d@\n", P2) : std::string("\n");
                } else {
                    l0 = F_("            %s%s = __builtin_amdgcn_mfma_f64_4x4x4f64(a%d, %s, %s, 0, 0, 0); @%d@\n", o.acc_first ? "double " : "",
                            o.acc.c_str(), t % PF, o.x.c_str(), o.acc_first ? "0.0" : o.acc.c_str(), t);
                }
                n_mfma++;
            } else if (o.kind == sched::K_MARK) {
                after_mark = true;
                continue;
            } else {
                l0 = "            " + o.text[0] + "\n";
                if (!o.text[1].empty()) l1 = "            " + o.text[1] + "\n";
            }
            if (!after_mark) head += l1.empty() ? l0 : l1;
            else { light += l0; full += l1.empty() ? l0 : l1; }
        }
        body = head + "            HITUPD;\n            if (all_hit) {\n" + light + "            } else {\n            double rd_ = 0.0, rp_ = 0.0;\n" + full +
               "            res |= (rd_ > tol_d) | (rp_ > tol_p);\n            }\n";
        if (getenv("SPCIES_BSP_VERBOSE")) {  // how often the same 4x4 block appears in the stream
            std::map<std::vector<double>, int> freq;
            for (const sched::Op &o : P.ops)
                if (o.kind == sched::K_MFMA) freq[std::vector<double>(o.blk, o.blk + 16)]++;
            std::vector<int> cnt;
            for (auto &kv : freq) cnt.push_back(kv.second);
            std::sort(cnt.rbegin(), cnt.rend());
            int acc = 0;
            fprintf(stderr, "[spcies bsp] %zu distinct blocks of %d; cumulative coverage of the most frequent:", cnt.size(), n_mfma);
            for (size_t k = 0; k < cnt.size() && k < 48; k++) { acc += cnt[k]; if (k % 4 == 3) fprintf(stderr, " %zu:%d", k + 1, acc); }
            fprintf(stderr, "\n");
        }
        if (getenv("SPCIES_BSP_VERBOSE")) {  // live values (two registers each) along the issue order
            std::vector<int> last_use(P.ids.size(), -1), first_def(P.ids.size(), -1);
            for (size_t k = 0; k < order.size(); k++) {
                const sched::Op &o = P.ops[order[k]];
                for (int v : o.reads) last_use[v] = (int)k;
                for (int v : o.writes) { if (first_def[v] < 0) first_def[v] = (int)k; last_use[v] = std::max(last_use[v], (int)k); }
            }
            int live = 0, peak = 0, peak_at = 0;
            std::vector<int> delta(order.size() + 1, 0);
            for (size_t v = 0; v < P.ids.size(); v++)
                if (first_def[v] >= 0) { delta[first_def[v]]++; delta[last_use[v]]--; }
            for (size_t k = 0; k < order.size(); k++) { live += delta[k]; if (live > peak) { peak = live; peak_at = (int)k; } }
            fprintf(stderr, "[spcies bsp] peak of %d temporaries live at operation %d of %zu (state arrays not counted)\n", peak, peak_at, order.size());
        }
        if (getenv("SPCIES_BSP_VERBOSE")) {
            int last = 0;
            for (const sched::Op &o : P.ops) last = std::max(last, o.issue + o.cost);
            fprintf(stderr, "[spcies bsp] scheduled %zu operations (%d products): model %d quads per iteration\n", P.ops.size(), n_mfma, last);
        }
    }
    if (!use_sched) {
    // ---- A. rhs = -bh, B. rhs += G q_hat  (column-oriented: q_hat of a slab is formed once, used, and dropped)
    body += "            // rhs = (-Gh Hh^-1) q_hat - bh\n";
    for (int Ib = 0; Ib < NR; Ib++) {
        bool any = false;
        for (int r = 4 * Ib; r < 4 * Ib + 4; r++) any |= (r < n) || (r == n_eq - 1) || (r > n_eq && r <= n_eq + n);
        if (any) {
            snprintf(line, sizeof(line), "            rh[%d] = -bh[%d];\n", Ib, (int)bh_slabs.size());
            bh_slabs.push_back(Ib);
        } else {
            snprintf(line, sizeof(line), "            rh[%d] = 0.0;\n", Ib);
        }
        body += line;
    }
    for (int J = 0; J < NP; J++) {
        if (bG.by_col[J].empty()) continue;
        char e[128];
        qhat_expr(J, e, sizeof(e));
        snprintf(line, sizeof(line), "            { const double qh = %s;\n", e);
        body += line;
        for (int Ib : bG.by_col[J]) {
            double blk[16];
            block_of(G, PR_, Ib, J, blk);
            snprintf(a1, sizeof(a1), "rh[%d]", Ib);
            MF(a1, emit_block(blk), "qh");
        }
        body += "            }\n";
        if (J % SEG_EVERY == SEG_EVERY - 1) body += "            SEG;\n";
    }
    body += "            SEG;\n            // W mu = rhs: forward substitution by blocks\n";
    std::vector<std::vector<double>> Linv(NR, std::vector<double>(16));
    // Column order ("right-looking", SPCIES_BSP_RL=0 restores the row order): x_J = Linv_JJ rh_J, then rh_I -= L_IJ x_J for the rows I
    // below.  In row order a block row is ONE chain of MFMAs on one accumulator (22 cycles per link instead of 16) that ends in the
    // product with the solution slab finished right before it; in column order consecutive products go to different accumulators and
    // only the chain D_J -> U_{J+1,J} -> D_{J+1} is dependent - its links are spaced by the other updates of the columns
    // (a queue ordered by the row they go to: every update of rh_I is out before D_I).
    const bool rl = !(getenv("SPCIES_BSP_RL") && getenv("SPCIES_BSP_RL")[0] == '0');
    struct Pend { int I, J; };
    auto run_columns = [&](bool forward) {
        std::vector<Pend> queue;  // kept sorted by distance of I from the current column
        auto pop_front = [&]() { Pend q = queue.front(); queue.erase(queue.begin()); return q; };
        auto emit_update = [&](const Pend &q) {
            double lj[16], o[16];
            if (forward) {
                block_of(L, RR, q.I, q.J, lj);
                for (int e = 0; e < 16; e++) o[e] = -lj[e];
            } else {  // rows I < J: -D_I (L_JI)'  (D = 1 / Dinv: the solve runs on x, the scaling sits in the diagonal block)
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
            snprintf(line, sizeof(line), "            { double xx = 0.0; MF(xx, a%d, rh[%d]); @%d@ rh[%d] = xx; }\n", (int)(tab.size() / 16) % PF, J,
                     (int)(tab.size() / 16), J);
            emit_block(d);
            body += line;
            n_mfma++;
            // this column's updates: the one into the next slab is the critical link, the others wait in the queue
            bool crit = false;
            std::vector<int> rows;
            if (forward) { for (int I : bL.by_col[J]) if (I > J) rows.push_back(I); }
            else { for (int I : bL.by_row[J]) if (I < J) rows.push_back(I); }
            // slot a: what still has to reach rh_next (else one update of an older column) hides the latency D_J -> U_{next,J}
            bool any = false;
            while (!queue.empty() && queue.front().I == nxt) { emit_update(pop_front()); any = true; }
            if (!any && !queue.empty()) emit_update(pop_front());
            for (int I : rows) {
                if (I == nxt) crit = true;
                else queue.push_back(Pend{I, J});
            }
            std::stable_sort(queue.begin(), queue.end(), [&](const Pend &x, const Pend &y) { return forward ? x.I < y.I : x.I > y.I; });
            if (crit) emit_update(Pend{nxt, J});
            // slot b: between U_{next,J} and D_next
            bool one = false;
            while (queue.size() > 1 || (!one && !queue.empty())) {
                if (queue.front().I == nxt) break;  // (cannot happen: those left in slot a)
                emit_update(pop_front());
                one = true;
            }
            if (step % SEG_EVERY == SEG_EVERY - 1) body += "            SEG;\n";
        }
        while (!queue.empty()) emit_update(pop_front());  // (empty by construction)
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
        snprintf(line, sizeof(line), "            { double acc = 0.0;\n");
        body += line;
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
        // Uinv_II = (Linv_II)';  Bk_II = Uinv_II diag(Dinv_I);  Bk_IJ = -Uinv_II U_IJ,  U_IJ = (L_JI)'
        double ui[16], d[16];
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 4; k++) ui[i * 4 + k] = Linv[Ib][k * 4 + i];
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 4; k++) d[i * 4 + k] = ui[i * 4 + k] * Dinv[4 * Ib + k];
        body += "            { double acc = 0.0;\n";
        snprintf(a2, sizeof(a2), "rh[%d]", Ib);
        MF("acc", emit_block(d), a2);
        for (int J : bL.by_col[Ib]) {  // blocks (J, Ib) of L with J > Ib  <=>  blocks (Ib, J) of U
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
    // ---- E. primal_hat = H q_hat + HG mu, row by row, each row consumed by the z / s update straight away.  A q_hat that a
    // LATER row needs from an EARLIER slab (off-diagonal blocks of -Hh^-1: the dense terminal weight) is saved before that
    // slab's state is overwritten.
    std::vector<int> last_use(NP, -1);
    for (int Ib = 0; Ib < NP; Ib++)
        for (int J : bH.by_row[Ib]) last_use[J] = std::max(last_use[J], Ib);
    for (int J = 0; J < NP; J++)
        if (last_use[J] > J) { const int idx = (int)saved.size(); saved[J] = idx; }
    body += "            // primal_hat = (-Hh^-1) q_hat + (-Hh^-1 Gh') mu; z: box, lambda; s: cone, mu; residuals\n";
    auto prim_row = [&](int Ib, const char *acc) {
        for (int J : bH.by_row[Ib]) {
            double blk[16];
            block_of(H, PR_, Ib, J, blk);
            char e[128];
            if (J < Ib && J < ZS) snprintf(e, sizeof(e), "sv[%d]", saved.at(J));  // (the s slabs are all updated at the end)
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
    // (round 3: the bounds of a slab are requested from LDS BEFORE its products - one wavefront per SIMD has nothing else to hide the
    // read behind; measured 10.05 -> 9.92 ms at C5.  Also tried: two accumulators per block row of the substitutions, used in turn - a
    // chain of v_mfma_f64_4x4x4 on one accumulator issues every 22 cycles, several chains every 16.8 - with the nearest dependence
    // last: 11.4 ms against 9.9 - slower, and so was the same program with the second accumulator left unused.)
    for (int Ib = 0; Ib < ZS; Ib++) {
        if (early_bounds) snprintf(line, sizeof(line), "            { double ph = 0.0; const double lbx = LBR(%d), ubx = UBR(%d);\n", Ib, Ib);
        else snprintf(line, sizeof(line), "            { double ph = 0.0; const double lbx = 0.0, ubx = 0.0; (void)lbx; (void)ubx;\n");
        body += line;
        prim_row(Ib, "ph");
        if (saved.count(Ib)) {
            char e[128];
            qhat_expr(Ib, e, sizeof(e));
            snprintf(line, sizeof(line), "              sv[%d] = %s;\n", saved[Ib], e);
            body += line;
        }
        snprintf(line, sizeof(line), "              ZUPD(%d, ph); }\n", Ib);
        body += line;
        if (Ib == 0) body += "/*SPLIT*/";  // the rest of the iteration exists twice: with and without residual checks (below)
        if (Ib % SEG_EVERY == SEG_EVERY - 1) body += "            SEG;\n";
    }
    body += "            { double sh[SS_];\n";
    for (int k = 0; k < SS; k++) {
        snprintf(line, sizeof(line), "              sh[%d] = 0.0;\n", k);
        body += line;
        snprintf(a1, sizeof(a1), "sh[%d]", k);
        prim_row(ZS + k, a1);
    }
    body += "              SUPD(sh); }\n";
    }  // (!use_sched)
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
    {  // resolve the refill markers.  The stream is padded to a multiple of PF positions (the holes past the last block consume
       // nothing), so that position t always sits in ring slot t % PF, also across the wrap into the next iteration: after
       // block t is consumed its slot takes position t + PF; the slots of the holes are refilled at the end of the iteration
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
    {  // Once the first slab's check has put every instance of the wavefront above its tolerance, no later check of this iteration
       // can change the outcome (the reference leaves its residual loops at the first hit): the rest of the iteration runs in a copy
       // without the checks (ZUPD_L / SUPD_L), chosen by ONE wave-uniform branch - a branch per slab would cut the straight-line
       // program into basic blocks the scheduler cannot interleave MFMA chains across (measured: slower than no branch at all)
        const size_t cut = body.find("/*SPLIT*/");
        if (cut != std::string::npos && !getenv("SPCIES_BSP_NOSPLIT")) {
            const std::string head = body.substr(0, cut), tail = body.substr(cut + 9);
            std::string light = tail;
            for (const char *nm : {"ZUPD(", "SUPD("}) {
                const std::string from = nm, to = std::string(nm).insert(4, "_L");
                for (size_t i = light.find(from); i != std::string::npos; i = light.find(from, i + to.size())) light.replace(i, from.size(), to);
            }
            body = head + "            HITUPD;\n            if (all_hit) {\n" + light + "            } else {\n" + tail + "            }\n";
        }
    }
    // ---- LB / UB rows of the z slabs (rows past dim - n - 1 are free; pads are pinned to 0 by 0 <= z <= 0)
    const int rc_lb = (int)tab.size();
    for (int r = 0; r < 4 * ZS; r++) tab.push_back(lb_rows[r]);
    const int rc_ub = (int)tab.size();
    for (int r = 0; r < 4 * ZS; r++) tab.push_back(ub_rows[r]);
    const int kb_off = (int)tab.size();
    tab.insert(tab.end(), kreg_blocks.begin(), kreg_blocks.end());
    for (double x : tab)
        if (!std::isfinite(x)) { p.why = "non-finite block"; return 0; }
    const size_t lds_bytes = tab.size() * sizeof(double);
    if (lds_bytes > 160 * 1024 - 1024 || p.n_blocks > 1536) { p.why = "block table exceeds the LDS"; return 0; }
    if (p.n_blocks <= PF) { p.why = "fewer blocks than the prefetch ring"; return 0; }
    if (NR + ZS + 2 * SS > 190) { p.why = "state does not fit the register file"; return 0; }
    // ---- source
    std::string s;
    auto def = [&](const char *name, long v) { snprintf(line, sizeof(line), "#define %s %ld\n", name, v); s += line; };
    def("ZS_", ZS); def("SS_", SS); def("NP_", NP); def("NR_", NR); def("NQ_", (long)qrow.size()); def("NSV_", (long)std::max<size_t>(saved.size(), 1));
    if (early_bounds) s += "#define EARLY_BOUNDS_ 1\n";
    def("TAB_DOUBLES_", (long)tab.size()); def("RC_LB_", rc_lb); def("RC_UB_", rc_ub); def("DIM_", dim); def("NSC_", n_s);
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
    def("KREG_", (long)(kreg_blocks.size() / 16)); def("KB_OFF_", kb_off);
    def("NBND_", bnd_in_regs ? (long)bnd_slab.size() : 1L);
    def("BND_IN_REGS_", bnd_in_regs ? 1 : 0);
    s += "static __device__ const int BNDSLAB_[NBND_] = {";
    for (size_t i = 0; i < (bnd_in_regs ? bnd_slab.size() : (size_t)1); i++) { snprintf(line, sizeof(line), "%s%d", i ? ", " : "", bnd_slab[i]); s += line; }
    s += "};\n";
    def("NKAP_", (long)kap_q.size());
    if (!kap_q.empty()) {
        s += "static __device__ const int KAPQ_[NKAP_] = {";
        for (size_t i = 0; i < kap_q.size(); i++) { snprintf(line, sizeof(line), "%s%d", i ? ", " : "", kap_q[i]); s += line; }
        s += "};\nstatic __device__ const int KAPB_[NKAP_] = {";
        for (size_t i = 0; i < kap_b.size(); i++) { snprintf(line, sizeof(line), "%s%d", i ? ", " : "", kap_b[i]); s += line; }
        s += "};\n";
    }
    s += "static __device__ const int KIA_[ZS_] = {";  // slab -> its kappa register (-1: plain coordinates)
    for (int J = 0; J < ZS; J++) { snprintf(line, sizeof(line), "%s%d", J ? ", " : "", kap_of[J]); s += line; }
    s += "};\nstatic __device__ const int BIA_[ZS_] = {";  // slab -> its bound pattern
    for (int J = 0; J < ZS; J++) { snprintf(line, sizeof(line), "%s%d", J ? ", " : "", bnd_in_regs ? bnd_of[J] : 0); s += line; }
    s += "};\n";
    def("NBH_", (long)bh_slabs.size());
    s += "static __device__ const int BHSLAB_[NBH_] = {";
    for (size_t i = 0; i < bh_slabs.size(); i++) { snprintf(line, sizeof(line), "%s%d", i ? ", " : "", bh_slabs[i]); s += line; }
    s += "};\n";
    s += "static __device__ const int QROW_[NQ_] = {";
    for (size_t i = 0; i < qrow.size(); i++) { snprintf(line, sizeof(line), "%s%d", i ? ", " : "", qrow[i]); s += line; }
    s += "};\n";
    s += R"SRC(
struct Args {
    int n, m, N, dim, n_s, n_eq, k_max, ref_stride, r_stride, pad;
    double tol_p, tol_d, rho, rho_i, sigma, sigma_i;
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
__device__ __forceinline__ void soc_bsp_body(const Args &p, const double *__restrict__ table_g, const double *__restrict__ cst,
                                             const double *__restrict__ x0g, const double *__restrict__ xrg,
                                             const double *__restrict__ urg, const double *__restrict__ rg,
                                             double *__restrict__ u_out, int *__restrict__ k_out, int *__restrict__ e_out,
                                             double *__restrict__ f0, double *__restrict__ f1, double *__restrict__ f2,
                                             double *__restrict__ f3, double *__restrict__ f4, double *__restrict__ f5) {
    __shared__ __attribute__((aligned(16))) double ldsr[2 * 4 * ZS_];
    __shared__ __attribute__((aligned(16))) double blk0[NB0_ * 16];
    __shared__ __attribute__((aligned(16))) double blk1[NB1_ * 16];
    __shared__ __attribute__((aligned(16))) double blk2[NB2_ * 16];
    for (int i = threadIdx.x; i < NB0_ * 16; i += 256) blk0[i] = table_g[i];
    for (int i = threadIdx.x; i < NB1_ * 16; i += 256) blk1[i] = table_g[512 * 16 + i];
    for (int i = threadIdx.x; i < NB2_ * 16; i += 256) blk2[i] = table_g[1024 * 16 + i];
    for (int i = threadIdx.x; i < 2 * 4 * ZS_; i += 256) ldsr[i] = table_g[RC_LB_ + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int n = p.n, m = p.m, nm = n + m, N = p.N, dim = p.dim, n_s = p.n_s, n_eq = p.n_eq;
    double *dump = const_cast<double *>(table_g) + TAB_DOUBLES_;  // (4 NP + 8 doubles behind the table: bsp::finish_soc)
    const double rho = p.rho, rho_i = p.rho_i, sigma = p.sigma, sigma_i = p.sigma_i, tol_p = p.tol_p, tol_d = p.tol_d;
    const double *cA = cst, *cQ = cA + n * n, *cR = cQ + n * n, *cT = cR + m * m, *cPhiP = cT + n * n;
    int ao = g * 4 + (lane & 3);
    const long n_tiles = (p.B + 15) / 16;
#if KREG_ > 0
    double kb[KREG_];  // blocks that repeat along the horizon: A operands in registers for the whole launch
#pragma unroll
    for (int u = 0; u < KREG_; u++) kb[u] = table_g[KB_OFF_ + u * 16 + ao];
#endif
#define BLK(arr, t) arr[(t) * 16 + ao]
#define PBLK(arr, pp) (*reinterpret_cast<const double2 *>(&arr[(pp) * 32 + 2 * ao]))
#define MF(acc, a, x) acc = __builtin_amdgcn_mfma_f64_4x4x4f64((a), (x), (acc), 0, 0, 0)
#define SEG __builtin_amdgcn_sched_barrier(0)
    for (long tile = (long)blockIdx.x * 4 + wave; tile < n_tiles; tile += (long)gridDim.x * 4) {
        const long inst = tile * 16 + c;
        const bool valid = inst < p.B;
        const long ii = valid ? inst : 0;
        const double *x0 = x0g + ii * n;
        const double *xr = p.ref_stride ? xrg + ii * n : xrg;
        const double *ur = p.ref_stride ? urg + ii * m : urg;
        const double r_ellip = rg[p.r_stride ? ii : 0];
        // per-instance constants: bh (rows of the right-hand side), q (one register per distinct slab pattern).  (gs: the lane's row group,
        // laundered per group of instances - the rows' addresses into A, Q, R, T, PhiP depend on the lane only, and left alone the compiler
        // forms all ~60 of them once in front of this loop and parks them in scratch memory for the whole launch: 488 B per lane)
        int gs = g;
        asm volatile("" : "+v"(gs));
        double bh[NBH_];
#pragma unroll
        for (int I = 0; I < NBH_; I++) {
            const int row = 4 * BHSLAB_[I] + gs;
            double v = 0.0;
            if (row < n) {
                for (int i = 0; i < n; i++) v -= cA[row * n + i] * x0[i];
            } else if (row == n_eq - 1) {
                v = r_ellip;
            } else if (row > n_eq && row <= n_eq + n) {
                const int jj = row - n_eq - 1;
                for (int i = 0; i < n; i++) v -= cPhiP[jj * n + i] * xr[i];
            }
            bh[I] = v;
        }
        double qv[NQ_];
#pragma unroll
        for (int u = 0; u < NQ_; u++) {
            const int j = QROW_[u] + gs;
            double v = 0.0;
            if (j < m) {
                for (int i = 0; i < m; i++) v += cR[j * m + i] * ur[i];
            } else if (j < m + (N - 1) * nm) {
                const int e = (j - m) % nm;
                if (e < n) {
                    for (int i = 0; i < n; i++) v += cQ[e * n + i] * xr[i];
                } else {
                    for (int i = 0; i < m; i++) v += cR[(e - n) * m + i] * ur[i];
                }
            } else if (j < m + (N - 1) * nm + n) {
                const int e = j - m - (N - 1) * nm;
                for (int i = 0; i < n; i++) v += cT[e * n + i] * xr[i];
            }
            qv[u] = v;
        }
        // state: w = z + lambda / sigma per z slab (z = clamp(w), lambda = sigma (w - z): the update w+ = z_hat + (w - clamp(w))
        // is the reference's z / lambda step, admm_mfma4.hpp); the cone slabs keep s and mu
        double w[ZS_], sc[SS_], mu[SS_], rh[NR_], sv[NSV_];
#pragma unroll
        for (int I = 0; I < ZS_; I++) w[I] = 0.0;
#pragma unroll
        for (int I = 0; I < SS_; I++) { sc[I] = 0.0; mu[I] = 0.0; }
        int go = g;  // (laundered once per iteration like ao: keeps LICM from hoisting every bound read out of the loop)
#define LBR(I) ldsr[4 * (I) + go]
#define UBR(I) ldsr[4 * ZS_ + 4 * (I) + go]
#if BND_IN_REGS_
        double lbv[NBND_], ubv[NBND_];  // the distinct (LB, UB) slab patterns
#pragma unroll
        for (int u = 0; u < NBND_; u++) {
            lbv[u] = ldsr[4 * BNDSLAB_[u] + g];
            ubv[u] = ldsr[4 * ZS_ + 4 * BNDSLAB_[u] + g];
        }
#endif
#if NKAP_ > 0  // unit-box slabs: kappa = q / (sigma D) - lb / D per (q pattern, bound pattern), -lb / D per bound pattern; w = 0 is w^ = kappa - lb / D
        double kapv[NKAP_], na3v[NBND_], dv[NBND_];  // (dv: the checked path's residuals and the record, in the caller's coordinates)
#pragma unroll
        for (int u = 0; u < NBND_; u++) {
            const double D_ = ubv[u] - lbv[u];
            dv[u] = D_;
            na3v[u] = (D_ > 0.0 && D_ < 1e6) ? -lbv[u] / D_ : 0.0;
        }
#pragma unroll
        for (int u = 0; u < NKAP_; u++) {
            const double D_ = ubv[KAPB_[u]] - lbv[KAPB_[u]];
            kapv[u] = qv[KAPQ_[u]] / (sigma * D_) + na3v[KAPB_[u]];
        }
#pragma unroll
        for (int I = 0; I < ZS_; I++)
            if (KIA_[I] >= 0) w[I] = kapv[KIA_[I]] + na3v[BIA_[I]];
#endif
        // z of slab I and w - z (= lambda / sigma) in the caller's coordinates from the state (exit, record: cold)
#if NKAP_ > 0
#define ZOF_(I) (KIA_[I] >= 0 ? __builtin_fma(UBR(I) - LBR(I), fmin(fmax(w[I] - kapv[KIA_[I] >= 0 ? KIA_[I] : 0], 0.0), 1.0), LBR(I)) : fmin(fmax(w[I], LBR(I)), UBR(I)))
#define WOF_(I) (KIA_[I] >= 0 ? __builtin_fma(UBR(I) - LBR(I), w[I] - kapv[KIA_[I] >= 0 ? KIA_[I] : 0], LBR(I)) : w[I])
#else
#define ZOF_(I) fmin(fmax(w[I], LBR(I)), UBR(I))
#define WOF_(I) w[I]
#endif
#ifdef EARLY_BOUNDS_  // the slab's bounds were read at the top of its block (lbx, ubx)
#define BND_LB(I) lbx
#define BND_UB(I) ubx
#else
#define BND_LB(I) LBR(I)
#define BND_UB(I) UBR(I)
#endif
#define QHZ(J) (qv[QI_##J] + sigma * (w[J] - 2.0 * fmin(fmax(w[J], LBR(J)), UBR(J))))
        bool active = valid, res = false, all_hit = false;
#define HITUPD                                                                                   \
    do {                                                                                         \
        unsigned long long hb_ = __ballot(res);                                                  \
        hb_ |= hb_ >> 32;                                                                        \
        hb_ |= hb_ >> 16;                                                                        \
        all_hit = (hb_ & 0xFFFFull) == 0xFFFFull;                                                \
    } while (0)
        int kk = 0;
        RING_INIT
        // ... without the residual check (every instance of the wavefront is above its tolerance already)
#define ZUPD_L(I, zh)                                                                            \
    do {                                                                                         \
        const double lb_ = BND_LB(I), ub_ = BND_UB(I);                                           \
        const double wo_ = w[I], zo_ = fmin(fmax(wo_, lb_), ub_);                                \
        w[I] = (zh) + (wo_ - zo_);                                                               \
        if (WANT_SOL) *((4 * (I) + 3 < DIM_ || 4 * (I) + g < DIM_) ? zhp + 4 * (I) : dump) = (zh); \
    } while (0)
        // z rows of slab I: box, lambda, residuals (:209-217, 246-248, 256-267)
#define ZUPD(I, zh)                                                                              \
    do {                                                                                         \
        const double lb_ = BND_LB(I), ub_ = BND_UB(I);                                           \
        const double wo_ = w[I], zo_ = fmin(fmax(wo_, lb_), ub_);                                \
        const double wn_ = (zh) + (wo_ - zo_), z_ = fmin(fmax(wn_, lb_), ub_);                   \
        w[I] = wn_;                                                                              \
        res |= (fabs(zo_ - z_) > tol_d) | (fabs(z_ - (zh)) > tol_p);                             \
        if (WANT_SOL) *((4 * (I) + 3 < DIM_ || 4 * (I) + g < DIM_) ? zhp + 4 * (I) : dump) = (zh); \
    } while (0)
        // the cone rows: s = proj_SOC(s_hat + mu / rho), mu, residuals (:220-242, 251-253)
#define SUPD(sh)                                                                                 \
    do {                                                                                         \
        double v_[SS_], nrm_ = 0.0;                                                              \
        _Pragma("unroll") for (int k_ = 0; k_ < SS_; k_++) {                                     \
            v_[k_] = (sh)[k_] + rho_i * mu[k_];                                                  \
            nrm_ += (k_ == 0 && g == 0) ? 0.0 : v_[k_] * v_[k_];                                 \
        }                                                                                        \
        nrm_ = xsum16_(nrm_);                                                                          \
        nrm_ = xsum32_(nrm_);                                                                          \
        const double s_norm_ = sqrt(nrm_), s0_ = row0_(v_[0]);                                   \
        _Pragma("unroll") for (int k_ = 0; k_ < SS_; k_++) {                                     \
            double v = v_[k_];                                                                   \
            if (s_norm_ <= s0_) {                                                                \
            } else if (s_norm_ <= -s0_) {                                                        \
                v = 0.0;                                                                         \
            } else {                                                                             \
                const double step_ = (s0_ + s_norm_) / (2 * s_norm_);                            \
                v = (k_ == 0 && g == 0) ? step_ * s_norm_ : step_ * v;                           \
            }                                                                                    \
            const double so_ = sc[k_], mu_ = mu[k_];                                             \
            sc[k_] = v;                                                                          \
            mu[k_] = mu_ + rho * ((sh)[k_] - v);                                                 \
            res |= (fabs(so_ - v) > tol_d) | (fabs(v - (sh)[k_]) > tol_p);                       \
            if (WANT_SOL) *((4 * k_ + 3 < NSC_ || 4 * k_ + g < NSC_) ? shp + 4 * k_ : dump) = (sh)[k_]; \
        }                                                                                        \
    } while (0)
#define SUPD_L(sh)                                                                                \
    do {                                                                                         \
        double v_[SS_], nrm_ = 0.0;                                                              \
        _Pragma("unroll") for (int k_ = 0; k_ < SS_; k_++) {                                     \
            v_[k_] = (sh)[k_] + rho_i * mu[k_];                                                  \
            nrm_ += (k_ == 0 && g == 0) ? 0.0 : v_[k_] * v_[k_];                                 \
        }                                                                                        \
        nrm_ = xsum16_(nrm_);                                                                          \
        nrm_ = xsum32_(nrm_);                                                                          \
        const double s_norm_ = sqrt(nrm_), s0_ = row0_(v_[0]);                                   \
        _Pragma("unroll") for (int k_ = 0; k_ < SS_; k_++) {                                     \
            double v = v_[k_];                                                                   \
            if (s_norm_ <= s0_) {                                                                \
            } else if (s_norm_ <= -s0_) {                                                        \
                v = 0.0;                                                                         \
            } else {                                                                             \
                const double step_ = (s0_ + s_norm_) / (2 * s_norm_);                            \
                v = (k_ == 0 && g == 0) ? step_ * s_norm_ : step_ * v;                           \
            }                                                                                    \
            const double so_ = sc[k_], mu_ = mu[k_];                                             \
            sc[k_] = v;                                                                          \
            mu[k_] = mu_ + rho * ((sh)[k_] - v);                                                 \
            (void)so_;                                                                           \
            if (WANT_SOL) *((4 * k_ + 3 < NSC_ || 4 * k_ + g < NSC_) ? shp + 4 * k_ : dump) = (sh)[k_]; \
        }                                                                                        \
    } while (0)
        while (true) {
            kk += 1;
            res = false;
            asm volatile("" : "+v"(ao), "+v"(go));
            // record of z_hat / s_hat: every iteration overwrites the instance's row; finished instances write to a dump row
            double *zhp = (WANT_SOL && active) ? f2 + inst * dim + g : dump;
            double *shp = (WANT_SOL && active) ? f3 + inst * n_s + g : dump;
)SRC";
    s += body;
    s += R"SRC(
            SEG;
            // exit test per instance (:269-286)
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
                    if (g < m) u_out[inst * m + g] = ZOF_(0);  // u = z[0 .. m)  (m <= 4: inside slab 0)
                    if (WANT_SOL) {
                        double *zp = f0 + inst * dim + g, *lp = f4 + inst * dim + g, *sp = f1 + inst * n_s + g, *mp = f5 + inst * n_s + g;
#pragma unroll
                        for (int I = 0; I < ZS_; I++)
                        {
                            const bool in_ = 4 * I + 3 < DIM_ || 4 * I + g < DIM_;
                            const double z_ = ZOF_(I);
                            *(in_ ? zp + 4 * I : dump) = z_;
                            *(in_ ? lp + 4 * I : dump) = sigma * (WOF_(I) - z_);
                        }
#pragma unroll
                        for (int k = 0; k < SS_; k++)
                        {
                            const bool in_ = 4 * k + 3 < NSC_ || 4 * k + g < NSC_;
                            *(in_ ? sp + 4 * k : dump) = sc[k];
                            *(in_ ? mp + 4 * k : dump) = mu[k];
                        }
                    }
                    active = false;
                }
            }
            if (!__any(active)) break;
        }
    }
}
extern "C" __global__ __launch_bounds__(256, 1) void soc_bsp_kernel(Args p, const double *table_g, const double *cst, const double *x0g,
                                                                   const double *xrg, const double *urg, const double *rg, double *u_out,
                                                                   int *k_out, int *e_out) {
    soc_bsp_body<false>(p, table_g, cst, x0g, xrg, urg, rg, u_out, k_out, e_out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}
extern "C" __global__ __launch_bounds__(256, 1) void soc_bsp_kernel_sol(Args p, const double *table_g, const double *cst, const double *x0g,
                                                                       const double *xrg, const double *urg, const double *rg, double *u_out,
                                                                       int *k_out, int *e_out, double *f0, double *f1, double *f2, double *f3,
                                                                       double *f4, double *f5) {
    soc_bsp_body<true>(p, table_g, cst, x0g, xrg, urg, rg, u_out, k_out, e_out, f0, f1, f2, f3, f4, f5);
}
)SRC";
    p.src = s;
    p.args = Args{n, m, N, dim, n_s, n_eq, c.k_max, 0, 0, 0, c.tol_p, c.tol_d, c.rho, c.rho_i, c.sigma, c.sigma_i, 0};
    if (m > 4) { p.why = "m > 4 (u rows outside slab 0)"; return 0; }
    if (const char *path = getenv("SPCIES_BSP_DUMP")) {  // kernel experiments: keep the generated program
        if (FILE *f = fopen(path, "w")) {
            fputs(s.c_str(), f);
            fclose(f);
        }
    }
    p.why = "not compiled yet";
    return 0;
}

// compiles p.src (hiprtc) and loads the module; *scratch = bytes of scratch memory per lane of the no-record kernel
inline int compile_program(Plan &p, int *scratch, const char *name0 = "soc_bsp_kernel", const char *name1 = "soc_bsp_kernel_sol") {
    if (p.module) rtc::unload_module(p.module);
    p.module = nullptr;
    // experiments: SPCIES_BSP_FLAGS holds extra compiler options, blank-separated
    std::vector<std::string> extra = rtc::split_flags(getenv("SPCIES_BSP_FLAGS"));
    if (p.scheduled && !getenv("SPCIES_BSP_KEEP_MISCHED")) {  // the order printed by bsp_sched.hpp is the order issued
        extra.push_back("-mllvm");
        extra.push_back("-enable-misched=false");
    }
    // (cached per process: the handles spcies_hip_create_multi builds for the same controller print the same program)
    int rc = rtc::compile_module(p.src.c_str(), "spcies_soc_bsp.hip", {name0, name1}, extra, &p.module, p.fn, true);
    if (rc) return rc;
    int local = 0;
    if (hipFuncGetAttribute(&local, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, p.fn[0]) != hipSuccess) local = 0;
    *scratch = local;
    return 0;
}

// device part: compile the program - with a shallower prefetch ring if the register allocator had to spill (scratch memory
// in the iteration loop costs more than the ring buys) - and upload the table and the dense setup constants
inline int finish_soc(Plan &p, const SocDev &c, const double *F, const int *I) {
    if (p.src.empty() || p.ok) return 0;
    if (p.d_table) hipFree(p.d_table);
    if (p.d_consts) hipFree(p.d_consts);
    p.d_table = p.d_consts = nullptr;
    int scratch = 0;
    int rc = compile_program(p, &scratch);
    if (rc) return rc;
    // a compiler that spills far more than the one this was tuned with (an older comgr loaded first: 1 KB and more, 77-99 ms) gets a
    // shorter ring
    if (!getenv("SPCIES_BSP_PF"))
        for (int pf : {-4, 12, 8, 4}) {  // (-4: the scheduled program with a ring of four pairs; then the round-2 form)
            if (scratch <= 640) break;
            if (pf < 0 && !p.scheduled) continue;
            p.legacy_order = pf > 0;
            rc = build_soc(p, c, F, I, pf < 0 ? -pf : pf);
            if (rc) return rc;
            if (p.src.empty()) return fail(SPCIES_HIP_ENOSUP, "BSP program: %s", p.why.c_str());
            rc = compile_program(p, &scratch);
            if (rc) return rc;
        }
    if (getenv("SPCIES_BSP_VERBOSE"))
        fprintf(stderr, "[spcies bsp] %d blocks, %d MFMAs per iteration, table %zu B, scratch %d B per lane\n", p.n_blocks, p.n_mfma,
                p.table.size() * sizeof(double), scratch);
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_table, (p.table.size() + 4 * (size_t)(p.ZS + p.SS) + 8) * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_table, p.table.data(), p.table.size() * sizeof(double), hipMemcpyHostToDevice));
    const int n = c.n, m = c.m;
    std::vector<double> cst;
    cst.insert(cst.end(), F + c.A, F + c.A + n * n);
    cst.insert(cst.end(), F + c.Q, F + c.Q + n * n);
    cst.insert(cst.end(), F + c.R, F + c.R + m * m);
    cst.insert(cst.end(), F + c.T, F + c.T + n * n);
    cst.insert(cst.end(), F + c.PhiP, F + c.PhiP + n * n);
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

// fields = z, s, z_hat, s_hat, lambda, mu (all or none)
inline int launch_soc(Plan &p, const SocDev &c, const double *x0, const double *xr, const double *ur, int ref_stride, const double *r,
                      int r_stride, long B, double *u, int *k, int *e, double *const *f, hipStream_t st) {
    if (!p.ok) return fail(SPCIES_HIP_ENOSUP, "BSP variant not available: %s", p.why.c_str());
    bool any = false, all = true;
    for (int i = 0; i < 6; i++) { any |= f[i] != nullptr; all &= f[i] != nullptr; }
    if (any && !all) return fail(SPCIES_HIP_EINVAL, "BSP variant: pass all six record fields or none");
    Args a = p.args;
    a.k_max = c.k_max; a.tol_p = c.tol_p; a.tol_d = c.tol_d;  // set_exit overrides
    a.ref_stride = ref_stride; a.r_stride = r_stride; a.B = B;
    const long n_tiles = (B + 15) / 16;
    long wgs = (n_tiles + 3) / 4;
    if (wgs > p.num_cu) wgs = p.num_cu;
    const double *table = p.d_table, *cst = p.d_consts;
    double *f0 = f[0], *f1 = f[1], *f2 = f[2], *f3 = f[3], *f4 = f[4], *f5 = f[5];
    void *params[] = {&a, &table, &cst, &x0, &xr, &ur, &r, &u, &k, &e, &f0, &f1, &f2, &f3, &f4, &f5};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel(p.fn[any ? 1 : 0], (unsigned)wgs, 1, 1, 256, 1, 1, 0, st, params, nullptr));
    return 0;
}

}  // namespace bsp
}  // namespace spcies
