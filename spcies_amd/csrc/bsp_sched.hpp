// Micro-operation form and (optional) issue-order scheduler of the BSP block programs (soc_bsp.hpp).
//
// One ADMM iteration of a BSP program is ~1 000 v_mfma_f64_4x4x4 block products, ~600 FP64 vector instructions (q_hat, box / cone
// updates, residuals) and a few LDS reads, in straight-line code with every operand a literal.  The generator emits it as a list of
// micro-operations - one statement each: a product, one FP64 instruction, an LDS read; every temporary a value of its own - with their
// read / write sets (`Program`).  That form is what keeps LLVM from mis-optimising the text (an or-chain of residuals sunk to the end of the
// iteration, the right-hand side's q_hat CSE'd with the primal phase's and kept alive across both solves) and what lets the generator
// decide the order: the program is compiled with the machine scheduler switched off, so the order printed is the order issued.
//
// Which order: the generator's own program order - vector instructions in runs per group of slabs, the substitutions' dependent chain
// spaced by the other updates of the columns - is the default.  `schedule()` list-schedules the operations on a model of the SIMD
// (SPCIES_BSP_REORDER=1); measured at C5 it is no faster than the grouped program order (7.8 ms both), because what it models - result
// latencies hidden behind other work - is not what costs time at one wavefront per SIMD (profiles/r03_microbench_issue.txt: a vector
// instruction does not overlap with the wavefront's own MFMA; alone between two MFMAs it costs 12 clocks, in a run 4).
//
// Machine model of `schedule()`, in issue slots of 4 clocks ("quads"):
//   v_mfma_f64_4x4x4 occupies the FP64 pipe for 4 quads; its result feeds the next product's accumulator after 4 quads, a
//   product's B operand after 6, a vector instruction after 7 (LLVM's hazard table for the 4-pass DGEMM, accumulator-file read included);
//   an FP64 vector instruction occupies the same pipe for 1 quad, result to a dependent vector instruction after 2, to a product's B
//   operand after 3; an LDS read returns after ~32 quads; `sw` quads are lost when a vector instruction follows a product.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace spcies {
namespace bsp {
namespace sched {

enum Kind { K_MFMA = 0, K_VALU = 1, K_LDS = 2, K_MARK = 3 };

struct Op {
    Kind kind = K_VALU;
    // statement text; [1]: the variant with residual checks where it differs ("" = same as [0]).  MFMA: built at emission.
    std::string text[2];
    std::vector<int> reads, writes;  // value ids
    // MFMA
    double blk[16];
    std::string acc, x;
    int acc_id = -1;       // value id of the accumulator (read-modify-write unless acc_first)
    bool acc_first = false;  // the accumulator starts at zero: this product declares it
    int cost = 1;          // quads of the FP64 pipe (VALU: number of FP64 instructions; MFMA: 4)
    bool pipe = true;      // occupies the FP64 pipe (false: integer / move / LDS)
    int order = 0;         // program order (the generator's)
    // filled by schedule()
    int issue = 0;
};

struct Program {
    std::vector<Op> ops;
    std::map<std::string, int> ids;
    int id(const std::string &name) {
        auto it = ids.find(name);
        if (it != ids.end()) return it->second;
        const int v = (int)ids.size();
        ids.emplace(name, v);
        return v;
    }
    int mark = -1;  // index of the branch marker (-1: none)

    // a vector / LDS statement: `text` (and its checked variant), the values it reads and writes
    int stmt(Kind kind, const std::string &light, const std::string &full, std::initializer_list<std::string> reads,
             std::initializer_list<std::string> writes, int cost, bool pipe = true) {
        Op o;
        o.kind = kind;
        o.text[0] = light;
        o.text[1] = full;
        for (const std::string &r : reads) o.reads.push_back(id(r));
        for (const std::string &w : writes) o.writes.push_back(id(w));
        o.cost = cost;
        o.pipe = pipe && kind != K_LDS;
        o.order = (int)ops.size();
        ops.push_back(o);
        return o.order;
    }
    // acc (+)= blk * x
    int mfma(const std::string &acc, bool first, const double *blk, const std::string &x) {
        Op o;
        o.kind = K_MFMA;
        std::memcpy(o.blk, blk, sizeof(o.blk));
        o.acc = acc;
        o.x = x;
        o.acc_id = id(acc);
        o.acc_first = first;
        o.reads.push_back(id(x));
        if (!first) o.reads.push_back(o.acc_id);
        o.writes.push_back(o.acc_id);
        o.cost = 4;
        o.order = (int)ops.size();
        ops.push_back(o);
        return o.order;
    }
};

struct Edge { int to, lat; };

struct Model {
    int lds = 32, mm_c = 4, mm_b = 6, mv = 7, vm = 2, vv = 1, valu_pipe = 1;
    int sw = 0;  // issue slots lost when a vector instruction follows a product (profiles/r03_microbench_issue.txt: 2)
    Model() {
        if (const char *ev = getenv("SPCIES_BSP_MODEL"))  // experiments: "lds,mm_c,mm_b,mv,vm,vv,valu_pipe,sw"
            sscanf(ev, "%d,%d,%d,%d,%d,%d,%d,%d", &lds, &mm_c, &mm_b, &mv, &vm, &vv, &valu_pipe, &sw);
    }
};
inline const Model &model() {
    static Model m;
    return m;
}

inline int raw_latency(const Op &p, const Op &s, int value) {
    const Model &m = model();
    if (p.kind == K_LDS) return m.lds;
    if (p.kind == K_MARK || s.kind == K_MARK) return 1;
    if (p.kind == K_MFMA) {
        if (s.kind == K_MFMA) return (s.acc_id == value && !s.acc_first) ? m.mm_c : m.mm_b;
        return m.mv;
    }
    // vector instruction(s): the last one of a multi-instruction statement produces the value
    if (s.kind == K_MFMA) return p.cost + m.vm;
    return p.cost + m.vv;
}

// Orders the operations.  `window`: an operation is not issued while more than `window` operations that precede it in program
// order are still waiting (bounds the live ranges the reordering creates).  Returns the permutation (indices into p.ops).
inline std::vector<int> schedule(Program &p, int window) {
    const int n = (int)p.ops.size();
    std::vector<std::vector<Edge>> succ(n);
    std::vector<int> npred(n, 0);
    {
        const int nv = (int)p.ids.size();
        std::vector<int> last_write(nv, -1);
        std::vector<std::vector<int>> readers(nv);
        auto edge = [&](int a, int b, int lat) {
            if (a == b) return;
            succ[a].push_back(Edge{b, lat});
            npred[b]++;
        };
        for (int i = 0; i < n; i++) {
            const Op &o = p.ops[i];
            for (int v : o.reads)
                if (last_write[v] >= 0) edge(last_write[v], i, raw_latency(p.ops[last_write[v]], o, v));
            for (int v : o.writes) {
                for (int r : readers[v]) edge(r, i, 1);                      // write after read
                if (last_write[v] >= 0) edge(last_write[v], i, 1);            // write after write
            }
            for (int v : o.reads) readers[v].push_back(i);
            for (int v : o.writes) {
                readers[v].clear();
                last_write[v] = i;
            }
        }
    }
    // priority: longest latency path to the end of the iteration
    std::vector<int> prio(n, 0);
    for (int i = n - 1; i >= 0; i--) {
        int best = p.ops[i].cost;
        for (const Edge &e : succ[i]) best = std::max(best, e.lat + prio[e.to]);
        prio[i] = best;
    }
    std::vector<int> earliest(n, 0), order;
    std::vector<char> done(n, 0);
    order.reserve(n);
    int t = 0, pipe_free = 0, first_waiting = 0;
    bool last_mfma = false;
    std::vector<int> ready;
    for (int i = 0; i < n; i++)
        if (npred[i] == 0) ready.push_back(i);
    while ((int)order.size() < n) {
        while (first_waiting < n && done[first_waiting]) first_waiting++;
        int best = -1, best_t = 0;
        for (int c : ready) {
            if (c > first_waiting + window) continue;
            const Op &o = p.ops[c];
            int est = std::max(earliest[c], t);
            if (o.pipe) est = std::max(est, pipe_free + ((o.kind == K_VALU && last_mfma) ? model().sw : 0));
            // (a run of vector instructions is continued before the matrix pipe is taken again: every switch costs)
            const bool cont = model().sw > 0 && !last_mfma && o.kind == K_VALU, bcont = best >= 0 && model().sw > 0 && !last_mfma && p.ops[best].kind == K_VALU;
            if (best < 0 || est < best_t || (est == best_t && ((cont && !bcont) || (cont == bcont && (prio[c] > prio[best] || (prio[c] == prio[best] && c < best)))))) {
                best = c;
                best_t = est;
            }
        }
        if (best < 0) {  // (everything ready is outside the window: take the oldest waiting operation's turn)
            best = *std::min_element(ready.begin(), ready.end());
            best_t = std::max({earliest[best], t, p.ops[best].pipe ? pipe_free : 0});
        }
        Op &o = p.ops[best];
        o.issue = best_t;
        t = best_t + (o.kind == K_MFMA ? 1 : std::max(1, o.cost));
        if (o.pipe) pipe_free = best_t + (o.kind == K_MFMA ? 4 : o.cost * model().valu_pipe);
        if (o.pipe) last_mfma = o.kind == K_MFMA;
        done[best] = 1;
        order.push_back(best);
        ready.erase(std::find(ready.begin(), ready.end(), best));
        for (const Edge &e : succ[best]) {
            earliest[e.to] = std::max(earliest[e.to], best_t + e.lat);
            if (--npred[e.to] == 0) ready.push_back(e.to);
        }
    }
    return order;
}

}  // namespace sched
}  // namespace bsp
}  // namespace spcies
