// Run-time specialisation of the MFMA4 kernel (admm_mfma4.hpp) for a controller whose (N, ceil(n/4), ceil((n+m)/4))
// is not among the shapes instantiated at build time.  Spcies is a code generator - its C platform prints one
// solver per controller - and the register-resident MFMA4 kernel is specialised on the horizon in the same way:
// the marked regions of the MFMA4 headers (gen_rtc_src.py -> mfma4_rtc_src.inc) are compiled with hiprtc for the
// shape at hand, loaded as a module and launched like the built-in instantiations.  Measured: about one second per
// controller at create time, 2.0-2.3x faster than MFMA4G on the shapes of tools/bench_rtc.py.  SPCIES_HIP_RTC=0 in
// the environment turns it off; libhiprtc.so is bound with dlopen on first use.
#pragma once
#include <dlfcn.h>
#include <limits.h>
#include <link.h>
#include <unistd.h>

#include "admm_mfma4.hpp"

namespace spcies {
namespace rtc {

static const char *const kMfma4Source =
#include "mfma4_rtc_src.inc"
    ;

struct Hiprtc {
    void *lib = nullptr;
    int (*create)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*add_name)(void *, const char *) = nullptr;
    int (*compile)(void *, int, const char **) = nullptr;
    int (*lowered)(void *, const char *, const char **) = nullptr;
    int (*code_size)(void *, size_t *) = nullptr;
    int (*code)(void *, char *) = nullptr;
    int (*log_size)(void *, size_t *) = nullptr;
    int (*log)(void *, char *) = nullptr;
    int (*destroy)(void **) = nullptr;
    char ***ns_environ = nullptr;  // &__environ of the private namespace's libc (dlmopen case)
    void sync_env() const {
        if (ns_environ) *ns_environ = environ;
    }
    int open() {
        if (lib) return 0;
        // A process that has loaded another ROCm user-space before us (PyTorch wheels bundle libhiprtc / libamd_comgr) hands us
        // THAT compiler by soname, whatever path we ask for - and the generated kernels are tuned against the installed one
        // (an older comgr spills the BSP program's state to scratch memory: 77-99 ms instead of 11 at C5 soc).  Only in such
        // a process the installation's hiprtc is opened in a link namespace of its own (dlmopen).  The namespace has its own
        // libc, whose view of the environment goes stale when the host program calls setenv: sync_env() before every call.
        {
            const char *root = getenv("ROCM_PATH");
            const std::string dir = std::string(root && *root ? root : "/opt/rocm") + "/lib/";
            char real[PATH_MAX];
            const std::string rdir = realpath(dir.c_str(), real) ? std::string(real) + "/" : dir;
            struct Probe { const std::string *a, *b; bool foreign; } probe{&dir, &rdir, false};
            dl_iterate_phdr(
                [](struct dl_phdr_info *info, size_t, void *data) {
                    Probe *pr = static_cast<Probe *>(data);
                    const char *nm = info->dlpi_name ? info->dlpi_name : "";
                    if ((strstr(nm, "libamd_comgr") || strstr(nm, "libhiprtc")) && strncmp(nm, pr->a->c_str(), pr->a->size()) != 0 &&
                        strncmp(nm, pr->b->c_str(), pr->b->size()) != 0)
                        pr->foreign = true;
                    return 0;
                },
                &probe);
            if (probe.foreign && !getenv("SPCIES_HIPRTC_SHARED_NAMESPACE")) {
                lib = dlmopen(LM_ID_NEWLM, (dir + "libhiprtc.so").c_str(), RTLD_NOW | RTLD_LOCAL);
                if (lib) ns_environ = (char ***)dlsym(lib, "__environ");
            }
        }
        for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            if (lib) break;
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!lib) return fail(SPCIES_HIP_ENOSUP, "run-time specialisation: cannot load libhiprtc.so (%s)", dlerror());
#define SPCIES_SYM(field, sym) field = (decltype(field))dlsym(lib, sym)
        SPCIES_SYM(create, "hiprtcCreateProgram");
        SPCIES_SYM(add_name, "hiprtcAddNameExpression");
        SPCIES_SYM(compile, "hiprtcCompileProgram");
        SPCIES_SYM(lowered, "hiprtcGetLoweredName");
        SPCIES_SYM(code_size, "hiprtcGetCodeSize");
        SPCIES_SYM(code, "hiprtcGetCode");
        SPCIES_SYM(log_size, "hiprtcGetProgramLogSize");
        SPCIES_SYM(log, "hiprtcGetProgramLog");
        SPCIES_SYM(destroy, "hiprtcDestroyProgram");
#undef SPCIES_SYM
        if (!create || !add_name || !compile || !lowered || !code_size || !code || !log_size || !log || !destroy)
            return fail(SPCIES_HIP_ENOSUP, "run-time specialisation: hiprtc symbols missing");
        return 0;
    }
};

inline Hiprtc &hiprtc() {  // one binding (and one link namespace) per process
    static Hiprtc rt;
    return rt;
}

struct Mfma4Module {
    hipModule_t module = nullptr;
    hipFunction_t fn[2] = {nullptr, nullptr};  // WANT_SOL = false, true
    bool ok = false;
};
inline void module_free(Mfma4Module &m) {
    if (m.module) hipModuleUnload(m.module);
    m.module = nullptr;
    m.ok = false;
}

// compile admm_mfma4_kernel<N, KX, KS, TERMINAL, false / true> for gfx950
inline int compile_mfma4(Mfma4Module &out, int N, int KX, int KS, bool terminal) {
    Hiprtc &rt = hiprtc();
    int rc = rt.open();
    if (rc) return rc;
    rt.sync_env();
    void *prog = nullptr;
    if (rt.create(&prog, kMfma4Source, "spcies_mfma4_rtc.hip", 0, nullptr, nullptr) != 0)
        return fail(SPCIES_HIP_EHIP, "hiprtcCreateProgram failed");
    char names[2][128];
    for (int s = 0; s < 2; s++) {
        snprintf(names[s], sizeof(names[s]), "spcies::admm_mfma4_kernel<%d, %d, %d, %s, %s>", N, KX, KS, terminal ? "true" : "false",
                 s ? "true" : "false");
        if (rt.add_name(prog, names[s]) != 0) {
            rt.destroy(&prog);
            return fail(SPCIES_HIP_EHIP, "hiprtcAddNameExpression failed");
        }
    }
    std::vector<std::string> extra;  // experiments: SPCIES_MFMA4_RTC_FLAGS holds extra options, blank-separated
    if (const char *ev = getenv("SPCIES_MFMA4_RTC_FLAGS")) {
        std::string tok;
        for (const char *c = ev;; c++) {
            if (*c == ' ' || *c == '\0') {
                if (!tok.empty()) extra.push_back(tok);
                tok.clear();
                if (!*c) break;
            } else {
                tok.push_back(*c);
            }
        }
    }
    std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-honor-nans", "-DSPCIES_RTC_STATIC_LDS=1"};
    for (const std::string &e : extra) opts.push_back(e.c_str());
    const int crc = rt.compile(prog, (int)opts.size(), opts.data());
    if (crc != 0) {
        size_t ls = 0;
        rt.log_size(prog, &ls);
        std::string lg(ls + 1, '\0');
        if (ls) rt.log(prog, &lg[0]);
        rt.destroy(&prog);
        return fail(SPCIES_HIP_EHIP, "hiprtcCompileProgram failed: %.400s", lg.c_str());
    }
    size_t cs = 0;
    rt.code_size(prog, &cs);
    std::vector<char> code(cs);
    rt.code(prog, code.data());
    std::string lowered[2];
    for (int s = 0; s < 2; s++) {
        const char *ln = nullptr;
        if (rt.lowered(prog, names[s], &ln) != 0 || !ln) {
            rt.destroy(&prog);
            return fail(SPCIES_HIP_EHIP, "hiprtcGetLoweredName failed");
        }
        lowered[s] = ln;
    }
    rt.destroy(&prog);
    SPCIES_HIP_CHECK(hipModuleLoadData(&out.module, code.data()));
    for (int s = 0; s < 2; s++) SPCIES_HIP_CHECK(hipModuleGetFunction(&out.fn[s], out.module, lowered[s].c_str()));
    out.ok = true;
    return 0;
}

// launch like launch_mfma4_shape (admm_mfma4.hpp); the table lives in static LDS
inline int launch_mfma4(const Mfma4Module &m, const Mfma4Plan &pl, const AdmmHost &a, const double *x0, const double *xr,
                        const double *ur, int ref_stride, long B, double *u, int *k, int *e, double *z, double *v, double *lam,
                        hipStream_t st) {
    const bool want_sol = (z || v || lam);
    if (want_sol && !(z && v && lam)) return fail(SPCIES_HIP_EINVAL, "MFMA4 variant: pass all of z, v, lambda or none");
    MfmaArgs args{a.n, a.m, a.k_max, a.tol, a.rho, a.rho_i, B, ref_stride};
    const long n_tiles = (B + 15) / 16;
    long wgs = (n_tiles + 3) / 4;
    if (wgs > pl.num_cu) wgs = pl.num_cu;
    const double *table = pl.d_table;
    double *dump = pl.d_table + pl.table_bytes / sizeof(double);
    void *params[] = {&args, &table, &x0, &xr, &ur, &u, &k, &e, &z, &v, &lam, &dump};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel(m.fn[want_sol ? 1 : 0], (unsigned)wgs, 1, 1, 256, 1, 1, 0, st, params, nullptr));
    return 0;
}

}  // namespace rtc
}  // namespace spcies
