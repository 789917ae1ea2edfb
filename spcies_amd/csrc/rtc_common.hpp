// hiprtc, bound at run time (dlopen): shared by every run-time specialised kernel of the library (MFMA4 shapes, BSP block
// programs, FUSED HMPC shapes).
#pragma once
#include <dlfcn.h>
#include <limits.h>
#include <link.h>
#include <unistd.h>

#include "common.hpp"

namespace spcies {
namespace rtc {

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


// Compile `src` for gfx950 and load it: `names` are name expressions (template instantiations) resolved to functions
inline int compile_module(const char *src, const char *fname, const std::vector<std::string> &names, const std::vector<std::string> &extra_opts,
                          hipModule_t *module, hipFunction_t *fns) {
    Hiprtc &rt = hiprtc();
    int rc = rt.open();
    if (rc) return rc;
    rt.sync_env();
    void *prog = nullptr;
    if (rt.create(&prog, src, fname, 0, nullptr, nullptr) != 0) return fail(SPCIES_HIP_EHIP, "hiprtcCreateProgram failed");
    for (const std::string &nm : names)
        if (rt.add_name(prog, nm.c_str()) != 0) {
            rt.destroy(&prog);
            return fail(SPCIES_HIP_EHIP, "hiprtcAddNameExpression failed");
        }
    std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-honor-nans"};
    for (const std::string &e : extra_opts) opts.push_back(e.c_str());
    if (rt.compile(prog, (int)opts.size(), opts.data()) != 0) {
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
    std::vector<std::string> lowered;
    for (const std::string &nm : names) {
        const char *ln = nullptr;
        if (rt.lowered(prog, nm.c_str(), &ln) != 0 || !ln) {
            rt.destroy(&prog);
            return fail(SPCIES_HIP_EHIP, "hiprtcGetLoweredName failed");
        }
        lowered.push_back(ln);
    }
    rt.destroy(&prog);
    SPCIES_HIP_CHECK(hipModuleLoadData(module, code.data()));
    for (size_t i = 0; i < lowered.size(); i++) SPCIES_HIP_CHECK(hipModuleGetFunction(&fns[i], *module, lowered[i].c_str()));
    return 0;
}

}  // namespace rtc
}  // namespace spcies
