// hiprtc, bound at run time (dlopen): shared by every run-time specialised kernel of the library (MFMA4 shapes, BSP block
// programs, FUSED HMPC shapes).
#pragma once
#include <dlfcn.h>
#include <limits.h>
#include <link.h>
#include <unistd.h>

#include <functional>
#include <map>
#include <memory>
#include <mutex>

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

// Everything that touches the binding - open(), the environment sync of the private link namespace, the compiler itself - runs
// under ONE process-wide lock: handles may be created from several host threads (spcies_hip_create_multi does: one per device).
inline std::mutex &rtc_mutex() {
    static std::mutex mu;
    return mu;
}

// Code objects compiled in this process, keyed by (source, file name, name expressions, options): the N handles of
// spcies_hip_create_multi - and any later handle for the same controller - compile once (4-10 s for a BSP / MFMA4R program) and
// load the same code object on their own device.  Entries are immutable once inserted and live as long as the process.
struct CodeObject {
    std::vector<char> code;
    std::vector<std::string> lowered;
};
inline std::map<std::string, std::shared_ptr<const CodeObject>> &code_cache() {
    static std::map<std::string, std::shared_ptr<const CodeObject>> cache;
    return cache;
}
struct CacheStats { long hits = 0, misses = 0; };
inline CacheStats &cache_stats() {
    static CacheStats st;
    return st;
}


inline std::vector<std::string> split_flags(const char *ev) {  // blank-separated compiler options of an experiment variable
    std::vector<std::string> out;
    std::string tok;
    for (const char *c = ev; c;) {
        if (*c == ' ' || *c == '\0') {
            if (!tok.empty()) out.push_back(tok);
            tok.clear();
            if (!*c) break;
        } else {
            tok.push_back(*c);
        }
        c++;
    }
    return out;
}

// Compile `src` for gfx950 (or take the code object this process compiled before) and load it on the current device: `names`
// are name expressions (template instantiations) resolved to functions
inline int compile_module(const char *src, const char *fname, const std::vector<std::string> &names, const std::vector<std::string> &extra_opts,
                          hipModule_t *module, hipFunction_t *fns, bool names_are_symbols = false) {
    // names_are_symbols: `names` are extern "C" kernels of the source (a generated program), taken as they are
    std::shared_ptr<const CodeObject> co;
    {
        std::lock_guard<std::mutex> lk(rtc_mutex());
        // the key holds the full text: a hash collision must not hand a controller somebody else's program
        std::string key = std::string(fname) + '\x1f';
        for (const std::string &nm : names) key += nm + '\x1e';
        key += '\x1f';
        for (const std::string &e : extra_opts) key += e + '\x1e';
        key += '\x1f';
        key += src;
        auto it = getenv("SPCIES_HIP_RTC_NOCACHE") ? code_cache().end() : code_cache().find(key);
        if (it != code_cache().end()) {
            co = it->second;
            cache_stats().hits++;
        } else {
            Hiprtc &rt = hiprtc();
            int rc = rt.open();
            if (rc) return rc;
            rt.sync_env();
            void *prog = nullptr;
            if (rt.create(&prog, src, fname, 0, nullptr, nullptr) != 0) return fail(SPCIES_HIP_EHIP, "hiprtcCreateProgram failed");
            for (const std::string &nm : names)
                if (!names_are_symbols && rt.add_name(prog, nm.c_str()) != 0) {
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
            auto fresh = std::make_shared<CodeObject>();
            size_t cs = 0;
            rt.code_size(prog, &cs);
            fresh->code.resize(cs);
            rt.code(prog, fresh->code.data());
            for (const std::string &nm : names) {
                const char *ln = names_are_symbols ? nm.c_str() : nullptr;
                if (!names_are_symbols && (rt.lowered(prog, nm.c_str(), &ln) != 0 || !ln)) {
                    rt.destroy(&prog);
                    return fail(SPCIES_HIP_EHIP, "hiprtcGetLoweredName failed");
                }
                fresh->lowered.push_back(ln);
            }
            rt.destroy(&prog);
            cache_stats().misses++;
            co = fresh;
            if (!getenv("SPCIES_HIP_RTC_NOCACHE")) code_cache().emplace(std::move(key), co);
        }
    }
    // loading is per device (the caller has made its device current) and needs no lock
    SPCIES_HIP_CHECK(hipModuleLoadData(module, co->code.data()));
    for (size_t i = 0; i < co->lowered.size(); i++) SPCIES_HIP_CHECK(hipModuleGetFunction(&fns[i], *module, co->lowered[i].c_str()));
    return 0;
}

}  // namespace rtc
}  // namespace spcies
