// hiprtc, bound at run time (dlopen): shared by every run-time specialised kernel of the library (MFMA4 shapes, BSP block
// programs, FUSED HMPC shapes).
#pragma once
#include <dlfcn.h>
#include <limits.h>
#include <link.h>
#include <poll.h>
#include <signal.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <functional>
#include <map>
#include <memory>
#include <mutex>

#include <sys/stat.h>

#include "../../include/spcies_hip.h"
#include "code_cache.hpp"
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
// under ONE process-wide lock: handles may be created from several host threads (spcies_hip_create_multi does: one per device), and
// neither the dlmopen'ed namespace's environment nor comgr's global state is documented as thread-safe.  Cache look-ups, disk hits
// and the statistics do NOT take it (code_cache.hpp has its own short lock).
inline std::mutex &rtc_mutex() {
    static std::mutex mu;
    return mu;
}

// Identity of the compiler that would run: hiprtc's version and the installed libraries' path, size and modification time - part
// of every cache key, so a ROCm update never serves code objects of the previous compiler (code_cache.hpp).
inline const std::string &compiler_identity() {
    static std::string id;
    static std::once_flag once;
    std::call_once(once, [] {
        Hiprtc &rt = hiprtc();
        std::string s = "hiprtc";
        if (rt.lib) {
            int (*version)(int *, int *) = (int (*)(int *, int *))dlsym(rt.lib, "hiprtcVersion");
            int major = 0, minor = 0;
            if (version && version(&major, &minor) == 0) s += " " + std::to_string(major) + "." + std::to_string(minor);
            Dl_info info;
            if (rt.create && dladdr((void *)rt.create, &info) && info.dli_fname) {
                char real[PATH_MAX];
                const std::string path = realpath(info.dli_fname, real) ? real : info.dli_fname;
                const std::string dir = path.substr(0, path.rfind('/') + 1);
                for (const std::string &f : {path, dir + "libamd_comgr.so"}) {
                    char r2[PATH_MAX];
                    const std::string rp = realpath(f.c_str(), r2) ? r2 : f;
                    struct stat st;
                    if (stat(rp.c_str(), &st) == 0)
                        s += " " + rp + ":" + std::to_string((long long)st.st_size) + ":" + std::to_string((long long)st.st_mtime);
                }
            }
        }
        id = s;
    });
    return id;
}

// A loaded module pins the code object it was created from (hipModuleLoadData is not documented to copy the image): the pin is
// dropped by unload_module, which every owner of a run-time compiled module calls instead of hipModuleUnload.
inline std::map<hipModule_t, std::shared_ptr<const CodeObject>> &module_pins() {
    static std::map<hipModule_t, std::shared_ptr<const CodeObject>> pins;
    return pins;
}
inline std::mutex &pins_mutex() {
    static std::mutex mu;
    return mu;
}
inline void unload_module(hipModule_t m) {
    if (!m) return;
    // Unload and un-pin under ONE lock, the lock compile_module loads and pins under: the runtime may hand the handle value of an
    // unloaded module to the next hipModuleLoadData at once (create_multi builds handles on parallel threads while another handle is
    // destroyed), and a late erase would then drop the NEW module's pin.  The code object outlives the unload (released last).
    std::shared_ptr<const CodeObject> keep;
    std::lock_guard<std::mutex> lk(pins_mutex());
    auto it = module_pins().find(m);
    if (it != module_pins().end()) {
        keep = std::move(it->second);
        module_pins().erase(it);
    }
    (void)hipModuleUnload(m);
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

// ---- the compiler in a process of its own (rtc_helper.cpp; SPCIES_HIP_RTC_ISOLATE=0: in-process as before) -----------------------------
// One helper per process, started at the first compilation that misses both caches and kept until the library is unloaded (its stdin
// closes: it exits).  Requests are serialised by rtc_mutex().  A helper that dies mid-request (compiler crash) fails THAT build - the caller's
// process goes on, AUTO falls back - and the next request starts a fresh one.  No helper binary next to the library: in-process.
struct HelperProc {
    pid_t pid = -1, owner = -1;  // owner: the process that started it (a fork()ed copy of this object must not share the parent's pipe)
    int to = -1, from = -1;
    bool looked = false;
    std::string path;
    ~HelperProc() { stop(); }
    void stop() {
        if (to >= 0) close(to);
        if (from >= 0) close(from);
        to = from = -1;
        if (pid > 0 && owner == getpid()) {
            int st = 0;
            for (int i = 0; i < 200 && waitpid(pid, &st, WNOHANG) == 0; i++) usleep(5000);  // (stdin closed: it leaves by itself)
            if (waitpid(pid, &st, WNOHANG) == 0) { kill(pid, SIGKILL); waitpid(pid, &st, 0); }
        }
        pid = -1;
    }
    bool find() {
        if (looked) return !path.empty();
        looked = true;
        if (const char *ev = getenv("SPCIES_HIP_RTC_ISOLATE"))
            if (ev[0] == '0') return false;
        if (const char *ev = getenv("SPCIES_HIP_RTC_HELPER")) {
            if (access(ev, X_OK) == 0) path = ev;
            return !path.empty();
        }
        Dl_info info;
        if (!dladdr((void *)&spcies_hip_abi_version, &info) || !info.dli_fname) return false;
        char real[PATH_MAX];
        const std::string lib = realpath(info.dli_fname, real) ? real : info.dli_fname;
        const std::string cand = lib.substr(0, lib.rfind('/') + 1) + "spcies_rtc_helper";
        if (access(cand.c_str(), X_OK) == 0) path = cand;
        return !path.empty();
    }
    bool start() {
        if (pid > 0 && owner != getpid()) {  // we are a fork()ed child of the process that owns this helper: leave it alone, get our own
            if (to >= 0) close(to);
            if (from >= 0) close(from);
            to = from = -1;
            pid = -1;
        }
        if (pid > 0) return true;
        int a[2], b[2];
        if (pipe2(a, O_CLOEXEC) != 0) return false;
        if (pipe2(b, O_CLOEXEC) != 0) { close(a[0]); close(a[1]); return false; }
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        posix_spawn_file_actions_adddup2(&fa, a[0], 0);  // (dup2 clears O_CLOEXEC on the new descriptor; every other one closes at exec)
        posix_spawn_file_actions_adddup2(&fa, b[1], 1);
        char *argv[] = {const_cast<char *>(path.c_str()), nullptr};
        pid_t child = -1;
        const int rc = posix_spawn(&child, path.c_str(), &fa, nullptr, argv, environ);
        posix_spawn_file_actions_destroy(&fa);
        close(a[0]);
        close(b[1]);
        if (rc != 0) { close(a[1]); close(b[0]); return false; }
        pid = child;
        owner = getpid();
        to = a[1];
        from = b[0];
        return true;
    }
    static bool wr(int fd, const void *p, size_t n) {
        const char *c = static_cast<const char *>(p);
        while (n) {
            const ssize_t r = write(fd, c, n);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) return false;
            c += r;
            n -= (size_t)r;
        }
        return true;
    }
    static bool rd(int fd, void *p, size_t n, int timeout_s) {
        char *c = static_cast<char *>(p);
        while (n) {
            struct pollfd pf{fd, POLLIN, 0};
            const int pr = poll(&pf, 1, timeout_s * 1000);
            if (pr < 0 && errno == EINTR) continue;
            if (pr <= 0) return false;
            const ssize_t r = read(fd, c, n);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) return false;
            c += r;
            n -= (size_t)r;
        }
        return true;
    }
};
inline HelperProc &helper_proc() {
    static HelperProc h;
    return h;
}
// 1: compiled by the helper (out filled); 0: no helper - compile in-process; -1: failed (g_last_error set)
inline int compile_in_helper(const std::string &hiprtc_path, const char *src, const char *fname, const std::vector<std::string> &names,
                             const std::vector<std::string> &opts, bool names_are_symbols, CodeObject &out) {
    HelperProc &h = helper_proc();
    if (!h.find() || hiprtc_path.empty()) return 0;
    struct sigaction ign{}, old{};
    ign.sa_handler = SIG_IGN;  // a helper that died must not kill the caller with SIGPIPE when the next request is written
    sigaction(SIGPIPE, &ign, &old);
    struct Restore { struct sigaction *o; ~Restore() { sigaction(SIGPIPE, o, nullptr); } } restore{&old};
    if (!h.start()) return 0;
    std::vector<char> req;
    auto u32 = [&](uint32_t v) { req.insert(req.end(), (char *)&v, (char *)&v + 4); };
    auto u64 = [&](uint64_t v) { req.insert(req.end(), (char *)&v, (char *)&v + 8); };
    auto str = [&](const std::string &s) { u64(s.size()); req.insert(req.end(), s.begin(), s.end()); };
    req.insert(req.end(), "SPCSRQ01", "SPCSRQ01" + 8);
    str(hiprtc_path);
    str(fname);
    u32((uint32_t)opts.size());
    for (const std::string &o : opts) str(o);
    u32((uint32_t)names.size());
    for (const std::string &n : names) str(n);
    u32(names_are_symbols ? 1u : 0u);
    str(src);
    const uint64_t bytes = req.size();
    uint64_t rb = 0;
    std::vector<char> resp;
    int timeout_s = 900;
    if (const char *ev = getenv("SPCIES_HIP_RTC_TIMEOUT_S")) timeout_s = std::max(1, atoi(ev));
    bool ok = HelperProc::wr(h.to, &bytes, 8) && HelperProc::wr(h.to, req.data(), req.size()) && HelperProc::rd(h.from, &rb, 8, timeout_s);
    if (ok && rb >= 8 && rb < (1ull << 31)) {
        resp.resize((size_t)rb);
        ok = HelperProc::rd(h.from, resp.data(), resp.size(), timeout_s);
    } else {
        ok = false;
    }
    if (!ok) {  // the compiler process died (or hung past the limit): this build fails, the caller lives
        int st = 0;
        const pid_t pid = h.pid;
        if (pid > 0 && waitpid(pid, &st, WNOHANG) == 0) { kill(pid, SIGKILL); waitpid(pid, &st, 0); }
        h.pid = -1;
        h.stop();
        if (WIFSIGNALED(st))
            return fail(SPCIES_HIP_EHIP, "hiprtc: the compiler process died with signal %d while compiling %s (the caller's process is unharmed; "
                                         "this variant counts as a failed build)", WTERMSIG(st), fname), -2;
        return fail(SPCIES_HIP_EHIP, "hiprtc: the compiler process ended (status %d) or did not answer within %d s while compiling %s", st, timeout_s, fname), -1;
    }
    size_t at = 0;
    auto g32 = [&]() { uint32_t v = 0; if (at + 4 <= resp.size()) memcpy(&v, resp.data() + at, 4); at += 4; return v; };
    auto gstr = [&]() {
        uint64_t n = 0;
        if (at + 8 <= resp.size()) memcpy(&n, resp.data() + at, 8);
        at += 8;
        std::string s;
        if (at + n <= resp.size()) s.assign(resp.data() + at, (size_t)n);
        at += (size_t)n;
        return s;
    };
    if (resp.size() < 8 || memcmp(resp.data(), "SPCSRS01", 8) != 0) return fail(SPCIES_HIP_EHIP, "hiprtc helper: malformed response"), -1;
    at = 8;
    const uint32_t rc = g32();
    const std::string log = gstr();
    if (rc != 0) return fail(SPCIES_HIP_EHIP, "%.460s", log.c_str()), -1;
    const uint32_t nl = g32();
    out.lowered.clear();
    for (uint32_t i = 0; i < nl; i++) out.lowered.push_back(gstr());
    const std::string code = gstr();
    if (at > resp.size() || code.empty()) return fail(SPCIES_HIP_EHIP, "hiprtc helper: truncated response"), -1;
    out.code.assign(code.begin(), code.end());
    return 1;
}
// path of the libhiprtc the binding resolved (what the helper is told to load)
inline std::string hiprtc_library_path() {
    Hiprtc &rt = hiprtc();
    Dl_info info;
    if (rt.create && dladdr((void *)rt.create, &info) && info.dli_fname) {
        char real[PATH_MAX];
        return realpath(info.dli_fname, real) ? real : info.dli_fname;
    }
    return "";
}

// Compile `src` for gfx950 - or take the code object from the process's cache or the on-disk cache (code_cache.hpp) - and load it
// on the current device: `names` are name expressions (template instantiations) resolved to functions
inline int compile_module(const char *src, const char *fname, const std::vector<std::string> &names, const std::vector<std::string> &extra_opts,
                          hipModule_t *module, hipFunction_t *fns, bool names_are_symbols = false) {
    // names_are_symbols: `names` are extern "C" kernels of the source (a generated program), taken as they are
    std::vector<std::string> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-honor-nans"};
    for (const std::string &e : extra_opts) opts.push_back(e);
    {   // the binding first: the key carries the identity of the compiler that would run
        std::lock_guard<std::mutex> lk(rtc_mutex());
        int rc = hiprtc().open();
        if (rc) return rc;
    }
    std::vector<std::string> key_names = names;
    key_names.push_back(names_are_symbols ? "#symbols" : "#expressions");
    const CacheKey key = make_key(compiler_identity(), fname, key_names, opts, src);
    auto compile = [&](CodeObject &out) -> int {
        std::lock_guard<std::mutex> lk(rtc_mutex());
        Hiprtc &rt = hiprtc();
        {   // the compiler process first (rtc_helper.cpp): 1 = done there, -1 = failed there (not retried in-process: a crash must stay out)
            const int hr = compile_in_helper(hiprtc_library_path(), src, fname, names, opts, names_are_symbols, out);
            if (hr == 1) return 0;
            if (hr == -2) return CodeCache::RC_COMPILER_DIED;  // (the disk cache marks the program: not tried again on this machine)
            if (hr < 0) return SPCIES_HIP_EHIP;
        }
        rt.sync_env();
        void *prog = nullptr;
        if (rt.create(&prog, src, fname, 0, nullptr, nullptr) != 0) return fail(SPCIES_HIP_EHIP, "hiprtcCreateProgram failed");
        for (const std::string &nm : names)
            if (!names_are_symbols && rt.add_name(prog, nm.c_str()) != 0) {
                rt.destroy(&prog);
                return fail(SPCIES_HIP_EHIP, "hiprtcAddNameExpression failed");
            }
        std::vector<const char *> copts;
        for (const std::string &o : opts) copts.push_back(o.c_str());
        if (rt.compile(prog, (int)copts.size(), copts.data()) != 0) {
            size_t ls = 0;
            rt.log_size(prog, &ls);
            std::string lg(ls + 1, '\0');
            if (ls) rt.log(prog, &lg[0]);
            rt.destroy(&prog);
            return fail(SPCIES_HIP_EHIP, "hiprtcCompileProgram failed: %.400s", lg.c_str());
        }
        size_t cs = 0;
        rt.code_size(prog, &cs);
        out.code.resize(cs);
        rt.code(prog, out.code.data());
        for (const std::string &nm : names) {
            const char *ln = names_are_symbols ? nm.c_str() : nullptr;
            if (!names_are_symbols && (rt.lowered(prog, nm.c_str(), &ln) != 0 || !ln)) {
                rt.destroy(&prog);
                return fail(SPCIES_HIP_EHIP, "hiprtcGetLoweredName failed");
            }
            out.lowered.push_back(ln);
        }
        rt.destroy(&prog);
        return 0;
    };
    std::shared_ptr<const CodeObject> co;
    const int rc = CodeCache::instance().get(key, compile, &co);
    if (rc == CodeCache::RC_COMPILER_DIED) return SPCIES_HIP_EHIP;  // (the message is the helper client's)
    if (rc == CodeCache::RC_KNOWN_CRASH)
        return fail(SPCIES_HIP_EHIP, "hiprtc: %s with these options killed the compiler process on this machine before (%s/%s.crashed marks it; delete the file to try again)",
                    fname, CodeCache::disk_dir().c_str(), key.digest.c_str());
    if (rc) return rc;
    if (co->lowered.size() != names.size())
        return fail(SPCIES_HIP_EHIP, "cached code object of %s has %zu kernels, %zu expected", fname, co->lowered.size(), names.size());
    // loading is per device (the caller has made its device current) and needs no lock
    {   // load and pin under the lock unload_module takes (see there)
        std::lock_guard<std::mutex> lk(pins_mutex());
        SPCIES_HIP_CHECK(hipModuleLoadData(module, co->code.data()));
        module_pins()[*module] = co;
    }
    for (size_t i = 0; i < co->lowered.size(); i++) {
        const hipError_t err = hipModuleGetFunction(&fns[i], *module, co->lowered[i].c_str());
        if (err != hipSuccess) {  // a loaded module must not outlive a failed look-up (nobody would unload it)
            unload_module(*module);
            *module = nullptr;
            return fail(SPCIES_HIP_EHIP, "hipModuleGetFunction(%s): %s", co->lowered[i].c_str(), hipGetErrorString(err));
        }
    }
    return 0;
}

}  // namespace rtc
}  // namespace spcies
