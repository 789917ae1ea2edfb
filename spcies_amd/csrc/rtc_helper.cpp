// spcies_rtc_helper - the compiler process of libspcies_hip.so's run-time specialised kernels (rtc_common.hpp, round 5).
//
// Spcies prints one C solver per controller and hands it to `mex` - a compiler in a process of its own.  The HIP platform's
// counterpart is hiprtc, and hiprtc runs INSIDE its caller: a compiler crash (ROCm 7.2's "Rewrite AGPR-Copy-MFMA" pass does crash on
// two admm_r instantiations under -amdgpu-mfma-vgpr-form, DESIGN.md 4.2b''') takes the MATLAB session / the Python process down with it,
// and a process that loaded another ROCm user space first (PyTorch wheels bundle an older libamd_comgr) hands hiprtc THAT compiler.
// This helper is started by the library (posix_spawn, two pipes on stdin / stdout), loads the installation's libhiprtc - the path the
// library resolved, nothing else - and serves compile requests until its stdin closes.  If it dies the library reports a failed build
// (AUTO falls back to the next variant) and starts a fresh one for the next request.  SPCIES_HIP_RTC_ISOLATE=0 compiles in-process.
//
// Protocol (host byte order, one request at a time):
//   request  = u64 bytes | "SPCSRQ01" | str lib | str file name | u32 n_opts, str* | u32 n_names, str* | u32 names_are_symbols | str source
//   response = u64 bytes | "SPCSRS01" | u32 rc | str log-or-error | u32 n_lowered, str* | str code          (str = u64 length + bytes)
#include <dlfcn.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

bool read_all(int fd, void *buf, size_t n) {
    char *p = static_cast<char *>(buf);
    while (n) {
        const ssize_t r = read(fd, p, n);
        if (r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}
bool write_all(int fd, const void *buf, size_t n) {
    const char *p = static_cast<const char *>(buf);
    while (n) {
        const ssize_t r = write(fd, p, n);
        if (r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}
struct Reader {
    const std::vector<char> &b;
    size_t at = 0;
    bool ok = true;
    explicit Reader(const std::vector<char> &buf) : b(buf) {}
    void raw(void *dst, size_t n) {
        if (at + n > b.size()) { ok = false; return; }
        memcpy(dst, b.data() + at, n);
        at += n;
    }
    uint32_t u32() { uint32_t v = 0; raw(&v, 4); return v; }
    uint64_t u64() { uint64_t v = 0; raw(&v, 8); return v; }
    std::string str() {
        const uint64_t n = u64();
        if (!ok || at + n > b.size()) { ok = false; return ""; }
        std::string s(b.data() + at, (size_t)n);
        at += (size_t)n;
        return s;
    }
};
void put_u32(std::vector<char> &o, uint32_t v) { o.insert(o.end(), (char *)&v, (char *)&v + 4); }
void put_u64(std::vector<char> &o, uint64_t v) { o.insert(o.end(), (char *)&v, (char *)&v + 8); }
void put_str(std::vector<char> &o, const std::string &s) { put_u64(o, s.size()); o.insert(o.end(), s.begin(), s.end()); }

struct Rt {
    void *lib = nullptr;
    std::string path;
    int (*create)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*add_name)(void *, const char *) = nullptr;
    int (*compile)(void *, int, const char **) = nullptr;
    int (*lowered)(void *, const char *, const char **) = nullptr;
    int (*code_size)(void *, size_t *) = nullptr;
    int (*code)(void *, char *) = nullptr;
    int (*log_size)(void *, size_t *) = nullptr;
    int (*log)(void *, char *) = nullptr;
    int (*destroy)(void **) = nullptr;
    bool open(const std::string &p, std::string &err) {
        if (lib && p == path) return true;
        lib = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!lib) { err = std::string("cannot load ") + p + ": " + dlerror(); return false; }
        path = p;
#define SYM(field, sym) field = (decltype(field))dlsym(lib, sym)
        SYM(create, "hiprtcCreateProgram"); SYM(add_name, "hiprtcAddNameExpression"); SYM(compile, "hiprtcCompileProgram");
        SYM(lowered, "hiprtcGetLoweredName"); SYM(code_size, "hiprtcGetCodeSize"); SYM(code, "hiprtcGetCode");
        SYM(log_size, "hiprtcGetProgramLogSize"); SYM(log, "hiprtcGetProgramLog"); SYM(destroy, "hiprtcDestroyProgram");
#undef SYM
        if (!create || !add_name || !compile || !lowered || !code_size || !code || !log_size || !log || !destroy) { err = "hiprtc symbols missing"; return false; }
        return true;
    }
};

}  // namespace

int main() {
    Rt rt;
    const int in = 0, out = dup(1);
    if (out < 0) return 2;
    dup2(2, 1);  // whatever the compiler prints to stdout must not corrupt the response stream
    // the library's process may hold GPU device files and sockets open without O_CLOEXEC: this process compiles, it must not keep them alive (nor
    // count as a user of the GPU)
    for (int fd = 3; fd < 4096; fd++)
        if (fd != out) close(fd);
    for (;;) {
        uint64_t bytes = 0;
        if (!read_all(in, &bytes, 8)) return 0;  // the library closed the pipe: done
        if (bytes < 8 || bytes > (1ull << 31)) return 3;
        std::vector<char> req((size_t)bytes);
        if (!read_all(in, req.data(), req.size())) return 3;
        Reader r(req);
        char magic[8];
        r.raw(magic, 8);
        std::vector<char> resp;
        resp.insert(resp.end(), "SPCSRS01", "SPCSRS01" + 8);
        auto fail = [&](const std::string &msg) {
            put_u32(resp, 1);
            put_str(resp, msg);
            put_u32(resp, 0);
            put_str(resp, "");
        };
        const std::string lib = r.str(), fname = r.str();
        std::vector<std::string> opts(r.u32()), names;
        for (auto &o : opts) o = r.str();
        names.resize(r.ok ? r.u32() : 0);
        for (auto &n : names) n = r.str();
        const bool symbols = r.u32() != 0;
        const std::string src = r.str();
        std::string err;
        if (!r.ok || memcmp(magic, "SPCSRQ01", 8) != 0) {
            fail("malformed request");
        } else if (src.find("//SPCIES_RTC_HELPER_SELFTEST_ABORT") != std::string::npos) {
            abort();  // tests/test_rtc_isolation.py: a compiler that dies mid-request
        } else if (!rt.open(lib, err)) {
            fail(err);
        } else {
            void *prog = nullptr;
            bool ok = rt.create(&prog, src.c_str(), fname.c_str(), 0, nullptr, nullptr) == 0;
            if (!ok) fail("hiprtcCreateProgram failed");
            for (size_t i = 0; ok && !symbols && i < names.size(); i++)
                if (rt.add_name(prog, names[i].c_str()) != 0) { fail("hiprtcAddNameExpression failed"); ok = false; }
            if (ok) {
                std::vector<const char *> copts;
                for (const std::string &o : opts) copts.push_back(o.c_str());
                if (rt.compile(prog, (int)copts.size(), copts.data()) != 0) {
                    size_t ls = 0;
                    rt.log_size(prog, &ls);
                    std::string lg(ls + 1, '\0');
                    if (ls) rt.log(prog, &lg[0]);
                    fail(std::string("hiprtcCompileProgram failed: ") + lg.c_str());
                    ok = false;
                }
            }
            if (ok) {
                size_t cs = 0;
                rt.code_size(prog, &cs);
                std::string code(cs, '\0');
                rt.code(prog, &code[0]);
                std::vector<std::string> low;
                for (const std::string &nm : names) {
                    const char *ln = symbols ? nm.c_str() : nullptr;
                    if (!symbols && (rt.lowered(prog, nm.c_str(), &ln) != 0 || !ln)) { ok = false; break; }
                    low.push_back(ln);
                }
                if (!ok) {
                    fail("hiprtcGetLoweredName failed");
                } else {
                    put_u32(resp, 0);
                    put_str(resp, "");
                    put_u32(resp, (uint32_t)low.size());
                    for (const std::string &l : low) put_str(resp, l);
                    put_str(resp, code);
                }
            }
            if (prog) rt.destroy(&prog);
        }
        const uint64_t rb = resp.size();
        if (!write_all(out, &rb, 8) || !write_all(out, resp.data(), resp.size())) return 4;
    }
}
