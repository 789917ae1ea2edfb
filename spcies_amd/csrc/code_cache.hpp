// Code objects of the run-time specialised kernels: a bounded in-memory cache in front of an on-disk cache.
//
// Spcies prints one C solver per controller and compiles it once (spcies_gen_controller -> mex); the HIP platform's run-time
// specialised kernels (MFMA4 shapes, MFMA4R, BSP block programs, FUSED shapes) cost 1-17 s of hiprtc per controller instead, and
// without this file every rank of a multi-GPU job, every MATLAB session and every test process paid that again.
//
//  * key    = SHA-256 over (compiler identity, file name, name expressions, options, source text).  The memory cache is keyed by
//             the digest and remembers the key's length and a second, independent 64-bit hash: a hit is accepted only when all
//             three agree (the full text - hundreds of KB for a generated block program - is not kept).
//  * memory = digest -> shared_ptr<CodeObject>, least-recently-used order, capped in bytes (SPCIES_HIP_RTC_CACHE_MB, default 256);
//             a code object stays alive while a loaded module pins it (rtc::unload_module drops the pin).
//  * disk   = <dir>/<digest>.hsaco, dir = $SPCIES_HIP_CACHE_DIR | $XDG_CACHE_HOME/spcies_hip | $HOME/.cache/spcies_hip
//             (SPCIES_HIP_DISK_CACHE=0 switches it off; an unusable directory does too, silently).  A file is written under a
//             temporary name and rename()d into place; look-up, compilation and write of one digest run under flock() on
//             <digest>.lock, so N processes that need the same program compile it once and the others read the file; the writer
//             unlinks the lock file (waiters re-check the inode).  Capped in bytes (SPCIES_HIP_DISK_CACHE_MB, default 4096, 0 = no
//             cap): after a write the least recently used files go until 80 % of the cap is left (prune_disk).
//  * the compiler runs outside the cache's own lock: look-ups and statistics never wait for a compilation; two threads that ask
//             for the same digest share one compilation (the second waits for the first one's result).
// Nothing here touches HIP or hiprtc: the compilation is a callback (tests/test_rtc_disk_cache.py drives it with a fake one).
#pragma once
#include <dirent.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace spcies {
namespace rtc {

struct CodeObject {
    std::vector<char> code;
    std::vector<std::string> lowered;
    size_t bytes() const {
        size_t b = code.size();
        for (const std::string &s : lowered) b += s.size();
        return b;
    }
};

// ---- SHA-256 (FIPS 180-4) -----------------------------------------------------------------------------------------------------
struct Sha256 {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint8_t buf[64];
    size_t fill = 0;
    uint64_t total = 0;
    static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t *p) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
            0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
            0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
            0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
            0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
            0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
        }
        h[0] += a, h[1] += b, h[2] += c, h[3] += d, h[4] += e, h[5] += f, h[6] += g, h[7] += hh;
    }
    void update(const void *data, size_t n) {
        const uint8_t *p = static_cast<const uint8_t *>(data);
        total += n;
        while (n) {
            const size_t take = std::min(n, (size_t)64 - fill);
            memcpy(buf + fill, p, take);
            fill += take, p += take, n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    std::string hex() {
        const uint64_t bits = total * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t len[8];
        for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(len, 8);
        char out[65];
        for (int i = 0; i < 8; i++) snprintf(out + 8 * i, 9, "%08x", h[i]);
        return std::string(out, 64);
    }
};

inline uint64_t fnv1a64(const void *data, size_t n, uint64_t seed = 0xcbf29ce484222325ull) {
    const uint8_t *p = static_cast<const uint8_t *>(data);
    uint64_t x = seed;
    for (size_t i = 0; i < n; i++) x = (x ^ p[i]) * 0x100000001b3ull;
    return x;
}

// what identifies one compilation
struct CacheKey {
    std::string digest;  // SHA-256, hex
    uint64_t length = 0, check = 0;
};

inline CacheKey make_key(const std::string &compiler_id, const char *fname, const std::vector<std::string> &names,
                         const std::vector<std::string> &opts, const char *src) {
    Sha256 sh;
    uint64_t len = 0, chk = 0xcbf29ce484222325ull;
    auto feed = [&](const void *p, size_t n) {
        sh.update(p, n);
        chk = fnv1a64(p, n, chk);
        len += n;
    };
    auto field = [&](const std::string &s) {
        const uint64_t n = s.size();
        feed(&n, sizeof n);  // length-prefixed: ("ab", "c") and ("a", "bc") differ
        feed(s.data(), s.size());
    };
    field("spcies-hip code object v1 gfx950");
    field(compiler_id);
    field(fname);
    { const uint64_t n = names.size(); feed(&n, sizeof n); }
    for (const std::string &s : names) field(s);
    { const uint64_t n = opts.size(); feed(&n, sizeof n); }
    for (const std::string &s : opts) field(s);
    field(src);
    CacheKey k;
    k.digest = sh.hex();
    k.length = len;
    k.check = chk;
    return k;
}

struct CacheStats {
    long mem_hits = 0, disk_hits = 0, compiles = 0, evictions = 0, disk_writes = 0, disk_errors = 0;
};

class CodeCache {
  public:
    using Compile = std::function<int(CodeObject &out)>;  // 0 on success; the error text is the callback's business

    static CodeCache &instance() {
        static CodeCache c;
        return c;
    }

    // the code object for `key`: from memory, from disk, or from `compile` (then stored in both).  *source (optional): 0 memory,
    // 1 disk, 2 compiled.
    int get(const CacheKey &key, const Compile &compile, std::shared_ptr<const CodeObject> *out, int *source = nullptr) {
        const bool use_mem = !getenv("SPCIES_HIP_RTC_NOCACHE");
        std::shared_ptr<InFlight> mine, theirs;
        {
            std::unique_lock<std::mutex> lk(mu_);
            if (use_mem) {
                auto it = map_.find(key.digest);
                if (it != map_.end() && it->second.length == key.length && it->second.check == key.check) {
                    lru_.splice(lru_.begin(), lru_, it->second.pos);
                    *out = it->second.co;
                    stats_.mem_hits++;
                    if (source) *source = 0;
                    return 0;
                }
                auto fl = inflight_.find(key.digest);
                if (fl != inflight_.end()) theirs = fl->second;
            }
            if (!theirs) {
                mine = std::make_shared<InFlight>();
                if (use_mem) inflight_[key.digest] = mine;
            }
        }
        if (theirs) {  // another thread of this process is producing it: share the outcome
            std::unique_lock<std::mutex> lk(theirs->mu);
            theirs->cv.wait(lk, [&] { return theirs->done; });
            if (theirs->rc == 0) {
                *out = theirs->co;
                std::lock_guard<std::mutex> g(mu_);
                stats_.mem_hits++;
                if (source) *source = 0;
                return 0;
            }
            // the other thread failed: its error text is thread-local over there - compile again here to report our own
            return get_uncached(key, compile, out, source);
        }
        int src = 2;
        std::shared_ptr<const CodeObject> co;
        const int rc = produce(key, compile, &co, &src);
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (rc == 0 && use_mem) insert_locked(key, co);
            if (use_mem) inflight_.erase(key.digest);
        }
        {
            std::lock_guard<std::mutex> lk(mine->mu);
            mine->done = true, mine->rc = rc, mine->co = co;
        }
        mine->cv.notify_all();
        if (rc) return rc;
        *out = co;
        if (source) *source = src;
        return 0;
    }

    CacheStats stats() {
        std::lock_guard<std::mutex> lk(mu_);
        return stats_;
    }
    size_t resident_bytes() {
        std::lock_guard<std::mutex> lk(mu_);
        return bytes_;
    }
    size_t resident_entries() {
        std::lock_guard<std::mutex> lk(mu_);
        return map_.size();
    }
    void clear_memory() {  // (tests: force the next look-up to the disk)
        std::lock_guard<std::mutex> lk(mu_);
        map_.clear();
        lru_.clear();
        bytes_ = 0;
    }

    // return codes a compile callback / get() may use next to the caller's own: the compiler process died on this program (the callback says so;
    // the disk cache then remembers it), and "it did before" (get() says so without compiling)
    static constexpr int RC_COMPILER_DIED = -7001, RC_KNOWN_CRASH = -7002;
    // directory of the disk cache ("" = off); created on first use
    static std::string disk_dir() {
        const char *sw = getenv("SPCIES_HIP_DISK_CACHE");
        if (sw && sw[0] == '0') return "";
        std::string dir;
        if (const char *d = getenv("SPCIES_HIP_CACHE_DIR")) dir = d;
        else if (const char *x = getenv("XDG_CACHE_HOME"); x && *x) dir = std::string(x) + "/spcies_hip";
        else if (const char *h = getenv("HOME"); h && *h) dir = std::string(h) + "/.cache/spcies_hip";
        if (dir.empty()) return "";
        // mkdir -p
        for (size_t i = 1; i <= dir.size(); i++)
            if (i == dir.size() || dir[i] == '/') {
                const std::string part = dir.substr(0, i);
                if (mkdir(part.c_str(), 0700) != 0 && errno != EEXIST) return "";
            }
        if (access(dir.c_str(), W_OK | X_OK) != 0) return "";
        return dir;
    }

  private:
    struct Entry {
        std::shared_ptr<const CodeObject> co;
        uint64_t length, check;
        std::list<std::string>::iterator pos;
    };
    struct InFlight {
        std::mutex mu;
        std::condition_variable cv;
        bool done = false;
        int rc = 0;
        std::shared_ptr<const CodeObject> co;
    };
    std::mutex mu_;
    std::map<std::string, Entry> map_;
    std::list<std::string> lru_;  // front = most recently used
    std::map<std::string, std::shared_ptr<InFlight>> inflight_;
    size_t bytes_ = 0;
    CacheStats stats_;

    static size_t cap_bytes() {
        const char *ev = getenv("SPCIES_HIP_RTC_CACHE_MB");
        const long mb = ev ? atol(ev) : 256;
        return (size_t)(mb < 0 ? 0 : mb) << 20;
    }
    void insert_locked(const CacheKey &key, const std::shared_ptr<const CodeObject> &co) {
        auto old = map_.find(key.digest);
        if (old != map_.end()) {
            bytes_ -= old->second.co->bytes();
            lru_.erase(old->second.pos);
            map_.erase(old);
        }
        lru_.push_front(key.digest);
        map_[key.digest] = Entry{co, key.length, key.check, lru_.begin()};
        bytes_ += co->bytes();
        const size_t cap = cap_bytes();
        while (bytes_ > cap && lru_.size() > 1) {  // the newest entry always stays
            auto victim = map_.find(lru_.back());
            bytes_ -= victim->second.co->bytes();
            map_.erase(victim);
            lru_.pop_back();
            stats_.evictions++;
        }
    }
    int get_uncached(const CacheKey &key, const Compile &compile, std::shared_ptr<const CodeObject> *out, int *source) {
        int src = 2;
        std::shared_ptr<const CodeObject> co;
        const int rc = produce(key, compile, &co, &src);
        if (rc) return rc;
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!getenv("SPCIES_HIP_RTC_NOCACHE")) insert_locked(key, co);
        }
        *out = co;
        if (source) *source = src;
        return 0;
    }

    // ---- disk ------------------------------------------------------------------------------------------------------------------
    // file: "SPCSCO01" | u64 key length | u64 key check | u32 n_names | { u32 len, bytes } * | u64 code bytes | code | u64 fnv of all before
    static bool read_file(const std::string &path, const CacheKey &key, CodeObject &co) {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return false;
        std::vector<char> all;
        char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) all.insert(all.end(), chunk, chunk + got);
        fclose(f);
        if (all.size() < 8 + 8 + 8 + 4 + 8 + 8) return false;
        const size_t body = all.size() - 8;
        uint64_t sum;
        memcpy(&sum, all.data() + body, 8);
        if (sum != fnv1a64(all.data(), body)) return false;
        size_t at = 0;
        auto take = [&](void *dst, size_t n) {
            if (at + n > body) return false;
            memcpy(dst, all.data() + at, n);
            at += n;
            return true;
        };
        char magic[8];
        uint64_t length, check, code_bytes;
        uint32_t n_names;
        if (!take(magic, 8) || memcmp(magic, "SPCSCO01", 8) != 0) return false;
        if (!take(&length, 8) || !take(&check, 8) || length != key.length || check != key.check) return false;
        if (!take(&n_names, 4) || n_names > 64) return false;
        co.lowered.clear();
        for (uint32_t i = 0; i < n_names; i++) {
            uint32_t len;
            if (!take(&len, 4) || at + len > body) return false;
            co.lowered.emplace_back(all.data() + at, len);
            at += len;
        }
        if (!take(&code_bytes, 8) || at + code_bytes != body) return false;
        co.code.assign(all.begin() + at, all.begin() + at + code_bytes);
        return true;
    }
    static bool write_file(const std::string &dir, const std::string &path, const CacheKey &key, const CodeObject &co) {
        std::vector<char> all;
        auto put = [&](const void *p, size_t n) { all.insert(all.end(), static_cast<const char *>(p), static_cast<const char *>(p) + n); };
        put("SPCSCO01", 8);
        put(&key.length, 8);
        put(&key.check, 8);
        const uint32_t n_names = (uint32_t)co.lowered.size();
        put(&n_names, 4);
        for (const std::string &s : co.lowered) {
            const uint32_t len = (uint32_t)s.size();
            put(&len, 4);
            put(s.data(), len);
        }
        const uint64_t code_bytes = co.code.size();
        put(&code_bytes, 8);
        put(co.code.data(), co.code.size());
        const uint64_t sum = fnv1a64(all.data(), all.size());
        put(&sum, 8);
        char tmpl[4096];
        snprintf(tmpl, sizeof tmpl, "%s/.tmp-%ld-XXXXXX", dir.c_str(), (long)getpid());
        const int fd = mkstemp(tmpl);
        if (fd < 0) return false;
        size_t done = 0;
        while (done < all.size()) {
            const ssize_t w = write(fd, all.data() + done, all.size() - done);
            if (w <= 0) { close(fd); unlink(tmpl); return false; }
            done += (size_t)w;
        }
        if (fsync(fd) != 0 || close(fd) != 0 || rename(tmpl, path.c_str()) != 0) { unlink(tmpl); return false; }
        return true;
    }
    // Size cap of the disk cache: SPCIES_HIP_DISK_CACHE_MB (default 4096; 0 = no cap).  Called after a write: when the *.hsaco files
    // of the directory exceed the cap, the least recently used (mtime: set at write and at every hit) are deleted until 80 % of it is
    // left; stale temporaries and lock files (older than an hour) go too.  One pruner at a time (a directory lock, not waited for);
    // a reader that loses its file to the pruner recompiles - a code object is never read half.
    static void prune_disk(const std::string &dir) {
        long cap_mb = 4096;
        if (const char *ev = getenv("SPCIES_HIP_DISK_CACHE_MB")) cap_mb = atol(ev);
        if (cap_mb <= 0) return;
        const std::string plock = dir + "/.prune.lock";
        const int fd = open(plock.c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
        if (fd < 0) return;
        if (flock(fd, LOCK_EX | LOCK_NB) != 0) { close(fd); return; }
        struct Item { std::string path; off_t size; time_t mtime; };
        std::vector<Item> files;
        unsigned long long total = 0;
        const time_t now = time(nullptr);
        if (DIR *d = opendir(dir.c_str())) {
            while (struct dirent *de = readdir(d)) {
                const std::string name = de->d_name;
                if (name == "." || name == ".." || name == ".prune.lock") continue;
                const std::string path = dir + "/" + name;
                struct stat st;
                if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) continue;
                const bool code = name.size() > 6 && name.compare(name.size() - 6, 6, ".hsaco") == 0;
                if (code) {
                    files.push_back({path, st.st_size, st.st_mtime});
                    total += (unsigned long long)st.st_size;
                } else if (now - st.st_mtime > 3600 && (name.compare(0, 5, ".tmp-") == 0 || (name.size() > 5 && name.compare(name.size() - 5, 5, ".lock") == 0))) {
                    unlink(path.c_str());  // a writer that died, or a lock nobody holds (a holder's inode check re-opens the name)
                }
            }
            closedir(d);
        }
        const unsigned long long cap = (unsigned long long)cap_mb << 20;
        if (total > cap) {
            std::sort(files.begin(), files.end(), [](const Item &a, const Item &b) { return a.mtime < b.mtime; });
            for (const Item &it : files) {
                if (total <= cap / 5 * 4) break;
                if (unlink(it.path.c_str()) == 0) total -= (unsigned long long)it.size;
            }
        }
        flock(fd, LOCK_UN);
        close(fd);
    }
    // memory missed: disk, else compile (under the digest's file lock when the disk cache is on)
    int produce(const CacheKey &key, const Compile &compile, std::shared_ptr<const CodeObject> *out, int *source) {
        const std::string dir = disk_dir();
        auto fresh = std::make_shared<CodeObject>();
        if (dir.empty()) {
            const int rc = compile(*fresh);
            if (rc) return rc;
            std::lock_guard<std::mutex> lk(mu_);
            stats_.compiles++;
            *out = fresh, *source = 2;
            return 0;
        }
        const std::string path = dir + "/" + key.digest + ".hsaco", lock_path = dir + "/" + key.digest + ".lock";
        // a program that killed the compiler process on this machine before (same compiler identity, options and text: the digest) is not
        // tried again - <digest>.crashed marks it; delete the file to retry
        const std::string crashed = dir + "/" + key.digest + ".crashed";
        if (access(crashed.c_str(), F_OK) == 0) return RC_KNOWN_CRASH;
        if (read_file(path, key, *fresh)) {  // the common warm case takes no lock: a file is complete once it has its name
            (void)utimensat(AT_FDCWD, path.c_str(), nullptr, 0);  // "used now": prune_disk deletes the least recently used first
            std::lock_guard<std::mutex> lk(mu_);
            stats_.disk_hits++;
            *out = fresh, *source = 1;
            return 0;
        }
        // The digest's lock file.  It is unlinked by the holder that wrote the code object (so the directory does not collect one
        // lock per controller ever compiled); a waiter that gets the lock on an inode that has lost its name opens the name again.
        int lock_fd = -1;
        for (int tries = 0; tries < 16; tries++) {
            lock_fd = open(lock_path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
            if (lock_fd < 0) break;
            while (flock(lock_fd, LOCK_EX) != 0 && errno == EINTR) {}
            struct stat a, b;
            if (fstat(lock_fd, &a) == 0 && stat(lock_path.c_str(), &b) == 0 && a.st_ino == b.st_ino && a.st_dev == b.st_dev) break;
            close(lock_fd);  // unlinked (or replaced) while we waited
            lock_fd = -1;
        }
        int rc = 0;
        bool wrote = false;
        if (lock_fd >= 0 && read_file(path, key, *fresh)) {  // somebody compiled it while we waited for the lock
            std::lock_guard<std::mutex> lk(mu_);
            stats_.disk_hits++;
            *source = 1;
        } else {
            *fresh = CodeObject();  // (a failed read may have filled part of it)
            rc = compile(*fresh);
            if (rc == RC_COMPILER_DIED) {
                const int fd = open(crashed.c_str(), O_CREAT | O_WRONLY | O_CLOEXEC, 0600);
                if (fd >= 0) close(fd);
            }
            if (rc == 0) {
                wrote = write_file(dir, path, key, *fresh);
                std::lock_guard<std::mutex> lk(mu_);
                stats_.compiles++;
                (wrote ? stats_.disk_writes : stats_.disk_errors)++;
                *source = 2;
            }
        }
        if (lock_fd >= 0) {
            if (wrote) unlink(lock_path.c_str());  // still holding it: the next waiter re-opens the name and finds the file
            flock(lock_fd, LOCK_UN);
            close(lock_fd);
        }
        if (wrote) prune_disk(dir);
        if (rc) return rc;
        *out = fresh;
        return 0;
    }
};

}  // namespace rtc
}  // namespace spcies
