// Host-side thread stress of libspcies_hip.so for the sanitizer builds (tools/sanitize.sh; CPU only - no GPU sanitizer on this pool).
//
// What runs without a device: the code-object cache (code_cache.hpp: memory LRU, on-disk files, flock, shared in-flight compilations,
// pruning) through the library's stand-in compiler hook, the blob parser and the host-side packers behind spcies_hip_create /
// spcies_hip_create_multi (they run before the first device call and end with ENODEV here), the thread-local last-error string and
// the sharding arithmetic.  Every one of them from several threads at once, on overlapping keys.
//
// usage: sanitize_stress <lib.so> <cache_dir> <threads> <rounds> [blob files ...]
#include <dlfcn.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/spcies_hip.h"  // the prototypes only: every entry point is bound with dlsym (the sanitizer build is not the linked one)
static decltype(&spcies_hip_rtc_cache_selftest) p_selftest;
static decltype(&spcies_hip_rtc_cache_stats_ex) p_stats;
static decltype(&spcies_hip_create) p_create;
static decltype(&spcies_hip_destroy) p_destroy;
static decltype(&spcies_hip_create_multi) p_create_multi;
static decltype(&spcies_hip_multi_destroy) p_multi_destroy;
static decltype(&spcies_hip_shard_range) p_shard;
static decltype(&spcies_hip_last_error) p_last_error;

template <class F>
static void bind(void *lib, const char *name, F &f) {
    f = (F)dlsym(lib, name);
    if (!f) { fprintf(stderr, "missing symbol %s\n", name); exit(2); }
}

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s lib.so cache_dir threads rounds [blobs...]\n", argv[0]); return 2; }
    setenv("SPCIES_HIP_CACHE_DIR", argv[2], 1);
    setenv("SPCIES_HIP_DISK_CACHE_MB", "1", 1);   // the pruner runs
    setenv("SPCIES_HIP_RTC_CACHE_MB", "1", 1);    // the memory cache evicts
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    bind(lib, "spcies_hip_rtc_cache_selftest", p_selftest);
    bind(lib, "spcies_hip_rtc_cache_stats_ex", p_stats);
    bind(lib, "spcies_hip_create", p_create);
    bind(lib, "spcies_hip_destroy", p_destroy);
    bind(lib, "spcies_hip_create_multi", p_create_multi);
    bind(lib, "spcies_hip_multi_destroy", p_multi_destroy);
    bind(lib, "spcies_hip_shard_range", p_shard);
    bind(lib, "spcies_hip_last_error", p_last_error);
    const int T = atoi(argv[3]), R = atoi(argv[4]);
    std::vector<std::string> blobs;
    for (int i = 5; i < argc; i++) {
        std::ifstream f(argv[i], std::ios::binary);
        blobs.emplace_back((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    }
    std::atomic<long> bad{0}, creates{0}, compiles_seen{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            unsigned seed = 1234u + 77u * (unsigned)t;
            auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
            for (int r = 0; r < R; r++) {
                // (a) the cache: a handful of keys shared by all threads, some big (eviction / pruning), some slow (in-flight sharing)
                const int keyid = (int)(rnd() % 12);
                std::string text = "stress key " + std::to_string(keyid) + " ";
                if (keyid % 3 == 0) text.append(400000, (char)('a' + keyid));
                int src = -1;
                unsigned long long chk = 0;
                const int rc = p_selftest(text.c_str(), keyid % 4 == 1 ? 3 : 0, (int)(rnd() % 7 == 0), &src, &chk);
                if (rc != 0 || src < 0 || src > 2) bad++;
                if (src == 2) compiles_seen++;
                // the same key twice in a row must give the same bytes
                unsigned long long chk2 = 0;
                if (p_selftest(text.c_str(), 0, 0, &src, &chk2) != 0 || chk2 != chk) bad++;
                // a failing compilation is an error for its caller only, and is not cached
                if (rnd() % 16 == 0) {
                    const std::string ft = "failing " + std::to_string(keyid);
                    if (p_selftest(ft.c_str(), -1, 0, &src, &chk) == 0) bad++;
                    if (!strstr(p_last_error(), "stand-in compiler")) bad++;
                }
                // (b) the parser and the host packers behind create / create_multi (ENODEV = -2 after a clean parse; anything but a crash for a mutant)
                if (!blobs.empty()) {
                    const std::string &b = blobs[rnd() % blobs.size()];
                    spcies_hip_handle h = nullptr;
                    int rc2 = p_create(b.data(), b.size(), 0, &h);
                    if (rc2 == 0) p_destroy(h);
                    else if (rc2 != -2) bad++;
                    std::string m = b;
                    for (int j = 0; j < 6; j++) m[rnd() % m.size()] = (char)rnd();
                    h = nullptr;
                    rc2 = p_create(m.data(), m.size(), 0, &h);
                    if (rc2 == 0) p_destroy(h);
                    const int ids[2] = {0, 0};
                    spcies_hip_multi_handle mh = nullptr;
                    rc2 = p_create_multi(b.data(), b.size(), ids, 2, &mh);
                    if (rc2 == 0) p_multi_destroy(mh);
                    creates += 3;
                }
                // (c) sharding arithmetic
                long lo = 0, cnt = 0, total = (long)(rnd() % 100000), sum = 0;
                const int G = 1 + (int)(rnd() % 8);
                for (int g = 0; g < G; g++) {
                    if (p_shard(total, G, g, &lo, &cnt) != 0 || lo != sum) bad++;
                    sum += cnt;
                }
                if (sum != total) bad++;
            }
        });
    for (auto &x : th) x.join();
    long st[6] = {0, 0, 0, 0, 0, 0};
    p_stats(st, 6);
    printf("sanitize_stress: %d threads x %d rounds, %ld create calls; cache: mem_hits %ld disk_hits %ld compiles %ld evictions %ld disk_writes %ld disk_errors %ld; "
           "inconsistencies %ld\n", T, R, creates.load(), st[0], st[1], st[2], st[3], st[4], st[5], bad.load());
    return bad.load() ? 1 : 0;
}
