// One-process multi-device entry points of the C-ABI (include/spcies_hip.h): what a mex gateway or a plain-C caller
// (examples/cl_in_C/main_cl_in_C.c:103 use case, batched) binds to use every GPU of a node without torchrun.
// create(blob, bytes, device_ids, n_dev), one host thread per device, contiguous shards of the host
// batch, no collective (the instances are independent: code_laxMPC_ADMM_C.c:58-77 keeps all state on the stack).
// Built on the single-device entry points only: one spcies_hip_handle per device, created from the same blob.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spcies_hip.h"
#include "common.hpp"

// One resident host thread per device beyond the first (the caller's thread drives device 0): a closed-loop caller solves a small batch
// every sample time, and starting G - 1 threads per call (rounds 1-3) cost more than the 50 us such a solve takes on the device.
// A job is posted under the mutex as a generation number; worker g runs job(g) and reports back.  Calls on ONE multi handle are
// serialised (call_mu): the single-device handles underneath own one stream and one scratch allocation each.
struct spcies_hip_multi_s {
    std::vector<spcies_hip_handle> h;
    std::vector<int> dev;
    std::vector<std::thread> workers;
    std::mutex mu, call_mu;
    std::condition_variable cv_job, cv_done;
    std::function<void(int)> job;
    unsigned long generation = 0;
    int pending = 0;
    bool stop = false;

    void start_workers() {
        for (int g = 1; g < (int)h.size(); g++)
            workers.emplace_back([this, g] {
                unsigned long seen = 0;
                for (;;) {
                    std::function<void(int)> fn;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_job.wait(lk, [&] { return stop || generation != seen; });
                        if (stop) return;
                        seen = generation;
                        fn = job;
                    }
                    fn(g);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
    }
    // runs fn(0) on the calling thread and fn(g) on worker g, returns when all are done
    void run_all(const std::function<void(int)> &fn) {
        if (workers.empty()) {
            fn(0);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            job = fn;
            pending = (int)workers.size();
            generation++;
        }
        cv_job.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    void stop_workers() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_job.notify_all();
        for (std::thread &t : workers) t.join();
        workers.clear();
    }
};

extern "C" {

int spcies_hip_shard_range(long B, int n_shards, int shard, long *begin, long *count) {
    if (B < 0 || n_shards <= 0 || shard < 0 || shard >= n_shards || !begin || !count)
        return spcies::fail(SPCIES_HIP_EINVAL, "shard_range: B >= 0, 0 <= shard < n_shards");
    const long base = B / n_shards, rem = B % n_shards;  // sizes differ by at most one, larger shards first
    *begin = shard * base + std::min<long>(shard, rem);
    *count = base + (shard < rem ? 1 : 0);
    return 0;
}

int spcies_hip_create_multi(const void *blob, size_t bytes, const int *device_ids, int n_dev, spcies_hip_multi_handle *out) {
    if (!out) return spcies::fail(SPCIES_HIP_EINVAL, "NULL out");
    *out = nullptr;
    if (n_dev <= 0) {  // every visible device
        int cnt = 0;
        int rc = spcies_hip_device_count(&cnt);
        if (rc) return rc;
        if (cnt <= 0) return spcies::fail(SPCIES_HIP_ENODEV, "no HIP device");
        n_dev = cnt;
        device_ids = nullptr;
    }
    // The first handle is built on the calling thread: whatever it specialises at run time (hiprtc: MFMA4 shapes, BSP / FUSED /
    // MFMA4R programs, seconds each) lands in the process-wide code-object cache (rtc_common.hpp).  The others are then built
    // side by side, one host thread per device, and take their code objects from the cache - N devices cost one compilation.
    spcies_hip_multi_s *m = new spcies_hip_multi_s;
    m->h.assign(n_dev, nullptr);
    m->dev.resize(n_dev);
    for (int i = 0; i < n_dev; i++) m->dev[i] = device_ids ? device_ids[i] : i;
    std::vector<int> rcs(n_dev, 0);
    std::vector<std::string> errs(n_dev);
    auto build = [&](int i) {
        rcs[i] = spcies_hip_create(blob, bytes, m->dev[i], &m->h[i]);
        if (rcs[i]) errs[i] = spcies_hip_last_error();  // thread-local: carried to the caller below
    };
    build(0);
    if (rcs[0] == 0) {
        std::vector<std::thread> th;
        for (int i = 1; i < n_dev; i++) th.emplace_back(build, i);
        for (std::thread &t : th) t.join();
    }
    for (int i = 0; i < n_dev; i++)
        if (rcs[i]) {
            const int rc = rcs[i];
            const std::string why = errs[i];
            for (spcies_hip_handle x : m->h)
                if (x) spcies_hip_destroy(x);
            const int d = m->dev[i];
            delete m;
            return spcies::fail(rc, "device %d (handle %d of %d): %s", d, i, n_dev, why.c_str());
        }
    m->start_workers();
    *out = m;
    return 0;
}

int spcies_hip_multi_destroy(spcies_hip_multi_handle m) {
    if (!m) return 0;
    m->stop_workers();
    int rc = 0;
    for (spcies_hip_handle x : m->h) {
        int r = spcies_hip_destroy(x);
        if (r) rc = r;
    }
    delete m;
    return rc;
}

int spcies_hip_multi_count(spcies_hip_multi_handle m, int *n_dev) {
    if (!m || !n_dev) return spcies::fail(SPCIES_HIP_EINVAL, "NULL argument");
    *n_dev = (int)m->h.size();
    return 0;
}

int spcies_hip_multi_get(spcies_hip_multi_handle m, int i, spcies_hip_handle *single) {
    if (!m || !single || i < 0 || i >= (int)m->h.size()) return spcies::fail(SPCIES_HIP_EINVAL, "multi_get: index out of range");
    *single = m->h[i];
    return 0;
}

int spcies_hip_multi_set_variant(spcies_hip_multi_handle m, int variant) {
    if (!m) return spcies::fail(SPCIES_HIP_EINVAL, "NULL handle");
    for (spcies_hip_handle x : m->h) {
        int rc = spcies_hip_set_variant(x, variant);
        if (rc) return rc;
    }
    return 0;
}

int spcies_hip_multi_set_exit(spcies_hip_multi_handle m, int k_max, double tol) {
    if (!m) return spcies::fail(SPCIES_HIP_EINVAL, "NULL handle");
    for (spcies_hip_handle x : m->h) {
        int rc = spcies_hip_set_exit(x, k_max, tol);
        if (rc) return rc;
    }
    return 0;
}

int spcies_hip_multi_solve_batch_ex(spcies_hip_multi_handle m, const double *x0, const double *xr, const double *ur, int ref_stride,
                                    const double *extra, int extra_stride, long extra_width, long B, double *u, int *k, int *e_flag,
                                    double *const *fields, int n_fields, spcies_hip_timing *timing) {
    if (!m || m->h.empty()) return spcies::fail(SPCIES_HIP_EINVAL, "NULL handle");
    if (B < 0) return spcies::fail(SPCIES_HIP_EINVAL, "negative batch");
    if (timing) *timing = spcies_hip_timing{0, 0, 0, 0};
    if (B == 0) return 0;
    // (checked here: a NULL would turn into a non-NULL pointer once a shard's offset is added)
    if (!x0 || !xr || !ur || !u || !k || !e_flag) return spcies::fail(SPCIES_HIP_EINVAL, "NULL buffer");
    spcies_hip_info info;
    int rc = spcies_hip_get_info(m->h[0], &info);
    if (rc) return rc;
    int nf = 0, dims[8] = {0};
    const char *names[8];
    rc = spcies_hip_get_sol_layout(m->h[0], &nf, dims, names);
    if (rc) return rc;
    if (fields && n_fields != nf) return spcies::fail(SPCIES_HIP_EINVAL, "this solver's record has %d fields", nf);
    if (extra && extra_stride) {  // a shard's offset into `extra` is the width the single-device entry point derives for itself
        long w = 1;
        rc = spcies_hip_get_extra_width(m->h[0], &w);
        if (rc) return rc;
        if (extra_width <= 0) extra_width = w;
        if (extra_width != w) return spcies::fail(SPCIES_HIP_EINVAL, "extra_width = %ld, this solver's per-instance extra input is %ld doubles", extra_width, w);
    }
    std::lock_guard<std::mutex> one_call(m->call_mu);
    const int G = (int)m->h.size();
    std::vector<int> rcs(G, 0);
    std::vector<std::string> errs(G);
    std::vector<spcies_hip_timing> tms(G, spcies_hip_timing{0, 0, 0, 0});
    const auto t0 = std::chrono::steady_clock::now();
    auto work = [&](int g) {
        long lo = 0, cnt = 0;
        spcies_hip_shard_range(B, G, g, &lo, &cnt);
        if (cnt == 0) return;
        const size_t n = (size_t)info.n, mm = (size_t)info.m;
        double *f[8] = {nullptr};
        for (int i = 0; fields && i < nf; i++) f[i] = fields[i] ? fields[i] + (size_t)lo * dims[i] : nullptr;
        rcs[g] = spcies_hip_solve_batch_ex(m->h[g], x0 + (size_t)lo * n, ref_stride ? xr + (size_t)lo * n : xr,
                                           ref_stride ? ur + (size_t)lo * mm : ur, ref_stride,
                                           (extra && extra_stride) ? extra + (size_t)lo * extra_width : extra, extra_stride, cnt,
                                           u + (size_t)lo * mm, k + lo, e_flag + lo, fields ? f : nullptr, nf, &tms[g]);
        if (rcs[g]) errs[g] = spcies_hip_last_error();  // thread-local: carried to the caller below
    };
    m->run_all(work);  // the calling thread drives device 0, the resident workers the others
    for (int g = 0; g < G; g++)
        if (rcs[g]) return spcies::fail(rcs[g], "device %d (shard %d of %d): %s", m->dev[g], g, G, errs[g].c_str());
    if (timing) {  // the shards run side by side: the slowest device sets each phase
        for (int g = 0; g < G; g++) {
            timing->update_time = std::max(timing->update_time, tms[g].update_time);
            timing->solve_time = std::max(timing->solve_time, tms[g].solve_time);
            timing->polish_time = std::max(timing->polish_time, tms[g].polish_time);
        }
        timing->run_time = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

int spcies_hip_multi_solve_batch(spcies_hip_multi_handle m, const double *x0, const double *xr, const double *ur, int ref_stride, long B,
                                 double *u, int *k, int *e_flag, double *z, double *v, double *lambda, spcies_hip_timing *timing) {
    if (!m || m->h.empty()) return spcies::fail(SPCIES_HIP_EINVAL, "NULL handle");
    int nf = 0, dims[8] = {0};
    const char *names[8];
    int rc = spcies_hip_get_sol_layout(m->h[0], &nf, dims, names);
    if (rc) return rc;
    if (!z && !v && !lambda) return spcies_hip_multi_solve_batch_ex(m, x0, xr, ur, ref_stride, nullptr, 0, 0, B, u, k, e_flag, nullptr, 0, timing);
    if (nf != 3) return spcies::fail(SPCIES_HIP_EINVAL, "this solver's record is not (z, v, lambda): use spcies_hip_multi_solve_batch_ex");
    double *f[3] = {z, v, lambda};
    return spcies_hip_multi_solve_batch_ex(m, x0, xr, ur, ref_stride, nullptr, 0, 0, B, u, k, e_flag, f, 3, timing);
}

}  // extern "C"
