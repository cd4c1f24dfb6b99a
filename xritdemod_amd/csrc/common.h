// common.h -- shared host-side plumbing of libxritdemod_amd (HIP runtime helpers,
// error text, grow-only device buffers, per-kernel HIP-event profiler).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/xritdemod_amd.h"

namespace xrit {

void set_error(const char *fmt, ...);
const char *get_error();

#define XR_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            ::xrit::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                              __LINE__);                                                    \
            return XRIT_E_HIP;                                                              \
        }                                                                                   \
    } while (0)

#define XR_TRY(expr)               \
    do {                           \
        int _r = (expr);           \
        if (_r != XRIT_OK) return _r; \
    } while (0)

// Grow-only device allocation, like the reference's checkAndResizeBuffers
// (demodulator.cpp:76-92): no allocation in process() once warmed up.
struct DevBuf {
    void  *p = nullptr;
    size_t bytes = 0;
    int reserve(size_t need)
    {
        if (need <= bytes) return XRIT_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t want = need + need / 8 + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return XRIT_E_NOMEM;
        }
        bytes = want;
        return XRIT_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Brackets kernel launches with HIP events on the launch stream.
struct Profiler {
    bool enabled = false;
    bool light = false;      // bracket only the kernel of the light list
    struct Rec { std::string name; hipEvent_t a, b; };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    std::map<std::string, std::pair<double, int>> acc;
    std::map<std::string, std::vector<float>> samples;      // every bracket's duration, in launch order
    std::vector<std::string> order;

    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
    // An event record is a barrier in the queue: the bracketed kernel cannot overlap the tail of its
    // predecessor or the head of its successor (~10-20 us per boundary at C2).  The light mode therefore
    // brackets a single kernel per call: the decimating FIR, the one launch that moves the input bytes.
    // (XRIT_LIGHT_LIST="name,name": other brackets for a light-mode measurement, read once)
    static bool in_light_list(const char *name)
    {
        static const char *extra = getenv("XRIT_LIGHT_LIST");
        if (extra) {
            const size_t n = strlen(name);
            for (const char *p = strstr(extra, name); p; p = strstr(p + 1, name))
                if ((p == extra || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
            return false;
        }
        return !strcmp(name, "fir_decim");
    }
    bool open = false;
    void begin(const char *name, hipStream_t s)
    {
        open = enabled && (!light || in_light_list(name));
        if (!open) return;
        Rec r{name, get(), get()};
        (void)hipEventRecord(r.a, s);
        pending.push_back(r);
    }
    void end(hipStream_t s)
    {
        if (!open) return;
        (void)hipEventRecord(pending.back().b, s);
        open = false;
    }
    // call after the stream has been synchronised; records of a stream that is still running (a front end that runs
    // ahead on the second stream) stay pending until a later call
    void collect()
    {
        std::vector<Rec> later;
        for (auto &r : pending) {
            if (hipEventQuery(r.b) != hipSuccess) { later.push_back(r); continue; }
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
                auto it = acc.find(r.name);
                if (it == acc.end()) { acc[r.name] = {ms, 1}; order.push_back(r.name); }
                else { it->second.first += ms; it->second.second += 1; }
                samples[r.name].push_back(ms);
            }
            pool.push_back(r.a);
            pool.push_back(r.b);
        }
        pending.swap(later);
    }
    void reset() { acc.clear(); samples.clear(); order.clear(); }
    ~Profiler()
    {
        for (auto &r : pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

struct ProfScope {
    Profiler *p; hipStream_t s;
    ProfScope(Profiler *p_, const char *name, hipStream_t s_) : p(p_), s(s_) { if (p) p->begin(name, s); }
    ~ProfScope() { if (p) p->end(s); }
};

#ifdef __HIPCC__
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory, i.e. it waits
// for every outstanding global load (s_waitcnt vmcnt(0)) -- which would serialise register prefetches that
// are meant to stay in flight across the barrier.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#endif

static inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace xrit
