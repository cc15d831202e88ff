// fy_common.hpp -- context, stream-ordered HBM buffers, error plumbing shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/filmyou.h"

namespace fy {

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);
const char* last_error();

struct Failure {
    int code;
};

#define FY_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            fy::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            throw fy::Failure{_e == hipErrorOutOfMemory ? FY_ERR_OUT_OF_MEMORY : FY_ERR_HIP};      \
        }                                                                                          \
    } while (0)

#define FY_FAIL(code, ...)            \
    do {                              \
        fy::set_error(__VA_ARGS__);   \
        throw fy::Failure{code};      \
    } while (0)

#define FY_KERNEL_CHECK() FY_HIP(hipGetLastError())

// ---------------------------------------------------------------- context
// One GPU, one stream, and a caching HBM allocator.  Every kernel and copy of a context runs on `stream`, so a block
// released by the host while work is still queued may be handed to a later allocation: the later user is ordered behind
// the earlier one by the stream (the semantics of hipFreeAsync, without the driver's pool: hipMallocAsync re-grew its
// pool at unpredictable steps and stalled whole jobs by 0.3-0.9 s when 14-16 GB buffers were split and re-requested).
// After the first job every request is served from the cache; fy_context_destroy returns the memory.
struct Context {
    int device = -1;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    size_t lds_per_block = 160 * 1024;
    size_t total_mem = 0;
    std::vector<hipStream_t> aux;               // extra streams ("lanes") a multi-cluster job overlaps its clusters on
    std::multimap<size_t, void*> free_blocks;   // capacity -> block
    std::map<void*, size_t> capacity;            // every block ever handed out
    int64_t fail_alloc_in = 0;                   // fault injection (fy_context_inject_alloc_failure): the n-th request from now fails

    void* alloc(size_t bytes) {
        const size_t want = (bytes + 255) & ~size_t(255);
        if (fail_alloc_in > 0 && --fail_alloc_in == 0) {
            set_error("HBM allocation of %zu bytes failed: injected fault", want);
            throw Failure{FY_ERR_OUT_OF_MEMORY};
        }
        auto it = free_blocks.lower_bound(want);
        if (it != free_blocks.end() && it->first <= want + want / 4 + (1u << 20)) {
            void* p = it->second;
            free_blocks.erase(it);
            return p;
        }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {   // give cached blocks back to the driver and retry once
            (void)hipGetLastError();
            (void)hipStreamSynchronize(stream);
            trim();
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error("HBM allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
            throw Failure{FY_ERR_OUT_OF_MEMORY};
        }
        capacity[p] = want;
        return p;
    }
    void release(void* p) {
        auto it = capacity.find(p);
        if (it != capacity.end()) free_blocks.emplace(it->second, p);
    }
    void trim() {   // the stream must be idle
        for (auto& kv : free_blocks) {
            (void)hipFree(kv.second);
            capacity.erase(kv.second);
        }
        free_blocks.clear();
    }
};

// ---------------------------------------------------------------- HBM buffer (stream-ordered alloc / free)
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    Context* ctx = nullptr;
    DevBuf() = default;
    DevBuf(Context* c, size_t count) { alloc(c, count); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept { *this = std::move(o); }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) {
            release();
            p = o.p; n = o.n; ctx = o.ctx;
            o.p = nullptr; o.n = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(Context* c, size_t count) {
        release();
        ctx = c;
        n = count;
        size_t bytes = (count ? count : 1) * sizeof(T);
        p = static_cast<T*>(c->alloc(bytes));
    }
    void zero() { FY_HIP(hipMemsetAsync(p, 0, (n ? n : 1) * sizeof(T), ctx->stream)); }
    void release() {
        if (p) ctx->release(p);
        p = nullptr;
        n = 0;
    }
    T* get() const { return p; }
    size_t size() const { return n; }
    size_t bytes() const { return n * sizeof(T); }
};

template <class T>
inline void d2h(Context* c, T* dst, const T* src, size_t count) {
    if (count) FY_HIP(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, c->stream));
}
template <class T>
inline void h2d(Context* c, T* dst, const T* src, size_t count) {
    if (count) FY_HIP(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
}
template <class T>
inline void d2d(Context* c, T* dst, const T* src, size_t count) {
    if (count) FY_HIP(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
}
inline void sync(Context* c) { FY_HIP(hipStreamSynchronize(c->stream)); }

template <class T>
inline T fetch(Context* c, const T* src) {
    T v;
    d2h(c, &v, src, 1);
    sync(c);
    return v;
}

// ---------------------------------------------------------------- HIP-event phase timers on the context's stream
struct EventTimer {
    Context* ctx;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
    explicit EventTimer(Context* c) : ctx(c) {}
    ~EventTimer() {
        for (auto& s : spans) { (void)hipEventDestroy(s.first); (void)hipEventDestroy(s.second); }
    }
    // a span lives on ONE stream (default: the context's); spans of different lanes may overlap in time
    size_t begin(hipStream_t st = nullptr) {
        hipEvent_t a, b;
        FY_HIP(hipEventCreate(&a));
        FY_HIP(hipEventCreate(&b));
        FY_HIP(hipEventRecord(a, st ? st : ctx->stream));
        spans.emplace_back(a, b);
        return spans.size() - 1;
    }
    void end(size_t i, hipStream_t st = nullptr) { FY_HIP(hipEventRecord(spans[i].second, st ? st : ctx->stream)); }
    // total milliseconds over all spans; the stream must have been synchronised
    double total_ms() {
        double t = 0;
        for (auto& s : spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, s.first, s.second) == hipSuccess) t += ms;
        }
        return t;
    }
    size_t count() const { return spans.size(); }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

}  // namespace fy

// ---------------------------------------------------------------- opaque ABI objects
struct fy_context {
    fy::Context c;
};

struct fy_ratings {
    fy::Context* ctx = nullptr;
    int64_t nnz = 0;
    fy::DevBuf<int32_t> user, item;
    fy::DevBuf<float> score;
    // largest user / item id over ALL entries (-1: none >= 0), found once when the ratings are put into HBM: the jobs pack
    // their sort keys into the bits these ids need
    int32_t max_user = -1, max_item = -1;
};
