// fy_common.hpp -- context, stream-ordered HBM buffers, error plumbing shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <exception>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/filmyou.h"

namespace fy {

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);
const char* last_error();
struct Tuning;
void load_tuning_from_env(Tuning& t);   // fy_api.hip

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

// ---------------------------------------------------------------- launch-shape knobs
// Defaults are the production values.  The environment overrides (FY_*) are TEST AND MEASUREMENT HOOKS: each
// one forces a path that the default heuristics would pick only at a larger size, so that the parity tests can drive every
// path at a size the oracle finishes; none of them changes a result beyond the summation order.  The environment is read ONCE,
// when the context is created (load_tuning_from_env, fy_api.hip; fy_context_reload_tuning reads it again) -- no job calls getenv.
struct Tuning {
    int force_select = 0;              // route every user through k_topn_select
    int pack24 = 1;                    // M rows as 24-bit floats (3 bytes per element): -25 % of the dominant traffic
    int pack24_min_items = 4096;       // ... for clusters with at least this many items
    int max_slices = 65536;            // user slices (workgroups) per column chunk
    int users_per_wave = 16;           // users a wave of the scoring kernel walks for one column chunk
    int64_t workspace_default = (int64_t)16 << 30;   // score scratch per batch of users
    int lanes = 4;                     // HIP streams the clusters of one job are spread over
    bool lanes_forced = false;         // FY_LANES given: panel mode does not lower it
    int prune = 1;                     // branch and bound over 256-column candidate blocks
    int prune_min_items = 8192;
    int seed_chunks = 0;               // 256-column chunks scored exactly before the bound pass (the most popular candidates);
                                       // 0 = from the list length: ~5 N columns (N = 50: one chunk, N = 100: two), at most four
    int cooc_block = 0;                // force the row kernel's workgroup size
    int cooc_max_ch = 19968;           // LDS accumulators of the row kernel: 156 KiB of 64-bit words of the 160 KiB LDS (ML-25M shape: three
                                       // column chunks instead of four, 19.7 -> 17.8 ms; smaller forces more chunks)
    bool cooc_max_ch_forced = false;   // FY_COOC_MAX_CH given (the item-similarity build has its own default)
    int sup_bounds = 1;                // FY_SUP_BOUNDS: one-cluster pruned jobs bound over <= 64 super-blocks inside the seed pass (0: a bound chunk per user over all blocks)
    int prep_packed = 1;               // FY_PREP_PACKED: fp16-exact scores ride in the low 16 bits of the prep's sort keys, the three nnz-sized sorts move keys alone
    int overlap_values = 1;            // FY_OVERLAP_VALUES: the per-rating values of the scoring kernels are computed on a side stream beside the one-cluster job's row kernel
    int shard_prep = 1;                // FY_SHARD_PREP: several ranks, clusters >= ranks: a rank preps its own clusters' ratings alone (fy_prep.hpp)
    int full_walk_sparse = 1;          // FY_FULL_WALK_SPARSE: unpruned clusters whose matrix has more elements than the cluster has pair visits walk full rows (no mirror pass)
    int refine = 1;                    // FY_REFINE: list rows whose score nearly cancels are scored again in fp64 from fp32 head rows (k_refine_rows)
    float refine_c = 2.0f;             // FY_REFINE_C: ... those with |score| < refine_c * sqrt(ratings of the user)
    int lazy_mirror = 1;               // FY_LAZY_MIRROR: pruned one-cluster jobs mirror only the column blocks somebody reads (0: the whole lower triangle)
    int seed_forced = 0;               // FY_SEED_CHUNKS given: prune whatever the list length
    int coop = 1;                      // cooperative scoring of clusters that span all ranks (needs fy_collectives)
    int coop_force = 0;                // cooperative path also with world == 1 (identity collectives)
    int cooc_pk = 1;                   // packed 4-byte CSR entries for the row kernel when the ratings are fp16-exact
    int cooc_f32 = 0;                  // row kernel accumulators in fp32 (ds_add_f32): MEASUREMENT ONLY -- 4x slower, see fy_cooc.hpp
    int cooc_half = 1;                 // symmetric walk (upper triangle + mirror pass) for clusters with packed rows
    int cooc_fx = 1;                   // fixed-point (ds_add_u64) accumulation in the packed walk
    int panel_min_clusters = 4;        // column-panel mode when at least this many clusters of the rank are pruned ones
    int prune_min_users = 600;         // clusters with fewer users take the plain full pass
    int panel_wide_below_users = 2500; // panel mode: clusters with fewer users keep twice the panel columns
    int panel_cols = 4096;             // columns of a row kept in panel mode (the seed columns and the popular blocks)
    bool bounded_tables = true;        // FY_BOUNDED_TABLES=0: exact table sizes (two host round trips per table)
    bool cooc_planes = true;           // FY_COOC_PLANES=0: linear accumulator layout (measurement only)
    int score_heavy = 512;             // users with more ratings are walked by a whole workgroup of the scoring kernel (0 = off)
    bool panel_repair = true;          // FY_PANEL_REPAIR=0: measurement only
    int panel_lanes = 2;               // job lanes when clusters run in panel mode (measured, 50 clusters: 1 lane 300 ms, 2: 213, 3: 230, 4: 240)
    int64_t flat_budget = 0;           // FY_FLAT_BUDGET_MB: bytes of matrices + score rows one flat batch may hold (0 = a quarter of the HBM, at most the workspace)
    int64_t panel_group_bytes = 0;     // FY_PANEL_GROUP_MB: bytes of panels + scoring scratch one group of panel-mode clusters may hold (0 = a third of the HBM)
    int debug_sync = 0;                // FY_DEBUG_SYNC=1: multi-cluster jobs drain the device after every step of a cluster and say on stderr where they are; 2: only the wall clock of the phases (drains the device at the phase ends)
    bool panel_sym = true;             // FY_PANEL_SYM=0: panel mode walks the head rows over all their chunks (round 2) instead of symmetrically over the panel's
    bool flat_batch = true;            // FY_FLAT=0: small unpruned clusters one after the other on the lanes (round 2) instead of one launch per kernel
    bool panel_multi_launch = true;    // FY_PANEL_MULTI_LAUNCH=0: one row-kernel launch per cluster also in two-phase panel mode
    bool panel_two_phase = true;       // FY_PANEL_TWO_PHASE=0: every cluster start to end on its lane (round 2)
    int panel_max_ch = 4096;           // chunk width of the row kernel in panel mode: five workgroups per CU (measured, 50 clusters, row kernel ms: 8192 -> 67, 6144 -> 54, 4096 -> 46)
    double max_surv_frac = 0.25;       // a pruned batch whose surviving blocks exceed this fraction falls back to the full pass
    // item-item similarity build (fy_itemsim.hip)
    int isim_heavy = 4096;             // row-at-a-time kernel: raters above which a row is split by column chunk
    int isim_gram = -1;                // symmetric Gram + band sweep: -1 = by size (cosine, >= isim_gram_min_items items), 0 = never, 1 = whenever possible
    int isim_gram_min_items = 4096;
    int isim_capg = 2040;              // candidates a row of the band sweep may collect before it is redone exactly (<= 2040: k_isim_finish sorts them in LDS)
    int isim_piece = 4096;             // columns per piece of the band sweep (measured, ML-25M shape, build ms: 2048 11.90, 4096 11.90, 8192 12.2, 16384 12.0, 32768 12.5, 65536 14.1)
    int isim_acc32 = 1;                // 32-bit fixed-point accumulators in the symmetric build's walk when the products are exact integers
};

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
    Tuning tune;                                 // launch-shape knobs, read from the environment when the context is created

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

// Declared right AFTER the host-side sources (std::vector, stack arrays) of queued hipMemcpyAsync uploads: when an exception unwinds the
// scope, the stream is drained before those sources are destroyed (objects die in reverse order of declaration).  On the normal
// path the scope's own synchronisation has already happened and this does nothing.
struct SyncOnUnwind {
    hipStream_t st;
    int n0;
    explicit SyncOnUnwind(hipStream_t s) : st(s), n0(std::uncaught_exceptions()) {}
    ~SyncOnUnwind() {
        if (std::uncaught_exceptions() > n0) (void)hipStreamSynchronize(st);
    }
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
    // every score (of ANY entry; NaN aside) is exactly representable in fp16 -- half stars, integers: found in the same pass; the jobs
    // then carry the rating inside their 64-bit sort keys (fy_prep.hip, packed mode)
    bool scores_fp16_exact = false;
    // what the last RM2 job over these ratings built from them and its clustering alone (fy_rm2.hip: RM2Static): a later job
    // with the same clustering starts from it.  Released with the ratings (before their context), by fy_ratings_drop_cache, and
    // by a caching job whose clustering / share does not match it (BEFORE that job builds its own: never two static sets in HBM).
    // The pointer is read and written under cache_mu; the jobs themselves are single-threaded per ratings object (include/filmyou.h).
    mutable std::shared_ptr<void> rm2_cache;
    mutable std::mutex cache_mu;
};
