// fy_api.hip -- the extern "C" surface declared in include/filmyou.h: context, ratings, results, error plumbing.
#include <algorithm>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <new>
#include <unordered_map>

#include "fy_prep.hpp"
#include "fy_rm2.hpp"

namespace fy {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }
}  // namespace fy

using namespace fy;

// the ONE place the library reads its environment (fy_context_create, fy_context_reload_tuning)
void fy::load_tuning_from_env(Tuning& t) {
    t = Tuning{};
    if (const char* e = getenv("FY_M24")) t.pack24 = atoi(e) != 0;
    if (const char* e = getenv("FY_M24_MIN_ITEMS")) t.pack24_min_items = std::max(0, atoi(e));
    if (const char* e = getenv("FY_SCORE_SLICES")) t.max_slices = atoi(e);
    if (const char* e = getenv("FY_SCORE_UPW")) { int v = atoi(e); if (v >= 1) t.users_per_wave = v; }
    if (const char* e = getenv("FY_PRUNE")) t.prune = atoi(e) != 0;
    if (const char* e = getenv("FY_LAZY_MIRROR")) t.lazy_mirror = atoi(e) != 0;
    if (const char* e = getenv("FY_REFINE")) t.refine = atoi(e) != 0;
    if (const char* e = getenv("FY_REFINE_C")) { double v = atof(e); if (v >= 0.0 && v < 1e6) t.refine_c = (float)v; }
    if (const char* e = getenv("FY_FULL_WALK_SPARSE")) t.full_walk_sparse = atoi(e) != 0;
    if (const char* e = getenv("FY_PREP_PACKED")) t.prep_packed = atoi(e) != 0;
    if (const char* e = getenv("FY_OVERLAP_VALUES")) t.overlap_values = atoi(e) != 0;
    if (const char* e = getenv("FY_SHARD_PREP")) t.shard_prep = atoi(e) != 0;
    if (const char* e = getenv("FY_SUP_BOUNDS")) t.sup_bounds = atoi(e) != 0;
    if (const char* e = getenv("FY_COOP")) t.coop = atoi(e) != 0;
    if (const char* e = getenv("FY_COOP_FORCE")) t.coop_force = atoi(e) != 0;
    if (const char* e = getenv("FY_COOC_PK")) t.cooc_pk = atoi(e) != 0;
    if (const char* e = getenv("FY_COOC_F32")) t.cooc_f32 = atoi(e) != 0;
    if (const char* e = getenv("FY_COOC_HALF")) t.cooc_half = atoi(e) != 0;
    if (const char* e = getenv("FY_COOC_FX")) t.cooc_fx = atoi(e) != 0;
    if (const char* e = getenv("FY_PANEL_MIN_CLUSTERS")) { int v = atoi(e); if (v >= 1) t.panel_min_clusters = v; }
    if (const char* e = getenv("FY_PANEL_COLS")) { int v = atoi(e); if (v >= 256) { t.panel_cols = v; t.panel_wide_below_users = 0; } }
    if (const char* e = getenv("FY_PANEL_WIDE_BELOW_USERS")) { int v = atoi(e); if (v >= 0) t.panel_wide_below_users = v; }
    if (const char* e = getenv("FY_PANEL_MAX_CH")) { int v = atoi(e); if (v >= 256) t.panel_max_ch = v; }
    if (const char* e = getenv("FY_BOUNDED_TABLES")) t.bounded_tables = atoi(e) != 0;
    if (const char* e = getenv("FY_COOC_PLANES")) t.cooc_planes = atoi(e) != 0;
    if (const char* e = getenv("FY_SCORE_HEAVY")) { int v = atoi(e); if (v >= 0) t.score_heavy = v; }
    if (const char* e = getenv("FY_PANEL_REPAIR")) t.panel_repair = atoi(e) != 0;
    if (const char* e = getenv("FY_PANEL_MULTI_LAUNCH")) t.panel_multi_launch = atoi(e) != 0;
    if (const char* e = getenv("FY_FLAT")) t.flat_batch = atoi(e) != 0;
    if (const char* e = getenv("FY_PANEL_SYM")) t.panel_sym = atoi(e) != 0;
    if (const char* e = getenv("FY_DEBUG_SYNC")) t.debug_sync = atoi(e);
    if (const char* e = getenv("FY_PANEL_GROUP_MB")) t.panel_group_bytes = std::max<int64_t>(0, (int64_t)atoll(e)) << 20;      // (0 = default)
    if (const char* e = getenv("FY_FLAT_BUDGET_MB")) t.flat_budget = std::max<int64_t>(0, (int64_t)atoll(e)) << 20;
    if (const char* e = getenv("FY_PANEL_TWO_PHASE")) t.panel_two_phase = atoi(e) != 0;
    if (const char* e = getenv("FY_PANEL_LANES")) { int v = atoi(e); if (v >= 1 && v <= 8) t.panel_lanes = v; }
    if (const char* e = getenv("FY_PRUNE_MIN_ITEMS")) { t.prune_min_items = atoi(e); t.prune_min_users = 0; }   // (a forced item threshold -- tests -- lifts the user threshold too)
    if (const char* e = getenv("FY_PRUNE_MIN_USERS")) { int v = atoi(e); if (v >= 0) t.prune_min_users = v; }
    if (const char* e = getenv("FY_SEED_CHUNKS")) { int v = atoi(e); if (v >= 0 && v <= 40) { t.seed_chunks = v; t.seed_forced = v > 0; } }     // (40 = SEED_CHUNKS_MAX, fy_rm2_kernels.hpp)
    if (const char* e = getenv("FY_WORKSPACE_GB")) { long v = atol(e); if (v >= 1) t.workspace_default = (int64_t)v << 30; }
    if (const char* e = getenv("FY_LANES")) { int v = atoi(e); if (v >= 1 && v <= 8) { t.lanes = v; t.lanes_forced = true; } }
    if (const char* e = getenv("FY_COOC_BLOCK")) { int v = atoi(e); if (v == 256 || v == 512 || v == 1024) t.cooc_block = v; }
    if (const char* e = getenv("FY_COOC_MAX_CH")) { int v = atoi(e); if (v >= 64 && v <= 20224) { t.cooc_max_ch = v; t.cooc_max_ch_forced = true; } }
    if (const char* e = getenv("FY_TOPN_FORCE_SELECT")) t.force_select = atoi(e) != 0;
    if (const char* e = getenv("FY_MAX_SURV_FRAC")) { double v = atof(e); if (v >= 0.0) t.max_surv_frac = v; }
    if (const char* e = getenv("FY_ISIM_HEAVY")) t.isim_heavy = std::max(0, atoi(e));
    if (const char* e = getenv("FY_ISIM_GRAM")) t.isim_gram = atoi(e) != 0;
    if (const char* e = getenv("FY_ISIM_GRAM_MIN_ITEMS")) t.isim_gram_min_items = std::max(0, atoi(e));
    if (const char* e = getenv("FY_ISIM_CAPG")) { int v = atoi(e); if (v >= 1 && v <= 2040) t.isim_capg = v; }
    if (const char* e = getenv("FY_ISIM_ACC32")) t.isim_acc32 = atoi(e) != 0;
    if (const char* e = getenv("FY_ISIM_PIECE")) { int v = atoi(e); if (v >= 64 && v <= 8192 && v % 64 == 0) t.isim_piece = v; }
}

// Every entry point that reaches the GPU selects its context's device first: allocations, new streams and kernel attributes
// go to the calling thread's CURRENT device, and a host that drives several GPUs from one process (or hands a job to
// another thread) would otherwise build one context's job in another GPU's memory.  The job handle is opaque in this file,
// so the contexts of live jobs are kept here (fy_rm2_prepare ... fy_rm2_job_destroy).
namespace {
// (never destroyed: a host may release its last job from a static destructor of its own, after this library's would have run)
std::mutex& g_jobs_mu = *new std::mutex;
std::unordered_map<const fy_rm2_job*, fy::Context*>& g_jobs = *new std::unordered_map<const fy_rm2_job*, fy::Context*>;
fy::Context* job_context(const fy_rm2_job* j) {
    std::lock_guard<std::mutex> g(g_jobs_mu);
    auto it = g_jobs.find(j);
    return it == g_jobs.end() ? nullptr : it->second;
}
inline void select_device(const fy::Context* ctx) {
    if (ctx) FY_HIP(hipSetDevice(ctx->device));
}
}  // namespace

// every entry point: no exception may cross the ABI
#define FY_TRY try {
#define FY_CATCH                                                           \
    }                                                                      \
    catch (const fy::Failure& f) { return f.code; }                        \
    catch (const std::bad_alloc&) {                                        \
        fy::set_error("host allocation failed");                           \
        return FY_ERR_OUT_OF_MEMORY;                                       \
    }                                                                      \
    catch (const std::exception& e) {                                      \
        fy::set_error("unexpected: %s", e.what());                         \
        return FY_ERR_HIP;                                                 \
    }                                                                      \
    return FY_OK;

extern "C" {

int fy_abi_version(void) { return FY_ABI_VERSION; }
const char* fy_last_error(void) { return fy::last_error(); }

int fy_context_create(int device_ordinal, fy_context** out) {
    if (!out) { set_error("out is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    FY_TRY
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        FY_FAIL(FY_ERR_NO_DEVICE, "no HIP device is visible (%s); this library has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    if (device_ordinal < 0 || device_ordinal >= n) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "device %d of %d", device_ordinal, n);
    FY_HIP(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    FY_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        FY_FAIL(FY_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 (MI355X) only", device_ordinal, prop.gcnArchName);
    std::unique_ptr<fy_context> c(new fy_context);
    c->c.device = device_ordinal;
    c->c.num_cus = prop.multiProcessorCount;
    c->c.total_mem = prop.totalGlobalMem;
    FY_HIP(hipStreamCreateWithFlags(&c->c.stream, hipStreamNonBlocking));
    fy::load_tuning_from_env(c->c.tune);
    *out = c.release();
    FY_CATCH
}

void fy_context_destroy(fy_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->c.device);
    if (c->c.stream) {
        for (hipStream_t x : c->c.aux) (void)hipStreamSynchronize(x);   // lanes of a multi-cluster job (a failed job may have left work there)
        (void)hipStreamSynchronize(c->c.stream);
        c->c.trim();
        for (auto& kv : c->c.capacity) (void)hipFree(kv.first);   // blocks still held by live objects: caller error, reclaimed anyway
        c->c.capacity.clear();
        for (hipStream_t x : c->c.aux) { (void)hipStreamSynchronize(x); (void)hipStreamDestroy(x); }
        (void)hipStreamDestroy(c->c.stream);
    }
    delete c;
}

int fy_context_inject_alloc_failure(fy_context* c, int64_t nth) {
    if (!c || nth < 0) { set_error("context is NULL or nth < 0"); return FY_ERR_INVALID_ARGUMENT; }
    c->c.fail_alloc_in = nth;
    return FY_OK;
}

int fy_context_reload_tuning(fy_context* c) {
    if (!c) { set_error("context is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    fy::load_tuning_from_env(c->c.tune);
    return FY_OK;
}

int fy_context_synchronize(fy_context* c) {
    if (!c) { set_error("context is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    FY_TRY
    select_device(&c->c);
    FY_HIP(hipStreamSynchronize(c->c.stream));
    FY_CATCH
}

void* fy_context_stream(fy_context* c) { return c ? (void*)c->c.stream : nullptr; }

int fy_ratings_create(fy_context* c, int64_t nnz, const int32_t* user, const int32_t* item, const float* score, int location,
                      fy_ratings** out) {
    if (!out) { set_error("out is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    if (!c) { set_error("context is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    if (nnz < 0 || (nnz > 0 && (!user || !item || !score))) { set_error("ratings arrays are NULL or nnz < 0"); return FY_ERR_INVALID_ARGUMENT; }
    if (location != FY_HOST && location != FY_DEVICE) { set_error("location must be FY_HOST or FY_DEVICE"); return FY_ERR_INVALID_ARGUMENT; }
    FY_TRY
    FY_HIP(hipSetDevice(c->c.device));
    std::unique_ptr<fy_ratings> r(new fy_ratings);
    r->ctx = &c->c;
    r->nnz = nnz;
    r->user.alloc(&c->c, (size_t)nnz);
    r->item.alloc(&c->c, (size_t)nnz);
    r->score.alloc(&c->c, (size_t)nnz);
    if (nnz) {
        const hipMemcpyKind k = location == FY_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
        FY_HIP(hipMemcpyAsync(r->user.get(), user, (size_t)nnz * 4, k, c->c.stream));
        FY_HIP(hipMemcpyAsync(r->item.get(), item, (size_t)nnz * 4, k, c->c.stream));
        FY_HIP(hipMemcpyAsync(r->score.get(), score, (size_t)nnz * 4, k, c->c.stream));
        FY_HIP(hipStreamSynchronize(c->c.stream));   // the caller may free its arrays when this returns
        fy::ratings_id_bounds(&c->c, r.get());
    }
    *out = r.release();
    FY_CATCH
}

void fy_ratings_destroy(fy_ratings* r) {
    if (!r) return;
    fy::Context* ctx = r->ctx;
    if (ctx) (void)hipSetDevice(ctx->device);
    delete r;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
}
int64_t fy_ratings_nnz(const fy_ratings* r) { return r ? r->nnz : 0; }
void fy_ratings_drop_cache(fy_ratings* r) {
    if (!r) return;
    if (r->ctx) (void)hipSetDevice(r->ctx->device);
    std::shared_ptr<void> old;
    {
        std::lock_guard<std::mutex> g(r->cache_mu);
        old.swap(r->rm2_cache);
    }
    if (old && r->ctx) (void)hipStreamSynchronize(r->ctx->stream);     // nothing queued may still read what is released next
    old.reset();
}

// ---------------------------------------------------------------- RM2
int fy_rm2_prepare(fy_context* c, const fy_rm2_params* p, const fy_ratings* r, int64_t n_map, const int32_t* map_user,
                   const int32_t* map_cluster, const int32_t* cluster_count, fy_rm2_job** out) {
    if (!out) { set_error("out is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    if (!c || !r) { set_error("context or ratings is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    if (r->ctx != &c->c) { set_error("ratings belong to another context"); return FY_ERR_INVALID_ARGUMENT; }
    FY_TRY
    FY_HIP(hipSetDevice(c->c.device));
    *out = fy::rm2_prepare(&c->c, p, r, n_map, map_user, map_cluster, cluster_count);
    {
        std::lock_guard<std::mutex> g(g_jobs_mu);
        g_jobs[*out] = &c->c;
    }
    FY_CATCH
}

int fy_rm2_partial_stats(fy_rm2_job* j, double** device_buf, int64_t* len) {
    if (!j || !device_buf || !len) { set_error("NULL argument"); return FY_ERR_INVALID_ARGUMENT; }
    FY_TRY
    select_device(job_context(j));
    fy::rm2_partial_stats(j, device_buf, len);
    FY_CATCH
}

int fy_rm2_stats_layout(fy_rm2_job* j, int64_t* n_item_slots, int64_t* n_user_slots) {
    if (!j || !n_item_slots || !n_user_slots) { set_error("NULL argument"); return FY_ERR_INVALID_ARGUMENT; }
    FY_TRY
    fy::rm2_stats_layout(j, n_item_slots, n_user_slots);
    FY_CATCH
}

int fy_rm2_set_global_stats(fy_rm2_job* j, const double* gathered_device, int32_t world) {
    if (!j || !gathered_device) { set_error("NULL argument"); return FY_ERR_INVALID_ARGUMENT; }
    FY_TRY
    select_device(job_context(j));
    fy::rm2_set_global_stats(j, gathered_device, world);
    FY_CATCH
}

int fy_rm2_set_collectives(fy_rm2_job* j, const fy_collectives* c) {
    if (!j || !c) { set_error("NULL argument"); return FY_ERR_INVALID_ARGUMENT; }
    FY_TRY
    fy::rm2_set_collectives(j, c);
    FY_CATCH
}

int fy_rm2_score(fy_rm2_job* j, fy_result** out) {
    if (!out) { set_error("out is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    if (!j) { set_error("job is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    FY_TRY
    select_device(job_context(j));
    *out = fy::rm2_score(j);
    FY_CATCH
}

void fy_rm2_job_destroy(fy_rm2_job* j) {
    if (!j) return;
    fy::Context* ctx = nullptr;
    {
        std::lock_guard<std::mutex> g(g_jobs_mu);
        auto it = g_jobs.find(j);
        if (it != g_jobs.end()) { ctx = it->second; g_jobs.erase(it); }
    }
    if (ctx) (void)hipSetDevice(ctx->device);
    fy::rm2_job_destroy(j);
}

int fy_rm2_run(const fy_rm2_params* p, int64_t nnz, const int32_t* user, const int32_t* item, const float* score, int64_t n_map,
               const int32_t* map_user, const int32_t* map_cluster, const int32_t* cluster_count, fy_result** out) {
    if (!out) { set_error("out is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    fy_context* c = nullptr;
    fy_ratings* r = nullptr;
    fy_rm2_job* j = nullptr;
    int rc = fy_context_create(0, &c);
    if (rc == FY_OK) rc = fy_ratings_create(c, nnz, user, item, score, FY_HOST, &r);
    if (rc == FY_OK) rc = fy_rm2_prepare(c, p, r, n_map, map_user, map_cluster, cluster_count, &j);
    if (rc == FY_OK) rc = fy_rm2_score(j, out);
    if (rc == FY_OK) {
        // the result outlives the context: bring everything to the host now
        (void)fy_result_key0(*out);
        (void)fy_result_user_sum(*out);
        (*out)->d_key0.release(); (*out)->d_key1.release(); (*out)->d_aux.release(); (*out)->d_value.release();
        (*out)->d_user_id.release(); (*out)->d_item_id.release(); (*out)->d_user_sum.release(); (*out)->d_icoll.release();
        (*out)->ctx = nullptr;
    }
    fy_rm2_job_destroy(j);
    fy_ratings_destroy(r);
    fy_context_destroy(c);
    return rc;
}

// ---------------------------------------------------------------- item-item similarity
int fy_itemsim_build(fy_context* c, const fy_itemsim_params* p, const fy_ratings* r, fy_result** out) {
    if (!out) { set_error("out is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    if (!c || !r || !p) { set_error("NULL argument"); return FY_ERR_INVALID_ARGUMENT; }
    if (r->ctx != &c->c) { set_error("ratings belong to another context"); return FY_ERR_INVALID_ARGUMENT; }
    FY_TRY
    FY_HIP(hipSetDevice(c->c.device));
    *out = fy::itemsim_build(&c->c, p, r);
    FY_CATCH
}

int fy_itemsim_run(const fy_itemsim_params* p, int64_t nnz, const int32_t* user, const int32_t* item, const float* score,
                   fy_result** out) {
    if (!out) { set_error("out is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    fy_context* c = nullptr;
    fy_ratings* r = nullptr;
    int rc = fy_context_create(0, &c);
    if (rc == FY_OK) rc = fy_ratings_create(c, nnz, user, item, score, FY_HOST, &r);
    if (rc == FY_OK) rc = fy_itemsim_build(c, p, r, out);
    if (rc == FY_OK) {
        (void)fy_result_key0(*out);
        (*out)->d_key0.release(); (*out)->d_key1.release(); (*out)->d_aux.release(); (*out)->d_value.release();
        (*out)->ctx = nullptr;
    }
    fy_ratings_destroy(r);
    fy_context_destroy(c);
    return rc;
}

int fy_cluster_assign(fy_context* c, int32_t n_rows, int32_t k, const double* H, int location, int32_t first_user,
                      int32_t cluster_offset, int32_t n_clusters, int32_t* user_out, int32_t* cluster_out, int32_t* count_inout) {
    if (!c) { set_error("context is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    FY_TRY
    FY_HIP(hipSetDevice(c->c.device));
    fy::cluster_assign(&c->c, n_rows, k, H, location, first_user, cluster_offset, n_clusters, user_out, cluster_out, count_inout);
    FY_CATCH
}

int fy_nmf_factorize(fy_context* c, const fy_nmf_params* p, const fy_ratings* r, double* H, double* W, fy_stats* st) {
    if (!c || !r) { set_error("context or ratings is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    if (r->ctx != &c->c) { set_error("ratings belong to another context"); return FY_ERR_INVALID_ARGUMENT; }
    FY_TRY
    FY_HIP(hipSetDevice(c->c.device));
    fy::nmf_factorize(&c->c, p, r, H, W, st);
    FY_CATCH
}

int fy_itemcf_recommend(fy_context* c, const fy_itemcf_params* p, const fy_ratings* r, fy_result* sims, fy_result** out) {
    if (!out) { set_error("out is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    if (!c || !r || !p || !sims) { set_error("NULL argument"); return FY_ERR_INVALID_ARGUMENT; }
    if (r->ctx != &c->c || sims->ctx != &c->c) { set_error("ratings / similarities belong to another context"); return FY_ERR_INVALID_ARGUMENT; }
    if (sims->kind != 1) { set_error("`similarities` is not the result of fy_itemsim_build"); return FY_ERR_INVALID_ARGUMENT; }
    FY_TRY
    FY_HIP(hipSetDevice(c->c.device));
    *out = fy::itemcf_recommend(&c->c, p, r, sims);
    FY_CATCH
}

// ---------------------------------------------------------------- results
static void rows_to_host(fy_result* r) {
    if (r->rows_on_host) return;
    r->rows_on_host = true;
    const size_t n = (size_t)r->n;
    r->h_key0.resize(n); r->h_key1.resize(n); r->h_aux.resize(n); r->h_value.resize(n);
    if (n == 0 || !r->ctx) return;
    try {
        select_device(r->ctx);
        fy::d2h(r->ctx, r->h_key0.data(), r->d_key0.get(), n);
        fy::d2h(r->ctx, r->h_key1.data(), r->d_key1.get(), n);
        fy::d2h(r->ctx, r->h_aux.data(), r->d_aux.get(), n);
        fy::d2h(r->ctx, r->h_value.data(), r->d_value.get(), n);
        fy::sync(r->ctx);
    } catch (const fy::Failure&) {
    }
}
static void sums_to_host(fy_result* r) {
    if (r->sums_on_host) return;
    r->sums_on_host = true;
    const size_t nu = r->d_user_id.size(), ni = r->d_item_id.size();
    r->h_user_id.resize(nu); r->h_user_sum.resize(nu); r->h_item_id.resize(ni); r->h_icoll.resize(ni);
    if (!r->ctx) return;
    try {
        select_device(r->ctx);
        fy::d2h(r->ctx, r->h_user_id.data(), r->d_user_id.get(), nu);
        fy::d2h(r->ctx, r->h_user_sum.data(), r->d_user_sum.get(), nu);
        fy::d2h(r->ctx, r->h_item_id.data(), r->d_item_id.get(), ni);
        if (r->d_icoll.size() == ni) fy::d2h(r->ctx, r->h_icoll.data(), r->d_icoll.get(), ni);
        fy::sync(r->ctx);
    } catch (const fy::Failure&) {
    }
}

int64_t fy_result_size(fy_result* r) { return r ? r->n : 0; }
const int32_t* fy_result_key0(fy_result* r) { if (!r) return nullptr; rows_to_host(r); return r->h_key0.data(); }
const int32_t* fy_result_key1(fy_result* r) { if (!r) return nullptr; rows_to_host(r); return r->h_key1.data(); }
const float* fy_result_value(fy_result* r) { if (!r) return nullptr; rows_to_host(r); return r->h_value.data(); }
const int32_t* fy_result_aux(fy_result* r) { if (!r) return nullptr; rows_to_host(r); return r->h_aux.data(); }
int64_t fy_result_n_users(fy_result* r) { if (!r) return 0; sums_to_host(r); return (int64_t)r->h_user_id.size(); }
const int32_t* fy_result_user_id(fy_result* r) { if (!r) return nullptr; sums_to_host(r); return r->h_user_id.data(); }
const double* fy_result_user_sum(fy_result* r) { if (!r) return nullptr; sums_to_host(r); return r->h_user_sum.data(); }
int64_t fy_result_n_items(fy_result* r) { if (!r) return 0; sums_to_host(r); return (int64_t)r->h_item_id.size(); }
const int32_t* fy_result_item_id(fy_result* r) { if (!r) return nullptr; sums_to_host(r); return r->h_item_id.data(); }
const double* fy_result_item_coll(fy_result* r) { if (!r) return nullptr; sums_to_host(r); return r->h_icoll.data(); }
double fy_result_total_sum(fy_result* r) { return r ? r->total_sum : 0.0; }
int fy_result_stats(fy_result* r, fy_stats* out) {
    if (!r || !out) { set_error("NULL argument"); return FY_ERR_INVALID_ARGUMENT; }
    *out = r->st;
    return FY_OK;
}
void fy_result_free(fy_result* r) {
    if (!r) return;
    fy::Context* ctx = r->ctx;
    if (ctx) (void)hipSetDevice(ctx->device);
    delete r;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
}

}  // extern "C"
