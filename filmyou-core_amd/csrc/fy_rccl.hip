// fy_rccl.hip -- compiled RCCL transport for fy_collectives (include/filmyou.h): ncclAllGather / ncclReduceScatter over xGMI,
// queued on the job's own HIP stream, so a collective is ordered behind the kernels that produce its input and in front of
// the kernels that consume its output without any host synchronisation.
//
// Replaces (re-expresses) the reference's shuffle / DistributedCache / counter exchange of jobs
// RM2-1 / RM2-2 (M/rm/RM2Job.java:130-149, 184-198, 260-263) for hosts that are not Python: the C++ mirror
// (filmyou-core_amd/host/filmyou_job.hpp) and the JNI shim use this instead of torch.distributed.
//
// librccl is opened at run time (dlopen): a single-GPU host needs no RCCL, and a process that already carries an RCCL
// (PyTorch bundles one under the same SONAME) keeps exactly one copy.
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>

#include "fy_common.hpp"

namespace {

// the slice of rccl.h this file needs (RCCL keeps NCCL's ABI: rccl.h:40-43, 187, 220, 260, 339, 448-466, 655, 678)
typedef struct { char internal[128]; } nccl_unique_id;
typedef void* nccl_comm;
enum { NCCL_SUCCESS = 0, NCCL_INT8 = 0, NCCL_FLOAT32 = 7, NCCL_SUM = 0 };

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm*, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
};

// opened once per process (std::call_once: the ThreadGroup test harness creates communicators from several threads); the
// loader's diagnostic of a failed dlopen / dlsym is captured where it happens (dlerror() clears itself when read)
struct RcclLoad {
    Rccl R;
    std::string why;
};
RcclLoad& rccl_load() {
    static RcclLoad L;
    static std::once_flag once;
    std::call_once(once, [] {
        Rccl& R = L.R;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)   // a copy already in the process (e.g. PyTorch's) is preferred
            if ((R.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
        if (!R.handle)
            for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                if ((R.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
                const char* e = dlerror();
                L.why = e ? e : "dlopen failed";
            }
        if (!R.handle) return;
#define FY_SYM(field, name)                                                          \
    if (R.handle) {                                                                  \
        R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.handle, name));        \
        if (!R.field) { const char* e = dlerror(); L.why = e ? e : name " is missing"; R.handle = nullptr; } \
    }
        FY_SYM(GetUniqueId, "ncclGetUniqueId")
        FY_SYM(CommInitRank, "ncclCommInitRank")
        FY_SYM(CommDestroy, "ncclCommDestroy")
        FY_SYM(GetErrorString, "ncclGetErrorString")
        FY_SYM(AllGather, "ncclAllGather")
        FY_SYM(ReduceScatter, "ncclReduceScatter")
#undef FY_SYM
    });
    return L;
}
Rccl* rccl() {
    RcclLoad& L = rccl_load();
    return L.R.handle ? &L.R : nullptr;
}

}  // namespace

struct fy_rccl {
    fy::Context* ctx = nullptr;
    nccl_comm comm = nullptr;
    int rank = 0, world = 1;
    int64_t calls_all_gather = 0, calls_reduce_scatter = 0, bytes = 0;
};

static int rccl_all_gather(void* user, const void* send, void* recv, int64_t bytes, void* stream) {
    fy_rccl* c = static_cast<fy_rccl*>(user);
    c->calls_all_gather++;
    c->bytes += bytes * c->world;
    return rccl()->AllGather(send, recv, (size_t)bytes, NCCL_INT8, c->comm, static_cast<hipStream_t>(stream));
}
static int rccl_reduce_scatter_f32(void* user, const float* send, float* recv, int64_t count, void* stream) {
    fy_rccl* c = static_cast<fy_rccl*>(user);
    c->calls_reduce_scatter++;
    c->bytes += 4 * count * c->world;
    return rccl()->ReduceScatter(send, recv, (size_t)count, NCCL_FLOAT32, NCCL_SUM, c->comm, static_cast<hipStream_t>(stream));
}

extern "C" {

int fy_rccl_unique_id(char* out128) {
    if (!out128) { fy::set_error("out128 is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    Rccl* R = rccl();
    if (!R) { fy::set_error("librccl.so could not be opened: %s", rccl_load().why.c_str()); return FY_ERR_COLLECTIVE; }
    nccl_unique_id id;
    const int rc = R->GetUniqueId(&id);
    if (rc != NCCL_SUCCESS) { fy::set_error("ncclGetUniqueId: %s", R->GetErrorString(rc)); return FY_ERR_COLLECTIVE; }
    std::memcpy(out128, id.internal, 128);
    return FY_OK;
}

int fy_rccl_create(fy_context* c, int rank, int world, const char* id128, fy_rccl** out) {
    if (!out) { fy::set_error("out is NULL"); return FY_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    if (!c || !id128 || world <= 0 || rank < 0 || rank >= world) { fy::set_error("fy_rccl_create: bad arguments (rank %d of %d)", rank, world); return FY_ERR_INVALID_ARGUMENT; }
    Rccl* R = rccl();
    if (!R) { fy::set_error("librccl.so could not be opened: %s", rccl_load().why.c_str()); return FY_ERR_COLLECTIVE; }
    if (hipSetDevice(c->c.device) != hipSuccess) { fy::set_error("hipSetDevice(%d) failed", c->c.device); return FY_ERR_HIP; }
    nccl_unique_id id;
    std::memcpy(id.internal, id128, 128);
    fy_rccl* h = new fy_rccl;
    h->ctx = &c->c;
    h->rank = rank;
    h->world = world;
    const int rc = R->CommInitRank(&h->comm, world, id, rank);   // collective over all ranks: blocks until they have all arrived
    if (rc != NCCL_SUCCESS) {
        fy::set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, R->GetErrorString(rc));
        delete h;
        return FY_ERR_COLLECTIVE;
    }
    *out = h;
    return FY_OK;
}

int fy_rccl_collectives(fy_rccl* h, fy_collectives* out) {
    if (!h || !out) { fy::set_error("NULL argument"); return FY_ERR_INVALID_ARGUMENT; }
    out->user = h;
    out->all_gather = rccl_all_gather;
    out->reduce_scatter_f32 = rccl_reduce_scatter_f32;
    return FY_OK;
}

int fy_rccl_counters(const fy_rccl* h, int64_t* all_gathers, int64_t* reduce_scatters, int64_t* payload_bytes) {
    if (!h) { fy::set_error("NULL argument"); return FY_ERR_INVALID_ARGUMENT; }
    if (all_gathers) *all_gathers = h->calls_all_gather;
    if (reduce_scatters) *reduce_scatters = h->calls_reduce_scatter;
    if (payload_bytes) *payload_bytes = h->bytes;
    return FY_OK;
}

// (the context must still be alive: the communicator's queued collectives live on its stream.  A host that has already destroyed
// the context -- interpreter teardown -- calls fy_rccl_detach_context first and only the communicator is released.)
void fy_rccl_detach_context(fy_rccl* h) {
    if (h) h->ctx = nullptr;
}

void fy_rccl_destroy(fy_rccl* h) {
    if (!h) return;
    if (h->ctx) (void)hipStreamSynchronize(h->ctx->stream);
    if (h->comm && rccl()) (void)rccl()->CommDestroy(h->comm);
    delete h;
}

}  // extern "C"
