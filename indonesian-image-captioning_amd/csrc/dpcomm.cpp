// Data-parallel gradient exchange through the C ABI: RCCL SUM all-reduce of gradient buckets on a library-owned
// communication stream, ordered against the caller's compute stream by HIP events (SURVEY.md 8b/8e:
// `dp_comm_{create,allreduce_bucket,destroy}`).  The reference has no distributed code at all; this is the exchange
// step of the one-process-per-GPU data-parallel train step (scnattn/dp.py uses it when SCNATTN_DP_BACKEND=cabi,
// torch.distributed -- the same RCCL -- otherwise).
//
// RCCL is bound at run time (dlopen("librccl.so.1") / dlsym): PyTorch ships its own copy of the library and a process
// must not end up with two; whichever copy is already mapped under that soname is the one used.
//
// xGMI is point-to-point (7 links x ~153 GB/s per GPU), so ring collectives are bound per link: callers hand over
// MiB-sized buckets (32 MiB by default in scnattn/dp.py), one ncclAllReduce each, in the order gradients appear.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <vector>
#include "../../include/scnattn.h"
#include "common.h"

namespace scn {

namespace {

// the handful of RCCL entry points used, with their published C signatures (rccl.h)
typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void* ncclComm_h;
typedef int (*fn_GetUniqueId)(ncclUniqueId_t*);
typedef int (*fn_CommInitRank)(ncclComm_h*, int, ncclUniqueId_t, int);
typedef int (*fn_CommDestroy)(ncclComm_h);
typedef int (*fn_AllReduce)(const void*, void*, size_t, int /*dtype*/, int /*op*/, ncclComm_h, hipStream_t);
typedef const char* (*fn_GetErrorString)(int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;     // ncclDataType_t::ncclFloat32, ncclRedOp_t::ncclSum

struct Rccl {
    void* h = nullptr;
    fn_GetUniqueId GetUniqueId = nullptr;
    fn_CommInitRank CommInitRank = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_AllReduce AllReduce = nullptr;
    fn_GetErrorString GetErrorString = nullptr;
};
std::mutex g_mu;
Rccl g_rccl;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rccl.h) return 0;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { set_error("dp_comm: cannot load librccl (%s)", dlerror()); return -3; }
    Rccl r;
    r.h = h;
    r.GetUniqueId = (fn_GetUniqueId)dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (fn_CommInitRank)dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (fn_CommDestroy)dlsym(h, "ncclCommDestroy");
    r.AllReduce = (fn_AllReduce)dlsym(h, "ncclAllReduce");
    r.GetErrorString = (fn_GetErrorString)dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) {
        set_error("dp_comm: librccl lacks an expected symbol");
        return -3;
    }
    g_rccl = r;
    return 0;
}

#define SCN_NCCL(expr)                                                                                  \
    do {                                                                                                \
        int _r = (expr);                                                                                \
        if (_r != 0) {                                                                                  \
            scn::set_error("%s:%d: %s -> rccl error %d (%s)", __FILE__, __LINE__, #expr, _r,            \
                           g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?");                   \
            return 1000 + _r;                                                                           \
        }                                                                                               \
    } while (0)

}  // namespace

}  // namespace scn

struct scnattn_dp_comm {
    void* comm = nullptr;
    hipStream_t stream = nullptr;     // communication stream in use
    hipStream_t own = nullptr;        // the library's own communication stream
    hipEvent_t ready = nullptr;       // recorded on the compute stream before a bucket is reduced
    hipEvent_t done = nullptr;        // recorded on the communication stream after the last bucket
    int world = 1, rank = 0, device = 0;
    long buckets = 0;
};

using namespace scn;

extern "C" {

int scnattn_dp_unique_id(char out[128]) {
    SCN_ARG(out, "dp_unique_id: NULL");
    SCN_TRY(load_rccl());
    ncclUniqueId_t id;
    SCN_NCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(out, id.internal, 128);
    return 0;
}

int scnattn_dp_comm_create(const char id[128], int world, int rank, scnattn_dp_comm** out) {
    SCN_ARG(id && out && world >= 1 && rank >= 0 && rank < world, "dp_comm_create: bad argument");
    SCN_TRY(load_rccl());
    scnattn_dp_comm* c = new scnattn_dp_comm();
    c->world = world;
    c->rank = rank;
    SCN_HIP(hipGetDevice(&c->device));
    ncclUniqueId_t uid;
    std::memcpy(uid.internal, id, 128);
    int rc = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (rc != 0) {
        set_error("dp_comm_create: ncclCommInitRank -> %d (%s)", rc, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
        delete c;
        return 1000 + rc;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking);
    c->stream = c->own;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        set_error("dp_comm_create: %s", hipGetErrorString(e));
        g_rccl.CommDestroy(c->comm);
        delete c;
        return (int)e;
    }
    *out = c;
    return 0;
}

int scnattn_dp_comm_allreduce_bucket(scnattn_dp_comm* c, void* compute_stream, float* buf, long n) {
    SCN_ARG(c && buf && n > 0, "dp_comm_allreduce_bucket: bad argument");
    // the bucket's gradients were produced on the compute stream: the communication stream waits for them, the
    // compute stream carries on with the rest of the backward pass
    SCN_HIP(hipEventRecord(c->ready, reinterpret_cast<hipStream_t>(compute_stream)));
    SCN_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
    SCN_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, c->comm, c->stream));
    c->buckets++;
    return 0;
}

int scnattn_dp_comm_finish(scnattn_dp_comm* c, void* compute_stream) {
    SCN_ARG(c, "dp_comm_finish: NULL");
    if (c->buckets == 0) return 0;
    SCN_HIP(hipEventRecord(c->done, c->stream));
    SCN_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(compute_stream), c->done, 0));
    c->buckets = 0;
    return 0;
}

int scnattn_dp_comm_set_stream(scnattn_dp_comm* c, void* stream) {
    SCN_ARG(c, "dp_comm_set_stream: NULL");
    SCN_ARG(c->buckets == 0, "dp_comm_set_stream: buckets in flight (call scnattn_dp_comm_finish first)");
    c->stream = stream ? reinterpret_cast<hipStream_t>(stream) : c->own;
    return 0;
}

int scnattn_dp_comm_world(const scnattn_dp_comm* c) { return c ? c->world : 0; }

int scnattn_dp_comm_destroy(scnattn_dp_comm* c) {
    if (!c) return 0;
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) g_rccl.CommDestroy(c->comm);
    c->stream = c->own;
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

}  // extern "C"
