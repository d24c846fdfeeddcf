// agh_comm.cpp -- the multi-GPU part of the C-ABI (include/agrep_hip.h): the only exchange of a
// sharded scan is the -c sum / the -l per-file hit vector (the plumbing it feeds is
// exec()'s per-file count printing, agrep.c:3444-3558).  Built directly on RCCL (ncclAllReduce
// over xGMI); RCCL is opened with dlopen at the first use, so single-GPU users of
// libagrep_hip.so never load it.
#include <dlfcn.h>
#include <errno.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <fcntl.h>
#include <stdlib.h>
#include <unistd.h>

#include <mutex>

#include "../../include/agrep_hip.h"

extern "C" __attribute__((visibility("hidden"))) void agh_set_error(const char *msg);   // agh_api.cpp: agh_last_error() text

static int cfail(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    agh_set_error(buf);
    errno = AGH_ERRNO;
    return -1;
}

namespace {
struct rccl_api {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
rccl_api R;
std::once_flag R_once;
bool R_ok = false;
char R_err[256] = "";

void load_rccl()
{
    // by soname first: a process that already holds an RCCL (PyTorch ships one with the same
    // soname) keeps exactly one copy
    const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *n : names) {
        R.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (R.handle) break;
    }
    if (!R.handle) {
        snprintf(R_err, sizeof(R_err), "librccl.so.1 not found: %s", dlerror());
        return;
    }
#define AGH_SYM(field, name)                                              \
    *(void **)(&R.field) = dlsym(R.handle, name);                         \
    if (!R.field) { snprintf(R_err, sizeof(R_err), "RCCL lacks %s", name); return; }
    AGH_SYM(GetUniqueId, "ncclGetUniqueId")
    AGH_SYM(CommInitRank, "ncclCommInitRank")
    AGH_SYM(CommInitAll, "ncclCommInitAll")
    AGH_SYM(CommDestroy, "ncclCommDestroy")
    AGH_SYM(AllReduce, "ncclAllReduce")
    AGH_SYM(GroupStart, "ncclGroupStart")
    AGH_SYM(GroupEnd, "ncclGroupEnd")
    AGH_SYM(GetErrorString, "ncclGetErrorString")
#undef AGH_SYM
    R_ok = true;
}

int need_rccl()
{
    std::call_once(R_once, load_rccl);
    if (!R_ok) return cfail("%s", R_err);
    return 0;
}
}   // namespace

// RCCL 2.27 prints a version banner on STDOUT when the first communicator is created.  A scan
// library must not write to its caller's stdout (agrep's stdout is the match list), so fd 1
// points at /dev/null while a communicator is being created (AGH_RCCL_BANNER=1 keeps it).
struct stdout_silencer {
    int saved = -1;
    stdout_silencer()
    {
        const char *e = getenv("AGH_RCCL_BANNER");
        if (e && e[0] == '1') return;
        fflush(stdout);                         // the caller's pending output goes where it belongs
        const int nul = open("/dev/null", O_WRONLY);
        if (nul < 0) return;
        saved = dup(1);
        if (saved >= 0) (void)dup2(nul, 1);
        close(nul);
    }
    ~stdout_silencer()
    {
        if (saved < 0) return;
        fflush(stdout);                         // the banner, if it was buffered
        (void)dup2(saved, 1);
        close(saved);
    }
};

#define NCCL_TRY(expr)                                                                     \
    do {                                                                                   \
        ncclResult_t r__ = (expr);                                                         \
        if (r__ != ncclSuccess)                                                            \
            return cfail("%s failed: %s (%s:%d)", #expr, R.GetErrorString(r__), __FILE__,  \
                         __LINE__);                                                        \
    } while (0)
#define HIPC_TRY(expr)                                                                     \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess)                                                             \
            return cfail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                         __LINE__);                                                        \
    } while (0)

struct agh_comm {
    agh_allreduce_fn custom = nullptr;      // agh_comm_init_custom: the caller's own transport instead of RCCL
    void *custom_ctx = nullptr;
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, nranks = 1;
    hipStream_t stream = nullptr;
    uint64_t *d_buf = nullptr;              // counts (2 x uint64) or file hits (bytes)
    size_t d_cap = 0;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;   // agh_comm_allreduce_dev: the caller's stream <-> ours
};

static int comm_setup(agh_comm *c, size_t bytes)
{
    HIPC_TRY(hipSetDevice(c->device));
    if (!c->stream) HIPC_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if (bytes > c->d_cap) {
        if (c->d_buf) (void)hipFree(c->d_buf);
        c->d_buf = nullptr;
        c->d_cap = 0;
        const size_t want = bytes < 4096 ? 4096 : bytes;
        HIPC_TRY(hipMalloc((void **)&c->d_buf, want));
        c->d_cap = want;
    }
    return 0;
}

extern "C" int agh_comm_unique_id(unsigned char id[AGH_UNIQUE_ID_BYTES])
{
    if (!id) return cfail("null argument");
    if (need_rccl()) return -1;
    static_assert(sizeof(ncclUniqueId) == AGH_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    NCCL_TRY(R.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return 0;
}

extern "C" agh_comm *agh_comm_init_rank(const unsigned char id[AGH_UNIQUE_ID_BYTES], int nranks,
                                        int rank)
{
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) {
        cfail("agh_comm_init_rank: bad arguments (nranks=%d, rank=%d)", nranks, rank);
        return nullptr;
    }
    if (need_rccl()) return nullptr;
    agh_comm *c = new agh_comm();
    if (hipGetDevice(&c->device) != hipSuccess) {
        cfail("no usable HIP device");
        delete c;
        return nullptr;
    }
    c->rank = rank;
    c->nranks = nranks;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t r;
    {
        stdout_silencer quiet;
        r = R.CommInitRank(&c->comm, nranks, u, rank);
    }
    if (r != ncclSuccess) {
        cfail("ncclCommInitRank failed: %s", R.GetErrorString(r));
        delete c;
        return nullptr;
    }
    return c;
}

// A communicator over the caller's own transport (MPI, gloo, a socket): every reduction of this file calls
// fn(ctx, buf, count, elem_bytes) on HOST memory -- elem_bytes 8: sum of uint64, 1: max of bytes -- and expects the
// reduced values back in buf.  For hosts without RCCL between the ranks (several nodes over Ethernet) and for
// running the N > 1 step of a scan with two ranks on one GPU (tests/test_gpu_records.py).
extern "C" agh_comm *agh_comm_init_custom(agh_allreduce_fn fn, void *ctx, int nranks, int rank)
{
    if (!fn || nranks < 1 || rank < 0 || rank >= nranks) {
        cfail("agh_comm_init_custom: bad arguments (nranks=%d, rank=%d)", nranks, rank);
        return nullptr;
    }
    agh_comm *c = new agh_comm();
    if (hipGetDevice(&c->device) != hipSuccess) {
        cfail("no usable HIP device");
        delete c;
        return nullptr;
    }
    c->custom = fn;
    c->custom_ctx = ctx;
    c->rank = rank;
    c->nranks = nranks;
    return c;
}

extern "C" int agh_comm_init_all(agh_comm **comms, int ndev, const int *devices)
{
    if (!comms || ndev < 1 || ndev > 64) return cfail("agh_comm_init_all: bad arguments");
    if (need_rccl()) return -1;
    ncclComm_t raw[64];
    int devs[64];
    for (int i = 0; i < ndev; ++i) devs[i] = devices ? devices[i] : i;
    {
        stdout_silencer quiet;
        ncclResult_t r = R.CommInitAll(raw, ndev, devs);
        if (r != ncclSuccess) return cfail("ncclCommInitAll failed: %s", R.GetErrorString(r));
    }
    for (int i = 0; i < ndev; ++i) {
        comms[i] = new agh_comm();
        comms[i]->comm = raw[i];
        comms[i]->device = devs[i];
        comms[i]->rank = i;
        comms[i]->nranks = ndev;
    }
    return 0;
}

extern "C" void agh_comm_free(agh_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm && R_ok) (void)R.CommDestroy(c->comm);
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    if (c->ev_out) (void)hipEventDestroy(c->ev_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int agh_comm_info(const agh_comm *c, int *rank, int *nranks, int *device)
{
    if (!c) return cfail("null communicator");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    if (device) *device = c->device;
    return 0;
}

// One all-reduce over `n` communicators of THIS process (n == 1: the usual one-process-per-GPU
// case; n > 1: one process driving several GPUs, grouped so that RCCL sees them together).
static int all_reduce_many(agh_comm *const *cs, int n, void *const *host, size_t count,
                           ncclDataType_t dt, ncclRedOp_t op, size_t elem)
{
    if (n >= 1 && cs[0] && cs[0]->custom) {     // the caller's transport: one communicator per process
        if (n != 1 || !host[0]) return cfail("a custom communicator reduces one rank per process");
        if (cs[0]->custom(cs[0]->custom_ctx, host[0], count, (int)elem)) return cfail("the caller's all-reduce failed");
        return 0;
    }
    if (need_rccl()) return -1;
    for (int i = 0; i < n; ++i) {
        if (!cs[i] || !host[i]) return cfail("null argument");
        if (comm_setup(cs[i], count * elem)) return -1;
        HIPC_TRY(hipMemcpyAsync(cs[i]->d_buf, host[i], count * elem, hipMemcpyHostToDevice, cs[i]->stream));
    }
    if (n > 1) NCCL_TRY(R.GroupStart());
    // inside the group nothing returns early: an open group would leave every later collective of
    // this thread undefined (and agh_comm_free could hang), so the first error is kept and the group
    // is closed before it is reported
    int rc = 0;
    for (int i = 0; i < n && rc == 0; ++i) {
        const hipError_t he = hipSetDevice(cs[i]->device);
        if (he != hipSuccess) {
            rc = cfail("hipSetDevice(%d) failed: %s", cs[i]->device, hipGetErrorString(he));
            break;
        }
        const ncclResult_t nr = R.AllReduce(cs[i]->d_buf, cs[i]->d_buf, count, dt, op, cs[i]->comm, cs[i]->stream);
        if (nr != ncclSuccess) rc = cfail("ncclAllReduce failed on rank %d: %s", cs[i]->rank, R.GetErrorString(nr));
    }
    if (n > 1) {
        const ncclResult_t ge = R.GroupEnd();
        if (ge != ncclSuccess && rc == 0) rc = cfail("ncclGroupEnd failed: %s", R.GetErrorString(ge));
    }
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        HIPC_TRY(hipSetDevice(cs[i]->device));
        HIPC_TRY(hipMemcpyAsync(host[i], cs[i]->d_buf, count * elem, hipMemcpyDeviceToHost, cs[i]->stream));
        HIPC_TRY(hipStreamSynchronize(cs[i]->stream));
    }
    return 0;
}

// internal (agh_api.cpp, agh_scan_device_reduce): uint64 sums in place -- on a device buffer, enqueued on the
// caller's stream behind the scan's kernels (no host round trip), or from host memory
extern "C" __attribute__((visibility("hidden"))) int agh_comm_allreduce_dev(agh_comm *c, uint64_t *d_buf, size_t count,
                                                                            hipStream_t st)
{
    if (!c || !d_buf) return cfail("null argument");
    if (c->custom) {                            // host transport: the values come down, are reduced, go back up
        uint64_t h[8];
        if (count > 8) return cfail("internal error: %zu values in a device all-reduce", count);
        HIPC_TRY(hipMemcpyAsync(h, d_buf, count * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIPC_TRY(hipStreamSynchronize(st));
        if (c->custom(c->custom_ctx, h, count, 8)) return cfail("the caller's all-reduce failed");
        HIPC_TRY(hipMemcpyAsync(d_buf, h, count * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        HIPC_TRY(hipStreamSynchronize(st));
        return 0;
    }
    if (need_rccl()) return -1;
    // RCCL runs on the communicator's own non-blocking stream, never on the caller's (which may be the legacy
    // default stream, with its implicit synchronisation against every blocking stream): the caller's stream
    // hands over through an event and takes the result back through another -- stream-ordered on both sides,
    // no host wait in between
    if (comm_setup(c, 64)) return -1;
    if (!c->ev_in) {
        HIPC_TRY(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
        HIPC_TRY(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
    }
    HIPC_TRY(hipEventRecord(c->ev_in, st));
    HIPC_TRY(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    NCCL_TRY(R.AllReduce(d_buf, d_buf, count, ncclUint64, ncclSum, c->comm, c->stream));
    HIPC_TRY(hipEventRecord(c->ev_out, c->stream));
    HIPC_TRY(hipStreamWaitEvent(st, c->ev_out, 0));
    return 0;
}

extern "C" __attribute__((visibility("hidden"))) int agh_comm_allreduce_host(agh_comm *c, uint64_t *v, size_t count)
{
    if (!c || !v) return cfail("null argument");
    int dev = 0;
    (void)hipGetDevice(&dev);
    void *h = v;
    const int rc = all_reduce_many(&c, 1, &h, count, ncclUint64, ncclSum, sizeof(uint64_t));
    (void)hipSetDevice(dev);
    return rc;
}

extern "C" int agh_reduce_counts(agh_comm *c, uint64_t counts[2])
{
    if (!c || !counts) return cfail("null argument");
    int dev = 0;
    (void)hipGetDevice(&dev);
    void *h = counts;
    const int rc = all_reduce_many(&c, 1, &h, 2, ncclUint64, ncclSum, sizeof(uint64_t));
    (void)hipSetDevice(dev);
    return rc;
}

extern "C" int agh_reduce_counts_all(agh_comm *const *comms, int n, uint64_t (*counts)[2])
{
    if (!comms || !counts || n < 1 || n > 64) return cfail("bad arguments");
    int dev = 0;
    (void)hipGetDevice(&dev);
    void *h[64];
    for (int i = 0; i < n; ++i) h[i] = counts[i];
    const int rc = all_reduce_many(comms, n, h, 2, ncclUint64, ncclSum, sizeof(uint64_t));
    (void)hipSetDevice(dev);
    return rc;
}

extern "C" int agh_reduce_file_hits(agh_comm *c, unsigned char *hits, size_t n_files)
{
    if (!c || (!hits && n_files)) return cfail("null argument");
    if (!n_files) return 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    void *h = hits;
    const int rc = all_reduce_many(&c, 1, &h, n_files, ncclUint8, ncclMax, 1);
    (void)hipSetDevice(dev);
    return rc;
}

extern "C" int agh_reduce_file_hits_all(agh_comm *const *comms, int n, unsigned char *const *hits,
                                        size_t n_files)
{
    if (!comms || !hits || n < 1 || n > 64) return cfail("bad arguments");
    if (!n_files) return 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    void *h[64];
    for (int i = 0; i < n; ++i) h[i] = hits[i];
    const int rc = all_reduce_many(comms, n, h, n_files, ncclUint8, ncclMax, 1);
    (void)hipSetDevice(dev);
    return rc;
}
