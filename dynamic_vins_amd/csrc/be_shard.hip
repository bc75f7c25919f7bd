// be_shard.hip — transports of the landmark-sharded window solve (north_star: "RCCL all-reduce of the reduced camera-pose Hessian").
//
// The exchange is an ALL-GATHER of one vector per rank followed by a rank-ordered sum on every rank (be_shard_finalize_kernel, be_solve.hip): an
// all-reduce whose result does not depend on the collective's internal algorithm (ring / tree / direct over the 7 xGMI links), so all ranks hold the same
// bits, take the same trust-region decisions and therefore enqueue the same number of exchanges — the property that keeps the device-resident dogleg
// loop free of host round trips when the window is sharded.
//
//   transport 1 (RCCL): ncclAllGather on the BA stream, ordered with the kernels around it; nothing is staged through the host.  librccl is resolved
//                       at run time (dlopen), first among the libraries the process has already loaded (a torch process brings its own), so the
//                       product library carries no link-time dependency on it.
//   transport 2 (host): the vector is copied to pinned memory, the caller's all-gather call-back runs (gloo, MPI, shared memory ...), the result is
//                       copied back.  One stream synchronisation per exchange: for tests (two processes on one GPU) and for hosts without RCCL.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "dv_internal.h"
#include "dv_ctx.h"
#include "be_kernels.h"

struct Id128 { char b[128]; };      // ncclUniqueId (passed by value to ncclCommInitRank)
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
}
static Rccl g_rccl;
static std::mutex g_rccl_mu;
static const int kNcclFloat64 = 8;      // ncclDataType_t: ncclFloat64 / ncclDouble

static bool rccl_load(std::string& err) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return true;
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void* h = nullptr;
    for (const char* nm : names) if ((h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;       // already in the process (torch's nccl backend)
    if (!h) for (const char* nm : names) if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { err = std::string("dv_dist: cannot load librccl: ") + dlerror(); return false; }
    Rccl r; r.lib = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) { err = "dv_dist: librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather"; return false; }
    g_rccl = r;
    return true;
}
static std::string rccl_err(int rc) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : ("ncclResult " + std::to_string(rc)); }

int be_dist_buffers(dv_ctx* ctx) {
    DvDist& d = ctx->dist;
    const size_t len = 8 * (size_t)BE_XS_LEN(BE_MAX_LM);
    DV_CHECK(d.xsend.ensure(len));
    DV_CHECK(d.xrecv.ensure(len * (size_t)d.world));
    if (d.transport == 2 && !d.h_send) {
        DV_CHECK(hipHostMalloc(&d.h_send, len, hipHostMallocDefault));
        DV_CHECK(hipHostMalloc(&d.h_recv, len * (size_t)d.world, hipHostMallocDefault));
    }
    return 0;
}
void be_dist_release(dv_ctx* ctx) {
    DvDist& d = ctx->dist;
    if (d.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d.comm);
    d.comm = nullptr;
    d.xsend.release(); d.xrecv.release();
    if (d.h_send) (void)hipHostFree(d.h_send);
    if (d.h_recv) (void)hipHostFree(d.h_recv);
    d.h_send = d.h_recv = nullptr; d.transport = 0; d.rank = 0; d.world = 1; d.fn = nullptr; d.user = nullptr;
}

int be_exchange(dv_ctx* ctx, size_t count, hipStream_t s) {
    DvDist& d = ctx->dist;
    d.exchanges++;
    if (d.transport == 1) {
        const int rc = g_rccl.AllGather(d.xsend.p, d.xrecv.p, count, kNcclFloat64, d.comm, s);
        if (rc != 0) DV_FAIL("dv_dist: ncclAllGather: " + rccl_err(rc));
        return 0;
    }
    if (d.transport == 2) {
        DV_CHECK(hipMemcpyAsync(d.h_send, d.xsend.p, 8 * count, hipMemcpyDeviceToHost, s));
        DV_CHECK(hipStreamSynchronize(s));
        if (d.fn(d.user, d.h_send, d.h_recv, 8 * count) != 0) DV_FAIL("dv_dist: the all-gather call-back failed");
        DV_CHECK(hipMemcpyAsync(d.xrecv.p, d.h_recv, 8 * count * (size_t)d.world, hipMemcpyHostToDevice, s));
        return 0;
    }
    DV_FAIL("dv_dist: no transport");
}

extern "C" {

int dv_dist_unique_id(uint8_t id[128]) {
    std::string err;
    if (!id || !rccl_load(err)) { dv_set_error(nullptr, err.empty() ? "dv_dist_unique_id: null argument" : err); return -1; }
    Id128 u; std::memset(&u, 0, sizeof(u));
    const int rc = g_rccl.GetUniqueId(&u);
    if (rc != 0) { dv_set_error(nullptr, "dv_dist_unique_id: " + rccl_err(rc)); return -1; }
    std::memcpy(id, u.b, 128);
    return 0;
}

int dv_dist_init_rccl(dv_ctx* ctx, int rank, int world, const uint8_t id[128]) {
    if (!ctx) return -1;
    if (!id || world < 1 || rank < 0 || rank >= world || world > 64) DV_FAIL("dv_dist_init_rccl: bad rank / world / id");
    if (ctx->be.pend->active) DV_FAIL("dv_dist_init_rccl: a solve is in flight");
    std::string err;
    if (!rccl_load(err)) DV_FAIL(err);
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    be_dist_release(ctx);
    Id128 u; std::memcpy(u.b, id, 128);
    void* comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, world, u, rank);
    if (rc != 0) DV_FAIL("dv_dist_init_rccl: ncclCommInitRank: " + rccl_err(rc));
    DvDist& d = ctx->dist;
    d.comm = comm; d.transport = 1; d.rank = rank; d.world = world;
    return be_dist_buffers(ctx);
}

int dv_dist_init_host(dv_ctx* ctx, int rank, int world, dv_allgather_fn fn, void* user) {
    if (!ctx) return -1;
    if (!fn || world < 1 || rank < 0 || rank >= world || world > 64) DV_FAIL("dv_dist_init_host: bad rank / world / call-back");
    if (ctx->be.pend->active) DV_FAIL("dv_dist_init_host: a solve is in flight");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    be_dist_release(ctx);
    DvDist& d = ctx->dist;
    d.transport = 2; d.rank = rank; d.world = world; d.fn = fn; d.user = user;
    return be_dist_buffers(ctx);
}

int dv_dist_shutdown(dv_ctx* ctx) {
    if (!ctx) return -1;
    if (ctx->be.pend->active) DV_FAIL("dv_dist_shutdown: a solve is in flight");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    DV_CHECK(hipStreamSynchronize(ctx->be_stream));
    be_dist_release(ctx);
    return 0;
}

int dv_dist_info(dv_ctx* ctx, int* rank, int* world, int* transport, long long* exchanges) {
    if (!ctx) return -1;
    if (rank) *rank = ctx->dist.rank;
    if (world) *world = ctx->dist.world;
    if (transport) *transport = ctx->dist.transport;
    if (exchanges) *exchanges = ctx->dist.exchanges;
    return 0;
}

// operator form of the exchange: S_g (host, n doubles: this rank's partial [S | g | cost]) -> the rank-ordered sum over all ranks, same bits everywhere
int dv_allreduce_reduced_system(dv_ctx* ctx, double* S_g, int n) {
    if (!ctx) return -1;
    DvDist& d = ctx->dist;
    if (!S_g || n < 1) DV_FAIL("dv_allreduce_reduced_system: null argument");
    if (d.transport == 0) { if (d.world == 1) return 0; DV_FAIL("dv_allreduce_reduced_system: dv_dist_init_* was not called"); }
    if (ctx->be.pend->active) DV_FAIL("dv_allreduce_reduced_system: a solve is in flight");
    if ((size_t)n > (size_t)BE_XS_LEN(BE_MAX_LM)) DV_FAIL("dv_allreduce_reduced_system: n exceeds the exchange buffer");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    hipStream_t s = ctx->be_stream;
    std::vector<double> all((size_t)n * d.world);
    if (d.transport == 2) {
        std::memcpy(d.h_send, S_g, 8 * (size_t)n);
        if (d.fn(d.user, d.h_send, d.h_recv, 8 * (size_t)n) != 0) DV_FAIL("dv_dist: the all-gather call-back failed");
        std::memcpy(all.data(), d.h_recv, 8 * all.size());
        d.exchanges++;
    } else {
        DV_CHECK(hipMemcpyAsync(d.xsend.p, S_g, 8 * (size_t)n, hipMemcpyHostToDevice, s));
        if (be_exchange(ctx, (size_t)n, s)) return -1;
        DV_CHECK(hipMemcpyAsync(all.data(), d.xrecv.p, 8 * all.size(), hipMemcpyDeviceToHost, s));
        DV_CHECK(hipStreamSynchronize(s));
    }
    for (int i = 0; i < n; ++i) { double v = all[i]; for (int r = 1; r < d.world; ++r) v += all[(size_t)r * n + i]; S_g[i] = v; }
    return 0;
}

}  // extern "C"
