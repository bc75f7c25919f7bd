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
//   transport 3 (peer): the one-shot exchange of SURVEY 5: every rank WRITES its vector straight into a window each peer exposes (hipIpc mapping: over the
//                       direct xGMI link between the two GPUs, one hop, no ring), then raises a per-sender sequence flag in that window; a rank's own wait
//                       kernel spins on its local flags and copies the gathered vectors to the exchange buffer.  Two launches per exchange, no
//                       library call, no host round trip.  142 KB per rank and exchange: latency-, not bandwidth-bound — which is the case ring
//                       all-gathers are worst at (world - 1 steps).  The window is double-buffered by sequence parity: a rank that is one exchange
//                       ahead writes the other half (it cannot be two ahead: exchange k + 2 needs every peer's flag k + 1, raised after that peer
//                       consumed exchange k).  UNMEASURED on a multi-GPU node (this pool hands out one GPU at a time): tested world = 1 and two
//                       processes sharing one GPU.
//   transport 2 (host): the vector is copied to pinned memory, the caller's all-gather call-back runs (gloo, MPI, shared memory ...), the result is
//                       copied back.  One stream synchronisation per exchange: for tests (two processes on one GPU) and for hosts without RCCL.
#include <dlfcn.h>
#include <algorithm>
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
    int (*CommCount)(void*, int*) = nullptr;
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
    r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) { err = "dv_dist: librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather"; return false; }
    g_rccl = r;
    return true;
}
static std::string rccl_err(int rc) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : ("ncclResult " + std::to_string(rc)); }

int be_dist_buffers(dv_ctx* ctx) {
    DvDist& d = ctx->dist;
    const size_t len = 8 * (size_t)BE_XS_BUF;
    DV_CHECK(d.xsend.ensure(len));
    DV_CHECK(d.xrecv.ensure(len * (size_t)d.world));
    DV_CHECK(d.qf.ensure(8 * 2 * (size_t)BE_QF_LEN));
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
    d.xsend.release(); d.xrecv.release(); d.qf.release();
    if (d.h_send) (void)hipHostFree(d.h_send);
    if (d.h_recv) (void)hipHostFree(d.h_recv);
    for (int r = 0; r < 64; ++r) { if (d.peer_win[r] && d.peer_win[r] != d.win) (void)hipIpcCloseMemHandle(d.peer_win[r]); d.peer_win[r] = nullptr; }
    if (d.win) (void)hipFree(d.win);
    if (d.peer_dead) (void)hipHostFree(d.peer_dead);
    d.peer_dead = nullptr;
    d.win = nullptr; d.win_slot = 0; d.seq = 0;
    d.h_send = d.h_recv = nullptr; d.transport = 0; d.rank = 0; d.world = 1; d.fn = nullptr; d.user = nullptr;
}

// ---- transport 3 ----
#define PEER_MAX 64
struct PeerArgs { double* win[PEER_MAX]; };      // every rank's window in this process's address space
__host__ __device__ inline size_t peer_flag_off(size_t slot, int world) { return (2 * (size_t)world * slot * 8 + 255) / 256 * 256; }      // bytes
// grid (chunks, world): block (b, r) writes this rank's vector into rank r's window, slot [parity][rank]
__global__ __launch_bounds__(256) void peer_push_kernel(PeerArgs pa, const double* __restrict__ src, int count, size_t slot, int rank, int world, int parity) {
    double* dst = pa.win[blockIdx.y] + ((size_t)parity * world + rank) * slot;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) dst[i] = src[i];
}
// block 0 first raises this rank's flag in every window (the push kernel before it on the stream has completed: its writes are visible system-wide); every
// block then waits until all senders' flags in the LOCAL window carry this exchange's sequence number and copies the gathered vectors out
__global__ __launch_bounds__(256) void peer_wait_kernel(PeerArgs pa, double* __restrict__ out, int count, size_t slot, int rank, int world, int parity, unsigned long long seq,
                                                        unsigned long long* __restrict__ dead, long long timeout_ticks) {
    const size_t foff = peer_flag_off(slot, world) / 8;
    if (blockIdx.x == 0 && (int)threadIdx.x < world) {
        unsigned long long* f = reinterpret_cast<unsigned long long*>(pa.win[threadIdx.x] + foff) + (size_t)parity * world + rank;
        __hip_atomic_store(f, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    unsigned long long* mine = reinterpret_cast<unsigned long long*>(pa.win[rank] + foff);
    if ((int)threadIdx.x < world) {
        unsigned long long* f = mine + (size_t)parity * world + threadIdx.x;
        const long long t0 = wall_clock64();
        // bounded by the constant 100 MHz clock, not by a spin count; a time-out is recorded in pinned host memory (the collect of the solve reads it: be_dist_check)
        // and is sticky — once a peer is known dead no later exchange waits for it again
        unsigned spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            if ((spins++ & 63u) == 0u && __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull) break;      // (a PCIe read: only now and then)
            if (wall_clock64() - t0 > timeout_ticks) { __hip_atomic_store(dead, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const double* src = pa.win[rank] + (size_t)parity * world * slot;
    for (int r = 0; r < world; ++r)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) out[(size_t)r * count + i] = src[(size_t)r * slot + i];
}
static int peer_exchange(dv_ctx* ctx, size_t count, hipStream_t s) {
    DvDist& d = ctx->dist;
    PeerArgs pa{};
    for (int r = 0; r < d.world; ++r) pa.win[r] = (double*)d.peer_win[r];
    const unsigned long long seq = ++d.seq; const int parity = (int)(seq & 1);
    const int chunks = (int)std::min<size_t>(16, (count + 1023) / 1024);
    hipLaunchKernelGGL(peer_push_kernel, dim3(chunks, d.world), dim3(256), 0, s, pa, (const double*)d.xsend.p, (int)count, d.win_slot, d.rank, d.world, parity);
    hipLaunchKernelGGL(peer_wait_kernel, dim3(chunks), dim3(256), 0, s, pa, (double*)d.xrecv.p, (int)count, d.win_slot, d.rank, d.world, parity, seq, d.peer_dead, d.peer_timeout_ticks);
    DV_CHECK(hipGetLastError());
    return 0;
}

int be_dist_check(dv_ctx* ctx) {
    const DvDist& d = ctx->dist;
    if (d.transport == 3 && d.peer_dead && *(volatile unsigned long long*)d.peer_dead != 0ull)
        DV_FAIL("dv_dist: a peer did not deliver its exchange vector (transport peer: wait timed out at exchange " + std::to_string(*(volatile unsigned long long*)d.peer_dead) + "); the gathered buffer was not used");
    return 0;
}

int be_exchange(dv_ctx* ctx, size_t count, hipStream_t s) {
    DvDist& d = ctx->dist;
    d.exchanges++;
    if (d.transport == 1) {
        const int rc = g_rccl.AllGather(d.xsend.p, d.xrecv.p, count, kNcclFloat64, d.comm, s);
        if (rc != 0) DV_FAIL("dv_dist: ncclAllGather: " + rccl_err(rc));
        return 0;
    }
    if (d.transport == 3) return peer_exchange(ctx, count, s);
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

// transport 3, step 1: allocate the window this rank exposes and hand out its IPC handle (64 bytes); the launcher all-gathers the handles
int dv_dist_peer_prepare(dv_ctx* ctx, int rank, int world, uint8_t handle[64]) {
    if (!ctx) return -1;
    if (!handle || world < 1 || rank < 0 || rank >= world || world > PEER_MAX) DV_FAIL("dv_dist_peer_prepare: bad rank / world / handle");
    if (ctx->be.pend->active) DV_FAIL("dv_dist_peer_prepare: a solve is in flight");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    be_dist_release(ctx);
    DvDist& d = ctx->dist;
    d.rank = rank; d.world = world; d.win_slot = (size_t)BE_XS_BUF;
    const size_t bytes = peer_flag_off(d.win_slot, world) + (2 * (size_t)world + 1) * 8 + 256;
    // fine-grained: remote writes and the local polling loads must be coherent inside a running kernel, not only at kernel boundaries
    // (no coarse-grained fall-back: in-kernel polling of remotely written flags is not coherent there — spurious time-outs or stale vectors)
    if (hipExtMallocWithFlags(&d.win, bytes, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); d.win = nullptr; DV_FAIL("dv_dist_peer_prepare: fine-grained device memory is not available on this device; use the rccl or host transport"); }
    DV_CHECK(hipHostMalloc((void**)&d.peer_dead, 64, hipHostMallocDefault));
    *d.peer_dead = 0ull;
    DV_CHECK(hipMemset(d.win, 0, bytes));
    hipIpcMemHandle_t h;
    DV_CHECK(hipIpcGetMemHandle(&h, d.win));
    std::memcpy(handle, &h, 64);
    return 0;
}
// step 2: handles[world][64] in rank order (this rank's own entry is not opened)
int dv_dist_init_peer(dv_ctx* ctx, const uint8_t* handles) {
    if (!ctx) return -1;
    DvDist& d = ctx->dist;
    if (!handles || !d.win) DV_FAIL("dv_dist_init_peer: dv_dist_peer_prepare was not called");
    if (ctx->be.pend->active) DV_FAIL("dv_dist_init_peer: a solve is in flight");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    for (int r = 0; r < d.world; ++r) {
        if (r == d.rank) { d.peer_win[r] = d.win; continue; }
        hipIpcMemHandle_t h; std::memcpy(&h, handles + 64 * (size_t)r, 64);
        DV_CHECK(hipIpcOpenMemHandle(&d.peer_win[r], h, hipIpcMemLazyEnablePeerAccess));
    }
    d.transport = 3;
    return be_dist_buffers(ctx);
}

// ranks of the RCCL communicator as the library itself reports them (ncclCommCount); 0 when the transport is not RCCL
int dv_dist_rccl_ranks(dv_ctx* ctx, int* n) {
    if (!ctx || !n) return -1;
    *n = 0;
    if (ctx->dist.transport != 1 || !ctx->dist.comm || !g_rccl.CommCount) return 0;
    const int rc = g_rccl.CommCount(ctx->dist.comm, n);
    if (rc != 0) DV_FAIL("dv_dist_rccl_ranks: ncclCommCount: " + rccl_err(rc));
    return 0;
}

int dv_dist_shutdown(dv_ctx* ctx) {
    if (!ctx) return -1;
    if (ctx->be.pend->active) DV_FAIL("dv_dist_shutdown: a solve is in flight");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    DV_CHECK(hipStreamSynchronize(ctx->be_stream));
    be_dist_release(ctx);
    return 0;
}

int dv_dist_exchange_bytes(dv_ctx* ctx, int n_landmarks, long long* system_bytes, long long* cost_bytes, long long* depth_bytes) {
    if (!ctx) return -1;
    const int world = ctx->dist.world > 0 ? ctx->dist.world : 1;
    if (system_bytes) *system_bytes = 8ll * BE_XS_LEN;                                   // per rank and linearisation: independent of the number of landmarks
    if (cost_bytes) *cost_bytes = 64;                                                     // one partial sum per rank (padded)
    if (depth_bytes) *depth_bytes = 8ll * std::max(1, (n_landmarks + world - 1) / world);   // once per solve, behind the last iteration slot
    return 0;
}
int dv_dist_info(dv_ctx* ctx, int* rank, int* world, int* transport, long long* exchanges) {
    if (!ctx) return -1;
    if (rank) *rank = ctx->dist.rank;
    if (world) *world = ctx->dist.world;
    if (transport) *transport = ctx->dist.transport;
    if (exchanges) *exchanges = ctx->dist.exchanges;
    if (be_dist_check(ctx)) return -1;      // a wait kernel gave up on a peer
    return 0;
}

// operator form of the exchange: S_g (host, n doubles: this rank's partial [S | g | cost]) -> the rank-ordered sum over all ranks, same bits everywhere
int dv_allreduce_reduced_system(dv_ctx* ctx, double* S_g, int n) {
    if (!ctx) return -1;
    DvDist& d = ctx->dist;
    if (!S_g || n < 1) DV_FAIL("dv_allreduce_reduced_system: null argument");
    if (d.transport == 0) { if (d.world == 1) return 0; DV_FAIL("dv_allreduce_reduced_system: dv_dist_init_* was not called"); }
    if (ctx->be.pend->active) DV_FAIL("dv_allreduce_reduced_system: a solve is in flight");
    if ((size_t)n > (size_t)BE_XS_BUF) DV_FAIL("dv_allreduce_reduced_system: n exceeds the exchange buffer");
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
        if (be_dist_check(ctx)) return -1;
    }
    for (int i = 0; i < n; ++i) { double v = all[i]; for (int r = 1; r < d.world; ++r) v += all[(size_t)r * n + i]; S_g[i] = v; }
    return 0;
}

}  // extern "C"
