// pgicp_comm.cpp -- the one collective of the path: the all-gather of loop-closure edge
// records between the GPUs of a node, over RCCL (xGMI).  One process per GPU; the communicator is bootstrapped from a
// unique id the caller passes between the processes by whatever channel it has (a file, MPI, a socket, torch's store).
// RCCL is opened at run time (dlopen), the first time a communicator is asked for: a single-GPU user of libpgicp never
// loads it, and a process that already holds an RCCL (PyTorch's) shares that one instead of mapping a second copy.
#include "pgicp.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

static_assert(PGICP_UNIQUE_ID_BYTES >= sizeof(ncclUniqueId), "unique id does not fit");

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot open librccl: ") + dlerror(); return false; }
        GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllGather || !GetErrorString) { err = "librccl lacks an expected symbol"; lib = nullptr; return false; }
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
thread_local std::string t_comm_err;

}  // namespace

struct pgicp_comm {
    ncclComm_t comm = nullptr;
    pgicp_ctx *ctx = nullptr;
    int world = 1, rank = 0, device = 0;
    void *d_send = nullptr, *d_recv = nullptr;
    size_t cap_send = 0, cap_recv = 0;
    std::vector<pgicp_edge> h_send, h_recv;
};

extern "C" {

const char *pgicp_comm_last_error(void) { return t_comm_err.c_str(); }

int pgicp_comm_unique_id(char id[PGICP_UNIQUE_ID_BYTES])
{
    if (!id) return PGICP_ERR_ARG;
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl.load()) { t_comm_err = g_rccl.err; return PGICP_ERR_HIP; }
    ncclUniqueId u;
    const ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) { t_comm_err = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return PGICP_ERR_HIP; }
    std::memset(id, 0, PGICP_UNIQUE_ID_BYTES);
    std::memcpy(id, &u, sizeof u);
    return PGICP_OK;
}

int pgicp_comm_create(pgicp_ctx *ctx, int world_size, int rank, const char id[PGICP_UNIQUE_ID_BYTES], pgicp_comm **out)
{
    if (!ctx || !out || !id || world_size < 1 || rank < 0 || rank >= world_size) { t_comm_err = "pgicp_comm_create: bad argument"; return PGICP_ERR_ARG; }
    *out = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        if (!g_rccl.load()) { t_comm_err = g_rccl.err; return PGICP_ERR_HIP; }
    }
    int device = 0;
    if (pgicp_ctx_device(ctx, &device) != PGICP_OK || hipSetDevice(device) != hipSuccess) { t_comm_err = "pgicp_comm_create: cannot select the context's device"; return PGICP_ERR_HIP; }
    (void)hipGetLastError();                        // RCCL reads the runtime's sticky last-error: a stale one from an earlier,
                                                    // already reported failure in this process would fail the initialisation
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    pgicp_comm *c = new pgicp_comm();
    c->ctx = ctx; c->world = world_size; c->rank = rank; c->device = device;
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world_size, u, rank);
    if (r != ncclSuccess) { t_comm_err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); delete c; return PGICP_ERR_HIP; }
    *out = c;
    return PGICP_OK;
}

void pgicp_comm_destroy(pgicp_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

int pgicp_comm_info(const pgicp_comm *c, int *world_size, int *rank)
{
    if (!c) return PGICP_ERR_ARG;
    if (world_size) *world_size = c->world;
    if (rank) *rank = c->rank;
    return PGICP_OK;
}

int pgicp_allgather_edges(pgicp_comm *c, const pgicp_edge *local, const int *pair_index, int n_local, int slots_per_rank, int n_total,
                          pgicp_edge *out)
{
    if (!c || n_local < 0 || slots_per_rank < n_local || n_total < 0 || !out || (n_local > 0 && (!local || !pair_index))) {
        t_comm_err = "pgicp_allgather_edges: bad argument";
        return PGICP_ERR_ARG;
    }
    for (int k = 0; k < n_local; k++)
        if (pair_index[k] < 0 || pair_index[k] >= n_total) { t_comm_err = "pgicp_allgather_edges: pair index out of range"; return PGICP_ERR_ARG; }
    if (hipSetDevice(c->device) != hipSuccess) { t_comm_err = "hipSetDevice failed"; return PGICP_ERR_HIP; }
    hipStream_t st = (hipStream_t)pgicp_ctx_stream(c->ctx);
    const size_t block = sizeof(pgicp_edge) * (size_t)slots_per_rank;
    // fixed-size blocks: this rank's edges first, then empty slots (pair index -1 in reserved[0])
    c->h_send.assign((size_t)slots_per_rank, pgicp_edge());
    for (int k = 0; k < slots_per_rank; k++) {
        pgicp_edge &e = c->h_send[k];
        if (k < n_local) { e = local[k]; e.reserved[0] = (double)pair_index[k]; }
        else { std::memset(&e, 0, sizeof e); e.from_id = -1; e.to_id = -1; e.status = -1; e.reserved[0] = -1.0; }
    }
    c->h_recv.resize((size_t)slots_per_rank * c->world);
    auto ensure = [&](void *&p, size_t &cap, size_t need) {
        if (need <= cap) return true;
        if (p) { (void)hipStreamSynchronize(st); (void)hipFree(p); p = nullptr; cap = 0; }
        if (hipMalloc(&p, need) != hipSuccess) return false;
        cap = need;
        return true;
    };
    if (slots_per_rank > 0) {
        if (!ensure(c->d_send, c->cap_send, block) || !ensure(c->d_recv, c->cap_recv, block * c->world)) { t_comm_err = "hipMalloc failed"; return PGICP_ERR_HIP; }
        if (hipMemcpyAsync(c->d_send, c->h_send.data(), block, hipMemcpyHostToDevice, st) != hipSuccess) { t_comm_err = "upload of the edge block failed"; return PGICP_ERR_HIP; }
        // ONE collective: world x block bytes, KB-sized, latency-bound on xGMI
        const ncclResult_t r = g_rccl.AllGather(c->d_send, c->d_recv, block, ncclChar, c->comm, st);
        if (r != ncclSuccess) { t_comm_err = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r); return PGICP_ERR_HIP; }
        if (hipMemcpyAsync(c->h_recv.data(), c->d_recv, block * c->world, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { t_comm_err = "download of the gathered edges failed"; return PGICP_ERR_HIP; }
    }
    // pair order; a pair nobody reported stays marked empty
    for (int i = 0; i < n_total; i++) { std::memset(&out[i], 0, sizeof out[i]); out[i].from_id = -1; out[i].to_id = -1; out[i].status = -1; out[i].reserved[0] = -1.0; }
    for (const pgicp_edge &e : c->h_recv) {
        const int i = (int)e.reserved[0];
        if (e.reserved[0] >= 0.0 && i < n_total) out[i] = e;
    }
    return PGICP_OK;
}

int pgicp_shard_slots(int n_pairs, const int64_t *cost, int world_size, int *slots)
{
    if (n_pairs < 0 || world_size <= 0 || !slots) return PGICP_ERR_ARG;
    int most = 0;
    for (int r = 0; r < world_size; r++) {
        int n = 0;
        const int st = pgicp_shard_pairs(n_pairs, cost, world_size, r, nullptr, 0, &n);
        if (st != PGICP_OK) return st;
        if (n > most) most = n;
    }
    *slots = most;
    return PGICP_OK;
}

}  // extern "C"
