// pgicp_comm.cpp -- the one collective of the path: the all-gather of loop-closure edge
// records between the GPUs of a node.  pgicp_allgather_edges = pack -> transport -> unpack:
//   pack      this rank's edges into a fixed block of slots_per_rank 512-byte records (pair index in reserved[0],
//             unused slots marked empty);
//   transport ONE all-gather of those blocks.  Two transports, chosen when the communicator is made:
//             * RCCL (pgicp_comm_create): ncclAllGather over xGMI on the context's stream.  One process per GPU; the
//               communicator is bootstrapped from a unique id the caller passes between the processes by whatever
//               channel it has (a file, MPI, a socket, torch's store).  RCCL is opened at run time (dlopen), the first
//               time a communicator is asked for: a single-GPU user of libpgicp never loads it, and a process that
//               already holds an RCCL (PyTorch's) shares that one instead of mapping a second copy.  Its five entry
//               points are declared here, so the library builds on a ROCm install without the RCCL headers.
//             * host (pgicp_comm_create_host): the blocks travel through a shared-memory file -- the "fake collective"
//               of SURVEY.md section 4 T4 / Appendix B.10.  It needs no device, so the slot / reorder logic of this
//               file is tested at world sizes > 1 on a CPU box, and a single-node job without RCCL can still gather.
//   unpack    the world x slots records into candidate order; pairs nobody reported stay marked empty.
#include "pgicp.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

namespace {

// ---- the part of RCCL's C interface this file uses (rccl.h: ncclUniqueId is 128 opaque bytes, results and data
//      types are plain enums; ncclSuccess = 0, ncclChar = 0) ----
struct RcclUniqueId { char internal[128]; };
typedef struct ncclComm *RcclComm;
typedef int (*fn_GetUniqueId)(RcclUniqueId *);
typedef int (*fn_CommInitRank)(RcclComm *, int, RcclUniqueId, int);
typedef int (*fn_CommDestroy)(RcclComm);
typedef int (*fn_AllGather)(const void *, void *, size_t, int, RcclComm, hipStream_t);
typedef const char *(*fn_GetErrorString)(int);
constexpr int kRcclSuccess = 0, kRcclChar = 0;
static_assert(PGICP_UNIQUE_ID_BYTES >= sizeof(RcclUniqueId), "unique id does not fit");

struct Rccl {
    void *lib = nullptr;
    fn_GetUniqueId GetUniqueId = nullptr;
    fn_CommInitRank CommInitRank = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_AllGather AllGather = nullptr;
    fn_GetErrorString GetErrorString = nullptr;
    std::string err;
    bool load()
    {
        if (lib) return true;
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) { err = std::string("cannot open librccl: ") + dlerror(); return false; }
        GetUniqueId = (fn_GetUniqueId)dlsym(h, "ncclGetUniqueId");
        CommInitRank = (fn_CommInitRank)dlsym(h, "ncclCommInitRank");
        CommDestroy = (fn_CommDestroy)dlsym(h, "ncclCommDestroy");
        AllGather = (fn_AllGather)dlsym(h, "ncclAllGather");
        GetErrorString = (fn_GetErrorString)dlsym(h, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllGather || !GetErrorString) {
            err = "librccl lacks an expected symbol";
            dlclose(h);
            return false;
        }
        lib = h;
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
thread_local std::string t_comm_err;

// ---- host transport: a shared-memory file ----
// [header | world blocks of max_slots records].  Every collective is a generation g = 1, 2, ...: a rank writes its
// block, publishes arrived[rank] = g, waits until every rank has, reads all blocks, publishes left[rank] = g; nobody
// writes generation g + 1 before every rank has left g.  The counters are lock-free 64-bit atomics in the mapping.
struct ShmHeader {
    std::atomic<uint64_t> magic;           // set last by rank 0, before the file gets its name
    uint64_t world, max_slots;
    std::atomic<uint64_t> attached;        // ranks attached to THIS file (the last one to detach removes it)
    // hello / ack: rank r writes a token of its own, the LIVE rank 0 answers with the same token.  A file left behind by a
    // crashed run has the magic and even counters, but nobody who answers: a rank that opened it never takes part in a
    // collective on it (see pgicp_comm_create_host).
    struct alignas(64) Line { std::atomic<uint64_t> arrived, left, hello, ack; } rank[1];     // `world` of them
};
constexpr uint64_t kShmMagic = 0x5047494350434F4DULL;      // "PGICPCOM"
constexpr double kShmTimeoutS = 120.0;

size_t shm_header_bytes(int world) { return (sizeof(ShmHeader) + sizeof(ShmHeader::Line) * (size_t)world + 4095) & ~(size_t)4095; }

template <typename F>
bool spin_until(F ready)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; !ready(); ++k) {
        if (k < 2000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((k & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kShmTimeoutS) return false;
    }
    return true;
}

void mark_empty(pgicp_edge &e)
{
    std::memset(&e, 0, sizeof e);
    e.from_id = -1; e.to_id = -1; e.status = -1;
}

}  // namespace

struct pgicp_comm {
    int world = 1, rank = 0;
    // RCCL transport
    RcclComm comm = nullptr;
    pgicp_ctx *ctx = nullptr;
    int device = 0;
    void *d_send = nullptr, *d_recv = nullptr;
    size_t cap_send = 0, cap_recv = 0;
    // staging of both transports: pinned for RCCL (async copies from pageable memory are staged by the runtime, on a
    // latency-bound path), plain for the host transport
    pgicp_edge *h_send = nullptr, *h_recv = nullptr;
    size_t cap_hsend = 0, cap_hrecv = 0;       // records
    // host transport
    bool host = false;
    char *shm = nullptr;
    size_t shm_bytes = 0;
    uint64_t generation = 0;
    int max_slots = 0;
    std::string shm_path;
};

namespace {

bool stage_ensure(pgicp_comm *c, pgicp_edge *&p, size_t &cap, size_t need)
{
    if (need <= cap) return true;
    if (p) { if (c->host) std::free(p); else (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t n = need + need / 4 + 16;
    if (c->host) p = (pgicp_edge *)std::malloc(sizeof(pgicp_edge) * n);
    else if (hipHostMalloc((void **)&p, sizeof(pgicp_edge) * n, hipHostMallocDefault) != hipSuccess) p = nullptr;
    if (!p) return false;
    cap = n;
    return true;
}

// pack: this rank's edges first, then empty slots; the pair index rides in reserved[0] (-1 = empty slot)
void pack_block(const pgicp_edge *local, const int *pair_index, int n_local, int slots, pgicp_edge *block)
{
    for (int k = 0; k < slots; k++) {
        pgicp_edge &e = block[k];
        if (k < n_local) { e = local[k]; e.reserved[0] = (double)pair_index[k]; }
        else { mark_empty(e); e.reserved[0] = -1.0; }
    }
}

// unpack: candidate order; a pair nobody reported stays marked empty; reserved[0] is the caller's again (0)
void unpack_blocks(const pgicp_edge *all, size_t n_records, int n_total, pgicp_edge *out)
{
    for (int i = 0; i < n_total; i++) mark_empty(out[i]);
    for (size_t k = 0; k < n_records; k++) {
        const pgicp_edge &e = all[k];
        const int i = (int)e.reserved[0];
        if (e.reserved[0] >= 0.0 && i < n_total) { out[i] = e; out[i].reserved[0] = 0.0; }
    }
}

int transport_rccl(pgicp_comm *c, int slots)
{
    if (hipSetDevice(c->device) != hipSuccess) { t_comm_err = "hipSetDevice failed"; return PGICP_ERR_HIP; }
    hipStream_t st = (hipStream_t)pgicp_ctx_stream(c->ctx);
    const size_t block = sizeof(pgicp_edge) * (size_t)slots;
    auto ensure = [&](void *&p, size_t &cap, size_t need) {
        if (need <= cap) return true;
        if (p) { (void)hipStreamSynchronize(st); (void)hipFree(p); p = nullptr; cap = 0; }
        if (hipMalloc(&p, need) != hipSuccess) return false;
        cap = need;
        return true;
    };
    if (!ensure(c->d_send, c->cap_send, block) || !ensure(c->d_recv, c->cap_recv, block * c->world)) { t_comm_err = "hipMalloc failed"; return PGICP_ERR_HIP; }
    if (hipMemcpyAsync(c->d_send, c->h_send, block, hipMemcpyHostToDevice, st) != hipSuccess) { t_comm_err = "upload of the edge block failed"; return PGICP_ERR_HIP; }
    // ONE collective: world x block bytes, KB-sized, latency-bound on xGMI
    const int r = g_rccl.AllGather(c->d_send, c->d_recv, block, kRcclChar, c->comm, st);
    if (r != kRcclSuccess) { t_comm_err = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r); return PGICP_ERR_HIP; }
    if (hipMemcpyAsync(c->h_recv, c->d_recv, block * c->world, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { t_comm_err = "download of the gathered edges failed"; return PGICP_ERR_HIP; }
    return PGICP_OK;
}

int transport_host(pgicp_comm *c, int slots)
{
    if (slots > c->max_slots) { t_comm_err = "pgicp_allgather_edges: slots_per_rank exceeds the host communicator's max_slots_per_rank"; return PGICP_ERR_ARG; }
    ShmHeader *H = (ShmHeader *)c->shm;
    pgicp_edge *area = (pgicp_edge *)(c->shm + shm_header_bytes(c->world));
    const uint64_t g = ++c->generation;
    // (every rank left generation g - 1 before anyone reaches this point with g: see the wait at the end)
    std::memcpy(area + (size_t)c->rank * c->max_slots, c->h_send, sizeof(pgicp_edge) * (size_t)slots);
    H->rank[c->rank].arrived.store(g, std::memory_order_release);
    for (int r = 0; r < c->world; r++)
        if (!spin_until([&] { return H->rank[r].arrived.load(std::memory_order_acquire) >= g; })) { t_comm_err = "host all-gather: rank " + std::to_string(r) + " did not arrive"; return PGICP_ERR_HIP; }
    for (int r = 0; r < c->world; r++)
        std::memcpy(c->h_recv + (size_t)r * slots, area + (size_t)r * c->max_slots, sizeof(pgicp_edge) * (size_t)slots);
    H->rank[c->rank].left.store(g, std::memory_order_release);
    for (int r = 0; r < c->world; r++)
        if (!spin_until([&] { return H->rank[r].left.load(std::memory_order_acquire) >= g; })) { t_comm_err = "host all-gather: rank " + std::to_string(r) + " did not finish"; return PGICP_ERR_HIP; }
    return PGICP_OK;
}

}  // namespace

extern "C" {

const char *pgicp_comm_last_error(void) { return t_comm_err.c_str(); }

int pgicp_comm_unique_id(char id[PGICP_UNIQUE_ID_BYTES])
{
    if (!id) return PGICP_ERR_ARG;
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl.load()) { t_comm_err = g_rccl.err; return PGICP_ERR_HIP; }
    RcclUniqueId u;
    const int r = g_rccl.GetUniqueId(&u);
    if (r != kRcclSuccess) { t_comm_err = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return PGICP_ERR_HIP; }
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
    RcclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    pgicp_comm *c = new pgicp_comm();
    c->ctx = ctx; c->world = world_size; c->rank = rank; c->device = device;
    const int r = g_rccl.CommInitRank(&c->comm, world_size, u, rank);
    if (r != kRcclSuccess) { t_comm_err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); delete c; return PGICP_ERR_HIP; }
    *out = c;
    return PGICP_OK;
}

int pgicp_comm_create_host(int world_size, int rank, const char *shm_path, int max_slots_per_rank, pgicp_comm **out)
{
    if (!out || !shm_path || !*shm_path || world_size < 1 || rank < 0 || rank >= world_size || max_slots_per_rank < 0) {
        t_comm_err = "pgicp_comm_create_host: bad argument";
        return PGICP_ERR_ARG;
    }
    *out = nullptr;
    const size_t bytes = shm_header_bytes(world_size) + sizeof(pgicp_edge) * (size_t)world_size * (size_t)std::max(1, max_slots_per_rank);
    // Rendezvous (collective: returns on every rank once all ranks are attached, like ncclCommInitRank).
    // Rank 0 makes a FRESH inode -- a stale file of that name (a crashed or timed-out run never removes its own) is
    // unlinked, the new one is written under a temporary name, its header completed, and only then renamed into place: a
    // file that carries the name is never truncated or half initialised under someone's mapping.  The other ranks say
    // hello with a token of their own and take part only once the live rank 0 has answered it; a rank that mapped a
    // stale inode gets no answer, notices that the name now belongs to another inode, and starts over with that one.
    const int slots_cap = std::max(1, max_slots_per_rank);
    auto map_fd = [&](int fd) -> void * {
        void *q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        return q == MAP_FAILED ? nullptr : q;
    };
    void *p = nullptr;
    if (rank == 0) {
        (void)unlink(shm_path);
        const std::string tmp = std::string(shm_path) + ".init." + std::to_string((long long)getpid());
        (void)unlink(tmp.c_str());
        const int fd = open(tmp.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { t_comm_err = std::string("pgicp_comm_create_host: cannot create ") + tmp; if (fd >= 0) { close(fd); (void)unlink(tmp.c_str()); } return PGICP_ERR_ARG; }
        p = map_fd(fd);
        close(fd);
        if (!p) { t_comm_err = "pgicp_comm_create_host: mmap failed"; (void)unlink(tmp.c_str()); return PGICP_ERR_ARG; }
        ShmHeader *H = (ShmHeader *)p;             // (a fresh file is zero-filled: all counters start at 0)
        H->world = (uint64_t)world_size; H->max_slots = (uint64_t)slots_cap;
        H->attached.store(1, std::memory_order_relaxed);
        H->magic.store(kShmMagic, std::memory_order_release);
        if (rename(tmp.c_str(), shm_path) != 0) { t_comm_err = std::string("pgicp_comm_create_host: cannot name ") + shm_path; munmap(p, bytes); (void)unlink(tmp.c_str()); return PGICP_ERR_ARG; }
        // answer every rank's hello
        for (int r = 1; r < world_size; r++) {
            uint64_t tok = 0;
            if (!spin_until([&] { tok = H->rank[r].hello.load(std::memory_order_acquire); return tok != 0; })) {
                t_comm_err = "pgicp_comm_create_host: rank " + std::to_string(r) + " never attached";
                munmap(p, bytes); (void)unlink(shm_path);
                return PGICP_ERR_ARG;
            }
            H->rank[r].ack.store(tok, std::memory_order_release);
        }
        if (!spin_until([&] { return H->attached.load(std::memory_order_acquire) >= (uint64_t)world_size; })) {
            t_comm_err = "pgicp_comm_create_host: not every rank attached";
            munmap(p, bytes); (void)unlink(shm_path);
            return PGICP_ERR_ARG;
        }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        const uint64_t token = (((uint64_t)getpid() << 32) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((uint64_t)rank << 56)) | 1u;
        bool shape_error = false;
        for (;;) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kShmTimeoutS) break;
            const int fd = open(shm_path, O_RDWR);
            struct stat sb;
            if (fd < 0 || fstat(fd, &sb) != 0 || (size_t)sb.st_size < bytes) {       // not there yet (or a stale file of another shape)
                if (fd >= 0) close(fd);
                std::this_thread::sleep_for(std::chrono::microseconds(200));
                continue;
            }
            void *q = map_fd(fd);
            close(fd);
            if (!q) { t_comm_err = "pgicp_comm_create_host: mmap failed"; return PGICP_ERR_ARG; }
            ShmHeader *H = (ShmHeader *)q;
            bool answered = false, replaced = false;
            if (H->magic.load(std::memory_order_acquire) == kShmMagic) {
                if (H->world != (uint64_t)world_size || H->max_slots != (uint64_t)slots_cap) shape_error = true;
                else {
                    shape_error = false;
                    H->rank[rank].hello.store(token, std::memory_order_release);
                    // wait for the answer; meanwhile watch whether the NAME still refers to this inode
                    for (int k = 0; !answered && !replaced; ++k) {
                        answered = H->rank[rank].ack.load(std::memory_order_acquire) == token;
                        if (answered) break;
                        if (k < 2000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(50));
                        if ((k & 255) == 255) {
                            struct stat now;
                            replaced = stat(shm_path, &now) != 0 || now.st_ino != sb.st_ino || now.st_dev != sb.st_dev;
                            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kShmTimeoutS) break;
                        }
                    }
                }
            }
            if (answered) { p = q; break; }
            munmap(q, bytes);
            if (!replaced) std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        if (!p) {
            t_comm_err = shape_error ? "pgicp_comm_create_host: the shared file belongs to a communicator of another shape"
                                     : std::string("pgicp_comm_create_host: no live rank 0 answered on ") + shm_path;
            return PGICP_ERR_ARG;
        }
        ((ShmHeader *)p)->attached.fetch_add(1, std::memory_order_acq_rel);
    }
    pgicp_comm *c = new pgicp_comm();
    c->host = true; c->world = world_size; c->rank = rank; c->shm = (char *)p; c->shm_bytes = bytes;
    c->max_slots = slots_cap; c->shm_path = shm_path;
    *out = c;
    return PGICP_OK;
}

void pgicp_comm_destroy(pgicp_comm *c)
{
    if (!c) return;
    if (c->host) {
        if (c->shm) {
            // the last rank to detach removes the file
            ShmHeader *H = (ShmHeader *)c->shm;
            const bool last = H->attached.fetch_sub(1, std::memory_order_acq_rel) == 1;
            munmap(c->shm, c->shm_bytes);
            if (last) (void)unlink(c->shm_path.c_str());
        }
        std::free(c->h_send); std::free(c->h_recv);
    } else {
        (void)hipSetDevice(c->device);
        if (c->d_send) (void)hipFree(c->d_send);
        if (c->d_recv) (void)hipFree(c->d_recv);
        if (c->h_send) (void)hipHostFree(c->h_send);
        if (c->h_recv) (void)hipHostFree(c->h_recv);
        if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    }
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
    const size_t n_all = (size_t)slots_per_rank * (size_t)c->world;
    if (slots_per_rank > 0) {
        if (!stage_ensure(c, c->h_send, c->cap_hsend, (size_t)slots_per_rank) || !stage_ensure(c, c->h_recv, c->cap_hrecv, n_all)) {
            t_comm_err = "pgicp_allgather_edges: cannot allocate the staging blocks";
            return PGICP_ERR_HIP;
        }
        pack_block(local, pair_index, n_local, slots_per_rank, c->h_send);
        const int st = c->host ? transport_host(c, slots_per_rank) : transport_rccl(c, slots_per_rank);
        if (st != PGICP_OK) return st;
    }
    unpack_blocks(c->h_recv, slots_per_rank > 0 ? n_all : 0, n_total, out);
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
