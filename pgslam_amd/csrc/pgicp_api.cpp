// pgicp_api.cpp -- host side of libpgicp: the extern "C" ABI of include/pgicp.h.
//
// Owns device memory (maps stay resident in HBM), sequences the kernels of
// kernels.hip on the context's HIP stream, and finishes the few host-side
// steps (un-centring the result, inverting the 6x6 covariance Hessian).
// There is no CPU compute path here: every stage runs on the GPU.
#include "pgicp.h"
#include "kernels.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <thread>
#include <limits>
#include <numeric>
#include <string>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

using namespace pgicp;

namespace {

// ---- allocation accounting (pgicp_debug_alloc_stats): hipMalloc / hipFree / hipHostMalloc are the calls of this library whose
// cost depends on the HOST (1 ms on some boxes, 45 ms on others; hipFree waits for the whole device) -- counted and timed, so
// that a slow pass of a caller can be told apart from a slow GPU
struct AllocStats { std::atomic<long long> n[4], ns[4]; };
AllocStats g_alloc;
struct AllocTimer {
    int k; std::chrono::steady_clock::time_point t0;
    explicit AllocTimer(int kind) : k(kind), t0(std::chrono::steady_clock::now()) {}
    ~AllocTimer() { g_alloc.n[k]++; g_alloc.ns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
inline hipError_t t_malloc(void **p, size_t bytes) { AllocTimer t(0); return hipMalloc(p, bytes); }
inline hipError_t t_free(void *p) { AllocTimer t(1); return hipFree(p); }
inline hipError_t t_host_malloc(void **p, size_t bytes, unsigned flags) { AllocTimer t(2); return hipHostMalloc(p, bytes, flags); }
inline hipError_t t_host_free(void *p) { AllocTimer t(3); return hipHostFree(p); }

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = t_free(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = t_malloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)t_free(p); p = nullptr; cap = 0; }
    template <typename U> U *as() const { return (U *)p; }
};

// the one device allocation of a (batch of) map(s); freed here only when nobody released it properly (an error path)
struct SharedBlock {
    char *p = nullptr;
    size_t bytes = 0;
    ~SharedBlock() { if (p) (void)t_free(p); }
};

template <typename T>
struct MapHost {
    bool used = false;
    int m = 0;
    T mean[3] = {0, 0, 0};
    bool has_nrm = false;
    typename Vec4<T>::type *pts = nullptr;
    typename Vec4<T>::type *nrm = nullptr;
    int *cell_start = nullptr;
    int *cell_start_f = nullptr;
    uint4 *sw = nullptr;            // MapDev::sw / ostart: the succinct table (then cell_start / cell_start_f are null)
    int *ostart = nullptr;
    int kx = 1;
    int *slot_of = nullptr;
    bool slot_of_made = false;      // (filled on first use: error_stats)
    int *sc_count = nullptr;
    int *near = nullptr;
    int *sc_dist = nullptr;
    int *sc_wit = nullptr;
    float *sc_ext = nullptr;
    unsigned *occ = nullptr;
    float *ptsf = nullptr;          // MapDev::ptsf (double maps)
    int first = 0;                  // MapDev::first
    // the one device allocation holding all of the above -- shared by the maps of one batched build and
    // returned to the pool (or freed) by whoever drops the last reference
    std::shared_ptr<SharedBlock> block;
    GridDesc<T> g{};
};

template <typename T>
struct State {
    std::vector<MapHost<T>> maps;
    DevBuf d_maps;                  // MapDev<T>[capacity]
    int d_maps_cap = 0;
    DevBuf rd_pre, rd_sorted, slot, d2, none_r, staging, stage_aux, nrm_pre, nrm_sorted;
};

struct ProfEvent {
    hipEvent_t a, b;
    int kid;
    int kid2 = -1;          // a second account the launch also goes to (-1: none)
    long long units;
    long long problems;
    long long map_points;
};

}  // namespace

struct pgicp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    pgicp_params prm{};
    State<float> f32;
    State<double> f64;
    DevBuf probs, src, partials, sums, sums2, small, stats, bdesc, tmp_a, tmp_b, tmp_c, tmp_d, tmp_e, tmp_f, tmp_r, tmp_w, tmp_p, tmp_n;
    DevBuf scan_pos;                // PGICP_SUM_ORDER_SCAN: the inverse of `order` (sorted position of every reading point)
    DevBuf qrow, qtmp, order, qcounts, qblock, qstart, qcursor, slow_list, slow_lb, slow_ring, slow2, active, sel_tables, queue;
    // host-input pipeline (pgicp_upload_*): a copy stream and two upload sets used alternately
    struct UploadSet {
        DevBuf dev;                     // device copies of the readings of one upload
        void *pin = nullptr;            // pinned staging for pageable sources
        size_t pin_cap = 0;
        size_t bytes = 0;               // bytes of `dev` in use
        hipEvent_t uploaded = nullptr;  // recorded on the copy stream after the last transfer of the upload
        hipEvent_t consumed = nullptr;  // recorded on the compute stream after the last kernel that reads the set
        bool pending = false, has_consumer = false;
    } up[2];
    hipStream_t copy_stream = nullptr;
    // pgicp_filter_cloud: four sets of device buffers used in turn (the filtered features of a call stay valid, as a device
    // reading, for the next three calls: a localizer that pre-processes scan k + 1 while scan k aligns, and now and then a
    // scan that was not pre-processed ahead, has three calls between making a reading and aligning it)
    struct FilterSet { DevBuf in_f, in_d, keep, pos, bsum, out_f, out_d, idx, drop; } fset[4];
    DevBuf robust_dev;              // RobustOutlierFilter: the pairs' absolute deviations from the median (the second selection's input)
    int fset_next = 0;
    int up_next = 0;
    int up_seen = 0;            // upload sets whose device pointers the running call was handed (see UploadUse)
    int *h_pinned = nullptr;        // pinned scratch for small D2H polls (64 ints)
    // pinned bounce buffers for the problem records of a batch (4.6 KB each: a Checker history rides in them): a megabyte
    // copied between device and PAGEABLE host memory takes the runtime's slow path -- 10-13 ms per step at 256-320
    // problems, none at 128 or 384 -- so the records travel through pinned memory in both directions
    char *h_up = nullptr, *h_down = nullptr;
    size_t h_up_cap = 0, h_down_cap = 0;
    // Bounce buffer (pinned): host memory the CALLER owns never reaches hipMemcpy*.  The runtime pins a pageable range of more
    // than ~128 KB on the fly (a userptr buffer object); when the caller later frees that memory (munmap, or a trimmed heap)
    // the MMU notifier makes the kernel driver EVICT every queue of the process for 25-40 ms.  Measured in round 5 (NOTES_r05,
    // profiles/r05_stall_diagnosis.txt): 10-100 such pauses in the first second of a process -- until glibc's dynamic mmap
    // threshold stops handing out mmap()ed chunks of a cloud's size -- made slam_100k's ICP phase 0.6 s on one box and 3.0 s on
    // another.  h2d() / d2h() below: memcpy through this buffer; copy-outs are completed by stream_sync().
    struct Bounce {
        char *p = nullptr;
        size_t cap = 0, off = 0;
        struct Out { void *dst; const char *src; size_t bytes; };
        std::vector<Out> outs;
        bool direct_out_pending = false;    // a small device -> host copy went straight to a caller's / a local buffer (see fail())
    } bounce;
    int *h_flag = nullptr;          // coherent pinned pair {problems done, stamp} the last kernel of an iteration writes
    int flag_stamp = 0;             // iterations enqueued so far == the value the last one's k_compact_active will store
    int *stamp_dev = nullptr;       // the same count on the device (k_compact_active advances it)
    // captured iterations (small batches): the launch sequence of one ICP iteration as an executable graph, keyed by
    // everything its kernel arguments depend on
    struct IterGraph { unsigned long long key = 0; hipGraphExec_t exec = nullptr; unsigned long long used = 0; };
    std::vector<IterGraph> iter_graphs;
    unsigned long long graph_clock = 0;
    int graph_max_problems = 0;     // batches of at most this many problems replay captured iterations (0: never -- the default:
                                    // measured on ROCm 7.2 / MI355X, replaying costs what the ten launches cost; PGICP_GRAPH_MAX_P)
    int graph_failures = 0;
    long long graph_captures = 0, graph_launches = 0;
    std::vector<SrcDesc> h_src;     // host copies of per-batch descriptors (uploaded asynchronously)
    std::vector<int> h_ident;
    int counters_clean = 0;         // the matcher's queue counters were zeroed by the last kernel of the previous iteration
    int seg_clean = 0;              // the matcher's SEGMENTED queue counters were zeroed by the previous iteration's first selection
    int poll_us = 400;              // how long the host polls h_flag before it blocks on the stream instead
    // selection hints (ProblemDev::qhint): what the previous call of the same kind found, per problem index -- [0] pgicp_align*,
    // [1] pgicp_partial_chain*; dropped when the chain changes.  sel_guess_first: the enqueued iteration's FIRST selection may take
    // the band path although the matcher is unseeded (every problem carries a hint for it).
    struct SelHint { double q[2][4]; };
    std::vector<SelHint> sel_hints[2];
    int sel_guess_first = 0;
    int sel_hints_on = 1;           // PGICP_SEL_HINTS=0 turns them off (A/B)
    int fast_rings_seeded = 0, fast_rings_unseeded = 0;      // 0: by the maps' cell size (BatchLayout::rings_*); PGICP_FAST_RINGS_* set them
    double cell_scale = 1.0;     // experiment knob PGICP_CELL_SCALE: the automatic grid cell times this
    int grid_kx = 4;                // x refinement of the table the matcher narrows ranges with (MapDev::cell_start_f)
    int table_mode = 0;             // the cell tables a map is built with: 0 / 1 dense (the default), 2 succinct (MapDev::sw);
                                    // PGICP_TABLES=dense|succinct, read when the context is made
    double near_frac = 0.2;         // MapDev::near looks this fraction of maxDist far (at most kNearReach cells)
    int bin_shift_add = 0;          // experiment knob PGICP_BIN_SHIFT_ADD: coarser (+) or finer (-) bins of the reading sort
    int med_rings = 4;              // rings a queued query may walk per lane before the wave-cooperative path takes it   // rings walked in the fast kernel before a query is queued
    bool prof_on = false;
    std::mutex prof_m;              // prof_events and the sums below: pgicp_profile_process collects from another thread
    std::vector<ProfEvent> prof_events;
    long long prof_launches[PGICP_PROF_COUNT] = {0};
    double prof_ms[PGICP_PROF_COUNT] = {0};
    long long prof_units[PGICP_PROF_COUNT] = {0};
    long long prof_problems[PGICP_PROF_COUNT] = {0};
    long long prof_map_points[PGICP_PROF_COUNT] = {0};      // reference points of the active problems of those launches
    long long prof_next_m = 0;                               // what the next matcher launch is charged with
};

namespace {

template <typename T> State<T> &state(pgicp_ctx *c);
template <> State<float> &state<float>(pgicp_ctx *c) { return c->f32; }
template <> State<double> &state<double>(pgicp_ctx *c) { return c->f64; }

// public map ids carry the precision in bit 30 so f32 and f64 maps never alias
template <typename T> constexpr int id_tag();
template <> constexpr int id_tag<float>() { return 0; }
template <> constexpr int id_tag<double>() { return 0x40000000; }

template <typename T>
int map_index(pgicp_ctx *c, int id)
{
    State<T> &S = state<T>(c);
    if (id < 0 || (id & 0x40000000) != id_tag<T>()) return -1;
    const int idx = id & 0x3FFFFFFF;
    if (idx >= (int)S.maps.size() || !S.maps[idx].used) return -1;
    return idx;
}

template <typename T>
MapHost<T> *get_map(pgicp_ctx *c, int id)
{
    const int idx = map_index<T>(c, id);
    return idx < 0 ? nullptr : &state<T>(c).maps[idx];
}

// Every failing exit of every entry point passes through here (directly, or through HIPC / XFER).  A device -> host copy may
// still be queued at that point: a bounce copy-out whose destination is a local of the failing call or a caller's buffer
// (pgicp_ctx::Bounce::outs), or a small direct hipMemcpyAsync into one.  Let the stream run dry so that nothing writes into
// memory that goes out of scope with the return, and DROP the queued copy-outs -- the call has failed, its outputs are void --
// so that no later stream_sync() (the next call's, bounce_take's wrap-around, pgicp_ctx_destroy) replays them.
int fail(pgicp_ctx *c, int code, const std::string &msg)
{
    if (c) {
        c->err = msg;
        if (!c->bounce.outs.empty() || c->bounce.direct_out_pending) {
            if (c->stream) (void)hipStreamSynchronize(c->stream);
            c->bounce.outs.clear();
            c->bounce.off = 0;
            c->bounce.direct_out_pending = false;
        }
    }
    return code;
}

#define HIPC(ctx, call)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(ctx, PGICP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));          \
    } while (0)

// process-wide view of the profile (pgicp_profile_process): totals of the contexts already destroyed plus, at the time
// of the call, of the live ones -- for callers that do not hold the contexts themselves (the C++ facade's ICP objects)
struct ProcessProfile {
    std::mutex m;
    std::vector<pgicp_ctx *> live;
    long long launches[PGICP_PROF_COUNT] = {0}, units[PGICP_PROF_COUNT] = {0}, problems[PGICP_PROF_COUNT] = {0}, map_points[PGICP_PROF_COUNT] = {0};
    double ms[PGICP_PROF_COUNT] = {0};
};
ProcessProfile &process_profile() { static ProcessProfile *p = new ProcessProfile(); return *p; }

struct ProfScope {
    pgicp_ctx *c;
    int kid;
    ProfEvent ev{};
    bool on;
    ProfScope(pgicp_ctx *c_, int kid_, long long units, long long problems = 1) : c(c_), kid(kid_), on(c_->prof_on)
    {
        if (!on) return;
        ev.kid = kid;
        ev.units = units;
        ev.problems = problems;
        ev.map_points = (kid_ == PGICP_PROF_KNN_GRID || kid_ == PGICP_PROF_KNN_BRUTE) ? c_->prof_next_m : 0;
        if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(ev.a, c->stream);
    }
    void also(int kid2) { ev.kid2 = kid2; }
    ~ProfScope();
};

// (with c->prof_m held) waits for the stream and folds the finished event pairs into the sums
void prof_collect_locked(pgicp_ctx *c)
{
    if (c->prof_events.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    // diagnostics (PGICP_STALL_LOG=<ms>): scopes and gaps between consecutive scopes of this context longer than that
    static const double stall_ms = std::getenv("PGICP_STALL_LOG") ? std::atof(std::getenv("PGICP_STALL_LOG")) : 0.0;
    const ProfEvent *prev = nullptr;
    for (auto &e : c->prof_events) {
        float ms = 0.f;
        if (stall_ms > 0.0 && prev) {
            float gap = 0.f;
            if (hipEventElapsedTime(&gap, prev->b, e.a) == hipSuccess && gap > stall_ms)
                std::fprintf(stderr, "pgicp stall: %.2f ms between scope %d and scope %d (context %p)\n", gap, prev->kid, e.kid, (void *)c);
        }
        prev = &e;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            if (stall_ms > 0.0 && ms > stall_ms) std::fprintf(stderr, "pgicp stall: scope %d took %.2f ms (context %p)\n", e.kid, ms, (void *)c);
            for (int kid : {e.kid, e.kid2}) {
                if (kid < 0) continue;
                c->prof_launches[kid] += 1;
                c->prof_ms[kid] += ms;
                c->prof_units[kid] += e.units;
                c->prof_problems[kid] += e.problems;
                c->prof_map_points[kid] += e.map_points;
            }
        }
    }
    for (auto &e : c->prof_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    c->prof_events.clear();
}
void prof_collect(pgicp_ctx *c)
{
    std::lock_guard<std::mutex> lock(c->prof_m);
    prof_collect_locked(c);
}
ProfScope::~ProfScope()
{
    if (!on) return;
    (void)hipEventRecord(ev.b, c->stream);
    std::lock_guard<std::mutex> lock(c->prof_m);
    c->prof_events.push_back(ev);
    // a context that profiles from its creation (PGICP_PROFILE_ALL: the facade's ICP objects) is never asked for its
    // numbers between calls: fold the events in now and then, or a long run holds hundreds of thousands of live events
    if (c->prof_events.size() >= 4096) prof_collect_locked(c);
}

static inline bool is_p2point(int m) { return m == PGICP_MINIMIZER_POINT_TO_POINT || m == PGICP_MINIMIZER_POINT_TO_POINT_WITH_COV; }
// the chain reads the reference's normals: every minimiser but the plain point-to-point one (its WithCov form estimates the
// covariance from them), and a SurfaceNormalOutlierFilter
static inline bool needs_ref_normals(const pgicp_params &p) { return p.error_minimizer != PGICP_MINIMIZER_POINT_TO_POINT || p.normal_max_angle > 0.0; }

template <typename T>
ChainDev<T> make_chain(const pgicp_params &p)
{
    ChainDev<T> ch;
    ch.max_dist = (T)p.max_dist;
    ch.max_dist2 = ch.max_dist * ch.max_dist;
    ch.trim_ratio = (T)p.trim_ratio;
    ch.trim_scale = (T)p.quantile_scale;
    {
        const T md = (T)p.outlier_max_dist;
        ch.outlier_max_d2 = (p.outlier_max_dist > 0.0 && std::isfinite(p.outlier_max_dist)) ? md * md : std::numeric_limits<T>::infinity();
    }
    ch.max_iters = p.max_iters;
    ch.smooth = p.smooth_length;
    ch.min_rot = p.min_diff_rot;
    ch.min_trans = p.min_diff_trans;
    ch.rank_rel_tol = 6.0 * (double)std::numeric_limits<T>::epsilon();
    ch.knn = std::max(1, p.knn);
    ch.minimizer = is_p2point(p.error_minimizer) ? 1 : 0;
    ch.robust.fct = p.robust_fct; ch.robust.mad = p.robust_scale == PGICP_ROBUST_SCALE_MAD ? 1 : 0;
    ch.robust.k = (T)p.robust_tuning; ch.robust.k2 = ch.robust.k * ch.robust.k;
    ch.robust.cut = (p.robust_approx > 0.0 && std::isfinite(p.robust_approx)) ? 1 : 0;
    { const T a = (T)p.robust_approx; ch.robust.a2 = a * a; }
    ch.force4dof = p.error_minimizer == PGICP_MINIMIZER_POINT_TO_PLANE_4DOF ? 1 : 0;
    ch.bound_rot = (p.bound_max_rot > 0.0 && std::isfinite(p.bound_max_rot)) ? p.bound_max_rot : 0.0;
    ch.bound_trans = (p.bound_max_trans > 0.0 && std::isfinite(p.bound_max_trans)) ? p.bound_max_trans : 0.0;
    ch.use_normals = p.normal_max_angle > 0.0 ? 1 : 0;
    ch.normal_cos = std::cos((T)p.normal_max_angle);          // in T, as the filter evaluates `cos(maxAngle)`
    ch.scan_pos = nullptr;                                     // (set by chain_of: it names a buffer of the context)
    return ch;
}

// the chain of a call that has sorted its readings (begin_batch): with PGICP_SUM_ORDER_SCAN the reduce kernels are handed the
// inverse of that sort (pgicp_ctx::scan_pos, filled by begin_batch)
template <typename T>
ChainDev<T> chain_of(pgicp_ctx *c, const pgicp_params &p)
{
    ChainDev<T> ch = make_chain<T>(p);
    ch.scan_pos = p.sum_order == PGICP_SUM_ORDER_SCAN ? c->scan_pos.as<int>() : nullptr;
    return ch;
}

double key_to_double(unsigned long long k)
{
    long long i = (k & 0x8000000000000000ULL) ? (long long)(k ^ 0x8000000000000000ULL) : (long long)~k;
    double d;
    std::memcpy(&d, &i, sizeof d);
    return d;
}

// A device pointer handed out by pgicp_upload_*: the context stream waits (on the device) for that upload.
// Returns the bit mask of upload sets the pointer belongs to.
int upload_wait(pgicp_ctx *c, const void *p)
{
    int mask = 0;
    for (int s = 0; s < 2; s++) {
        pgicp_ctx::UploadSet &U = c->up[s];
        if (!U.pending || !U.dev.p) continue;
        const char *q = (const char *)p;
        if (q >= (const char *)U.dev.p && q < (const char *)U.dev.p + U.bytes) {
            (void)hipStreamWaitEvent(c->stream, U.uploaded, 0);
            mask |= 1 << s;
        }
    }
    return mask;
}
// ... and after the last kernel that reads them has been queued, the sets may be overwritten once it has run
void upload_consumed(pgicp_ctx *c, int mask)
{
    for (int s = 0; s < 2; s++)
        if (mask & (1 << s)) { (void)hipEventRecord(c->up[s].consumed, c->stream); c->up[s].has_consumer = true; }
}

// ---- transfers between the device and host memory the caller owns (see pgicp_ctx::Bounce) ----
constexpr size_t kDirectCopyBytes = 64 << 10;        // below this the runtime stages the copy itself (it pins from ~128 KB on)
static void bounce_drain(pgicp_ctx *c)               // the stream is idle: finish the copy-outs, the whole buffer is free again
{
    for (auto &o : c->bounce.outs) std::memcpy(o.dst, o.src, o.bytes);
    c->bounce.outs.clear();
    c->bounce.off = 0;
    c->bounce.direct_out_pending = false;
}
static hipError_t stream_sync(pgicp_ctx *c)
{
    const hipError_t e = hipStreamSynchronize(c->stream);
    bounce_drain(c);
    return e;
}
static int bounce_take(pgicp_ctx *c, size_t bytes, char **out)
{
    pgicp_ctx::Bounce &B = c->bounce;
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (B.off + need > B.cap) {
        HIPC(c, stream_sync(c));                     // wrap around: what is in flight lands first
        if (need > B.cap) {
            if (B.p) (void)t_host_free(B.p);
            B.p = nullptr; B.cap = 0;
            const size_t want = need + need / 4 + (1u << 20);
            HIPC(c, t_host_malloc((void **)&B.p, want, hipHostMallocDefault));
            B.cap = want;
        }
    }
    *out = B.p + B.off;
    B.off += need;
    return PGICP_OK;
}
// host -> device on the context's stream; `src` may be reused as soon as the call returns
static int h2d(pgicp_ctx *c, void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes == 0) return PGICP_OK;
    if (bytes < kDirectCopyBytes) { HIPC(c, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream)); return PGICP_OK; }
    char *b = nullptr;
    { const int st = bounce_take(c, bytes, &b); if (st) return st; }
    std::memcpy(b, src_host, bytes);
    HIPC(c, hipMemcpyAsync(dst_dev, b, bytes, hipMemcpyHostToDevice, c->stream));
    return PGICP_OK;
}
// device -> host on the context's stream; `dst` holds the data after the next stream_sync(c)
static int d2h(pgicp_ctx *c, void *dst_host, const void *src_dev, size_t bytes)
{
    if (bytes == 0) return PGICP_OK;
    if (bytes < kDirectCopyBytes) {
        c->bounce.direct_out_pending = true;
        HIPC(c, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        return PGICP_OK;
    }
    char *b = nullptr;
    { const int st = bounce_take(c, bytes, &b); if (st) return st; }
    HIPC(c, hipMemcpyAsync(b, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    c->bounce.outs.push_back({dst_host, b, bytes});
    return PGICP_OK;
}
#define XFER(ctx, call) do { const int st_ = (call); if (st_) return st_; } while (0)

// Every entry point that may be handed a device pointer of pgicp_upload_* declares one of these: to_device() makes the
// compute stream wait for the upload (on the device) and notes the set; when the call returns -- its kernels queued, most
// calls synchronised -- the set is marked consumed, so that the next upload into it waits for them.
struct UploadUse {
    pgicp_ctx *c;
    explicit UploadUse(pgicp_ctx *c_) : c(c_) { if (c) c->up_seen = 0; }
    ~UploadUse() { if (c && c->up_seen) { upload_consumed(c, c->up_seen); c->up_seen = 0; } }
    void touch(const void *p) { if (c && p && (c->up[0].pending || c->up[1].pending)) c->up_seen |= upload_wait(c, p); }
};

// Bring a strided point buffer onto the device if it is host memory.
// Returns the device pointer to use (either the caller's or the staging copy).
template <typename T>
int to_device(pgicp_ctx *c, const T *p, int stride, int n, int mem, DevBuf &stage, size_t stage_off_bytes, const T **out)
{
    if (mem == PGICP_DEVICE) {
        if (c->up[0].pending || c->up[1].pending) c->up_seen |= upload_wait(c, p);
        *out = p;
        return PGICP_OK;
    }
    const size_t bytes = sizeof(T) * ((size_t)(n - 1) * stride + 3);
    XFER(c, h2d(c, (char *)stage.p + stage_off_bytes, p, bytes));
    *out = (const T *)((char *)stage.p + stage_off_bytes);
    return PGICP_OK;
}

static int pinned_ensure(pgicp_ctx *c, char **buf, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return PGICP_OK;
    HIPC(c, stream_sync(c));                 // a transfer out of / into the old buffer may still be queued
    if (*buf) (void)t_host_free(*buf);
    *buf = nullptr; *cap = 0;
    HIPC(c, t_host_malloc((void **)buf, bytes + bytes / 4 + 4096, hipHostMallocDefault));
    *cap = bytes + bytes / 4 + 4096;
    return PGICP_OK;
}

size_t staged_bytes(size_t elt, int stride, int n)
{
    size_t b = elt * ((size_t)(n - 1) * stride + 3);
    return (b + 255) & ~(size_t)255;
}

template <typename T>
int sync_maps_table(pgicp_ctx *c)
{
    State<T> &S = state<T>(c);
    const int n = (int)S.maps.size();
    if (n == 0) return PGICP_OK;
    if (n > S.d_maps_cap) {
        HIPC(c, stream_sync(c));
        HIPC(c, S.d_maps.ensure(sizeof(MapDev<T>) * (size_t)(n + 16)));
        S.d_maps_cap = n + 16;
    }
    std::vector<MapDev<T>> h(n);
    for (int i = 0; i < n; i++) {
        const MapHost<T> &m = S.maps[i];
        h[i].pts = m.pts; h[i].nrm = m.nrm; h[i].ptsf = m.ptsf; h[i].cell_start = m.cell_start; h[i].g = m.g; h[i].m = m.used ? m.m : 0;
        h[i].cell_start_f = m.cell_start_f; h[i].kx = m.kx; h[i].sw = m.sw; h[i].ostart = m.ostart;
        h[i].sc_count = m.sc_count;
        h[i].slot_of = m.slot_of;
        h[i].near = m.near;
        h[i].sc_dist = m.sc_dist;
        h[i].sc_wit = m.sc_wit;
        h[i].sc_ext = m.sc_ext;
        h[i].occ = m.occ;
        h[i].first = m.first;
        h[i].nsx = (m.g.nx + 7) >> 3; h[i].nsy = (m.g.ny + 7) >> 3; h[i].nsz = (m.g.nz + 7) >> 3;
    }
    XFER(c, h2d(c, S.d_maps.p, h.data(), sizeof(MapDev<T>) * n));
    HIPC(c, stream_sync(c));   // h goes out of scope
    return PGICP_OK;
}

// Freed map blocks are kept for reuse (hipFree waits for the whole device, and loop closing / the streaming mapper create
// and destroy an index per candidate pair / per keyframe).  The pool belongs to the DEVICE, not to a context: in the
// streaming mapper a builder context allocates every block and the serving context -- which received the map through
// pgicp_map_transfer -- releases it; with per-context pools the blocks piled up where nobody allocates.  A pooled block
// carries an event recorded on the releasing context's stream; whoever takes it makes its own stream wait for that event,
// so work still queued on the old owner's stream is ordered before the new contents.
// A batch of 128 loop-closure maps is ONE block of 3-4 GB: with a 2 GiB pool it was never recycled, every batch paid a
// hipMalloc and a (device-synchronising) hipFree of that size -- 1 ms on some hosts, 45 ms on others, which made the
// loop-closing workload run at 4 700 or at 1 700 pairs/s depending on the box.  The pool may hold an eighth of the
// device's memory (36 GB of an MI355X's 288), and a block is allocated an eighth larger than asked for, so that the
// slightly larger batch that follows fits the block of the one before.
constexpr size_t kPoolLimitFallback = (size_t)8 << 30;
struct PooledBlock { char *p; hipEvent_t released; };
struct DevicePool {
    std::mutex m;
    std::multimap<size_t, PooledBlock> blocks;
    size_t bytes = 0;
    size_t limit = 0;               // set on first use: total device memory / 8
    int contexts = 0;
};
// one pool per device, made on first use (never destroyed: contexts may outlive static destruction order)
DevicePool &pool_of(int device)
{
    static std::mutex m;
    static std::vector<DevicePool *> pools;
    std::lock_guard<std::mutex> lock(m);
    if ((size_t)device >= pools.size()) pools.resize((size_t)device + 1, nullptr);
    if (!pools[device]) pools[device] = new DevicePool();
    return *pools[device];
}

int block_alloc(pgicp_ctx *c, size_t bytes, char **out, size_t *got)
{
    DevicePool &dp = pool_of(c->device);
    {
        std::lock_guard<std::mutex> lock(dp.m);
        auto it = dp.blocks.lower_bound(bytes);
        if (it != dp.blocks.end() && it->first <= bytes + bytes / 2 + (1u << 20)) {
            *out = it->second.p; *got = it->first;
            const hipEvent_t ev = it->second.released;
            dp.bytes -= it->first;
            dp.blocks.erase(it);
            if (ev) {
                (void)hipStreamWaitEvent(c->stream, ev, 0);
                (void)hipEventDestroy(ev);             // released once the recorded work has completed
            }
            return PGICP_OK;
        }
    }
    const size_t padded = bytes + bytes / 8;
    if (t_malloc((void **)out, padded) == hipSuccess) { *got = padded; return PGICP_OK; }
    (void)hipGetLastError();
    HIPC(c, t_malloc((void **)out, bytes));
    *got = bytes;
    return PGICP_OK;
}

void block_release(pgicp_ctx *c, char *p, size_t bytes)
{
    if (!p) return;
    if (c) {
        DevicePool &dp = pool_of(c->device);
        std::lock_guard<std::mutex> lock(dp.m);
        if (dp.limit == 0) {
            size_t free_b = 0, total_b = 0;
            dp.limit = hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b ? total_b / 8 : kPoolLimitFallback;
        }
        if (dp.bytes + bytes <= dp.limit) {
            PooledBlock b{p, nullptr};
            if (hipEventCreateWithFlags(&b.released, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(b.released, c->stream);
            else b.released = nullptr;
            dp.blocks.emplace(bytes, b);
            dp.bytes += bytes;
            return;
        }
    }
    (void)t_free(p);
}

// Work on the context stream that uses the map is ordered before whatever reuses the block (block_release records an
// event there); work queued on OTHER streams must have been waited for by the caller.
template <typename T>
void free_map(pgicp_ctx *c, MapHost<T> &m)
{
    if (m.block && m.block.use_count() == 1 && m.block->p) {
        block_release(c, m.block->p, m.block->bytes);
        m.block->p = nullptr;
    }
    m = MapHost<T>();
}

// one reference cloud of a batched map creation
template <typename T>
struct MapSrc {
    const T *xyz; int xyz_stride; const T *nrm; int nrm_stride; int m;
};

// Builds n maps with ONE host round trip: all centroid/bbox kernels first, one copy back of the
// statistics, then every grid build back to back on the context stream.  Each map's arrays live in a
// single device allocation.
template <typename T>
int map_create_batch(pgicp_ctx *c, int n, const MapSrc<T> *src, int mem, int center, int *map_ids)
{
    if (!c || n <= 0 || !src || !map_ids) return fail(c, PGICP_ERR_ARG, "pgicp_map_create: bad argument");
    for (int k = 0; k < n; k++)
        if (!src[k].xyz || src[k].m <= 0 || src[k].xyz_stride < 3 || (src[k].nrm && src[k].nrm_stride < 3))
            return fail(c, PGICP_ERR_ARG, "pgicp_map_create: bad argument");
    HIPC(c, hipSetDevice(c->device));
    State<T> &S = state<T>(c);
    UploadUse use(c);
    using V4 = typename Vec4<T>::type;
    // ---- phase 1: inputs on the device, centroid + bbox of every cloud ----
    std::vector<const T *> d_xyz(n), d_nrm(n, nullptr);
    if (mem == PGICP_HOST) {
        size_t tot = 0;
        for (int k = 0; k < n; k++)
            tot += staged_bytes(sizeof(T), src[k].xyz_stride, src[k].m) + (src[k].nrm ? staged_bytes(sizeof(T), src[k].nrm_stride, src[k].m) : 0);
        HIPC(c, S.staging.ensure(tot));
    }
    size_t soff = 0;
    for (int k = 0; k < n; k++) {
        if (mem == PGICP_DEVICE && (c->up[0].pending || c->up[1].pending)) {       // clouds handed out by pgicp_upload_*
            (void)upload_wait(c, src[k].xyz);                                       // (this call ends with a stream synchronisation:
            if (src[k].nrm) (void)upload_wait(c, src[k].nrm);                       //  nothing of it reads them afterwards)
        }
        int st = to_device<T>(c, src[k].xyz, src[k].xyz_stride, src[k].m, mem, S.staging, soff, &d_xyz[k]);
        if (st) return st;
        if (mem == PGICP_HOST) soff += staged_bytes(sizeof(T), src[k].xyz_stride, src[k].m);
        if (src[k].nrm) {
            st = to_device<T>(c, src[k].nrm, src[k].nrm_stride, src[k].m, mem, S.staging, soff, &d_nrm[k]);
            if (st) return st;
            if (mem == PGICP_HOST) soff += staged_bytes(sizeof(T), src[k].nrm_stride, src[k].m);
        }
    }
    std::vector<unsigned long long> h_stats((size_t)9 * n);
    for (int k = 0; k < n; k++) {
        unsigned long long *s = h_stats.data() + 9 * k;
        s[0] = s[1] = s[2] = 0; s[3] = s[4] = s[5] = ~0ULL; s[6] = s[7] = s[8] = 0;
    }
    std::vector<BuildDesc<T>> descs(n);
    int max_m = 0;
    for (int k = 0; k < n; k++) {
        BuildDesc<T> &d = descs[k];
        std::memset(&d, 0, sizeof d);
        d.xyz = d_xyz[k]; d.nrm = d_nrm[k]; d.xstride = src[k].xyz_stride; d.nstride = src[k].nrm_stride; d.m = src[k].m;
        max_m = std::max(max_m, src[k].m);
    }
    HIPC(c, c->stats.ensure(sizeof(unsigned long long) * 9 * (size_t)n));
    HIPC(c, c->bdesc.ensure(sizeof(BuildDesc<T>) * (size_t)n));
    XFER(c, h2d(c, c->stats.p, h_stats.data(), sizeof(unsigned long long) * 9 * n));
    XFER(c, h2d(c, c->bdesc.p, descs.data(), sizeof(BuildDesc<T>) * n));
    {
        ProfScope ps(c, PGICP_PROF_GRID_BUILD, max_m, n);
        launch_centroid_bbox_batch<T>(c->stream, c->bdesc.as<BuildDesc<T>>(), n, max_m, c->stats.as<unsigned long long>());
    }
    XFER(c, d2h(c, h_stats.data(), c->stats.p, sizeof(unsigned long long) * 9 * n));
    HIPC(c, stream_sync(c));

    // ---- phase 2: grids; every cloud gets its slice of ONE allocation and ONE set of build launches ----
    std::vector<MapHost<T>> Ms(n);
    long long tot_m = 0, tot_c = 0, tot_s = 0, tot_f = 0, tot_o = 0, tot_b = 0, tot_w = 0;
    // The succinct table (MapDev::sw) is an OPTION (PGICP_TABLES=succinct), the dense tables the default for every map:
    // measured in round 6 (profiles/r06_experiments/succinct_tables.txt) the succinct table makes a 100 k-pt map's tables 1.1 MB
    // instead of 14 MB and the batched build 0.9 ms shorter per 512 maps, but the fast matcher pays a third dependent load and
    // ~290 more instructions: +7 % per launch in loop closing (its traffic -13 %: the launches' bytes are results and points,
    // not tables), -6 % headline, -5 % streaming.  It is what a map too sparse for dense tables would be built with.
    const bool succ = c->table_mode == 2;
    int max_cells = 0, max_nsc = 0, max_bins = 0, max_blocks = 0;
    const int kx = std::max(1, std::min(8, c->grid_kx));
    bool any_nrm = false;
    for (int k = 0; k < n; k++) {
        MapHost<T> &M = Ms[k];
        const int m = src[k].m;
        const unsigned long long *st = h_stats.data() + 9 * k;
        M.used = true; M.m = m; M.has_nrm = src[k].nrm != nullptr;
        any_nrm = any_nrm || M.has_nrm;
        double lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            // the centroid is an order-independent fixed-point sum (2^-24 units in an int64): sum |v| 2^24 must stay
            // below 2^63 -- e.g. a million points at ECEF / UTM-scale coordinates would wrap silently.  Refused, not wrong.
            const double vmax = std::max(std::fabs((double)key_to_double(st[3 + a])), std::fabs((double)key_to_double(st[6 + a])));
            if (vmax * (double)m >= 5.0e11)
                return fail(c, PGICP_ERR_ARG, "pgicp_map_create: |coordinate| x points = " + std::to_string(vmax * (double)m) +
                                              " overflows the centroid's fixed-point sum (limit 5e11): express the cloud in a local frame first");
            const double mean_d = ((double)(long long)st[a] / 16777216.0) / (double)m;
            M.mean[a] = center ? (T)mean_d : (T)0;
            // rounding is monotone: min/max of fl(x - mean) are fl(min - mean), fl(max - mean)
            lo[a] = (double)((T)key_to_double(st[3 + a]) - M.mean[a]);
            hi[a] = (double)((T)key_to_double(st[6 + a]) - M.mean[a]);
            if (!(lo[a] <= hi[a]) || !std::isfinite(lo[a]) || !std::isfinite(hi[a]))
                return fail(c, PGICP_ERR_ARG, "pgicp_map_create: non-finite coordinates");
        }
        const double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        double h = c->prm.grid_cell;
        if (!(h > 0)) {
            // Range-scan clouds sit on surfaces (and, ring by ring, on curves): cells sized for ~2
            // points per cell of the projected bounding-box area end up holding ~15-20 points where
            // the data actually is (measured on the Velodyne-shaped benchmark map).
            const double area = ex * ey + ey * ez + ex * ez;
            h = std::sqrt(2.0 * std::max(area, 1e-12) / (double)m);
            const double diag = std::sqrt(ex * ex + ey * ey + ez * ez);
            h *= c->cell_scale;
            if (!(h > diag * 1e-4)) h = std::max(diag * 1e-4, 1e-6);
        }
        for (;;) {   // bound the dense cell table
            const double nx = std::floor(ex / h) + 1, ny = std::floor(ey / h) + 1, nz = std::floor(ez / h) + 1;
            if (nx <= 65535 && ny <= 65535 && nz <= 65535 && nx * ny * nz <= 67108864.0) break;
            h *= 1.26;
        }
        GridDesc<T> &g = M.g;
        g.h = (T)h; g.inv_h = (T)1 / g.h; g.margin = (T)0.02 * g.h;
        g.ox = (T)lo[0]; g.oy = (T)lo[1]; g.oz = (T)lo[2];
        g.nx = (int)std::floor(ex / (double)g.h) + 1; g.ny = (int)std::floor(ey / (double)g.h) + 1;
        g.nz = (int)std::floor(ez / (double)g.h) + 1;
        BuildDesc<T> &d = descs[k];
        d.g = g;
        for (int a = 0; a < 3; a++) d.mean[a] = M.mean[a];
        d.ncells = g.nx * g.ny * g.nz;
        d.nsc = ((g.nx + 7) >> 3) * ((g.ny + 7) >> 3) * ((g.nz + 7) >> 3);
        d.pbase = tot_m; d.cbase = tot_c; d.sbase = tot_s; d.fbase = tot_f; d.obase = tot_o; d.bbase = tot_b; d.wbase = tot_w;
        d.kx = kx; d.ncells_f = d.ncells * kx;
        d.nbins = (d.ncells_f >> 9) + 1;                 // bins of the build's counting sort: 512 fine cells (kMapBinShift), sentinel included
        M.kx = kx;
        // a first candidate farther than a fraction of maxDist prunes little: do not look for one beyond that
        const double reach_len = std::isfinite(c->prm.max_dist) ? c->near_frac * c->prm.max_dist : 1e30;
        d.near_reach = (int)std::min((double)kNearReach, std::max(2.0, std::ceil(reach_len / (double)g.h)));
        tot_m += m; tot_c += (long long)d.ncells + 1; tot_s += d.nsc;
        tot_f += ((long long)d.ncells_f + 1 + 3) & ~3LL;     // (every cloud's fine table starts on a 16-byte boundary: k_mfill stores four entries at a time)
        tot_o += (long long)(d.ncells >> 5) + 2;         // (a range test reads one word past the last cell's)
        tot_b += d.nbins;
        tot_w += (long long)(d.ncells_f >> 6) + 2;       // (the sentinel cell's group, and one to spare)
        max_cells = std::max(max_cells, d.ncells);
        max_bins = std::max(max_bins, d.nbins);
        max_nsc = std::max(max_nsc, d.nsc);
        max_blocks = std::max(max_blocks, ((g.nx + 1) >> 1) * ((g.ny + 1) >> 1) * ((g.nz + 1) >> 1));
    }
    if (tot_m > 0x7FFFFFF0LL || tot_c > 0x7FFFFFF0LL) {
        // the concatenated index space must fit an int: an oversized batch is built in two halves (each may split again)
        if (n == 1) return fail(c, PGICP_ERR_ARG, "pgicp_map_create: cloud too large");
        const int half = n / 2;
        const int st1 = map_create_batch<T>(c, half, src, mem, center, map_ids);
        if (st1) return st1;
        return map_create_batch<T>(c, n - half, src + half, mem, center, map_ids + half);
    }
    const size_t n_scratch = (size_t)std::max(tot_b + 2, tot_c);
    HIPC(c, c->tmp_a.ensure(sizeof(int) * (size_t)tot_m));                               // fine-cell keys: by point, then by slot
    HIPC(c, c->tmp_b.ensure(sizeof(int) * n_scratch));                                   // bin counts, then sweep scratch
    HIPC(c, c->tmp_c.ensure(sizeof(int) * ((size_t)std::max(tot_b + 1, succ ? tot_m + 1 : 0LL) / kScanChunkHost + 2)));  // the scans' block sums
    if (succ) {                                                                           // first-of-cell flags, and their ranks
        HIPC(c, c->tmp_f.ensure(sizeof(int) * (size_t)(tot_m + 1)));
        HIPC(c, c->tmp_r.ensure(sizeof(int) * (size_t)(tot_m + 1)));
    }
    HIPC(c, c->tmp_d.ensure(sizeof(int) * n_scratch));                                   // bin starts, then sweep scratch
    HIPC(c, c->tmp_e.ensure(sizeof(int) * (size_t)tot_m));                               // arrival positions
    HIPC(c, c->tmp_w.ensure(sizeof(unsigned long long) * (size_t)tot_m));                // (fine cell, index) words of the cell sort
    HIPC(c, c->tmp_p.ensure(sizeof(V4) * 2 * (size_t)tot_m));                            // the records in cloud order: (point, normal) pairs
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // (succinct: no dense tables -- b_cs keeps one entry so that the layout below stays one expression)
    const size_t b_pts = up(sizeof(V4) * (size_t)tot_m), b_nrm = any_nrm ? up(2 * sizeof(V4) * (size_t)tot_m) : 0, b_cs = up(sizeof(int) * (size_t)(succ ? 1 : tot_c)),
                 b_slot = up(sizeof(int) * (size_t)tot_m), b_sc = up(sizeof(int) * (size_t)tot_s), b_near = up(sizeof(int) * (size_t)tot_c),
                 b_csf = (kx > 1 && !succ) ? up(sizeof(int) * (size_t)tot_f) : 0, b_occ = up(sizeof(unsigned) * (size_t)tot_o),
                 b_ext = up(sizeof(float) * 6 * (size_t)tot_s), b_ptsf = sizeof(T) == 8 ? up(sizeof(float) * 4 * (size_t)tot_m + 64) : 0,
                 b_sw = succ ? up(sizeof(uint4) * (size_t)tot_w) : 0, b_ost = succ ? up(sizeof(int) * (size_t)(tot_m + 2)) : 0;
    auto blk = std::make_shared<SharedBlock>();
    { const int ast = block_alloc(c, b_pts + b_nrm + b_cs + b_slot + 3 * b_sc + b_near + b_csf + b_occ + b_ext + b_ptsf + b_sw + b_ost, &blk->p, &blk->bytes); if (ast) return ast; }
    char *base = blk->p;
    V4 *g_pts = (V4 *)base, *g_nrm = any_nrm ? (V4 *)(base + b_pts) : nullptr;
    int *g_cs = (int *)(base + b_pts + b_nrm), *g_slot = (int *)(base + b_pts + b_nrm + b_cs),
        *g_sc = (int *)(base + b_pts + b_nrm + b_cs + b_slot), *g_near = (int *)(base + b_pts + b_nrm + b_cs + b_slot + b_sc),
        *g_scd = (int *)(base + b_pts + b_nrm + b_cs + b_slot + b_sc + b_near),
        *g_wit = (int *)(base + b_pts + b_nrm + b_cs + b_slot + 2 * b_sc + b_near),
        *g_csf = kx > 1 ? (int *)(base + b_pts + b_nrm + b_cs + b_slot + 3 * b_sc + b_near) : g_cs;
    unsigned *g_occ = (unsigned *)(base + b_pts + b_nrm + b_cs + b_slot + 3 * b_sc + b_near + b_csf);
    float *g_ext = (float *)(base + b_pts + b_nrm + b_cs + b_slot + 3 * b_sc + b_near + b_csf + b_occ);
    float *g_ptsf = b_ptsf ? (float *)(base + b_pts + b_nrm + b_cs + b_slot + 3 * b_sc + b_near + b_csf + b_occ + b_ext) : nullptr;   // (16-byte records: blocks are 256-byte aligned)
    uint4 *g_sw = succ ? (uint4 *)(base + b_pts + b_nrm + b_cs + b_slot + 3 * b_sc + b_near + b_csf + b_occ + b_ext + b_ptsf) : nullptr;
    int *g_ost = succ ? (int *)(base + b_pts + b_nrm + b_cs + b_slot + 3 * b_sc + b_near + b_csf + b_occ + b_ext + b_ptsf + b_sw) : nullptr;
    for (int k = 0; k < n; k++) {
        MapHost<T> &M = Ms[k];
        const BuildDesc<T> &d = descs[k];
        M.block = blk;
        M.pts = g_pts;                               // shared by the batch; this map's points start at slot `first`
        M.nrm = M.has_nrm ? g_nrm : nullptr;
        M.first = (int)d.pbase;
        M.cell_start = succ ? nullptr : g_cs + d.cbase;
        M.cell_start_f = succ ? nullptr : g_csf + d.fbase;
        M.sw = succ ? g_sw + d.wbase : nullptr;
        M.ostart = g_ost;
        M.slot_of = g_slot + d.pbase;
        M.sc_count = g_sc + d.sbase;
        M.near = g_near + d.cbase;
        M.sc_dist = g_scd + d.sbase;
        M.sc_wit = g_wit + d.sbase;
        M.sc_ext = g_ext + 6 * d.sbase;
        M.occ = g_occ + d.obase;
        M.ptsf = g_ptsf;                             // (indexed by slot, like pts)
    }
    XFER(c, h2d(c, c->bdesc.p, descs.data(), sizeof(BuildDesc<T>) * n));
    {
        ProfScope ps(c, PGICP_PROF_GRID_BUILD, tot_m, n);
        launch_grid_build_batch<T>(c->stream, c->bdesc.as<BuildDesc<T>>(), n, tot_m, tot_b, tot_s, max_m, max_cells, max_bins, max_nsc, max_blocks, kx == 4 ? 1 : 0, c->tmp_a.as<int>(),
                                   c->tmp_b.as<int>(), c->tmp_c.as<int>(), g_cs, g_csf, c->tmp_d.as<int>(), c->tmp_e.as<int>(), c->tmp_w.as<unsigned long long>(), c->tmp_p.as<V4>(), c->tmp_n.as<V4>(), g_pts, g_nrm,
                                   g_slot, g_sc, g_near, g_scd, g_wit, g_occ, tot_o, g_ext, g_ptsf, g_sw, g_ost, succ ? c->tmp_f.as<int>() : nullptr, succ ? c->tmp_r.as<int>() : nullptr);
    }
    HIPC(c, stream_sync(c));          // `descs` (host) feeds an async copy
    HIPC(c, hipGetLastError());
    // ---- phase 3: register ----
    for (int k = 0; k < n; k++) {
        int id = -1;
        for (size_t i = 0; i < S.maps.size(); i++) if (!S.maps[i].used) { id = (int)i; break; }
        if (id < 0) { S.maps.push_back(MapHost<T>()); id = (int)S.maps.size() - 1; }
        S.maps[id] = Ms[k];
        map_ids[k] = id | id_tag<T>();
    }
    return sync_maps_table<T>(c);
}

template <typename T>
int map_create(pgicp_ctx *c, const T *xyz, int xyz_stride, const T *nrm, int nrm_stride, int m, int mem, int center,
               int *map_id)
{
    if (!map_id) return fail(c, PGICP_ERR_ARG, "pgicp_map_create: bad argument");
    MapSrc<T> s{xyz, xyz_stride, nrm, nrm_stride, m};
    return map_create_batch<T>(c, 1, &s, mem, center, map_id);
}

void translation(const double *t3, double sign, double *T)
{
    mat4_identity(T);
    T[3] = sign * t3[0]; T[7] = sign * t3[1]; T[11] = sign * t3[2];
}

struct BatchLayout {
    int P = 0;
    int max_n = 0;
    int max_rows = 1;
    int bin_shift = 2;          // reading-sort bins are (1 << bin_shift)^3 map cells
    long long total = 0;
    int knn = 1;                // pairs per reading point (ChainDev::knn)
    bool normals = false;       // the readings' normals travel with them (a SurfaceNormalOutlierFilter is in the chain)
    int table_kinds = 0;        // bit 0: some map of the batch carries the dense cell tables, bit 1: some map the succinct one (MapDev::sw)
    double max_h = 0.0;         // largest cell edge among the batch's maps
    // Rings of cells the fast matcher walks beyond the 27-cell block before it queues a query: 3 unseeded / 1 seeded on the 9 cm
    // cells of a 1 M-pt map (measured, rounds 3 and 4); the 28 cm cells of a 100 k-pt keyframe map -- loop closing -- are
    // cheaper to walk than to queue: 5 / 2 there (round 5, tools/r5_lc_knobs.sh: +3...6 %).  The environment knobs override.
    int rings_unseeded = 3, rings_seeded = 1;
    int max_pairs() const { return (int)std::min<long long>((long long)max_n * knn, 0x7FFFFFFFLL); }
};

// Prepare a batch: stage readings, fill + upload ProblemDev, run the prologue
// transform.  `Tpre_of(p, out16)` provides each problem's pre-transform.
template <typename T, typename F>
int batch_begin(pgicp_ctx *c, int P, const pgicp_problem *pr, F Tpre_of, BatchLayout &L, std::vector<ProblemDev> &hp, int hint_kind = -1)
{
    if (P > 65535) return fail(c, PGICP_ERR_ARG, "pgicp: at most 65535 problems per batch (the problem index is a launch-grid dimension)");
    State<T> &S = state<T>(c);
    L.P = P; L.max_n = 0; L.max_rows = 1; L.total = 0; L.table_kinds = 0;
    L.knn = std::max(1, c->prm.knn);
    L.normals = c->prm.normal_max_angle > 0.0;
    if (L.normals)
        for (int p = 0; p < P; p++)
            if (!pr[p].normals || pr[p].nstride < 3)
                return fail(c, PGICP_ERR_ARG, "SurfaceNormalOutlierFilter: the reading of problem " + std::to_string(p) + " has no normals (pgicp_problem.normals)");
    size_t stage_total = 0;
    double dens = 0.0;                         // largest reading-to-map size ratio of the batch
    for (int p = 0; p < P; p++) {
        if (pr[p].n <= 0 || !pr[p].reading || pr[p].stride < 3)
            return fail(c, PGICP_ERR_ARG, "pgicp: bad reading in problem " + std::to_string(p));
        MapHost<T> *M = get_map<T>(c, pr[p].map_id);
        if (!M) return fail(c, PGICP_ERR_ARG, "pgicp: unknown map id " + std::to_string(pr[p].map_id));
        L.max_n = std::max(L.max_n, pr[p].n);
        L.table_kinds |= M->sw ? 2 : 1;
        L.max_h = std::max(L.max_h, (double)M->g.h);
        dens = std::max(dens, (double)pr[p].n / (double)M->m);
        L.total += pr[p].n;
        if (pr[p].mem == PGICP_HOST) stage_total += staged_bytes(sizeof(T), pr[p].stride, pr[p].n) +
                                                    (L.normals ? staged_bytes(sizeof(T), pr[p].nstride, pr[p].n) : 0);
    }
    L.rings_unseeded = c->fast_rings_unseeded > 0 ? c->fast_rings_unseeded : (L.max_h >= 0.18 ? 5 : 3);
    L.rings_seeded = c->fast_rings_seeded > 0 ? c->fast_rings_seeded : (L.max_h >= 0.18 ? 2 : 1);
    if ((long long)L.max_n * L.knn > 0x7FFFFFF0LL || L.total * L.knn > 0x7FFFFFF0LL)
        return fail(c, PGICP_ERR_ARG, "pgicp: points x knn exceeds 2^31: split the batch");
    // Bins of the reading sort are blocks of map cells.  The rank pass costs O(bin population) per
    // point and map cells hold ~17 points where the data is, so a reading as dense as its map gets
    // single-cell bins and a sparse one 4x4x4-cell bins; the bin table is kept under 2^26 entries.
    L.bin_shift = std::max(0, std::min(4, (dens <= 0.25 ? 2 : (dens <= 2.0 ? 1 : 0)) + c->bin_shift_add));
    for (;; ++L.bin_shift) {
        const int s = L.bin_shift, r = (1 << s) - 1;
        L.max_rows = 1;
        for (int p = 0; p < P; p++) {
            const GridDesc<T> &g = get_map<T>(c, pr[p].map_id)->g;
            const long long rows = (long long)((g.nx + r) >> s) * ((g.ny + r) >> s) * ((g.nz + r) >> s);
            L.max_rows = (int)std::max<long long>(L.max_rows, std::min<long long>(rows, 1LL << 30));
        }
        if (s >= 2 || ((long long)L.max_rows < (1LL << 25) && (long long)L.max_rows * P <= (1LL << 26))) break;
    }
    // (the bin table is indexed with ints on the device)
    if ((long long)L.max_rows * P > 0x7FFFFFFFLL)
        return fail(c, PGICP_ERR_ARG, "pgicp: batch too large for the reading sort (problems x map blocks = " +
                                      std::to_string((long long)L.max_rows * P) + " > 2^31 - 1): split the batch");
    HIPC(c, S.rd_pre.ensure(sizeof(typename Vec4<T>::type) * (size_t)L.total));
    HIPC(c, S.rd_sorted.ensure(sizeof(T) * 3 * (size_t)L.total));
    {
        const size_t nbins = (size_t)P * L.max_rows;
        HIPC(c, c->qrow.ensure(sizeof(int) * (size_t)L.total));
        // (the sort's words; afterwards the selection's key list: one key per PAIR)
        HIPC(c, c->qtmp.ensure(std::max(sizeof(unsigned long long) * (size_t)L.total, sizeof(T) * (size_t)L.total * L.knn)));
        if (c->prm.robust_fct != PGICP_ROBUST_NONE) HIPC(c, c->robust_dev.ensure(sizeof(T) * (size_t)L.total * L.knn));
        HIPC(c, c->order.ensure(sizeof(int) * (size_t)L.total));
        if (c->prm.sum_order == PGICP_SUM_ORDER_SCAN) HIPC(c, c->scan_pos.ensure(sizeof(int) * (size_t)L.total));
        HIPC(c, c->slow_list.ensure(sizeof(int2) * (size_t)L.total));
        HIPC(c, c->slow_lb.ensure(sizeof(T) * (size_t)L.total));
        HIPC(c, c->slow_ring.ensure(sizeof(int) * (size_t)L.total));
        HIPC(c, c->slow2.ensure(sizeof(int) * (size_t)L.total));
        HIPC(c, c->active.ensure(sizeof(int) * (size_t)P));
        HIPC(c, c->qcounts.ensure(sizeof(int) * nbins));
        HIPC(c, c->qstart.ensure(sizeof(int) * (nbins + 1)));
        HIPC(c, c->qblock.ensure(sizeof(int) * (nbins / kScanChunkHost + 2)));
    }
    HIPC(c, S.slot.ensure(sizeof(int) * (size_t)L.total * L.knn));
    HIPC(c, S.d2.ensure(sizeof(T) * (size_t)L.total * L.knn));
    if (L.normals) {
        HIPC(c, S.nrm_pre.ensure(sizeof(typename Vec4<T>::type) * (size_t)L.total));
        HIPC(c, S.nrm_sorted.ensure(sizeof(T) * 3 * (size_t)L.total));
    }
    HIPC(c, S.none_r.ensure(sizeof(T) * (size_t)L.total));
    // (the problem records and the readings' source descriptors share one buffer: one upload -- batch_begin)
    const size_t src_off = (sizeof(ProblemDev) * (size_t)P + 63) & ~(size_t)63;
    HIPC(c, c->probs.ensure(src_off + sizeof(SrcDesc) * (size_t)P));
    HIPC(c, c->partials.ensure(sizeof(double) * (size_t)P * reduce_blocks(L.max_pairs()) * kCovTerms));
    HIPC(c, c->sums.ensure(sizeof(double) * (size_t)P * kCovTerms));
    HIPC(c, c->small.ensure(256));
    HIPC(c, c->sel_tables.ensure(trim_select_table_bytes(P)));
    HIPC(c, c->queue.ensure(knn_queue_bytes(P, L.max_n, sizeof(T))));
    if (stage_total) HIPC(c, S.staging.ensure(stage_total));

    hp.assign(P, ProblemDev());
    std::vector<SrcDesc> &hs = c->h_src;        // context members: they outlive the async uploads below
    hs.assign(P, SrcDesc());
    size_t soff = 0;
    long long off = 0;
    int up_mask = 0;
    for (int p = 0; p < P; p++) {
        const T *d_rd = nullptr;
        if (pr[p].mem == PGICP_DEVICE && (c->up[0].pending || c->up[1].pending)) up_mask |= upload_wait(c, pr[p].reading);
        int st = to_device<T>(c, (const T *)pr[p].reading, pr[p].stride, pr[p].n, pr[p].mem, S.staging, soff, &d_rd);
        if (st) return st;
        if (pr[p].mem == PGICP_HOST) soff += staged_bytes(sizeof(T), pr[p].stride, pr[p].n);
        hs[p].ptr = d_rd; hs[p].stride = pr[p].stride; hs[p].nptr = nullptr; hs[p].nstride = 0;
        if (L.normals) {
            const T *d_nr = nullptr;
            if (pr[p].mem == PGICP_DEVICE && (c->up[0].pending || c->up[1].pending)) up_mask |= upload_wait(c, pr[p].normals);
            st = to_device<T>(c, (const T *)pr[p].normals, pr[p].nstride, pr[p].n, pr[p].mem, S.staging, soff, &d_nr);
            if (st) return st;
            if (pr[p].mem == PGICP_HOST) soff += staged_bytes(sizeof(T), pr[p].nstride, pr[p].n);
            hs[p].nptr = d_nr; hs[p].nstride = pr[p].nstride;
        }
        ProblemDev &D = hp[p];
        std::memset(&D, 0, sizeof D);
        D.map = map_index<T>(c, pr[p].map_id); D.n = pr[p].n; D.off = off; D.knn = L.knn;
        off += pr[p].n;
        if (hint_kind >= 0 && c->sel_hints_on && (int)c->sel_hints[hint_kind].size() == P) std::memcpy(D.qhint, c->sel_hints[hint_kind][(size_t)p].q, sizeof D.qhint);
        Tpre_of(p, D.Tpre);
        mat4_identity(D.T_iter); mat4_identity(D.T_prev); mat4_identity(D.dT);
        for (int i = 0; i < 12; i++) { D.Tcur[i] = D.T_iter[i]; D.Tcur_f[i] = (float)D.T_iter[i]; }
        checker_init(D.chk);
    }
    // One upload and one kernel set a batch up (round 6; three copies and three memsets before: six launches a call, twelve per
    // scan of the facade, which is bound by its launches at sensor size): the problem records and the source descriptors
    // travel as ONE pinned block into ONE buffer; k_batch_setup writes the identity list of active problems and clears the
    // small counters, the selection tables and the reading sort's bin counts.
    { const int pst = pinned_ensure(c, &c->h_up, &c->h_up_cap, src_off + sizeof(SrcDesc) * (size_t)P); if (pst) return pst; }
    std::memcpy(c->h_up, hp.data(), sizeof(ProblemDev) * (size_t)P);
    std::memcpy(c->h_up + src_off, hs.data(), sizeof(SrcDesc) * (size_t)P);
    HIPC(c, hipMemcpyAsync(c->probs.p, c->h_up, src_off + sizeof(SrcDesc) * (size_t)P, hipMemcpyHostToDevice, c->stream));
    const SrcDesc *src_dev = (const SrcDesc *)((const char *)c->probs.p + src_off);
    std::vector<int> &ident = c->h_ident;
    ident.resize(P);
    std::iota(ident.begin(), ident.end(), 0);
    launch_batch_setup(c->stream, c->active.as<int>(), P, c->small.as<int>(), 64, c->sel_tables.as<int>(), (long long)(trim_select_table_bytes(P) / sizeof(int)),
                       c->qcounts.as<int>(), (long long)P * L.max_rows);
    c->counters_clean = 1;
    c->seg_clean = 0;               // a new batch: its first matcher launch clears the segmented counters itself
    {
        ProfScope ps(c, PGICP_PROF_PRETRANSFORM, L.total, P);
        // pre-transform + ordering of each reading by (block of map cells, cell, index), once per scan: waves stay spatially
        // coherent for every iteration
        launch_query_sort<T>(c->stream, c->probs.as<ProblemDev>(), src_dev, S.d_maps.template as<MapDev<T>>(),
                             S.rd_pre.template as<typename Vec4<T>::type>(), S.rd_sorted.template as<T>(), c->qrow.as<int>(), c->qtmp.as<unsigned long long>(),
                             c->order.as<int>(), c->qcounts.as<int>(), c->qblock.as<int>(), c->qstart.as<int>(),
                             P, L.max_n, L.max_rows, L.bin_shift, L.normals ? S.nrm_pre.template as<typename Vec4<T>::type>() : nullptr,
                             L.normals ? S.nrm_sorted.template as<T>() : nullptr);
        if (c->prm.sum_order == PGICP_SUM_ORDER_SCAN)
            launch_invert_order(c->stream, c->probs.as<ProblemDev>(), c->order.as<int>(), c->scan_pos.as<int>(), P, L.max_n);
        upload_consumed(c, up_mask);                 // its first kernel is the only one that reads the readings where they lie
        c->up_seen = 0;
    }
    // (no synchronisation here: hp is the caller's, hs / ident are context members, and every caller ends with
    // a stream synchronisation before it returns -- a wait at this point idles the GPU for ~25 us per scan)
    if (const char *e = std::getenv("PGICP_TRACE_ORIG")) {     // diagnostics build: narrate one query of problem 0
        HIPC(c, stream_sync(c));
        std::vector<int> ord(hp[0].n);
        (void)hipMemcpy(ord.data(), c->order.as<int>(), sizeof(int) * ord.size(), hipMemcpyDeviceToHost);
        const int want = std::atoi(e);
        for (int j = 0; j < (int)ord.size(); j++) if (ord[j] == want) { (void)knn_trace_set(j); break; }
    }
    return PGICP_OK;
}

template <typename T>
void enqueue_iteration(pgicp_ctx *c, const BatchLayout &L, const ChainDev<T> &ch, bool with_solve, long long act_units,
                       long long act_probs, int use_seed)
{
    // only the problems still iterating are launched: `active` lists them first (k_compact_active)
    const int nA = (int)act_probs;
    const int *active = c->active.as<int>();
    State<T> &S = state<T>(c);
    const MapDev<T> *maps = S.d_maps.template as<MapDev<T>>();
    ProblemDev *probs = c->probs.as<ProblemDev>();
    const T *rd_nrm = L.normals ? S.nrm_sorted.template as<T>() : nullptr;
    if (L.knn > 1) {
        // KDTreeMatcher.knn > 1: the top-K matcher finds every pair exactly (no queue, no lazy resolution), one selection
        // over the knn * N distances, the minimiser over the knn * N pairs
        {
            ProfScope ps(c, PGICP_PROF_KNN_GRID, act_units, act_probs);
            (void)launch_knn_topk<T>(c->stream, probs, maps, S.rd_sorted.template as<T>(), S.slot.template as<int>(), S.d2.template as<T>(), ch, nA, L.max_n, active);
        }
        {
            ProfScope ps(c, PGICP_PROF_TRIM, act_units, act_probs);
            launch_trim_select<T>(c->stream, probs, S.d2.template as<T>(), ch, nA, L.max_pairs(), 0, active, c->sel_tables.as<int>(), c->qtmp.p, nullptr, use_seed);
        }
        {
            ProfScope ps(c, PGICP_PROF_REDUCE, act_units, act_probs);
            launch_reduce<T>(c->stream, probs, maps, S.rd_sorted.template as<T>(), rd_nrm, S.slot.template as<int>(), S.d2.template as<T>(),
                             c->partials.as<double>(), nA, L.max_pairs(), active, ch);
        }
        if (with_solve) {
            ProfScope ps(c, PGICP_PROF_SOLVE, act_units, act_probs);
            launch_solve<T>(c->stream, probs, c->partials.as<double>(), ch, c->small.as<int>(), nA, L.max_pairs(), active, nullptr, nullptr, nullptr);
            launch_compact_active(c->stream, probs, L.P, c->active.as<int>(), c->h_flag, c->stamp_dev, c->small.as<int>() + 16);
        }
        return;
    }
    {
        ProfScope ps(c, c->prm.matcher == PGICP_MATCHER_BRUTE ? PGICP_PROF_KNN_BRUTE : PGICP_PROF_KNN_GRID, act_units, act_probs);
        if (!use_seed && c->prm.matcher == PGICP_MATCHER_GRID) ps.also(PGICP_PROF_KNN_GRID_UNSEEDED);
        launch_knn<T>(c->stream, c->prm.matcher, probs, maps, S.rd_sorted.template as<T>(), S.slot.template as<int>(),
                      S.d2.template as<T>(), ch, nA, L.max_n, use_seed, c->small.as<int>() + 16, c->slow_list.as<int2>(),
                      c->slow_lb.as<T>(), c->slow_ring.as<int>(), use_seed ? L.rings_seeded : L.rings_unseeded, active, S.none_r.template as<T>(),
                      L.P, c->queue.p, c->seg_clean ? 0 : 1, L.table_kinds);
        c->counters_clean = 0;
    }
    {
        ProfScope ps(c, PGICP_PROF_TRIM, act_units, act_probs);
        // (with the grid matcher this selection also clears the matcher's segmented queue counters for the next iteration)
        const bool grid = c->prm.matcher == PGICP_MATCHER_GRID;
        launch_trim_select<T>(c->stream, probs, S.d2.template as<T>(), ch, nA, L.max_n, 0, active, c->sel_tables.as<int>(), c->qtmp.p,
                              grid ? (int *)c->queue.p : nullptr, ch.robust.fct != 0 ? 0 : (use_seed || c->sel_guess_first));      // (every selection but a run's first starts from the last one's result,
                                                                                          // ProblemDev::qraw; a run's first from the previous call's, ::qhint, if there is one)
        c->seg_clean = grid ? 1 : 0;
    }
    const bool robust = ch.robust.fct != 0;
    // RobustOutlierFilter: nothing is trimmed -- the threshold the lazy path resolves queued queries up to is +inf from here on
    if (robust) launch_robust_open(c->stream, probs, active, nA);
    if (c->prm.matcher == PGICP_MATCHER_GRID) {
        // lazy resolution: only queued queries whose lower bound is within the threshold just
        // selected (an upper bound of the final one) are searched exactly; then the threshold
        // is re-selected if anything changed.  Kept pairs / threshold / n_finite stay exact.
        {
            ProfScope ps(c, PGICP_PROF_KNN_SLOW, act_units, act_probs);
            launch_knn_med<T>(c->stream, probs, maps, S.rd_sorted.template as<T>(), S.slot.template as<int>(), S.d2.template as<T>(),
                              ch, c->small.as<int>() + 16, c->slow_list.as<int2>(), c->slow_lb.as<T>(), c->slow_ring.as<int>(),
                              c->slow2.as<int>(), c->med_rings, use_seed, S.none_r.template as<T>());
            launch_knn_slow<T>(c->stream, probs, maps, S.rd_sorted.template as<T>(), S.slot.template as<int>(),
                               S.d2.template as<T>(), ch, c->small.as<int>() + 16, c->slow_list.as<int2>(), c->slow_lb.as<T>(),
                               c->slow2.as<int>(), 0, S.none_r.template as<T>());
            static const bool each_pass = std::getenv("PGICP_PHASE_EACH_PASS") != nullptr;     // diagnostics builds only
            if (each_pass) {
                unsigned long long ph[48];
                (void)stream_sync(c);
                if (knn_phase_read(ph, 1) == 0)
                    std::fprintf(stderr, "    pass (seeded %d): wave-per-query kernel entries=%llu, longest wave %llu, longest entry %llu cycles; per entry: walk %.0f\n",
                                 use_seed, ph[46], ph[44], ph[45], (double)ph[36] / (double)std::max<unsigned long long>(1ULL, ph[46]));
            }
        }
        ProfScope ps(c, PGICP_PROF_TRIM, 0, 0);
        if (!robust) launch_trim_select<T>(c->stream, probs, S.d2.template as<T>(), ch, nA, L.max_n, 1, active, c->sel_tables.as<int>(), c->qtmp.p, nullptr, 1);
    }
    if (robust) {
        // the scale of this iteration from the (now exact) distances: median, absolute deviations, their median
        ProfScope ps(c, PGICP_PROF_TRIM, 0, 0);
        launch_robust_scale<T>(c->stream, probs, S.d2.template as<T>(), c->robust_dev.as<T>(), ch, nA, L.max_n, active, c->sel_tables.as<int>(), c->qtmp.p);
    }
    {
        ProfScope ps(c, PGICP_PROF_REDUCE, act_units, act_probs);
        launch_reduce<T>(c->stream, probs, maps, S.rd_sorted.template as<T>(), rd_nrm, S.slot.template as<int>(), S.d2.template as<T>(),
                         c->partials.as<double>(), nA, L.max_n, active, ch);
    }
    if (with_solve) {
        ProfScope ps(c, PGICP_PROF_SOLVE, act_units, act_probs);
        if (L.P == 1) {
            // one problem: the solve kernel also does the bookkeeping of k_compact_active (one launch less per iteration)
            launch_solve<T>(c->stream, probs, c->partials.as<double>(), ch, c->small.as<int>(), nA, L.max_n, active, c->h_flag, c->stamp_dev,
                            c->small.as<int>() + 16);
        } else {
            launch_solve<T>(c->stream, probs, c->partials.as<double>(), ch, c->small.as<int>(), nA, L.max_n, active, nullptr, nullptr, nullptr);
            launch_compact_active(c->stream, probs, L.P, c->active.as<int>(), c->h_flag, c->stamp_dev, c->small.as<int>() + 16);
        }
    }
}

// One ICP iteration.  For a batch of a few problems (a streamed scan, a SLAM front end) -- a chain of ten small launches --
// the sequence can be captured once per shape into a hipGraph and replayed (PGICP_GRAPH_MAX_P = largest such batch).
// Everything a kernel argument depends on is in the key; a context keeps the eight graphs used last.  OFF by default:
// built, parity-green (74 GPU tests with it on) and measured on ROCm 7.2 / MI355X -- 4 725 replays in a 1 500-scan SLAM
// run, 860-990 scans/s with and without, streaming 1 066 vs 1 103: hipGraphLaunch of ten kernel nodes costs the host and
// the GPU what ten launches cost there, and the workloads are not bound by launching (GPU 52 % busy in the SLAM run; the
// rest is host logic between ICP calls).
template <typename T>
void one_iteration(pgicp_ctx *c, const BatchLayout &L, const ChainDev<T> &ch, bool with_solve, long long act_units,
                   long long act_probs, int use_seed)
{
    State<T> &S = state<T>(c);
    const bool want_graph = L.P <= c->graph_max_problems && !c->prof_on && c->graph_failures < 3;
    bool done = false;
    if (want_graph) {
        // FNV-1a over the argument values: shapes, switches, chain parameters, and every buffer the launches name
        unsigned long long key = 1469598103934665603ULL;
        auto mix = [&key](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; i++) { key ^= b[i]; key *= 1099511628211ULL; } };
        // (sel_guess_first decides the `guess` argument of the iteration's first selection launch: the hinted and the
        //  unhinted first iteration are different graphs)
        const int shape[13] = {(int)sizeof(T), L.P, (int)act_probs, L.max_n, use_seed, with_solve ? 1 : 0, c->med_rings, L.rings_seeded,
                               L.rings_unseeded, c->prm.matcher, c->seg_clean, c->sel_guess_first, c->sel_hints_on};
        mix(shape, sizeof shape);
        mix(&ch, sizeof ch);
        const void *bufs[] = {c->probs.p, S.d_maps.p, S.rd_sorted.p, S.slot.p, S.d2.p, S.none_r.p, c->small.p, c->slow_list.p, c->slow_lb.p,
                              c->slow_ring.p, c->slow2.p, c->active.p, c->queue.p, c->sel_tables.p, c->qtmp.p, c->partials.p, c->h_flag, c->stamp_dev,
                              c->robust_dev.p, L.normals ? S.nrm_sorted.p : nullptr};
        mix(bufs, sizeof bufs);
        pgicp_ctx::IterGraph *g = nullptr;
        for (auto &e : c->iter_graphs) if (e.exec && e.key == key) { g = &e; break; }
        if (!g) {
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                enqueue_iteration<T>(c, L, ch, with_solve, act_units, act_probs, use_seed);
                ok = hipStreamEndCapture(c->stream, &graph) == hipSuccess && graph != nullptr;
            }
            if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
            if (ok) {
                if (c->iter_graphs.size() < 8) { c->iter_graphs.push_back(pgicp_ctx::IterGraph()); g = &c->iter_graphs.back(); }
                else {
                    g = &c->iter_graphs[0];
                    for (auto &e : c->iter_graphs) if (e.used < g->used) g = &e;
                    if (g->exec) (void)hipGraphExecDestroy(g->exec);
                }
                g->key = key; g->exec = exec;
                c->graph_captures++;
            } else {
                (void)hipGetLastError();
                c->graph_failures++;
            }
        }
        if (g) {
            g->used = ++c->graph_clock;
            if (hipGraphLaunch(g->exec, c->stream) == hipSuccess) { done = true; c->graph_launches++; }
            else { (void)hipGetLastError(); c->graph_failures++; }
        }
    }
    if (!done) enqueue_iteration<T>(c, L, ch, with_solve, act_units, act_probs, use_seed);
    c->counters_clean = with_solve ? 1 : 0;      // k_compact_active clears the matcher's queue counters
    c->seg_clean = c->prm.matcher == PGICP_MATCHER_GRID ? 1 : 0;     // (a replayed graph did not pass through enqueue_iteration)
    if (with_solve) ++c->flag_stamp;
}

// every problem of the batch carries a hint for the first selection of its first pass
static bool hints_cover_first(const std::vector<ProblemDev> &hp)
{
    for (const ProblemDev &D : hp) if (!(D.qhint[0][0] > 0.0)) return false;
    return !hp.empty();
}
// what this call's selections found becomes the next call's hints (problem by problem; a call with another number of problems starts over)
static void hints_store(pgicp_ctx *c, int kind, const std::vector<ProblemDev> &hp)
{
    std::vector<pgicp_ctx::SelHint> &H = c->sel_hints[kind];
    H.resize(hp.size());
    for (size_t p = 0; p < hp.size(); p++) std::memcpy(H[p].q, hp[p].qrec, sizeof H[p].q);
}

// number of finished problems after the iteration just enqueued (its k_compact_active carries c->flag_stamp)
static int wait_iteration_flag(pgicp_ctx *c)
{
    const int want = c->flag_stamp;
    const auto t0 = std::chrono::steady_clock::now();
    for (int spin = 0;; ++spin) {
        if (__atomic_load_n(&c->h_flag[1], __ATOMIC_ACQUIRE) == want) return __atomic_load_n(&c->h_flag[0], __ATOMIC_RELAXED);
        if ((spin & 63) == 63) {
            const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            if (us > c->poll_us) break;
            std::this_thread::yield();
        }
    }
    if (stream_sync(c) != hipSuccess) { fail(c, PGICP_ERR_HIP, "pgicp: stream synchronisation failed"); return -1; }
    if (__atomic_load_n(&c->h_flag[1], __ATOMIC_ACQUIRE) != want) {
        // a launch failed somewhere: the device's stamp and the host's are out of step.  Take the device's, so that the
        // context's NEXT call can work again instead of timing out for ever.
        int dev_stamp = 0;
        if (hipMemcpy(&dev_stamp, c->stamp_dev, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) c->flag_stamp = dev_stamp;
        fail(c, PGICP_ERR_HIP, "pgicp: iteration flag not written");
        return -1;
    }
    return __atomic_load_n(&c->h_flag[0], __ATOMIC_RELAXED);
}

// `residual` / `res_ratio` / `res_status` (optional, P entries each): after the ICP, LoopCloser::ComputeResidualError's chain
// on the result (LoopCloser.hpp:343-365) -- see pgicp_align_residual_batch_*.
template <typename T>
int align_batch(pgicp_ctx *c, int P, const pgicp_problem *pr, double *T_out, pgicp_stats *stats, double *residual = nullptr,
                double *res_ratio = nullptr, int *res_status = nullptr)
{
    if (!c || P <= 0 || !pr || !T_out) return fail(c, PGICP_ERR_ARG, "pgicp_align_batch: bad argument");
    HIPC(c, hipSetDevice(c->device));
    State<T> &S = state<T>(c);
    const pgicp_params &prm = c->prm;
    std::vector<std::vector<double>> Tref(P, std::vector<double>(16));
    BatchLayout L;
    std::vector<ProblemDev> hp;
    for (int p = 0; p < P; p++) {
        MapHost<T> *M = get_map<T>(c, pr[p].map_id);
        if (!M) return fail(c, PGICP_ERR_ARG, "pgicp_align: unknown map id");
        if (!M->has_nrm && needs_ref_normals(prm))
            return fail(c, PGICP_ERR_ARG, "pgicp_align: reference has no normals descriptor");
    }
    static const bool host_timing = std::getenv("PGICP_HOST_TIMING") != nullptr;      // diagnostics
    const auto ht0 = std::chrono::steady_clock::now();
    int st = batch_begin<T>(c, P, pr, [&](int p, double *Tpre) {
        MapHost<T> *M = get_map<T>(c, pr[p].map_id);
        double mean[3] = {(double)M->mean[0], (double)M->mean[1], (double)M->mean[2]};
        double Tm_inv[16];
        translation(mean, -1.0, Tm_inv);
        translation(mean, +1.0, Tref[p].data());
        mat4_mul(Tm_inv, pr[p].T_init, Tpre);
    }, L, hp, 0);
    if (st) return st;
    const bool hint_first = hints_cover_first(hp);
    long long total_m = 0;                       // (profile only) reference points over the batch's problems
    if (c->prof_on) for (int p = 0; p < P; p++) total_m += get_map<T>(c, pr[p].map_id)->m;
    const auto ht1 = std::chrono::steady_clock::now();
    const ChainDev<T> ch = chain_of<T>(c, prm);
    const int every = std::max(1, prm.check_every);
    // active-problem accounting for the profile: exact when check_every == 1 and all
    // readings have the same size (the benchmark's case), an upper bound otherwise
    int n_done = 0;
    for (int it = 0; it < prm.max_iters; it++) {
        const long long act_p = P - n_done;
        c->prof_next_m = total_m * act_p / P;
        c->sel_guess_first = it == 0 && hint_first ? 1 : 0;
        one_iteration<T>(c, L, ch, true, L.total * act_p / P, act_p, it > 0 ? 1 : 0);
        c->sel_guess_first = 0;
        if ((it + 1) % every == 0 || it + 1 == prm.max_iters) {
            // k_compact_active stores {problems done, stamp} straight into pinned host memory: polling it spares
            // the copy and the wake-up of a blocking wait (30-40 us of idle GPU per iteration of a single scan)
            int got = wait_iteration_flag(c);
            if (got < 0) return PGICP_ERR_HIP;
            n_done = got;
            if (n_done >= P) break;
        }
    }
    const auto ht2 = std::chrono::steady_clock::now();
    const bool with_cov = prm.error_minimizer != PGICP_MINIMIZER_POINT_TO_POINT;      // (PointToPoint: the base class's zeros; its WithCov form: the same Censi estimate)
    if (with_cov) {
        ProfScope ps(c, PGICP_PROF_COV, L.total, P);
        launch_cov<T>(c->stream, c->probs.as<ProblemDev>(), S.d_maps.template as<MapDev<T>>(), S.rd_sorted.template as<T>(),
                      L.normals ? S.nrm_sorted.template as<T>() : nullptr, S.slot.template as<int>(), S.d2.template as<T>(),
                      c->partials.as<double>(), c->sums.as<double>(), P, L.max_pairs(), ch);
    } else HIPC(c, hipMemsetAsync(c->sums.p, 0, sizeof(double) * (size_t)P * kCovTerms, c->stream));
    std::vector<double> cs((size_t)P * kCovTerms);
    { const int pst = pinned_ensure(c, &c->h_down, &c->h_down_cap, sizeof(ProblemDev) * (size_t)P); if (pst) return pst; }
    HIPC(c, hipMemcpyAsync(c->h_down, c->probs.p, sizeof(ProblemDev) * P, hipMemcpyDeviceToHost, c->stream));
    XFER(c, d2h(c, cs.data(), c->sums.p, sizeof(double) * cs.size()));
    std::vector<double> rsys;
    if (residual || res_ratio || res_status) {
        // The residual check of the result: one more pass of the chain WITHOUT a solve, with the final transform (it is in
        // ProblemDev::Tcur) -- transform, match, outlier weights, error elements.  The pass is SEEDED with the last
        // iteration's correspondences, which the final increment (below the convergence thresholds) hardly moves: as a
        // separate, unseeded chain it cost a fifth of a loop-closure batch.  Seeds are candidates only: the matches are exact.
        launch_reopen(c->stream, c->probs.as<ProblemDev>(), P);
        XFER(c, h2d(c, c->active.p, c->h_ident.data(), sizeof(int) * P));
        c->prof_next_m = total_m;
        one_iteration<T>(c, L, ch, false, L.total, P, 1);
        HIPC(c, c->sums2.ensure(sizeof(double) * (size_t)P * kSys));
        launch_sum_partials(c->stream, c->partials.as<double>(), reduce_blocks(L.max_pairs()), kSys, c->probs.as<ProblemDev>(), 0,
                            c->sums2.as<double>(), P);
        rsys.resize((size_t)P * kSys);
        XFER(c, d2h(c, rsys.data(), c->sums2.p, sizeof(double) * rsys.size()));
    }
    HIPC(c, stream_sync(c));
    HIPC(c, hipGetLastError());
    std::memcpy(hp.data(), c->h_down, sizeof(ProblemDev) * (size_t)P);
    hints_store(c, 0, hp);
    for (int p = 0; p < P && !rsys.empty(); p++) {
        const double *rs = rsys.data() + (size_t)p * kSys;
        const bool ok = hp[p].status == PGICP_ST_OK && rs[28] > 0.0;
        if (res_status) res_status[p] = ok ? PGICP_OK : PGICP_ERR_NO_MATCH;
        if (res_ratio) res_ratio[p] = ok ? rs[27] / ((double)pr[p].n * L.knn) : 0.0;
        if (residual) residual[p] = ok ? rs[29] : std::numeric_limits<double>::infinity();
    }
    if (host_timing) {
        const auto ht3 = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "align_batch P=%d: begin %.2f ms, iterations %.2f ms, tail %.2f ms\n", P, ms(ht0, ht1), ms(ht1, ht2), ms(ht2, ht3));
    }
    int worst = PGICP_OK;
    for (int p = 0; p < P; p++) {
        const ProblemDev &D = hp[p];
        double *To = T_out + 16 * p;
        pgicp_stats s;
        std::memset(&s, 0, sizeof s);
        s.status = D.status;
        s.iterations = D.iters; s.converged = D.converged; s.max_iter_reached = D.max_iter_reached;
        s.status = D.status == PGICP_ST_BOUND ? PGICP_ERR_BOUND : D.status;
        s.overlap = D.sys[27] / ((double)D.n * L.knn); s.residual = D.sys[29]; s.trim_limit = D.limit;
        s.n_kept = D.n_kept; s.n_finite = D.n_finite;
        if (D.status == PGICP_ST_OK) {
            double t1[16];
            mat4_mul(D.T_iter, D.Tpre, t1);
            mat4_mul(Tref[p].data(), t1, To);
            double H[36], G[36], Hi[36], tmp[36];
            sys_to_full(cs.data() + (size_t)p * kCovTerms, H);
            sys_to_full(cs.data() + (size_t)p * kCovTerms + 21, G);
            if (!with_cov) { /* zeros */ }
            else if (inverse6(H, Hi)) {
                const double s2 = prm.sensor_std_dev * prm.sensor_std_dev;
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 6; j++) {
                        double a = 0.0;
                        for (int k = 0; k < 6; k++) a += Hi[i * 6 + k] * G[k * 6 + j];
                        tmp[i * 6 + j] = a;
                    }
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 6; j++) {
                        double a = 0.0;
                        for (int k = 0; k < 6; k++) a += tmp[i * 6 + k] * Hi[k * 6 + j];
                        s.cov[i * 6 + j] = s2 * a;
                    }
            } else
                for (int i = 0; i < 6; i++) s.cov[i * 6 + i] = std::numeric_limits<double>::max();
        } else {
            mat4_identity(To);
            if (worst == PGICP_OK) worst = s.status;
        }
        if (stats) stats[p] = s;
    }
    if (worst != PGICP_OK)
        return fail(c, worst, worst == PGICP_ERR_NO_MATCH ? "ICP: no point to minimize (ConvergenceError)"
                           : worst == PGICP_ERR_BOUND ? "ICP: BoundTransformationChecker: limit exceeded (ConvergenceError)"
                                                      : "ICP: NaN in transformation checker (ConvergenceError)");
    return PGICP_OK;
}

template <typename T>
int icp_pair(pgicp_ctx *c, const T *reading, int rd_stride, int n, const T *ref_xyz, int ref_stride, const T *ref_nrm,
             int nrm_stride, int m, int mem, const double *T_init, double *T_out, pgicp_stats *stats)
{
    int id = -1;
    int st = map_create<T>(c, ref_xyz, ref_stride, ref_nrm, nrm_stride, m, mem, 1, &id);
    if (st) return st;
    pgicp_problem pr;
    std::memset(&pr, 0, sizeof pr);
    pr.map_id = id; pr.reading = reading; pr.stride = rd_stride; pr.n = n; pr.mem = mem;
    std::memcpy(pr.T_init, T_init, sizeof pr.T_init);
    st = align_batch<T>(c, 1, &pr, T_out, stats);
    (void)stream_sync(c);
    if (MapHost<T> *mh = get_map<T>(c, id)) free_map(c, *mh);
    return st;
}

// matcher-only / partial chain on one problem
template <typename T>
int run_partial(pgicp_ctx *c, int map_id, const T *reading, int stride, int n, int mem, const double *Tmove,
                BatchLayout &L, std::vector<ProblemDev> &hp)
{
    MapHost<T> *M = get_map<T>(c, map_id);
    if (!M) return fail(c, PGICP_ERR_ARG, "pgicp: unknown map id");
    pgicp_problem pr;
    std::memset(&pr, 0, sizeof pr);
    pr.map_id = map_id; pr.reading = reading; pr.stride = stride; pr.n = n; pr.mem = mem;
    mat4_identity(pr.T_init);
    if (Tmove) std::memcpy(pr.T_init, Tmove, sizeof pr.T_init);
    // (the matcher alone: no outlier filter runs here, so a SurfaceNormalOutlierFilter's normals are not asked for)
    struct NoNormals { pgicp_ctx *c; double a; ~NoNormals() { c->prm.normal_max_angle = a; } } restore{c, c->prm.normal_max_angle};
    c->prm.normal_max_angle = 0.0;
    int st = batch_begin<T>(c, 1, &pr, [&](int, double *Tpre) {
        double mean[3] = {(double)M->mean[0], (double)M->mean[1], (double)M->mean[2]};
        double Tm_inv[16];
        translation(mean, -1.0, Tm_inv);
        mat4_mul(Tm_inv, pr.T_init, Tpre);
    }, L, hp);
    if (st) return st;
    State<T> &S = state<T>(c);
    const ChainDev<T> ch = chain_of<T>(c, c->prm);
    const MapDev<T> *maps = S.d_maps.template as<MapDev<T>>();
    ProblemDev *probs = c->probs.as<ProblemDev>();
    if (L.knn > 1) {
        ProfScope ps(c, PGICP_PROF_KNN_GRID, n);
        if (launch_knn_topk<T>(c->stream, probs, maps, S.rd_sorted.template as<T>(), S.slot.template as<int>(), S.d2.template as<T>(), ch, 1, n,
                               c->active.as<int>()) != 0)
            return fail(c, PGICP_ERR_ARG, "KDTreeMatcher.knn exceeds PGICP_MAX_KNN");
        return PGICP_OK;
    }
    {
        ProfScope ps(c, c->prm.matcher == PGICP_MATCHER_BRUTE ? PGICP_PROF_KNN_BRUTE : PGICP_PROF_KNN_GRID, n);
        launch_knn<T>(c->stream, c->prm.matcher, probs, maps, S.rd_sorted.template as<T>(), S.slot.template as<int>(),
                      S.d2.template as<T>(), ch, 1, n, 0, c->small.as<int>() + 16, c->slow_list.as<int2>(), c->slow_lb.as<T>(),
                      c->slow_ring.as<int>(), L.rings_unseeded, c->active.as<int>(), S.none_r.template as<T>(), 1, c->queue.p, 1, L.table_kinds);
        c->seg_clean = 0;
        // public matcher output / partial chain: resolve every queued query exactly
        if (c->prm.matcher == PGICP_MATCHER_GRID)
            launch_knn_slow<T>(c->stream, probs, maps, S.rd_sorted.template as<T>(), S.slot.template as<int>(),
                               S.d2.template as<T>(), ch, c->small.as<int>() + 16, c->slow_list.as<int2>(), c->slow_lb.as<T>(),
                               c->slow2.as<int>(), 1, S.none_r.template as<T>());
    }
    return PGICP_OK;
}

template <typename T>
int match(pgicp_ctx *c, int map_id, const T *reading, int stride, int n, int mem, const double *Tm, int32_t *ids, T *dist2)
{
    if (!c || !reading || n <= 0 || !ids || !dist2) return fail(c, PGICP_ERR_ARG, "pgicp_match: bad argument");
    HIPC(c, hipSetDevice(c->device));
    BatchLayout L;
    std::vector<ProblemDev> hp;
    int st = run_partial<T>(c, map_id, reading, stride, n, mem, Tm, L, hp);
    if (st) return st;
    State<T> &S = state<T>(c);
    const size_t np = (size_t)n * L.knn;                     // knn entries per reading point
    HIPC(c, c->tmp_a.ensure(sizeof(int) * np));
    HIPC(c, c->tmp_b.ensure(sizeof(T) * np));
    launch_unpermute<T>(c->stream, S.d_maps.template as<MapDev<T>>(), map_index<T>(c, map_id), c->order.as<int>(),
                        S.slot.template as<int>(), S.d2.template as<T>(), n, L.knn, c->tmp_a.as<int>(), c->tmp_b.as<T>());
    if (mem == PGICP_DEVICE) {
        HIPC(c, hipMemcpyAsync(ids, c->tmp_a.p, sizeof(int) * np, hipMemcpyDeviceToDevice, c->stream));
        HIPC(c, hipMemcpyAsync(dist2, c->tmp_b.p, sizeof(T) * np, hipMemcpyDeviceToDevice, c->stream));
    } else {
        XFER(c, d2h(c, ids, c->tmp_a.p, sizeof(int) * np));
        XFER(c, d2h(c, dist2, c->tmp_b.p, sizeof(T) * np));
    }
    HIPC(c, stream_sync(c));
    HIPC(c, hipGetLastError());
    return PGICP_OK;
}

// transform -> match -> trimmed outlier filter -> error elements for P (map, reading, T) problems in
// one device pass; the same lazily-exact matcher pipeline one ICP iteration uses, without the solve.
template <typename T>
int partial_chain_batch(pgicp_ctx *c, int P, const pgicp_problem *pr, double *ratio, double *residual, int *status)
{
    if (!c || P <= 0 || !pr) return fail(c, PGICP_ERR_ARG, "pgicp_partial_chain: bad argument");
    HIPC(c, hipSetDevice(c->device));
    for (int p = 0; p < P; p++) {
        MapHost<T> *M = get_map<T>(c, pr[p].map_id);
        if (!M) return fail(c, PGICP_ERR_ARG, "pgicp_partial_chain: unknown map id");
        if (!M->has_nrm && needs_ref_normals(c->prm))
            return fail(c, PGICP_ERR_ARG, "pgicp: reference has no normals descriptor");
    }
    BatchLayout L;
    std::vector<ProblemDev> hp;
    int st = batch_begin<T>(c, P, pr, [&](int p, double *Tpre) {
        MapHost<T> *M = get_map<T>(c, pr[p].map_id);
        double mean[3] = {(double)M->mean[0], (double)M->mean[1], (double)M->mean[2]};
        double Tm_inv[16];
        translation(mean, -1.0, Tm_inv);
        mat4_mul(Tm_inv, pr[p].T_init, Tpre);
    }, L, hp, 1);
    if (st) return st;
    const ChainDev<T> ch = chain_of<T>(c, c->prm);
    c->sel_guess_first = hints_cover_first(hp) ? 1 : 0;
    one_iteration<T>(c, L, ch, false, L.total, P, 0);
    c->sel_guess_first = 0;
    launch_sum_partials(c->stream, c->partials.as<double>(), reduce_blocks(L.max_pairs()), kSys, c->probs.as<ProblemDev>(), 0,
                        c->sums.as<double>(), P);
    std::vector<double> sys((size_t)P * kSys);
    XFER(c, d2h(c, sys.data(), c->sums.p, sizeof(double) * sys.size()));
    { const int pst = pinned_ensure(c, &c->h_down, &c->h_down_cap, sizeof(ProblemDev) * (size_t)P); if (pst) return pst; }
    HIPC(c, hipMemcpyAsync(c->h_down, c->probs.p, sizeof(ProblemDev) * P, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, stream_sync(c));
    HIPC(c, hipGetLastError());
    std::memcpy(hp.data(), c->h_down, sizeof(ProblemDev) * (size_t)P);
    hints_store(c, 1, hp);
    int worst = PGICP_OK;
    for (int p = 0; p < P; p++) {
        const double *s = sys.data() + (size_t)p * kSys;
        const bool ok = hp[p].n_finite != 0 && s[28] > 0.0;
        if (status) status[p] = ok ? PGICP_OK : PGICP_ERR_NO_MATCH;
        if (!ok) worst = PGICP_ERR_NO_MATCH;
        if (ratio) ratio[p] = ok ? s[27] / ((double)pr[p].n * L.knn) : 0.0;
        if (residual) residual[p] = ok ? s[29] : 0.0;
    }
    if (worst != PGICP_OK) return fail(c, worst, "no point to minimize (ConvergenceError)");
    return PGICP_OK;
}

template <typename T>
int partial_chain(pgicp_ctx *c, int map_id, const T *reading, int stride, int n, int mem, const double *Tm, double *ratio,
                  double *residual)
{
    if (!c || !reading || n <= 0) return fail(c, PGICP_ERR_ARG, "pgicp_partial_chain: bad argument");
    pgicp_problem pr;
    std::memset(&pr, 0, sizeof pr);
    pr.map_id = map_id; pr.reading = reading; pr.stride = stride; pr.n = n; pr.mem = mem;
    mat4_identity(pr.T_init);
    if (Tm) std::memcpy(pr.T_init, Tm, sizeof pr.T_init);
    return partial_chain_batch<T>(c, 1, &pr, ratio, residual, nullptr);
}

template <typename T>
int outlier_weights(pgicp_ctx *c, const T *dist2, int n, int mem, T *weights, T *limit, int *n_finite)
{
    if (!c || !dist2 || n <= 0) return fail(c, PGICP_ERR_ARG, "pgicp_outlier_weights: bad argument");
    HIPC(c, hipSetDevice(c->device));
    const T *d_d2 = dist2;
    T *d_w = weights;
    if (mem == PGICP_HOST) {
        HIPC(c, c->tmp_a.ensure(sizeof(T) * (size_t)n));
        HIPC(c, c->tmp_b.ensure(sizeof(T) * (size_t)n));
        XFER(c, h2d(c, c->tmp_a.p, dist2, sizeof(T) * (size_t)n));
        d_d2 = c->tmp_a.as<T>();
        d_w = weights ? c->tmp_b.as<T>() : nullptr;
    }
    HIPC(c, c->small.ensure(256));
    // RobustOutlierFilter in the chain: its weights (the median and the median absolute deviation of the finite distances
    // by the same block selection); `limit` comes back +inf -- the filter has none
    const bool robust = c->prm.robust_fct != PGICP_ROBUST_NONE;
    if (robust) HIPC(c, c->robust_dev.ensure(sizeof(T) * (size_t)n));
    {
        ProfScope ps(c, PGICP_PROF_TRIM, n);
        if (robust) launch_robust_raw<T>(c->stream, d_d2, n, make_chain<T>(c->prm).robust, c->robust_dev.as<T>(), c->small.as<T>(), d_w);
        else launch_trim_raw<T>(c->stream, d_d2, n, (T)c->prm.trim_ratio, (T)c->prm.quantile_scale, c->small.as<T>(), d_w);
    }
    T h[2];
    XFER(c, d2h(c, h, c->small.p, sizeof h));
    if (mem == PGICP_HOST && weights)
        XFER(c, d2h(c, weights, d_w, sizeof(T) * (size_t)n));
    HIPC(c, stream_sync(c));
    HIPC(c, hipGetLastError());
    if (limit) *limit = robust ? std::numeric_limits<T>::infinity() : h[0];
    if (n_finite) *n_finite = (int)h[1];
    if ((int)h[1] == 0) return fail(c, PGICP_ERR_NO_MATCH, "no outlier to filter (ConvergenceError)");
    return PGICP_OK;
}

template <typename T>
int error_stats(pgicp_ctx *c, int map_id, const T *reading, int stride, int n, int mem, const int32_t *ids, const T *w,
                double *ratio, double *residual, double *sys_out)
{
    if (!c || !reading || !ids || !w || n <= 0 || stride < 3) return fail(c, PGICP_ERR_ARG, "pgicp_error_stats: bad argument");
    HIPC(c, hipSetDevice(c->device));
    MapHost<T> *M = get_map<T>(c, map_id);
    if (!M || (!M->has_nrm && !is_p2point(c->prm.error_minimizer))) return fail(c, PGICP_ERR_ARG, "pgicp_error_stats: unknown map or no normals");
    State<T> &S = state<T>(c);
    const int K = std::max(1, c->prm.knn);                   // ids / w: knn entries per reading point
    const T *d_rd = reading, *d_w = w;
    const int *d_ids = ids;
    UploadUse use(c);
    if (mem == PGICP_DEVICE) use.touch(reading);
    if (mem == PGICP_HOST) {
        HIPC(c, S.staging.ensure(staged_bytes(sizeof(T), stride, n)));
        int st = to_device<T>(c, reading, stride, n, mem, S.staging, 0, &d_rd);
        if (st) return st;
        HIPC(c, c->tmp_a.ensure(sizeof(int) * (size_t)n * K));
        HIPC(c, c->tmp_b.ensure(sizeof(T) * (size_t)n * K));
        XFER(c, h2d(c, c->tmp_a.p, ids, sizeof(int) * (size_t)n * K));
        XFER(c, h2d(c, c->tmp_b.p, w, sizeof(T) * (size_t)n * K));
        d_ids = c->tmp_a.as<int>();
        d_w = c->tmp_b.as<T>();
    }
    if (!M->slot_of_made) {
        launch_slot_of<T>(c->stream, M->pts, M->first, M->m, M->slot_of);
        M->slot_of_made = true;
    }
    const int nb = (int)(((long long)n * K + kReduceSpan - 1) / kReduceSpan);
    HIPC(c, c->partials.ensure(sizeof(double) * (size_t)nb * kSys));
    HIPC(c, c->sums.ensure(sizeof(double) * kCovTerms));
    {
        ProfScope ps(c, PGICP_PROF_REDUCE, n);
        launch_error_stats<T>(c->stream, S.d_maps.template as<MapDev<T>>(), map_index<T>(c, map_id), M->slot_of, d_rd, stride, d_ids, d_w, n, K,
                              M->mean, c->partials.as<double>(), c->sums.as<double>(), is_p2point(c->prm.error_minimizer) ? 1 : 0);
    }
    double sys[kSys];
    XFER(c, d2h(c, sys, c->sums.p, sizeof sys));
    HIPC(c, stream_sync(c));
    HIPC(c, hipGetLastError());
    if (sys_out) std::memcpy(sys_out, sys, sizeof sys);
    if (!(sys[28] > 0.0)) return fail(c, PGICP_ERR_NO_MATCH, "no point to minimize (ConvergenceError)");
    if (ratio) *ratio = sys[27] / ((double)n * K);
    if (residual) *residual = sys[29];
    return PGICP_OK;
}

bool is_rigid(const double *T)
{
    // RigidTransformation::checkParameters: R^T R ~ I and det(R) ~ +1
    double err = 0.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0.0;
            for (int k = 0; k < 3; k++) s += T[k * 4 + i] * T[k * 4 + j];
            err = std::max(err, std::fabs(s - (i == j ? 1.0 : 0.0)));
        }
    const double det = T[0] * (T[5] * T[10] - T[6] * T[9]) - T[1] * (T[4] * T[10] - T[6] * T[8]) +
                       T[2] * (T[4] * T[9] - T[5] * T[8]);
    return err < 1e-3 && std::fabs(det - 1.0) < 1e-3;
}

template <typename T>
int transform(pgicp_ctx *c, const double *T16, const T *in, int in_stride, T *out, int out_stride, int n, int rotate_only,
              int mem)
{
    if (!c || !T16 || !in || !out || n < 0 || in_stride < 3 || out_stride < 3)
        return fail(c, PGICP_ERR_ARG, "pgicp_transform: bad argument");
    if (!is_rigid(T16)) return fail(c, PGICP_ERR_NOT_RIGID, "pgicp_transform: transformation is not rigid");
    if (n == 0) return PGICP_OK;
    HIPC(c, hipSetDevice(c->device));
    State<T> &S = state<T>(c);
    UploadUse use(c);
    if (mem == PGICP_DEVICE) {
        use.touch(in);
        launch_transform<T>(c->stream, in, in_stride, out, out_stride, n, T16, rotate_only);
        HIPC(c, stream_sync(c));
        return PGICP_OK;
    }
    const size_t bi = staged_bytes(sizeof(T), in_stride, n), bo = staged_bytes(sizeof(T), out_stride, n);
    HIPC(c, S.staging.ensure(bi));
    if (in == out && in_stride == out_stride) {
        // in place: the staged copy is input and output (a thread reads and writes its own point), the caller's other
        // rows travel with it -- one upload instead of two
        const T *d_io = nullptr;
        int st0 = to_device<T>(c, in, in_stride, n, mem, S.staging, 0, &d_io);
        if (st0) return st0;
        launch_transform<T>(c->stream, d_io, in_stride, const_cast<T *>(d_io), out_stride, n, T16, rotate_only);
        XFER(c, d2h(c, out, d_io, sizeof(T) * ((size_t)(n - 1) * out_stride + 3)));
        HIPC(c, stream_sync(c));
        HIPC(c, hipGetLastError());
        return PGICP_OK;
    }
    HIPC(c, S.stage_aux.ensure(bo));
    const T *d_in = nullptr;
    int st = to_device<T>(c, in, in_stride, n, mem, S.staging, 0, &d_in);
    if (st) return st;
    // keep the caller's other rows (e.g. the homogeneous 1) when out is strided
    XFER(c, h2d(c, S.stage_aux.p, out, sizeof(T) * ((size_t)(n - 1) * out_stride + 3)));
    launch_transform<T>(c->stream, d_in, in_stride, S.stage_aux.template as<T>(), out_stride, n, T16, rotate_only);
    XFER(c, d2h(c, out, S.stage_aux.p, sizeof(T) * ((size_t)(n - 1) * out_stride + 3)));
    HIPC(c, stream_sync(c));
    HIPC(c, hipGetLastError());
    return PGICP_OK;
}

template <typename T>
int build_local_map(pgicp_ctx *c, int n_kf, const T *const *xyz, const T *const *nrm, const int *sx, const int *sn,
                    const int *counts, const double *T_ref_kf, T *out_xyz, int out_stride, T *out_nrm, int out_nstride, int mem)
{
    if (!c || n_kf <= 0 || !xyz || !counts || !T_ref_kf || !out_xyz || out_stride < 3 || (out_nrm && out_nstride < 3))
        return fail(c, PGICP_ERR_ARG, "pgicp_build_local_map: bad argument");
    HIPC(c, hipSetDevice(c->device));
    State<T> &S = state<T>(c);
    UploadUse use(c);
    long long total = 0;
    size_t stage = 0;
    for (int k = 0; k < n_kf; k++) {
        if (counts[k] <= 0 || sx[k] < 3) return fail(c, PGICP_ERR_ARG, "pgicp_build_local_map: bad keyframe");
        total += counts[k];
        if (mem == PGICP_HOST) stage += staged_bytes(sizeof(T), sx[k], counts[k]) + (nrm ? staged_bytes(sizeof(T), sn[k], counts[k]) : 0);
    }
    T *d_ox = out_xyz, *d_on = out_nrm;
    const size_t ob = sizeof(T) * ((size_t)(total - 1) * out_stride + 3);
    const size_t onb = out_nrm ? sizeof(T) * ((size_t)(total - 1) * out_nstride + 3) : 0;
    if (mem == PGICP_HOST) {
        HIPC(c, S.staging.ensure(stage));
        HIPC(c, S.stage_aux.ensure(((ob + 255) & ~(size_t)255) + onb + 256));
        d_ox = S.stage_aux.template as<T>();
        d_on = out_nrm ? (T *)((char *)S.stage_aux.p + ((ob + 255) & ~(size_t)255)) : nullptr;
        XFER(c, h2d(c, d_ox, out_xyz, ob));
        if (out_nrm) XFER(c, h2d(c, d_on, out_nrm, onb));
    }
    size_t soff = 0;
    long long off = 0;
    double I[16];
    mat4_identity(I);
    for (int k = 0; k < n_kf; k++) {
        const double *Tk = k == 0 ? I : T_ref_kf + 16 * k;     // LocalMap.hpp:214: the reference cloud is copied as is
        if (k > 0 && !is_rigid(Tk)) return fail(c, PGICP_ERR_NOT_RIGID, "pgicp_build_local_map: transformation is not rigid");
        const T *d_x = nullptr, *d_n = nullptr;
        int st = to_device<T>(c, xyz[k], sx[k], counts[k], mem, S.staging, soff, &d_x);
        if (st) return st;
        if (mem == PGICP_HOST) soff += staged_bytes(sizeof(T), sx[k], counts[k]);
        launch_transform<T>(c->stream, d_x, sx[k], d_ox + off * out_stride, out_stride, counts[k], Tk, 0);
        if (nrm && out_nrm) {
            st = to_device<T>(c, nrm[k], sn[k], counts[k], mem, S.staging, soff, &d_n);
            if (st) return st;
            if (mem == PGICP_HOST) soff += staged_bytes(sizeof(T), sn[k], counts[k]);
            launch_transform<T>(c->stream, d_n, sn[k], d_on + off * out_nstride, out_nstride, counts[k], Tk, 1);
        }
        off += counts[k];
    }
    if (mem == PGICP_HOST) {
        XFER(c, d2h(c, out_xyz, d_ox, ob));
        if (out_nrm) XFER(c, d2h(c, out_nrm, d_on, onb));
    }
    HIPC(c, stream_sync(c));
    HIPC(c, hipGetLastError());
    return PGICP_OK;
}

template <typename T>
int surface_normals(pgicp_ctx *c, const T *xyz, int stride, int n, int mem, int knn, double max_dist, T *out_nrm,
                    int out_stride, T *out_eig, int32_t *out_ids, T *out_d2)
{
    if (!c || !xyz || n <= 0 || stride < 3 || !out_nrm || out_stride < 3 || knn < 1 || knn > 32 || !(max_dist > 0))
        return fail(c, PGICP_ERR_ARG, "pgicp_surface_normals: bad argument (1 <= knn <= 32)");
    HIPC(c, hipSetDevice(c->device));
    int id = -1;
    int st = map_create<T>(c, xyz, stride, nullptr, 0, n, mem, 0, &id);       // uncentred: coordinates stay exact
    if (st) return st;
    State<T> &S = state<T>(c);
    T *d_nrm = out_nrm, *d_eig = out_eig, *d_d2 = out_d2;
    int32_t *d_ids = out_ids;
    const size_t b_nrm = sizeof(T) * ((size_t)(n - 1) * out_stride + 3), b_eig = sizeof(T) * 3 * (size_t)n,
                 b_ids = sizeof(int32_t) * (size_t)knn * n, b_d2 = sizeof(T) * (size_t)knn * n;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    if (mem == PGICP_HOST) {
        HIPC(c, S.stage_aux.ensure(up(b_nrm) + up(b_eig) + up(b_ids) + up(b_d2) + 256));
        char *base = (char *)S.stage_aux.p;
        d_nrm = (T *)base;
        d_eig = out_eig ? (T *)(base + up(b_nrm)) : nullptr;
        d_ids = out_ids ? (int32_t *)(base + up(b_nrm) + up(b_eig)) : nullptr;
        d_d2 = out_d2 ? (T *)(base + up(b_nrm) + up(b_eig) + up(b_ids)) : nullptr;
        if (out_stride > 3) XFER(c, h2d(c, d_nrm, out_nrm, b_nrm));   // keep the caller's padding
    }
    {
        ProfScope ps(c, PGICP_PROF_NORMALS, n);
        if (launch_surface_normals<T>(c->stream, S.d_maps.template as<MapDev<T>>(), map_index<T>(c, id), n, knn, (T)max_dist,
                                      std::numeric_limits<T>::epsilon(), d_nrm, out_stride, d_eig, d_ids, d_d2) != 0)
            return fail(c, PGICP_ERR_ARG, "pgicp_surface_normals: knn > 32");
    }
    if (mem == PGICP_HOST) {
        XFER(c, d2h(c, out_nrm, d_nrm, b_nrm));
        if (out_eig) XFER(c, d2h(c, out_eig, d_eig, b_eig));
        if (out_ids) XFER(c, d2h(c, out_ids, d_ids, b_ids));
        if (out_d2) XFER(c, d2h(c, out_d2, d_d2, b_d2));
    }
    HIPC(c, stream_sync(c));
    HIPC(c, hipGetLastError());
    if (MapHost<T> *mh = get_map<T>(c, id)) free_map(c, *mh);
    return PGICP_OK;
}

// the filter list of pgicp_filter_cloud*, checked, as the launcher takes it
template <typename T>
int filter_specs(pgicp_ctx *c, int nf, const pgicp_filter *f, int *types, double *params)
{
    for (int k = 0; k < nf; k++) {
        if (f[k].type < PGICP_FILTER_IDENTITY || f[k].type > PGICP_FILTER_MAX_POINT_COUNT) return fail(c, PGICP_ERR_ARG, "pgicp_filter_cloud: unknown filter type");
        if (f[k].type == PGICP_FILTER_FIX_STEP && !(f[k].p[0] >= 1.0 && f[k].p[0] <= 2147483647.0))
            return fail(c, PGICP_ERR_ARG, "pgicp_filter_cloud: FixStep needs 1 <= step <= INT_MAX");
        if ((f[k].type == PGICP_FILTER_MAX_DIST || f[k].type == PGICP_FILTER_MIN_DIST) &&
            !(f[k].p[1] == 0.0 || f[k].p[1] == 1.0 || f[k].p[1] == 2.0 || f[k].p[1] == 3.0))
            return fail(c, PGICP_ERR_ARG, "pgicp_filter_cloud: Min/MaxDist p[1] must be dim + 1 in {0 (radius), 1, 2, 3}");
        if (f[k].type == PGICP_FILTER_RANDOM_SAMPLING && !(f[k].p[0] >= 0.0 && f[k].p[0] <= 1.0))
            return fail(c, PGICP_ERR_ARG, "pgicp_filter_cloud: RandomSampling needs 0 <= prob <= 1");
        if ((f[k].type == PGICP_FILTER_RANDOM_SAMPLING || f[k].type == PGICP_FILTER_MAX_POINT_COUNT) &&
            !(f[k].p[1] >= 0.0 && f[k].p[1] < 9007199254740992.0 && f[k].p[1] == std::floor(f[k].p[1])))
            return fail(c, PGICP_ERR_ARG, "pgicp_filter_cloud: the sampler's seed must be an integer in [0, 2^53)");
        if (f[k].type == PGICP_FILTER_MAX_POINT_COUNT && !(f[k].p[0] >= 1.0 && f[k].p[0] <= 2147483647.0))
            return fail(c, PGICP_ERR_ARG, "pgicp_filter_cloud: MaxPointCount needs 1 <= maxCount <= INT_MAX");
        types[k] = f[k].type;
        std::memcpy(params + 8 * k, f[k].p, sizeof f[k].p);
        if (f[k].type == PGICP_FILTER_MAX_POINT_COUNT) params[8 * k + 2] = sizeof(T) == 4 ? 1.0 : 0.0;   // prob = T(maxCount) / T(N)
    }
    return PGICP_OK;
}

template <typename T>
int filter_cloud(pgicp_ctx *c, int nf, const pgicp_filter *f, const T *feat, int frows, const T *desc, int drows, int n, const double *T16,
                 int rot0, int rot1, T *out_feat, T *out_desc, int32_t *kept_idx, int *n_out, const T **dev_feat)
{
    if (!c || nf < 0 || nf > PGICP_MAX_FILTERS || (nf && !f) || !feat || frows < 3 || n <= 0 || (desc && drows <= 0) || !out_feat || !n_out ||
        (desc && !out_desc) || (rot0 >= 0 && (!desc || rot0 + 3 > drows)) || (rot1 >= 0 && (!desc || rot1 + 3 > drows)))
        return fail(c, PGICP_ERR_ARG, "pgicp_filter_cloud: bad argument");
    if (T16 && !is_rigid(T16)) return fail(c, PGICP_ERR_NOT_RIGID, "pgicp_filter_cloud: transformation is not rigid");
    int types[PGICP_MAX_FILTERS];
    double params[8 * PGICP_MAX_FILTERS];
    { const int st = filter_specs<T>(c, nf, f, types, params); if (st) return st; }
    HIPC(c, hipSetDevice(c->device));
    pgicp_ctx::FilterSet &S = c->fset[c->fset_next];
    c->fset_next = (c->fset_next + 1) & 3;
    // Without a transformation (none, or the identity -- a sensor at the robot's origin) the kept points are the input's
    // own: only the features travel (the predicates look at nothing else, and the ICP wants them on the device anyway), and
    // what comes back is the list of kept indices, if anything was dropped at all -- the host copies compact themselves.
    bool ident = T16 == nullptr;
    if (T16) {
        ident = true;
        for (int i = 0; i < 12; i++) ident = ident && T16[i] == (i % 5 == 0 ? 1.0 : 0.0);
    }
    const bool dev_desc = desc && !ident;
    const size_t bf = sizeof(T) * (size_t)frows * n, bd = dev_desc ? sizeof(T) * (size_t)drows * n : 0;
    HIPC(c, S.in_f.ensure(bf)); HIPC(c, S.out_f.ensure(bf));
    if (dev_desc) { HIPC(c, S.in_d.ensure(bd)); HIPC(c, S.out_d.ensure(bd)); }
    HIPC(c, S.keep.ensure(sizeof(int) * ((size_t)n + 1))); HIPC(c, S.pos.ensure(sizeof(int) * ((size_t)n + 1)));
    HIPC(c, S.bsum.ensure(sizeof(int) * ((size_t)n / kScanChunkHost + 4)));
    const bool want_idx = kept_idx || ident;
    if (want_idx) HIPC(c, S.idx.ensure(sizeof(int) * (size_t)n));
    XFER(c, h2d(c, S.in_f.p, feat, bf));
    if (dev_desc) XFER(c, h2d(c, S.in_d.p, desc, bd));
    // (without a transformation and without a caller's kept_idx the host side only has to close the gaps of the dropped points: a
    // short ascending list of them comes back with the count -- the kept indices, 4 bytes a point, are fetched only when that
    // list would not hold them)
    constexpr int kDropCap = 4096, kDropHead = 256;
    const bool want_drop = ident && !kept_idx;
    if (want_drop) HIPC(c, S.drop.ensure(sizeof(int) * kDropCap));
    launch_filter_cloud<T>(c->stream, S.in_f.as<T>(), frows, frows, dev_desc ? S.in_d.as<T>() : nullptr, drows, n, nf, types, params,
                           ident ? nullptr : T16, rot0, rot1, S.keep.as<int>(), S.pos.as<int>(), S.bsum.as<int>(), S.out_f.as<T>(),
                           dev_desc ? S.out_d.as<T>() : nullptr, want_idx ? S.idx.as<int>() : nullptr, want_drop ? S.drop.as<int>() : nullptr,
                           want_drop ? std::min(kDropCap, n) : 0);
    int kept = 0;
    int drop_list[kDropCap];
    XFER(c, d2h(c, &kept, S.pos.as<int>() + n, sizeof(int)));
    if (want_drop) XFER(c, d2h(c, drop_list, S.drop.p, sizeof(int) * (size_t)std::min(kDropHead, n)));
    HIPC(c, stream_sync(c));
    HIPC(c, hipGetLastError());
    *n_out = kept;
    if (ident && want_drop && kept < n && n - kept <= std::min(kDropCap, n)) {
        const int nd = n - kept;
        if (nd > kDropHead) { XFER(c, d2h(c, drop_list + kDropHead, S.drop.as<int>() + kDropHead, sizeof(int) * (size_t)(nd - kDropHead))); HIPC(c, stream_sync(c)); }
        // the kept runs between consecutive dropped points, moved as blocks (out may alias the input: towards the front only)
        if (out_feat != feat) std::memcpy(out_feat, feat, sizeof(T) * (size_t)frows * (size_t)drop_list[0]);
        if (desc && out_desc != desc) std::memcpy(out_desc, desc, sizeof(T) * (size_t)drows * (size_t)drop_list[0]);
        int dst = drop_list[0];
        for (int k = 0; k < nd; k++) {
            const int from = drop_list[k] + 1, to = k + 1 < nd ? drop_list[k + 1] : n;
            if (to > from) {
                std::memmove(out_feat + (size_t)dst * frows, feat + (size_t)from * frows, sizeof(T) * (size_t)frows * (size_t)(to - from));
                if (desc) std::memmove(out_desc + (size_t)dst * drows, desc + (size_t)from * drows, sizeof(T) * (size_t)drows * (size_t)(to - from));
                dst += to - from;
            }
        }
    } else if (ident) {
        std::vector<int> hidx;
        const int *idx = nullptr;
        if (kept < n || kept_idx) {
            if (!want_idx) return fail(c, PGICP_ERR_HIP, "pgicp_filter_cloud: internal (kept indices not made)");
            int *dst = kept_idx;
            if (!dst) { hidx.resize((size_t)std::max(kept, 1)); dst = hidx.data(); }
            if (kept > 0) { XFER(c, d2h(c, dst, S.idx.p, sizeof(int) * (size_t)kept)); HIPC(c, stream_sync(c)); }
            idx = dst;
        }
        if (kept == n) {
            if (out_feat != feat) std::memcpy(out_feat, feat, bf);
            if (desc && out_desc != desc) std::memcpy(out_desc, desc, sizeof(T) * (size_t)drows * n);
        } else {
            // (ascending indices: compaction in place moves every point towards the front)
            for (int k = 0; k < kept; k++) {
                std::memmove(out_feat + (size_t)k * frows, feat + (size_t)idx[k] * frows, sizeof(T) * (size_t)frows);
                if (desc) std::memmove(out_desc + (size_t)k * drows, desc + (size_t)idx[k] * drows, sizeof(T) * (size_t)drows);
            }
        }
    } else if (kept > 0) {
        XFER(c, d2h(c, out_feat, S.out_f.p, sizeof(T) * (size_t)frows * kept));
        if (desc) XFER(c, d2h(c, out_desc, S.out_d.p, sizeof(T) * (size_t)drows * kept));
        if (kept_idx) XFER(c, d2h(c, kept_idx, S.idx.p, sizeof(int) * (size_t)kept));
        HIPC(c, stream_sync(c));
    }
    if (dev_feat) *dev_feat = S.out_f.as<T>();
    return PGICP_OK;
}

// pgicp_filter_cloud_dev: the same device pass for a caller that keeps the host side to itself -- no transformation, the
// caller's arrays are not written: the kept count, the device copy of the filtered features, and the ASCENDING indices of the
// dropped points (a range sensor's input filters drop a handful of a scan's points: the caller closes those gaps in its own
// arrays -- while the ICP already runs on the device copy, if it likes)
template <typename T>
int filter_cloud_dev(pgicp_ctx *c, int nf, const pgicp_filter *f, const T *feat, int frows, int n, int32_t *dropped_idx, int dropped_cap,
                     int *n_dropped, int *n_out, const T **dev_feat)
{
    if (!c || nf < 0 || nf > PGICP_MAX_FILTERS || (nf && !f) || !feat || frows < 3 || n <= 0 || !n_out || !n_dropped || dropped_cap < 0 ||
        (dropped_cap && !dropped_idx))
        return fail(c, PGICP_ERR_ARG, "pgicp_filter_cloud_dev: bad argument");
    int types[PGICP_MAX_FILTERS];
    double params[8 * PGICP_MAX_FILTERS];
    { const int st = filter_specs<T>(c, nf, f, types, params); if (st) return st; }
    HIPC(c, hipSetDevice(c->device));
    pgicp_ctx::FilterSet &S = c->fset[c->fset_next];
    c->fset_next = (c->fset_next + 1) & 3;
    const size_t bf = sizeof(T) * (size_t)frows * n;
    const int cap = std::min(dropped_cap, n);
    HIPC(c, S.in_f.ensure(bf)); HIPC(c, S.out_f.ensure(bf));
    HIPC(c, S.keep.ensure(sizeof(int) * ((size_t)n + 1))); HIPC(c, S.pos.ensure(sizeof(int) * ((size_t)n + 1)));
    HIPC(c, S.bsum.ensure(sizeof(int) * ((size_t)n / kScanChunkHost + 4)));
    HIPC(c, S.drop.ensure(sizeof(int) * (size_t)std::max(cap, 1)));
    XFER(c, h2d(c, S.in_f.p, feat, bf));
    launch_filter_cloud<T>(c->stream, S.in_f.as<T>(), frows, frows, nullptr, 0, n, nf, types, params, nullptr, -1, -1, S.keep.as<int>(), S.pos.as<int>(),
                           S.bsum.as<int>(), S.out_f.as<T>(), nullptr, nullptr, cap ? S.drop.as<int>() : nullptr, cap);
    // the kept count and the head of the dropped list in ONE round trip (a second one only when more than that were dropped)
    int kept = 0;
    const int head = std::min(cap, 256);
    XFER(c, d2h(c, &kept, S.pos.as<int>() + n, sizeof(int)));
    if (head) XFER(c, d2h(c, dropped_idx, S.drop.p, sizeof(int) * (size_t)head));
    HIPC(c, stream_sync(c));
    HIPC(c, hipGetLastError());
    const int nd = n - kept;
    if (nd > head && nd <= cap) { XFER(c, d2h(c, dropped_idx + head, S.drop.as<int>() + head, sizeof(int) * (size_t)(nd - head))); HIPC(c, stream_sync(c)); }
    *n_out = kept;
    *n_dropped = nd;                          // (more than dropped_cap: the list is incomplete -- the caller takes pgicp_filter_cloud)
    if (dev_feat) *dev_feat = S.out_f.as<T>();
    return PGICP_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// extern "C" surface
// ---------------------------------------------------------------------------
template <typename T>
int map_create_batch_abi(pgicp_ctx *c, int n, const T *const *xyz, const int *xs, const T *const *nrm, const int *ns,
                                const int *m, int mem, int center, int *ids)
{
    if (!c || n <= 0 || !xyz || !xs || !m || !ids || (nrm && !ns)) return fail(c, PGICP_ERR_ARG, "pgicp_map_create_batch: bad argument");
    if (n > 65535) return fail(c, PGICP_ERR_ARG, "pgicp_map_create_batch: at most 65535 clouds per call");
    std::vector<MapSrc<T>> src(n);
    for (int k = 0; k < n; k++) src[k] = MapSrc<T>{xyz[k], xs[k], nrm ? nrm[k] : nullptr, nrm ? ns[k] : 0, m[k]};
    return map_create_batch<T>(c, n, src.data(), mem, center, ids);
}
template <typename T>
int debug_last_matches(pgicp_ctx *c, int problem, int32_t *ids, T *dist2)
{
    if (!c || problem < 0 || !ids || !dist2) return PGICP_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    State<T> &S = state<T>(c);
    ProblemDev D;
    HIPC(c, hipMemcpy(&D, c->probs.as<ProblemDev>() + problem, sizeof D, hipMemcpyDeviceToHost));
    const size_t np = (size_t)D.n * D.knn;                   // (knn entries per point)
    HIPC(c, c->tmp_a.ensure(sizeof(int) * np));
    HIPC(c, c->tmp_b.ensure(sizeof(T) * np));
    // `order` holds positions in the batch-wide sorted arrays: point the kernel at this problem's slice
    launch_unpermute<T>(c->stream, S.d_maps.template as<MapDev<T>>(), D.map, c->order.as<int>() + D.off,
                        S.slot.template as<int>() + D.off * D.knn, S.d2.template as<T>() + D.off * D.knn, D.n, D.knn, c->tmp_a.as<int>(), c->tmp_b.as<T>());
    XFER(c, d2h(c, ids, c->tmp_a.p, sizeof(int) * np));
    XFER(c, d2h(c, dist2, c->tmp_b.p, sizeof(T) * np));
    HIPC(c, stream_sync(c));
    return PGICP_OK;
}

template <typename T>
int map_transfer_impl(pgicp_ctx *from, int id, pgicp_ctx *to, int *new_id)
{
    MapHost<T> *src = get_map<T>(from, id);
    if (!src) return -1;
    hipEvent_t ev;
    HIPC(to, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPC(to, hipEventRecord(ev, from->stream));
    HIPC(to, hipStreamWaitEvent(to->stream, ev, 0));
    HIPC(to, hipEventDestroy(ev));                 // released once the recorded work has completed
    State<T> &S = state<T>(to);
    int slot = -1;
    for (size_t i = 0; i < S.maps.size(); i++) if (!S.maps[i].used) { slot = (int)i; break; }
    if (slot < 0) { S.maps.push_back(MapHost<T>()); slot = (int)S.maps.size() - 1; }
    S.maps[slot] = *src;
    *src = MapHost<T>();                           // ownership of the device block moves with the record
    *new_id = slot | id_tag<T>();
    return sync_maps_table<T>(to);
}

namespace {
template <typename T>
int upload(pgicp_ctx *c, int n, const T *const *host, const int *stride, const int *npts, int mem, const T **dev_ptrs)
{
    if (!c || n <= 0 || !host || !stride || !npts || !dev_ptrs || (mem != PGICP_HOST && mem != PGICP_HOST_PINNED))
        return fail(c, PGICP_ERR_ARG, "pgicp_upload: bad argument");
    size_t total = 0;
    for (int k = 0; k < n; k++) {
        if (!host[k] || npts[k] <= 0 || stride[k] < 3) return fail(c, PGICP_ERR_ARG, "pgicp_upload: bad reading " + std::to_string(k));
        total += staged_bytes(sizeof(T), stride[k], npts[k]);
    }
    HIPC(c, hipSetDevice(c->device));
    const int set = c->up_next;                    // flipped only when the upload has been queued completely
    pgicp_ctx::UploadSet &U = c->up[set];
    std::vector<const T *> ptrs((size_t)n);
    // the set is overwritten: the calls that read its last contents must have consumed them (device-side wait), and a
    // transfer out of its pinned staging must have left it (host-side wait: two uploads ago, over long since)
    if (U.has_consumer) HIPC(c, hipStreamWaitEvent(c->copy_stream, U.consumed, 0));
    if (U.pending) HIPC(c, hipEventSynchronize(U.uploaded));
    if (total > U.dev.cap) {
        // growing means freeing (hipFree waits for the device): nothing may still read the old block
        HIPC(c, stream_sync(c));
        HIPC(c, hipStreamSynchronize(c->copy_stream));
        HIPC(c, U.dev.ensure(total));
    }
    if (mem == PGICP_HOST && total > U.pin_cap) {
        if (U.pin) HIPC(c, t_host_free(U.pin));
        U.pin = nullptr; U.pin_cap = 0;
        HIPC(c, t_host_malloc(&U.pin, total + total / 4, hipHostMallocDefault));
        U.pin_cap = total + total / 4;
    }
    size_t off = 0;
    struct Piece { char *dst; const char *src; size_t len; };
    std::vector<Piece> pieces;                     // pageable sources: what goes into the pinned staging buffer
    for (int k = 0; k < n;) {
        const size_t bytes = sizeof(T) * ((size_t)(npts[k] - 1) * stride[k] + 3), slot = staged_bytes(sizeof(T), stride[k], npts[k]);
        // pinned sources of equal size at equal spacing (scans carved out of one pinned block) go as ONE 2-D transfer:
        // a copy command per scan costs the host 10-30 us each, which a batch of 128 scans does not hide
        int run = 1;
        if (mem == PGICP_HOST_PINNED && k + 1 < n && npts[k + 1] == npts[k] && stride[k + 1] == stride[k]) {
            const ptrdiff_t pitch = (const char *)host[k + 1] - (const char *)host[k];
            if (pitch >= (ptrdiff_t)bytes) {
                run = 2;
                while (k + run < n && npts[k + run] == npts[k] && stride[k + run] == stride[k] &&
                       (const char *)host[k + run] - (const char *)host[k + run - 1] == pitch)
                    ++run;
                HIPC(c, hipMemcpy2DAsync((char *)U.dev.p + off, slot, host[k], (size_t)pitch, bytes, (size_t)run, hipMemcpyHostToDevice, c->copy_stream));
            } else
                run = 1;
        }
        if (run == 1) {
            if (mem == PGICP_HOST) {
                const size_t piece = (size_t)1 << 20;
                for (size_t b = 0; b < bytes; b += piece)
                    pieces.push_back({(char *)U.pin + off + b, (const char *)host[k] + b, std::min(piece, bytes - b)});
            } else HIPC(c, hipMemcpyAsync((char *)U.dev.p + off, host[k], bytes, hipMemcpyHostToDevice, c->copy_stream));
        }
        for (int j = 0; j < run; j++) { ptrs[k + j] = (const T *)((char *)U.dev.p + off); off += slot; }
        k += run;
    }
    if (mem == PGICP_HOST) {
        // one core copies 25-30 GB/s, and the caller's thread is the one that drives the ICP iterations: a large batch is
        // staged by a few threads side by side (the sources are the caller's again when this call returns)
        const unsigned hw = std::thread::hardware_concurrency();
        const int workers = total < ((size_t)8 << 20) ? 1 : (int)std::min<size_t>(std::min<unsigned>(hw ? hw : 1, 8u), pieces.size());
        auto stage = [&pieces, workers](int w) {
            for (size_t i = (size_t)w; i < pieces.size(); i += (size_t)workers) std::memcpy(pieces[i].dst, pieces[i].src, pieces[i].len);
        };
        std::vector<std::thread> pool;
        for (int w = 1; w < workers; w++) pool.emplace_back(stage, w);
        stage(0);
        for (auto &t : pool) t.join();
        HIPC(c, hipMemcpyAsync(U.dev.p, U.pin, total, hipMemcpyHostToDevice, c->copy_stream));
    }
    HIPC(c, hipEventRecord(U.uploaded, c->copy_stream));
    U.bytes = total;
    U.pending = true;
    U.has_consumer = false;
    c->up_next = set ^ 1;
    for (int k = 0; k < n; k++) dev_ptrs[k] = ptrs[k];
    return PGICP_OK;
}
}  // namespace


extern "C" {

int pgicp_abi_version(void) { return PGICP_ABI_VERSION; }

int pgicp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void pgicp_default_params(pgicp_params *p)
{
    if (!p) return;
    // libpointmatcher defaults for the chain pgslam instantiates
    p->knn = 1;
    p->epsilon = 0.0;
    p->max_dist = std::numeric_limits<double>::infinity();
    p->trim_ratio = 0.85;
    p->quantile_scale = 1.0;
    p->outlier_max_dist = 0.0;
    p->max_iters = 40;
    p->min_diff_rot = 0.001;
    p->min_diff_trans = 0.001;
    p->smooth_length = 3;
    p->sensor_std_dev = 0.01;
    p->matcher = PGICP_MATCHER_GRID;
    p->grid_cell = 0.0;
    p->check_every = 1;
    p->error_minimizer = PGICP_MINIMIZER_POINT_TO_PLANE;
    p->robust_fct = PGICP_ROBUST_NONE; p->robust_tuning = 1.0; p->robust_scale = PGICP_ROBUST_SCALE_MAD; p->robust_approx = 0.0;
    p->sum_order = PGICP_SUM_ORDER_SORTED;
    p->bound_max_rot = 0.0;
    p->bound_max_trans = 0.0;
    p->normal_max_angle = 0.0;
}

static int ctx_create_impl(int device, int high_priority, pgicp_ctx **out);
int pgicp_ctx_create(int device, pgicp_ctx **out) { return ctx_create_impl(device, 0, out); }
int pgicp_ctx_create_priority(int device, int high_priority, pgicp_ctx **out) { return ctx_create_impl(device, high_priority, out); }
static int ctx_create_impl(int device, int high_priority, pgicp_ctx **out)
{
    if (!out) return PGICP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return PGICP_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return PGICP_ERR_NO_DEVICE;
    pgicp_ctx *c = new pgicp_ctx();
    c->device = device;
    pgicp_default_params(&c->prm);
    if (const char *e = std::getenv("PGICP_FAST_RINGS_SEEDED")) c->fast_rings_seeded = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("PGICP_KX")) c->grid_kx = std::atoi(e);
    if (const char *e = std::getenv("PGICP_TABLES")) c->table_mode = !std::strcmp(e, "dense") ? 1 : !std::strcmp(e, "succinct") ? 2 : 0;
    if (const char *e = std::getenv("PGICP_CELL_SCALE")) { const double v = std::atof(e); if (v > 0.05 && v < 20.0) c->cell_scale = v; }
    if (const char *e = std::getenv("PGICP_NEAR_FRAC")) c->near_frac = std::atof(e);
    if (const char *e = std::getenv("PGICP_BIN_SHIFT_ADD")) c->bin_shift_add = std::atoi(e);
    if (const char *e = std::getenv("PGICP_MED_RINGS")) c->med_rings = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("PGICP_POLL_US")) c->poll_us = std::atoi(e);
    if (const char *e = std::getenv("PGICP_SEL_HINTS")) c->sel_hints_on = std::atoi(e);
    if (const char *e = std::getenv("PGICP_FAST_RINGS_UNSEEDED")) c->fast_rings_unseeded = std::max(1, std::atoi(e));
    // a high-priority context: its (short) launches are scheduled ahead of the other contexts' queued work -- the MT facade's
    // input stage next to the localizer's ICP
    int prio_least = 0, prio_greatest = 0;
    if (high_priority) (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    bool ok = (high_priority ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest) : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) == hipSuccess &&
              hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) == hipSuccess &&
              t_host_malloc((void **)&c->h_pinned, 64 * sizeof(int), hipHostMallocDefault) == hipSuccess;
    for (int s = 0; ok && s < 2; s++)
        ok = hipEventCreateWithFlags(&c->up[s].uploaded, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->up[s].consumed, hipEventDisableTiming) == hipSuccess;
    { std::lock_guard<std::mutex> lock(pool_of(device).m); pool_of(device).contexts++; }
    if (!ok) {
        pgicp_ctx_destroy(c);
        return PGICP_ERR_HIP;
    }
    // the polled iteration flag wants fine-grained (coherent) pinned memory; plain pinned memory also works with the
    // stream wait the poll falls back to, so a runtime that refuses the flags is not fatal
    if (t_host_malloc((void **)&c->h_flag, 64 * sizeof(int), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        c->poll_us = 0;
        if (t_host_malloc((void **)&c->h_flag, 64 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
            c->h_flag = nullptr;
            pgicp_ctx_destroy(c);               // (streams, events, pinned memory and the device pool's context count)
            return PGICP_ERR_HIP;
        }
    }
    c->h_flag[0] = 0; c->h_flag[1] = 0;
    if (t_malloc((void **)&c->stamp_dev, 256) != hipSuccess || hipMemset(c->stamp_dev, 0, 256) != hipSuccess) {
        pgicp_ctx_destroy(c);
        return PGICP_ERR_HIP;
    }
    if (const char *e = std::getenv("PGICP_GRAPH_MAX_P")) c->graph_max_problems = std::atoi(e);
    if (const char *e = std::getenv("PGICP_PROFILE_ALL")) c->prof_on = std::atoi(e) != 0;      // see pgicp_profile_process
    { ProcessProfile &pp = process_profile(); std::lock_guard<std::mutex> lock(pp.m); pp.live.push_back(c); }
    *out = c;
    return PGICP_OK;
}

void pgicp_ctx_destroy(pgicp_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->bounce.outs.clear();         // (every successful call has drained its own; whatever is left belongs to nobody)
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    prof_collect(c);
    {
        ProcessProfile &pp = process_profile();
        std::lock_guard<std::mutex> lock(pp.m);
        for (size_t i = 0; i < pp.live.size(); i++) if (pp.live[i] == c) { pp.live.erase(pp.live.begin() + i); break; }
        for (int k = 0; k < PGICP_PROF_COUNT; k++) {
            pp.launches[k] += c->prof_launches[k]; pp.ms[k] += c->prof_ms[k]; pp.units[k] += c->prof_units[k];
            pp.problems[k] += c->prof_problems[k]; pp.map_points[k] += c->prof_map_points[k];
        }
    }
    for (int s = 0; s < 2; s++) {
        c->up[s].dev.release();
        if (c->up[s].pin) (void)t_host_free(c->up[s].pin);
        if (c->up[s].uploaded) (void)hipEventDestroy(c->up[s].uploaded);
        if (c->up[s].consumed) (void)hipEventDestroy(c->up[s].consumed);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto &m : c->f32.maps) free_map<float>(nullptr, m);
    for (auto &m : c->f64.maps) free_map<double>(nullptr, m);
    {
        DevicePool &dp = pool_of(c->device);
        std::lock_guard<std::mutex> lock(dp.m);
        if (--dp.contexts <= 0) {                      // the device's last context: give the pooled blocks back
            for (auto &kv : dp.blocks) { if (kv.second.released) (void)hipEventDestroy(kv.second.released); (void)t_free(kv.second.p); }
            dp.blocks.clear();
            dp.bytes = 0;
            dp.contexts = 0;
        }
    }
    for (DevBuf *b : {&c->f32.d_maps, &c->f32.rd_pre, &c->f32.slot, &c->f32.d2, &c->f32.staging, &c->f32.stage_aux,
                      &c->f64.d_maps, &c->f64.rd_pre, &c->f64.slot, &c->f64.d2, &c->f64.staging, &c->f64.stage_aux,
                      &c->probs, &c->src, &c->partials, &c->sums, &c->sums2, &c->small, &c->stats, &c->bdesc, &c->tmp_a, &c->tmp_b, &c->tmp_c, &c->tmp_d, &c->tmp_e, &c->tmp_f, &c->tmp_r, &c->tmp_w, &c->tmp_p, &c->tmp_n,
                      &c->f32.rd_sorted, &c->f64.rd_sorted, &c->f32.nrm_pre, &c->f32.nrm_sorted, &c->f64.nrm_pre, &c->f64.nrm_sorted, &c->qrow, &c->qtmp, &c->order, &c->scan_pos, &c->qcounts, &c->qblock, &c->qstart,
                      &c->qcursor, &c->slow_list, &c->slow_lb, &c->slow_ring, &c->slow2, &c->active, &c->sel_tables, &c->queue, &c->f32.none_r, &c->f64.none_r})
        b->release();
    if (c->h_pinned) (void)t_host_free(c->h_pinned);
    if (c->h_up) (void)t_host_free(c->h_up);
    if (c->h_down) (void)t_host_free(c->h_down);
    if (c->h_flag) (void)t_host_free(c->h_flag);
    if (c->bounce.p) (void)t_host_free(c->bounce.p);
    if (std::getenv("PGICP_GRAPH_DEBUG"))
        std::fprintf(stderr, "pgicp context %p: %lld iteration graphs captured, %lld replayed, %d failures\n", (void *)c, c->graph_captures, c->graph_launches, c->graph_failures);
    for (auto &fs : c->fset)
        for (DevBuf *b : {&fs.in_f, &fs.in_d, &fs.keep, &fs.pos, &fs.bsum, &fs.out_f, &fs.out_d, &fs.idx}) b->release();
    for (auto &g : c->iter_graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (c->stamp_dev) (void)t_free(c->stamp_dev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int pgicp_upload_f32(pgicp_ctx *c, int n, const float *const *host, const int *stride, const int *npts, int mem, const float **dev_ptrs)
{
    return upload<float>(c, n, host, stride, npts, mem, dev_ptrs);
}
int pgicp_upload_f64(pgicp_ctx *c, int n, const double *const *host, const int *stride, const int *npts, int mem, const double **dev_ptrs)
{
    return upload<double>(c, n, host, stride, npts, mem, dev_ptrs);
}
int pgicp_host_alloc(pgicp_ctx *c, size_t bytes, void **out)
{
    if (!c || !out || bytes == 0) return fail(c, PGICP_ERR_ARG, "pgicp_host_alloc: bad argument");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, t_host_malloc(out, bytes, hipHostMallocDefault));
    return PGICP_OK;
}
int pgicp_host_free(pgicp_ctx *c, void *p)
{
    if (!c) return PGICP_ERR_ARG;
    if (p) HIPC(c, t_host_free(p));
    return PGICP_OK;
}

int pgicp_device_alloc(pgicp_ctx *c, size_t bytes, void **out)
{
    if (!c || !out || bytes == 0) return fail(c, PGICP_ERR_ARG, "pgicp_device_alloc: bad argument");
    HIPC(c, hipSetDevice(c->device));
    if (t_malloc(out, bytes) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return fail(c, PGICP_ERR_HIP, "pgicp_device_alloc: out of device memory"); }
    return PGICP_OK;
}
int pgicp_device_free(pgicp_ctx *c, void *p)
{
    // (the context may be NULL: an owner that outlives its context -- a keyframe freed at process exit -- still frees)
    if (!p) return PGICP_OK;
    if (!c) return t_free(p) == hipSuccess ? PGICP_OK : PGICP_ERR_HIP;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, t_free(p));
    return PGICP_OK;
}
int pgicp_device_copy(pgicp_ctx *c, void *dst, const void *src, size_t bytes, int kind)
{
    if (!c || (bytes && (!dst || !src)) || kind < PGICP_COPY_TO_DEVICE || kind > PGICP_COPY_ON_DEVICE)
        return fail(c, PGICP_ERR_ARG, "pgicp_device_copy: bad argument");
    if (!bytes) return PGICP_OK;
    HIPC(c, hipSetDevice(c->device));
    const hipMemcpyKind k = kind == PGICP_COPY_TO_DEVICE ? hipMemcpyHostToDevice : kind == PGICP_COPY_FROM_DEVICE ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (kind == PGICP_COPY_TO_DEVICE) XFER(c, h2d(c, dst, src, bytes));
    else if (kind == PGICP_COPY_FROM_DEVICE) XFER(c, d2h(c, dst, src, bytes));
    else HIPC(c, hipMemcpyAsync(dst, src, bytes, k, c->stream));
    HIPC(c, stream_sync(c));
    return PGICP_OK;
}

const char *pgicp_status_string(int status)
{
    switch (status) {
    case PGICP_OK: return "ok";
    case PGICP_ERR_NO_MATCH: return "no point to minimize (ConvergenceError)";
    case PGICP_ERR_NAN: return "NaN in the transformation checkers (ConvergenceError)";
    case PGICP_ERR_ARG: return "bad argument";
    case PGICP_ERR_HIP: return "HIP runtime error";
    case PGICP_ERR_NO_DEVICE: return "no usable gfx950 device (there is no CPU fallback)";
    case PGICP_ERR_NOT_RIGID: return "transformation is not rigid";
    case PGICP_ERR_BOUND: return "BoundTransformationChecker: limit exceeded (ConvergenceError)";
    default: return "unknown status";
    }
}

const char *pgicp_last_error(const pgicp_ctx *c) { return c ? c->err.c_str() : "null context"; }
void *pgicp_ctx_stream(pgicp_ctx *c) { return c ? (void *)c->stream : nullptr; }
int pgicp_ctx_device(const pgicp_ctx *c, int *device) { if (!c || !device) return PGICP_ERR_ARG; *device = c->device; return PGICP_OK; }

int pgicp_ctx_synchronize(pgicp_ctx *c)
{
    if (!c) return PGICP_ERR_ARG;
    HIPC(c, stream_sync(c));
    return PGICP_OK;
}

int pgicp_set_params(pgicp_ctx *c, const pgicp_params *p)
{
    if (!c || !p) return PGICP_ERR_ARG;
    if (p->knn < 1 || p->knn > PGICP_MAX_KNN) return fail(c, PGICP_ERR_ARG, "KDTreeMatcher.knn must be in [1, " + std::to_string(PGICP_MAX_KNN) + "]");
    if (p->error_minimizer != PGICP_MINIMIZER_POINT_TO_PLANE && p->error_minimizer != PGICP_MINIMIZER_POINT_TO_POINT &&
        p->error_minimizer != PGICP_MINIMIZER_POINT_TO_PLANE_4DOF && p->error_minimizer != PGICP_MINIMIZER_POINT_TO_POINT_WITH_COV)
        return fail(c, PGICP_ERR_ARG, "unknown error minimizer");
    if (p->bound_max_rot < 0.0 || p->bound_max_rot != p->bound_max_rot || p->bound_max_trans < 0.0 || p->bound_max_trans != p->bound_max_trans)
        return fail(c, PGICP_ERR_ARG, "BoundTransformationChecker limits must be >= 0");
    if (p->normal_max_angle < 0.0 || p->normal_max_angle != p->normal_max_angle) return fail(c, PGICP_ERR_ARG, "SurfaceNormalOutlierFilter.maxAngle must be >= 0");
    // (libnabo's epsilon ALLOWS a neighbour up to (1 + epsilon) times farther than the nearest; the exact neighbour meets that
    // for every epsilon >= 0, so the value is accepted and the search stays exact: the results are those of epsilon = 0)
    if (!(p->epsilon >= 0.0)) return fail(c, PGICP_ERR_ARG, "KDTreeMatcher.epsilon must be >= 0");
    if (!(p->max_dist > 0.0)) return fail(c, PGICP_ERR_ARG, "KDTreeMatcher.maxDist must be > 0");
    if (!(p->trim_ratio > 0.0 && p->trim_ratio <= 1.0)) return fail(c, PGICP_ERR_ARG, "TrimmedDistOutlierFilter.ratio must be in (0,1]");
    if (!(p->quantile_scale > 0.0) || !std::isfinite(p->quantile_scale)) return fail(c, PGICP_ERR_ARG, "MedianDistOutlierFilter.factor (quantile_scale) must be positive and finite");
    if (p->outlier_max_dist < 0.0 || p->outlier_max_dist != p->outlier_max_dist) return fail(c, PGICP_ERR_ARG, "MaxDistOutlierFilter.maxDist must be >= 0");
    if (p->max_iters < 1) return fail(c, PGICP_ERR_ARG, "CounterTransformationChecker.maxIterationCount must be >= 1");
    if (p->smooth_length < 1 || p->smooth_length > kHist - 1)
        return fail(c, PGICP_ERR_ARG, "DifferentialTransformationChecker.smoothLength must be in [1,15]");
    if (p->matcher != PGICP_MATCHER_GRID && p->matcher != PGICP_MATCHER_BRUTE) return fail(c, PGICP_ERR_ARG, "unknown matcher");
    if (p->grid_cell < 0.0) return fail(c, PGICP_ERR_ARG, "grid_cell must be >= 0");
    if (p->sum_order != PGICP_SUM_ORDER_SORTED && p->sum_order != PGICP_SUM_ORDER_SCAN) return fail(c, PGICP_ERR_ARG, "sum_order must be PGICP_SUM_ORDER_SORTED or PGICP_SUM_ORDER_SCAN");
    if (p->robust_fct < PGICP_ROBUST_NONE || p->robust_fct > PGICP_ROBUST_L1 || (p->robust_scale != PGICP_ROBUST_SCALE_NONE && p->robust_scale != PGICP_ROBUST_SCALE_MAD) ||
        (p->robust_fct != PGICP_ROBUST_NONE && !(p->robust_tuning > 0.0)) || p->robust_approx < 0.0 || p->robust_approx != p->robust_approx)
        return fail(c, PGICP_ERR_ARG, "RobustOutlierFilter: unknown robustFct / scaleEstimator, or tuning <= 0, or approximation < 0");
    if (p->robust_fct != PGICP_ROBUST_NONE && (p->trim_ratio != 1.0 || (p->quantile_scale != 1.0 && p->quantile_scale > 0.0)))
        return fail(c, PGICP_ERR_ARG, "RobustOutlierFilter: no quantile filter (TrimmedDist / MedianDist) beside it (trim_ratio must be 1)");
    if (p->robust_fct != PGICP_ROBUST_NONE && p->knn > 1)
        return fail(c, PGICP_ERR_ARG, "RobustOutlierFilter: knn > 1 is not supported with it");
    {   // another chain: its thresholds are not this one's (field by field: the struct has padding)
        const pgicp_params &a = c->prm, &b = *p;
        const bool same = a.knn == b.knn && a.max_dist == b.max_dist && a.trim_ratio == b.trim_ratio && a.outlier_max_dist == b.outlier_max_dist &&
                          a.quantile_scale == b.quantile_scale && a.error_minimizer == b.error_minimizer && a.normal_max_angle == b.normal_max_angle &&
                          a.matcher == b.matcher && a.grid_cell == b.grid_cell && a.robust_fct == b.robust_fct && a.robust_scale == b.robust_scale;
        if (!same) { c->sel_hints[0].clear(); c->sel_hints[1].clear(); }
    }
    c->prm = *p;
    if (c->prm.check_every < 1) c->prm.check_every = 1;
    return PGICP_OK;
}

int pgicp_get_params(const pgicp_ctx *c, pgicp_params *p)
{
    if (!c || !p) return PGICP_ERR_ARG;
    *p = c->prm;
    return PGICP_OK;
}

int pgicp_map_create_f32(pgicp_ctx *c, const float *xyz, int xs, const float *nrm, int ns, int m, int mem, int center, int *id)
{ return map_create<float>(c, xyz, xs, nrm, ns, m, mem, center, id); }
int pgicp_map_create_f64(pgicp_ctx *c, const double *xyz, int xs, const double *nrm, int ns, int m, int mem, int center, int *id)
{ return map_create<double>(c, xyz, xs, nrm, ns, m, mem, center, id); }

int pgicp_map_create_batch_f32(pgicp_ctx *c, int n, const float *const *xyz, const int *xs, const float *const *nrm,
                               const int *ns, const int *m, int mem, int center, int *ids)
{ return map_create_batch_abi<float>(c, n, xyz, xs, nrm, ns, m, mem, center, ids); }
int pgicp_map_create_batch_f64(pgicp_ctx *c, int n, const double *const *xyz, const int *xs, const double *const *nrm,
                               const int *ns, const int *m, int mem, int center, int *ids)
{ return map_create_batch_abi<double>(c, n, xyz, xs, nrm, ns, m, mem, center, ids); }

int pgicp_map_destroy(pgicp_ctx *c, int id)
{
    if (!c) return PGICP_ERR_ARG;
    (void)hipSetDevice(c->device);
    // a map id is unique across precisions only per State; try both, f32 first
    if (MapHost<float> *m = get_map<float>(c, id)) { free_map(c, *m); return PGICP_OK; }
    if (MapHost<double> *m = get_map<double>(c, id)) { free_map(c, *m); return PGICP_OK; }
    return fail(c, PGICP_ERR_ARG, "pgicp_map_destroy: unknown map id");
}

int pgicp_map_transfer(pgicp_ctx *from, int id, pgicp_ctx *to, int *new_id)
{
    if (!from || !to || !new_id) return PGICP_ERR_ARG;
    if (from == to) { *new_id = id; return PGICP_OK; }
    if (from->device != to->device) return fail(to, PGICP_ERR_ARG, "pgicp_map_transfer: contexts are on different devices");
    (void)hipSetDevice(to->device);
    int st = map_transfer_impl<float>(from, id, to, new_id);
    if (st == -1) st = map_transfer_impl<double>(from, id, to, new_id);
    if (st == -1) return fail(to, PGICP_ERR_ARG, "pgicp_map_transfer: unknown map id");
    return st;
}

int pgicp_map_size(pgicp_ctx *c, int id, int *m)
{
    if (!c || !m) return PGICP_ERR_ARG;
    if (MapHost<float> *h = get_map<float>(c, id)) { *m = h->m; return PGICP_OK; }
    if (MapHost<double> *h = get_map<double>(c, id)) { *m = h->m; return PGICP_OK; }
    return fail(c, PGICP_ERR_ARG, "pgicp_map_size: unknown map id");
}

static pgicp_problem one_problem(int map_id, const void *rd, int stride, int n, int mem, const double *T)
{
    pgicp_problem p;
    std::memset(&p, 0, sizeof p);
    p.map_id = map_id; p.reading = rd; p.stride = stride; p.n = n; p.mem = mem;
    if (T) std::memcpy(p.T_init, T, sizeof p.T_init); else mat4_identity(p.T_init);
    return p;
}

int pgicp_align_f32(pgicp_ctx *c, int map_id, const float *rd, int stride, int n, int mem, const double T_init[16],
                    double T_out[16], pgicp_stats *stats)
{
    if (!T_init) return fail(c, PGICP_ERR_ARG, "pgicp_align: T_init is null");
    pgicp_problem p = one_problem(map_id, rd, stride, n, mem, T_init);
    return align_batch<float>(c, 1, &p, T_out, stats);
}
int pgicp_align_f64(pgicp_ctx *c, int map_id, const double *rd, int stride, int n, int mem, const double T_init[16],
                    double T_out[16], pgicp_stats *stats)
{
    if (!T_init) return fail(c, PGICP_ERR_ARG, "pgicp_align: T_init is null");
    pgicp_problem p = one_problem(map_id, rd, stride, n, mem, T_init);
    return align_batch<double>(c, 1, &p, T_out, stats);
}
int pgicp_align_batch_f32(pgicp_ctx *c, int P, const pgicp_problem *pr, double *T_out, pgicp_stats *stats)
{ return align_batch<float>(c, P, pr, T_out, stats); }
int pgicp_align_batch_f64(pgicp_ctx *c, int P, const pgicp_problem *pr, double *T_out, pgicp_stats *stats)
{ return align_batch<double>(c, P, pr, T_out, stats); }

int pgicp_align_residual_batch_f32(pgicp_ctx *c, int P, const pgicp_problem *pr, double *T_out, pgicp_stats *stats, double *residual,
                                   double *ratio, int *status)
{
    if (!residual) return fail(c, PGICP_ERR_ARG, "pgicp_align_residual_batch: residual is null");
    return align_batch<float>(c, P, pr, T_out, stats, residual, ratio, status);
}
int pgicp_align_residual_batch_f64(pgicp_ctx *c, int P, const pgicp_problem *pr, double *T_out, pgicp_stats *stats, double *residual,
                                   double *ratio, int *status)
{
    if (!residual) return fail(c, PGICP_ERR_ARG, "pgicp_align_residual_batch: residual is null");
    return align_batch<double>(c, P, pr, T_out, stats, residual, ratio, status);
}

int pgicp_icp_pair_f32(pgicp_ctx *c, const float *rd, int rs, int n, const float *rx, int xs, const float *rn, int ns, int m,
                       int mem, const double T_init[16], double T_out[16], pgicp_stats *stats)
{ return icp_pair<float>(c, rd, rs, n, rx, xs, rn, ns, m, mem, T_init, T_out, stats); }
int pgicp_icp_pair_f64(pgicp_ctx *c, const double *rd, int rs, int n, const double *rx, int xs, const double *rn, int ns,
                       int m, int mem, const double T_init[16], double T_out[16], pgicp_stats *stats)
{ return icp_pair<double>(c, rd, rs, n, rx, xs, rn, ns, m, mem, T_init, T_out, stats); }

int pgicp_match_f32(pgicp_ctx *c, int map_id, const float *rd, int stride, int n, int mem, const double *T, int32_t *ids, float *d2)
{ return match<float>(c, map_id, rd, stride, n, mem, T, ids, d2); }
int pgicp_match_f64(pgicp_ctx *c, int map_id, const double *rd, int stride, int n, int mem, const double *T, int32_t *ids, double *d2)
{ return match<double>(c, map_id, rd, stride, n, mem, T, ids, d2); }

int pgicp_outlier_weights_f32(pgicp_ctx *c, const float *d2, int n, int mem, float *w, float *limit, int *nf)
{ return outlier_weights<float>(c, d2, n, mem, w, limit, nf); }
int pgicp_outlier_weights_f64(pgicp_ctx *c, const double *d2, int n, int mem, double *w, double *limit, int *nf)
{ return outlier_weights<double>(c, d2, n, mem, w, limit, nf); }

int pgicp_error_stats_f32(pgicp_ctx *c, int map_id, const float *rd, int stride, int n, int mem, const int32_t *ids,
                          const float *w, double *ratio, double *residual, double sys[30])
{ return error_stats<float>(c, map_id, rd, stride, n, mem, ids, w, ratio, residual, sys); }
int pgicp_error_stats_f64(pgicp_ctx *c, int map_id, const double *rd, int stride, int n, int mem, const int32_t *ids,
                          const double *w, double *ratio, double *residual, double sys[30])
{ return error_stats<double>(c, map_id, rd, stride, n, mem, ids, w, ratio, residual, sys); }

int pgicp_partial_chain_f32(pgicp_ctx *c, int map_id, const float *rd, int stride, int n, int mem, const double *T,
                            double *ratio, double *residual)
{ return partial_chain<float>(c, map_id, rd, stride, n, mem, T, ratio, residual); }
int pgicp_partial_chain_f64(pgicp_ctx *c, int map_id, const double *rd, int stride, int n, int mem, const double *T,
                            double *ratio, double *residual)
{ return partial_chain<double>(c, map_id, rd, stride, n, mem, T, ratio, residual); }

int pgicp_partial_chain_batch_f32(pgicp_ctx *c, int P, const pgicp_problem *pr, double *ratio, double *residual, int *status)
{ return partial_chain_batch<float>(c, P, pr, ratio, residual, status); }
int pgicp_partial_chain_batch_f64(pgicp_ctx *c, int P, const pgicp_problem *pr, double *ratio, double *residual, int *status)
{ return partial_chain_batch<double>(c, P, pr, ratio, residual, status); }

int pgicp_surface_normals_f32(pgicp_ctx *c, const float *xyz, int stride, int n, int mem, int knn, double max_dist,
                              float *out_nrm, int out_stride, float *out_eig, int32_t *out_ids, float *out_d2)
{ return surface_normals<float>(c, xyz, stride, n, mem, knn, max_dist, out_nrm, out_stride, out_eig, out_ids, out_d2); }
int pgicp_surface_normals_f64(pgicp_ctx *c, const double *xyz, int stride, int n, int mem, int knn, double max_dist,
                              double *out_nrm, int out_stride, double *out_eig, int32_t *out_ids, double *out_d2)
{ return surface_normals<double>(c, xyz, stride, n, mem, knn, max_dist, out_nrm, out_stride, out_eig, out_ids, out_d2); }

int pgicp_transform_f32(pgicp_ctx *c, const double T[16], const float *in, int is, float *out, int os, int n, int ro, int mem)
{ return transform<float>(c, T, in, is, out, os, n, ro, mem); }
int pgicp_transform_f64(pgicp_ctx *c, const double T[16], const double *in, int is, double *out, int os, int n, int ro, int mem)
{ return transform<double>(c, T, in, is, out, os, n, ro, mem); }

int pgicp_build_local_map_f32(pgicp_ctx *c, int n_kf, const float *const *xyz, const float *const *nrm, const int *sx,
                              const int *sn, const int *counts, const double *T_ref_kf, float *ox, int os, float *on, int ons, int mem)
{ return build_local_map<float>(c, n_kf, xyz, nrm, sx, sn, counts, T_ref_kf, ox, os, on, ons, mem); }
int pgicp_build_local_map_f64(pgicp_ctx *c, int n_kf, const double *const *xyz, const double *const *nrm, const int *sx,
                              const int *sn, const int *counts, const double *T_ref_kf, double *ox, int os, double *on, int ons, int mem)
{ return build_local_map<double>(c, n_kf, xyz, nrm, sx, sn, counts, T_ref_kf, ox, os, on, ons, mem); }

int pgicp_filter_cloud_f32(pgicp_ctx *c, int nf, const pgicp_filter *f, const float *feat, int frows, const float *desc, int drows, int n,
                           const double *T, int r0, int r1, float *of, float *od, int32_t *idx, int *n_out, const float **dev)
{ return filter_cloud<float>(c, nf, f, feat, frows, desc, drows, n, T, r0, r1, of, od, idx, n_out, dev); }
int pgicp_filter_cloud_f64(pgicp_ctx *c, int nf, const pgicp_filter *f, const double *feat, int frows, const double *desc, int drows, int n,
                           const double *T, int r0, int r1, double *of, double *od, int32_t *idx, int *n_out, const double **dev)
{ return filter_cloud<double>(c, nf, f, feat, frows, desc, drows, n, T, r0, r1, of, od, idx, n_out, dev); }
int pgicp_filter_cloud_dev_f32(pgicp_ctx *c, int nf, const pgicp_filter *f, const float *feat, int frows, int n, int32_t *dropped, int cap, int *n_dropped,
                               int *n_out, const float **dev)
{ return filter_cloud_dev<float>(c, nf, f, feat, frows, n, dropped, cap, n_dropped, n_out, dev); }
int pgicp_filter_cloud_dev_f64(pgicp_ctx *c, int nf, const pgicp_filter *f, const double *feat, int frows, int n, int32_t *dropped, int cap, int *n_dropped,
                               int *n_out, const double **dev)
{ return filter_cloud_dev<double>(c, nf, f, feat, frows, n, dropped, cap, n_dropped, n_out, dev); }

int pgicp_shard_pairs(int n_pairs, const int64_t *cost, int world, int rank, int *out_idx, int cap, int *n_out)
{
    if (n_pairs < 0 || world <= 0 || rank < 0 || rank >= world || !n_out) return PGICP_ERR_ARG;
    // longest-processing-time first: sort by cost descending (index ascending on
    // ties), give each pair to the currently least loaded rank (lowest rank on ties)
    std::vector<int> order(n_pairs);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        const int64_t ca = cost ? cost[a] : 1, cb = cost ? cost[b] : 1;
        return ca > cb;
    });
    std::vector<int64_t> load(world, 0);
    std::vector<int> count(world, 0);
    int n = 0;
    for (int i : order) {
        int best = 0;
        for (int r = 1; r < world; r++)
            if (load[r] < load[best] || (load[r] == load[best] && count[r] < count[best])) best = r;
        load[best] += cost ? cost[i] : 1;
        count[best]++;
        if (best == rank) {
            if (out_idx && n < cap) out_idx[n] = i;
            n++;
        }
    }
    *n_out = n;
    if (out_idx && n > cap) return PGICP_ERR_ARG;
    if (out_idx) std::sort(out_idx, out_idx + n);
    return PGICP_OK;
}

int pgicp_check_icp_result(const pgicp_stats *icp, double residual_error, double overlap_threshold,
                           double residual_error_threshold)
{
    if (!icp) return 0;
    if (icp->status != PGICP_OK) return 0;
    if (icp->max_iter_reached) return 0;                       // LoopCloser.hpp:317
    if (icp->overlap < overlap_threshold) return 0;            // LoopCloser.hpp:331
    if (residual_error > residual_error_threshold) return 0;   // LoopCloser.hpp:335
    return 1;
}

int pgicp_debug_alloc_stats(long long out[8])
{
    if (!out) return PGICP_ERR_ARG;
    for (int k = 0; k < 4; k++) { out[2 * k] = g_alloc.n[k].load(); out[2 * k + 1] = g_alloc.ns[k].load(); }
    return PGICP_OK;
}

int pgicp_debug_counters(pgicp_ctx *c, int out[4])
{
    if (!c || !out) return PGICP_ERR_ARG;
    out[0] = out[1] = out[2] = out[3] = 0;
    HIPC(c, hipSetDevice(c->device));
    if (c->small.p)           // (a context that has not aligned anything yet has no counters)
        HIPC(c, hipMemcpy(out, c->small.as<int>() + (c->counters_clean ? 24 : 16), 4 * sizeof(int), hipMemcpyDeviceToHost));
    out[3] = sel_fallbacks_read(1);       // selections whose guess was off (k_sel_final2) since the last call, all contexts of the device
    if (std::getenv("PGICP_KNN_STATS_DUMP")) {          // diagnostics builds only
        unsigned long long s[56];
        (void)hipDeviceSynchronize();
        {
            unsigned long long ph[48];
            if (knn_phase_read(ph, 1) == 0) {
                std::fprintf(stderr, "  wave-per-query kernel: waves=%llu entries=%llu; cycles per ENTRY by phase [fetch, seed, look-ups, points, super-cell walk, finish]:", ph[47], ph[46]);
                for (int k = 0; k < 6; k++) std::fprintf(stderr, " %.0f", (double)ph[32 + k] / (double)std::max<unsigned long long>(1ULL, ph[46]));
                std::fprintf(stderr, "; longest wave %llu, longest entry %llu cycles\n", ph[44], ph[45]);
                for (int u = 0; u < 2; u++) {
                    const double nw = (double)std::max<unsigned long long>(1ULL, ph[16 * u + 15]);
                    std::fprintf(stderr, "  fast kernel, %s: waves=%llu; cycles per wave by phase [set-up, near, own row, row tables, flat walk, rings, finish]:", u ? "seeded passes" : "unseeded pass", ph[16 * u + 15]);
                    for (int k = 0; k < 7; k++) std::fprintf(stderr, " %.0f", (double)ph[16 * u + k] / nw);
                    std::fprintf(stderr, "\n");
                }
            }
        }
        if (knn_stats_read(s, 1) == 0) {
            std::fprintf(stderr, "knn_stats waves=%llu a1_max=%llu a1_sum=%llu flat_iters_max=%llu a2_sum=%llu b_cand=%llu unresolved=%llu b_lanes=%llu b_max=%llu tot_max=%llu\n",
                         s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8], s[9]);
            std::fprintf(stderr, "  slow: entries=%llu cycles_sum=%llu cycles_max=%llu (worst: supercells=%llu rows=%llu trips=%llu exist_only=%llu found=%llu) trips_sum=%llu\n",
                         s[10], s[11], s[12], s[13] >> 40, (s[13] >> 20) & 0xFFFFF, s[13] & 0xFFFFF, s[14] >> 32, s[14] & 1, s[15]);
            std::fprintf(stderr, "  slow kinds: bounded stage=%llu capped=%llu exist_only=%llu; super-cells (or stage trips)=%llu rows with points=%llu\n", s[51], s[52], s[53], s[54], s[55]);
            std::fprintf(stderr, "  med: searched=%llu existence_unknown=%llu resolved=%llu candidates=%llu; wave ticks(100MHz) max=%llu sum=%llu waves=%llu\n", s[44], s[45], s[46], s[47], s[48], s[49], s[50]);
            std::fprintf(stderr, "  own-row hist (0,1,2-3,4-7,...):");
            for (int i = 0; i < 12; i++) std::fprintf(stderr, " %llu", s[16 + i]);
            std::fprintf(stderr, "\n  flat hist:");
            for (int i = 0; i < 12; i++) std::fprintf(stderr, " %llu", s[32 + i]);
            std::fprintf(stderr, "\n");
        }
    }
    return PGICP_OK;
}

// diagnostics builds only (not part of the ABI): per-query candidate dump of the fast matcher
int pgicp_debug_dump_setup(long long total, int passes) { return knn_dump_setup(total, passes); }
int pgicp_debug_dump_read(unsigned *cnt, float *d2, long long n) { return knn_dump_read(cnt, d2, n); }

int pgicp_debug_reading_order(pgicp_ctx *c, int problem, int32_t *order)
{
    if (!c || problem < 0 || !order) return PGICP_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    ProblemDev D;
    HIPC(c, hipMemcpy(&D, c->probs.as<ProblemDev>() + problem, sizeof D, hipMemcpyDeviceToHost));
    if (D.n <= 0 || !c->order.p) return fail(c, PGICP_ERR_ARG, "pgicp_debug_reading_order: no reading was sorted for this problem");
    XFER(c, d2h(c, order, c->order.as<int>() + D.off, sizeof(int) * (size_t)D.n));
    HIPC(c, stream_sync(c));
    return PGICP_OK;
}
int pgicp_debug_last_matches_f32(pgicp_ctx *c, int problem, int32_t *ids, float *dist2) { return debug_last_matches<float>(c, problem, ids, dist2); }
int pgicp_debug_last_matches_f64(pgicp_ctx *c, int problem, int32_t *ids, double *dist2) { return debug_last_matches<double>(c, problem, ids, dist2); }

int pgicp_profile_enable(pgicp_ctx *c, int on)
{
    if (!c) return PGICP_ERR_ARG;
    if (!on) prof_collect(c);
    c->prof_on = on != 0;
    return PGICP_OK;
}

int pgicp_profile_process(int kid, long long *launches, double *total_ms, long long *units, long long *problems, long long *map_points)
{
    if (kid < 0 || kid >= PGICP_PROF_COUNT) return PGICP_ERR_ARG;
    ProcessProfile &pp = process_profile();
    std::lock_guard<std::mutex> lock(pp.m);
    long long l = pp.launches[kid], u = pp.units[kid], pr = pp.problems[kid], mp = pp.map_points[kid];
    double ms = pp.ms[kid];
    for (pgicp_ctx *c : pp.live) {            // (a context that is mid-call on another thread: its events so far are collected
        (void)hipSetDevice(c->device);        //  under its profile lock -- a wait on its stream, no access to its other state)
        std::lock_guard<std::mutex> plock(c->prof_m);
        prof_collect_locked(c);
        l += c->prof_launches[kid]; ms += c->prof_ms[kid]; u += c->prof_units[kid]; pr += c->prof_problems[kid]; mp += c->prof_map_points[kid];
    }
    if (launches) *launches = l;
    if (total_ms) *total_ms = ms;
    if (units) *units = u;
    if (problems) *problems = pr;
    if (map_points) *map_points = mp;
    return PGICP_OK;
}

int pgicp_profile_reset(pgicp_ctx *c)
{
    if (!c) return PGICP_ERR_ARG;
    prof_collect(c);
    for (int i = 0; i < PGICP_PROF_COUNT; i++) { c->prof_launches[i] = 0; c->prof_ms[i] = 0; c->prof_units[i] = 0; c->prof_problems[i] = 0; c->prof_map_points[i] = 0; }
    return PGICP_OK;
}

int pgicp_profile_get(pgicp_ctx *c, int kid, long long *launches, double *total_ms, long long *units, long long *problems)
{
    if (!c || kid < 0 || kid >= PGICP_PROF_COUNT) return PGICP_ERR_ARG;
    prof_collect(c);
    if (launches) *launches = c->prof_launches[kid];
    if (total_ms) *total_ms = c->prof_ms[kid];
    if (units) *units = c->prof_units[kid];
    if (problems) *problems = c->prof_problems[kid];
    return PGICP_OK;
}

}  // extern "C"
