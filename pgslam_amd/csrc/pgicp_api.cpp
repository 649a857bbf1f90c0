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

// The host side in parts (one translation unit: the parts share pgicp_ctx and its helpers; split in round 6 along the file's own
// sections -- it had grown to 2 700 lines):
#include "api_context.inc"            // context, allocation accounting, fail() / HIPC / XFER, the pinned bounce buffer, the profile
#include "api_maps.inc"               // the device table of maps, the block pool, map_create_batch (the index build)
#include "api_icp.inc"                // BatchLayout, batch_begin, one iteration, align_batch, icp_pair
#include "api_stages.inc"             // match, partial chain, outlier weights, error statistics, transform, local maps, normals
#include "api_filters_uploads.inc"    // pgicp_filter_cloud*, batched map ABI, last-call diagnostics, map transfer, pgicp_upload_*

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

int pgicp_partial_chain_seeded_f32(pgicp_ctx *c, int map_id, const float *rd, int stride, int n, int mem, const double *T, pgicp_ctx *src, int n_seg,
                                   const int32_t *src_start, const int32_t *dst_start, double *ratio, double *residual)
{ return partial_chain_seeded<float>(c, map_id, rd, stride, n, mem, T, src, n_seg, src_start, dst_start, ratio, residual); }
int pgicp_partial_chain_seeded_f64(pgicp_ctx *c, int map_id, const double *rd, int stride, int n, int mem, const double *T, pgicp_ctx *src, int n_seg,
                                   const int32_t *src_start, const int32_t *dst_start, double *ratio, double *residual)
{ return partial_chain_seeded<double>(c, map_id, rd, stride, n, mem, T, src, n_seg, src_start, dst_start, ratio, residual); }
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

