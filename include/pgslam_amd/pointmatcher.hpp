// pointmatcher.hpp -- drop-in for the part of libpointmatcher's object model that
// pgslam drives on its ICP hot path, forwarding every numeric stage to the
// MI355X library through the C ABI of include/pgicp.h.
//
// Same spelling as PointMatcher<T> (reference src/pgslam/types.h:19-29 aliases
// it as PM): DataPoints, Matches, OutlierWeights, Transformation(s),
// DataPointsFilters, Matcher, OutlierFilters, ErrorMinimizer (+ErrorElements),
// TransformationCheckers, ICPChainBase, ICP, ICPSequence, ConvergenceError,
// PM::get().REG(Transformation).create("RigidTransformation").
// Call sites mirrored: Localizer.hpp:20,70,77,103,106,126,148,168,238,254,278,
// 309-347; LoopCloser.hpp:73,98,108,317,331,346-362; LocalMap.hpp:37,97,222.
//
// Header only; link with -lpgicp.  There is no CPU fallback: constructing an ICP
// object without a usable GPU throws.
#pragma once
#include <cmath>
#include <cstring>
#include <istream>
#include <limits>
#include <functional>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../pgicp.h"
#include "matrix.hpp"
#include "yaml_lite.hpp"

namespace pgslam_amd {

// one lazily created context for objects that are not ICP chains (standalone
// RigidTransformation, LocalMap assembly); every ICP chain owns its own context
// and stream, as every pgslam worker owns its own ICP object.
inline pgicp_ctx *default_context(int device = 0)
{
    struct Holder {
        pgicp_ctx *c = nullptr;
        ~Holder() { if (c) pgicp_ctx_destroy(c); }
    };
    static thread_local Holder h;
    if (!h.c) {
        const int st = pgicp_ctx_create(device, &h.c);
        if (st != PGICP_OK) throw std::runtime_error("pgslam_amd: no usable MI355X device (pgicp_ctx_create failed, code " + std::to_string(st) + ")");
    }
    return h.c;
}

//! A device context made at FIRST USE: an ICP chain (and therefore a default-constructed pgslam facade, as the
//! reference's tests/instantiation.cpp constructs them) touches no GPU until it is asked to compute something.
struct LazyContext {
    int device = 0;
    int high_priority = 0;                      // pgicp_ctx_create_priority: a stream of the device's highest priority
    mutable pgicp_ctx *p = nullptr;
    explicit LazyContext(int d = 0, int prio = 0) : device(d), high_priority(prio) {}
    LazyContext(const LazyContext &) = delete;
    LazyContext &operator=(const LazyContext &) = delete;
    ~LazyContext() { destroy(); }
    bool made() const { return p != nullptr; }
    pgicp_ctx *get() const
    {
        if (!p) {
            const int st = high_priority ? pgicp_ctx_create_priority(device, 1, &p) : pgicp_ctx_create(device, &p);
            if (st != PGICP_OK) { p = nullptr; throw std::runtime_error(std::string("pgslam_amd: cannot create a device context: ") + pgicp_status_string(st)); }
        }
        return p;
    }
    operator pgicp_ctx *() const { return get(); }
    void destroy() { if (p) { pgicp_ctx_destroy(p); p = nullptr; } }
};

//! A cloud that stays in device memory between calls (pgicp_device_alloc): points as [n][xs], normals (optional) as [n][ns].
//! What a keyframe of the local map keeps (LocalMap.hpp:209-224 walks the keyframes at every rebuild; here the walk is one
//! device pass over resident clouds: pgicp_build_local_map with mem = PGICP_DEVICE) and what an assembled map is handed to
//! ICPSequence::setMap as.  Grow-only when reused as an assembly buffer.
template <typename T>
struct DeviceCloud {
    T *xyz = nullptr, *nrm = nullptr;
    int n = 0, xs = 3, ns = 3;
    size_t cap_x = 0, cap_n = 0;                    // elements allocated
    bool has_nrm = false;                           // the CURRENT contents carry normals (a reused buffer may hold an older cloud's)
    DeviceCloud() {}
    DeviceCloud(const DeviceCloud &) = delete;
    DeviceCloud &operator=(const DeviceCloud &) = delete;
    ~DeviceCloud() { release(); }
    void release()
    {
        if (xyz) pgicp_device_free(nullptr, xyz);
        if (nrm) pgicp_device_free(nullptr, nrm);
        xyz = nrm = nullptr; n = 0; cap_x = cap_n = 0; has_nrm = false;
    }
    static void fail(pgicp_ctx *c, int st, const char *what)
    {
        if (st != PGICP_OK) throw std::runtime_error(std::string(what) + ": " + pgicp_status_string(st) + " (" + pgicp_last_error(c) + ")");
    }
    //! room for `count` points (and normals) at the given strides; contents are undefined afterwards
    void reserve(pgicp_ctx *c, int count, int xstride, bool normals, int nstride)
    {
        const size_t need_x = (size_t)count * xstride, need_n = normals ? (size_t)count * nstride : 0;
        if (need_x > cap_x) {
            if (xyz) pgicp_device_free(c, xyz);
            xyz = nullptr; cap_x = 0;
            fail(c, pgicp_device_alloc(c, sizeof(T) * (need_x + need_x / 8), (void **)&xyz), "DeviceCloud: device allocation");
            cap_x = need_x + need_x / 8;
        }
        if (need_n > cap_n) {
            if (nrm) pgicp_device_free(c, nrm);
            nrm = nullptr; cap_n = 0;
            fail(c, pgicp_device_alloc(c, sizeof(T) * (need_n + need_n / 8), (void **)&nrm), "DeviceCloud: device allocation");
            cap_n = need_n + need_n / 8;
        }
        n = count; xs = xstride; ns = nstride; has_nrm = normals;
    }
    bool hasNormals() const { return has_nrm && nrm != nullptr && cap_n >= (size_t)n * ns && n > 0; }
    //! host -> device, strides kept as they are on the host (`x` holds (count - 1) * xstride + 3 elements at least)
    void upload(pgicp_ctx *c, const T *x, int xstride, const T *nr, int nstride, int count)
    {
        reserve(c, count, xstride, nr != nullptr, nstride);
        if (count <= 0) return;
        fail(c, pgicp_device_copy(c, xyz, x, sizeof(T) * ((size_t)(count - 1) * xstride + 3), PGICP_COPY_TO_DEVICE), "DeviceCloud: upload");
        if (nr) fail(c, pgicp_device_copy(c, nrm, nr, sizeof(T) * ((size_t)(count - 1) * nstride + 3), PGICP_COPY_TO_DEVICE), "DeviceCloud: upload");
    }
};

template <typename T> struct Abi;
template <> struct Abi<float> {
    static int map_create(pgicp_ctx *c, const float *x, int xs, const float *n, int ns, int m, int center, int *id) { return pgicp_map_create_f32(c, x, xs, n, ns, m, PGICP_HOST, center, id); }
    static int align(pgicp_ctx *c, int id, const float *r, int s, int n, const double *Ti, double *To, pgicp_stats *st) { return pgicp_align_f32(c, id, r, s, n, PGICP_HOST, Ti, To, st); }
    static int align_dev(pgicp_ctx *c, int id, const float *r, int s, int n, const double *Ti, double *To, pgicp_stats *st) { return pgicp_align_f32(c, id, r, s, n, PGICP_DEVICE, Ti, To, st); }
    static int align_nrm(pgicp_ctx *c, int id, const float *r, int s, int n, const float *nr, int ns, const double *Ti, double *To, pgicp_stats *st)
    {
        pgicp_problem p;
        std::memset(&p, 0, sizeof p);
        p.map_id = id; p.reading = r; p.stride = s; p.n = n; p.mem = PGICP_HOST; p.normals = nr; p.nstride = ns;
        std::memcpy(p.T_init, Ti, sizeof p.T_init);
        return pgicp_align_batch_f32(c, 1, &p, To, st);
    }
    static int upload(pgicp_ctx *c, const float *host, int stride, int n, const float **dev) { return pgicp_upload_f32(c, 1, &host, &stride, &n, PGICP_HOST, dev); }
    static int match(pgicp_ctx *c, int id, const float *r, int s, int n, int32_t *ids, float *d2) { return pgicp_match_f32(c, id, r, s, n, PGICP_HOST, nullptr, ids, d2); }
    static int weights(pgicp_ctx *c, const float *d2, int n, float *w, float *lim, int *nf) { return pgicp_outlier_weights_f32(c, d2, n, PGICP_HOST, w, lim, nf); }
    static int stats(pgicp_ctx *c, int id, const float *r, int s, int n, const int32_t *ids, const float *w, double *ratio, double *res, double *sys) { return pgicp_error_stats_f32(c, id, r, s, n, PGICP_HOST, ids, w, ratio, res, sys); }
    static int map_create_batch(pgicp_ctx *c, int k, const float *const *x, const int *xs, const float *const *n, const int *ns, const int *m, int center, int *ids) { return pgicp_map_create_batch_f32(c, k, x, xs, n, ns, m, PGICP_HOST, center, ids); }
    static int normals(pgicp_ctx *c, const float *x, int xs, int n, int knn, double md, float *out, int os, float *eig) { return pgicp_surface_normals_f32(c, x, xs, n, PGICP_HOST, knn, md, out, os, eig, nullptr, nullptr); }
    static int normals_ids(pgicp_ctx *c, const float *x, int xs, int n, int knn, double md, float *out, int os, float *eig, int32_t *ids) { return pgicp_surface_normals_f32(c, x, xs, n, PGICP_HOST, knn, md, out, os, eig, ids, nullptr); }
    static int partial(pgicp_ctx *c, int id, const float *r, int s, int n, const double *Tm, double *ratio, double *res) { return pgicp_partial_chain_f32(c, id, r, s, n, PGICP_HOST, Tm, ratio, res); }
    static int partial_dev(pgicp_ctx *c, int id, const float *r, int s, int n, const double *Tm, double *ratio, double *res) { return pgicp_partial_chain_f32(c, id, r, s, n, PGICP_DEVICE, Tm, ratio, res); }
    static int partial_seeded_dev(pgicp_ctx *c, int id, const float *r, int s, int n, const double *Tm, pgicp_ctx *src, int ns, const int32_t *a, const int32_t *b, double *ratio, double *res) { return pgicp_partial_chain_seeded_f32(c, id, r, s, n, PGICP_DEVICE, Tm, src, ns, a, b, ratio, res); }
    static int transform(pgicp_ctx *c, const double *Tm, const float *in, int is, float *out, int os, int n, int ro) { return pgicp_transform_f32(c, Tm, in, is, out, os, n, ro, PGICP_HOST); }
    static int local_map(pgicp_ctx *c, int k, const float *const *x, const float *const *n, const int *sx, const int *sn, const int *cnt, const double *Ts, float *ox, int os, float *on, int ons) { return pgicp_build_local_map_f32(c, k, x, n, sx, sn, cnt, Ts, ox, os, on, ons, PGICP_HOST); }
    static int local_map_dev(pgicp_ctx *c, int k, const float *const *x, const float *const *n, const int *sx, const int *sn, const int *cnt, const double *Ts, float *ox, int os, float *on, int ons) { return pgicp_build_local_map_f32(c, k, x, n, sx, sn, cnt, Ts, ox, os, on, ons, PGICP_DEVICE); }
    static int map_create_dev(pgicp_ctx *c, const float *x, int xs, const float *n, int ns, int m, int center, int *id) { return pgicp_map_create_f32(c, x, xs, n, ns, m, PGICP_DEVICE, center, id); }
    static int transform_dev(pgicp_ctx *c, const double *T16, const float *in, int is, float *out, int os, int n, int rotate_only) { return pgicp_transform_f32(c, T16, in, is, out, os, n, rotate_only, PGICP_DEVICE); }
    static int last_matches(pgicp_ctx *c, int problem, int32_t *ids, float *d2) { return pgicp_debug_last_matches_f32(c, problem, ids, d2); }
};
template <> struct Abi<double> {
    static int map_create(pgicp_ctx *c, const double *x, int xs, const double *n, int ns, int m, int center, int *id) { return pgicp_map_create_f64(c, x, xs, n, ns, m, PGICP_HOST, center, id); }
    static int align(pgicp_ctx *c, int id, const double *r, int s, int n, const double *Ti, double *To, pgicp_stats *st) { return pgicp_align_f64(c, id, r, s, n, PGICP_HOST, Ti, To, st); }
    static int align_dev(pgicp_ctx *c, int id, const double *r, int s, int n, const double *Ti, double *To, pgicp_stats *st) { return pgicp_align_f64(c, id, r, s, n, PGICP_DEVICE, Ti, To, st); }
    static int align_nrm(pgicp_ctx *c, int id, const double *r, int s, int n, const double *nr, int ns, const double *Ti, double *To, pgicp_stats *st)
    {
        pgicp_problem p;
        std::memset(&p, 0, sizeof p);
        p.map_id = id; p.reading = r; p.stride = s; p.n = n; p.mem = PGICP_HOST; p.normals = nr; p.nstride = ns;
        std::memcpy(p.T_init, Ti, sizeof p.T_init);
        return pgicp_align_batch_f64(c, 1, &p, To, st);
    }
    static int upload(pgicp_ctx *c, const double *host, int stride, int n, const double **dev) { return pgicp_upload_f64(c, 1, &host, &stride, &n, PGICP_HOST, dev); }
    static int match(pgicp_ctx *c, int id, const double *r, int s, int n, int32_t *ids, double *d2) { return pgicp_match_f64(c, id, r, s, n, PGICP_HOST, nullptr, ids, d2); }
    static int weights(pgicp_ctx *c, const double *d2, int n, double *w, double *lim, int *nf) { return pgicp_outlier_weights_f64(c, d2, n, PGICP_HOST, w, lim, nf); }
    static int stats(pgicp_ctx *c, int id, const double *r, int s, int n, const int32_t *ids, const double *w, double *ratio, double *res, double *sys) { return pgicp_error_stats_f64(c, id, r, s, n, PGICP_HOST, ids, w, ratio, res, sys); }
    static int map_create_batch(pgicp_ctx *c, int k, const double *const *x, const int *xs, const double *const *n, const int *ns, const int *m, int center, int *ids) { return pgicp_map_create_batch_f64(c, k, x, xs, n, ns, m, PGICP_HOST, center, ids); }
    static int normals(pgicp_ctx *c, const double *x, int xs, int n, int knn, double md, double *out, int os, double *eig) { return pgicp_surface_normals_f64(c, x, xs, n, PGICP_HOST, knn, md, out, os, eig, nullptr, nullptr); }
    static int normals_ids(pgicp_ctx *c, const double *x, int xs, int n, int knn, double md, double *out, int os, double *eig, int32_t *ids) { return pgicp_surface_normals_f64(c, x, xs, n, PGICP_HOST, knn, md, out, os, eig, ids, nullptr); }
    static int partial(pgicp_ctx *c, int id, const double *r, int s, int n, const double *Tm, double *ratio, double *res) { return pgicp_partial_chain_f64(c, id, r, s, n, PGICP_HOST, Tm, ratio, res); }
    static int partial_dev(pgicp_ctx *c, int id, const double *r, int s, int n, const double *Tm, double *ratio, double *res) { return pgicp_partial_chain_f64(c, id, r, s, n, PGICP_DEVICE, Tm, ratio, res); }
    static int partial_seeded_dev(pgicp_ctx *c, int id, const double *r, int s, int n, const double *Tm, pgicp_ctx *src, int ns, const int32_t *a, const int32_t *b, double *ratio, double *res) { return pgicp_partial_chain_seeded_f64(c, id, r, s, n, PGICP_DEVICE, Tm, src, ns, a, b, ratio, res); }
    static int transform(pgicp_ctx *c, const double *Tm, const double *in, int is, double *out, int os, int n, int ro) { return pgicp_transform_f64(c, Tm, in, is, out, os, n, ro, PGICP_HOST); }
    static int local_map(pgicp_ctx *c, int k, const double *const *x, const double *const *n, const int *sx, const int *sn, const int *cnt, const double *Ts, double *ox, int os, double *on, int ons) { return pgicp_build_local_map_f64(c, k, x, n, sx, sn, cnt, Ts, ox, os, on, ons, PGICP_HOST); }
    static int local_map_dev(pgicp_ctx *c, int k, const double *const *x, const double *const *n, const int *sx, const int *sn, const int *cnt, const double *Ts, double *ox, int os, double *on, int ons) { return pgicp_build_local_map_f64(c, k, x, n, sx, sn, cnt, Ts, ox, os, on, ons, PGICP_DEVICE); }
    static int map_create_dev(pgicp_ctx *c, const double *x, int xs, const double *n, int ns, int m, int center, int *id) { return pgicp_map_create_f64(c, x, xs, n, ns, m, PGICP_DEVICE, center, id); }
    static int transform_dev(pgicp_ctx *c, const double *T16, const double *in, int is, double *out, int os, int n, int rotate_only) { return pgicp_transform_f64(c, T16, in, is, out, os, n, rotate_only, PGICP_DEVICE); }
    static int last_matches(pgicp_ctx *c, int problem, int32_t *ids, double *d2) { return pgicp_debug_last_matches_f64(c, problem, ids, d2); }
};

}  // namespace pgslam_amd

template <typename T>
struct PointMatcher {
    typedef T ScalarType;
    typedef pgslam_amd::Mat<T> Matrix;
    typedef pgslam_amd::Mat<int> IntMatrix;
    typedef Matrix TransformationParameters;
    typedef Matrix OutlierWeights;
    using A = pgslam_amd::Abi<T>;

    //! what libpointmatcher throws when a stage cannot proceed; pgslam never catches it
    struct ConvergenceError : std::runtime_error {
        explicit ConvergenceError(const std::string &r) : std::runtime_error(r) {}
    };
    static void check(pgicp_ctx *c, int st)
    {
        if (st == PGICP_OK) return;
        const std::string msg = c ? pgicp_last_error(c) : "pgicp error";
        if (st == PGICP_ERR_NO_MATCH || st == PGICP_ERR_NAN || st == PGICP_ERR_BOUND) throw ConvergenceError(msg);
        if (st == PGICP_ERR_NOT_RIGID) throw std::runtime_error("RigidTransformation: " + msg);
        throw std::runtime_error("pgicp (" + std::to_string(st) + "): " + msg);
    }

    // ------------------------------------------------------------------ DataPoints
    struct DataPoints {
        struct Label {
            std::string text;
            size_t span;
            Label(const std::string &t = "", size_t s = 0) : text(t), span(s) {}
            bool operator==(const Label &o) const { return text == o.text && span == o.span; }
        };
        struct Labels : std::vector<Label> {
            bool contains(const std::string &t) const { for (auto &l : *this) if (l.text == t) return true; return false; }
            size_t totalDim() const { size_t d = 0; for (auto &l : *this) d += l.span; return d; }
        };
        Matrix features;          //!< (dim+1) x N homogeneous coordinates, column major: one point = 4 contiguous values
        Labels featureLabels;
        Matrix descriptors;       //!< D x N
        Labels descriptorLabels;

        DataPoints() {}
        DataPoints(const Matrix &f, const Labels &fl) : features(f), featureLabels(fl) {}
        DataPoints(const Matrix &f, const Labels &fl, const Matrix &d, const Labels &dl) : features(f), featureLabels(fl), descriptors(d), descriptorLabels(dl) {}
        //! convenience: N x 3 points (+ optional N x 3 normals), row i = point i
        static DataPoints fromXYZ(const T *xyz, int n, const T *normals = nullptr)
        {
            DataPoints dp;
            dp.features = Matrix(4, n);
            for (int i = 0; i < n; i++) { for (int a = 0; a < 3; a++) dp.features(a, i) = xyz[3 * i + a]; dp.features(3, i) = T(1); }
            dp.featureLabels.push_back(Label("x", 1)); dp.featureLabels.push_back(Label("y", 1));
            dp.featureLabels.push_back(Label("z", 1)); dp.featureLabels.push_back(Label("pad", 1));
            if (normals) {
                dp.descriptors = Matrix(3, n);
                for (int i = 0; i < n; i++) for (int a = 0; a < 3; a++) dp.descriptors(a, i) = normals[3 * i + a];
                dp.descriptorLabels.push_back(Label("normals", 3));
            }
            return dp;
        }
        unsigned getNbPoints() const { return (unsigned)features.cols(); }
        unsigned getEuclideanDim() const { return features.rows() > 0 ? (unsigned)features.rows() - 1 : 0; }
        bool descriptorExists(const std::string &name) const { return descriptorLabels.contains(name); }
        int getDescriptorStartingRow(const std::string &name) const
        {
            int row = 0;
            for (auto &l : descriptorLabels) { if (l.text == name) return row; row += (int)l.span; }
            throw std::runtime_error("DataPoints: descriptor " + name + " not found");
        }
        int getDescriptorDimension(const std::string &name) const
        {
            for (auto &l : descriptorLabels) if (l.text == name) return (int)l.span;
            return 0;
        }
        //! copy of the rows of one descriptor (Eigen returns a block view; reads are equivalent)
        Matrix getDescriptorViewByName(const std::string &name) const
        {
            return descriptors.block(getDescriptorStartingRow(name), 0, getDescriptorDimension(name), descriptors.cols());
        }
        void addDescriptor(const std::string &name, const Matrix &d)
        {
            if (d.cols() != features.cols()) throw std::runtime_error("DataPoints::addDescriptor: wrong number of points");
            if (descriptorExists(name)) throw std::runtime_error("DataPoints::addDescriptor: " + name + " exists");
            const int old = descriptors.rows();
            Matrix nd(old + d.rows(), features.cols());
            for (int j = 0; j < features.cols(); j++) {
                for (int i = 0; i < old; i++) nd(i, j) = descriptors(i, j);
                for (int i = 0; i < d.rows(); i++) nd(old + i, j) = d(i, j);
            }
            descriptors = nd;
            descriptorLabels.push_back(Label(name, d.rows()));
        }
        //! what a filter that "allocates" a descriptor does upstream (DataPoints::allocateDescriptors): a descriptor of that name and
        //! span that already exists is REUSED -- its rows are overwritten --, one of another span is an error, a new one is appended
        void setDescriptor(const std::string &name, const Matrix &d)
        {
            if (!descriptorExists(name)) { addDescriptor(name, d); return; }
            if (d.cols() != features.cols()) throw std::runtime_error("DataPoints::setDescriptor: wrong number of points");
            if (getDescriptorDimension(name) != (int)d.rows()) throw std::runtime_error("DataPoints::setDescriptor: " + name + " exists with another dimension");
            const int r0 = getDescriptorStartingRow(name);
            for (int j = 0; j < features.cols(); j++) for (int i = 0; i < d.rows(); i++) descriptors(r0 + i, j) = d(i, j);
        }
        //! append dp; only descriptors present in BOTH clouds (same name and span) are kept
        void concatenate(const DataPoints &dp)
        {
            if (features.cols() == 0) { *this = dp; return; }
            if (dp.features.rows() != features.rows()) throw std::runtime_error("DataPoints::concatenate: feature dimensions differ");
            const int n0 = features.cols(), n1 = dp.features.cols();
            Matrix f(features.rows(), n0 + n1);
            std::memcpy(f.data(), features.data(), sizeof(T) * features.size());
            std::memcpy(f.data() + features.size(), dp.features.data(), sizeof(T) * dp.features.size());
            Labels keep;
            int rows = 0;
            for (auto &l : descriptorLabels)
                if (dp.descriptorLabels.contains(l.text) && dp.getDescriptorDimension(l.text) == (int)l.span) { keep.push_back(l); rows += (int)l.span; }
            Matrix d(rows, n0 + n1);
            int r = 0;
            for (auto &l : keep) {
                const int a = getDescriptorStartingRow(l.text), b = dp.getDescriptorStartingRow(l.text);
                for (int k = 0; k < (int)l.span; k++) {
                    for (int j = 0; j < n0; j++) d(r + k, j) = descriptors(a + k, j);
                    for (int j = 0; j < n1; j++) d(r + k, n0 + j) = dp.descriptors(b + k, j);
                }
                r += (int)l.span;
            }
            features = f; descriptors = d; descriptorLabels = keep;
        }
        // (pointer, stride) views handed to the C ABI
        const T *xyzPtr() const { return features.data(); }
        int xyzStride() const { return features.rows(); }
        const T *normalsPtr() const { return descriptorExists("normals") ? descriptors.data() + getDescriptorStartingRow("normals") : nullptr; }
        int normalsStride() const { return descriptors.rows(); }
    };

    // ------------------------------------------------------------------ Matches
    struct Matches {
        typedef Matrix Dists;
        typedef IntMatrix Ids;
        static constexpr int InvalidId = -1;
        static T InvalidDist() { return std::numeric_limits<T>::infinity(); }
        Dists dists;      //!< knn x N SQUARED distances
        Ids ids;          //!< knn x N indices into the reference
        Matches() {}
        Matches(int knn, int n) : dists(knn, n), ids(knn, n) {}
    };

    // ------------------------------------------------------------------ Transformation
    struct Transformation {
        virtual ~Transformation() {}
        virtual DataPoints compute(const DataPoints &input, const TransformationParameters &parameters) const = 0;
        virtual bool checkParameters(const TransformationParameters &parameters) const = 0;
    };
    struct RigidTransformation : Transformation {
        const pgslam_amd::LazyContext *ctx;          // the owning chain's context (made at first use), or none
        explicit RigidTransformation(const pgslam_amd::LazyContext *c = nullptr) : ctx(c) {}
        bool checkParameters(const TransformationParameters &p) const override
        {
            double e = 0;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    double s = 0;
                    for (int k = 0; k < 3; k++) s += (double)p(k, i) * (double)p(k, j);
                    e = std::max(e, std::fabs(s - (i == j ? 1.0 : 0.0)));
                }
            return e < 1e-3;
        }
        //! features' = T * features; the normals / observationDirections descriptors are rotated
        DataPoints compute(const DataPoints &input, const TransformationParameters &parameters) const override
        {
            pgicp_ctx *c = ctx ? ctx->get() : pgslam_amd::default_context();
            DataPoints out(input);
            double Tm[16];
            pgslam_amd::to_row_major16(parameters, Tm);
            const int n = (int)input.getNbPoints();
            if (n == 0) return out;
            // `out` is a copy of `input`: transformed in place (the call uploads its data once, not input AND output)
            check(c, A::transform(c, Tm, out.features.data(), out.features.rows(), out.features.data(), out.features.rows(), n, 0));
            for (const char *name : {"normals", "observationDirections"}) {
                if (!input.descriptorExists(name) || input.getDescriptorDimension(name) != 3) continue;
                const int row = input.getDescriptorStartingRow(name);
                check(c, A::transform(c, Tm, out.descriptors.data() + row, out.descriptors.rows(), out.descriptors.data() + row,
                                      out.descriptors.rows(), n, 1));
            }
            return out;
        }
    };
    struct Transformations : std::vector<std::shared_ptr<Transformation>> {
        //! LoopCloser.hpp:352 -- temp_icp.transformations.apply(reading, T)
        void apply(DataPoints &cloud, const TransformationParameters &parameters) const
        {
            DataPoints t(cloud);
            for (auto &tr : *this) t = tr->compute(t, parameters);
            cloud = t;
        }
    };

    // ------------------------------------------------------------------ DataPointsFilters
    struct DataPointsFilter {
        virtual ~DataPointsFilter() {}
        virtual void init() {}
        virtual void inPlaceFilter(DataPoints &cloud) = 0;
        //! the filter as an entry of pgicp_filter_cloud's list (the device pass keeps the points this host version keeps);
        //! false: the filter has no device form (it adds descriptors, or needs neighbours)
        virtual bool deviceSpec(pgicp_filter &) const { return false; }
        //! called once after a device pass that stood in for inPlaceFilter (filters whose parameters evolve from cloud to cloud)
        virtual void afterDevicePass() {}
    };
    struct IdentityDataPointsFilter : DataPointsFilter {
        void inPlaceFilter(DataPoints &) override {}
        bool deviceSpec(pgicp_filter &f) const override { std::memset(&f, 0, sizeof f); f.type = PGICP_FILTER_IDENTITY; return true; }
    };
    //! keeps the columns for which `keep(j)` holds, in order (features and descriptors)
    template <typename Pred>
    static void compactColumns(DataPoints &c, Pred keep)
    {
        const int n = (int)c.features.cols();
        int k = 0;
        for (int j = 0; j < n; j++) {
            if (!keep(j)) continue;
            if (k != j) {
                for (int i = 0; i < c.features.rows(); i++) c.features(i, k) = c.features(i, j);
                for (int i = 0; i < c.descriptors.rows(); i++) c.descriptors(i, k) = c.descriptors(i, j);
            }
            k++;
        }
        c.features.conservativeResize(c.features.rows(), k);
        if (c.descriptors.rows()) c.descriptors.conservativeResize(c.descriptors.rows(), k);
    }
    //! [EXT] MaxDistDataPointsFilter{maxDist, dim} / MinDistDataPointsFilter{minDist, dim} (DataPointsFilters/MaxDist.cpp, MinDist.cpp):
    //! dim = -1: keeps points whose distance to the origin -- features.col(i).head(3).norm() in T -- is below |maxDist| (above
    //! |minDist|), strictly; dim = 0..2: keeps features(dim, i) < maxDist (> minDist).  A NaN fails either comparison.
    struct DistLimitDataPointsFilter : DataPointsFilter {
        T limit; bool keepInside; int dim;
        DistLimitDataPointsFilter(T l, bool inside, int d = -1) : limit(l), keepInside(inside), dim(d) {}
        bool deviceSpec(pgicp_filter &f) const override
        {
            std::memset(&f, 0, sizeof f);
            f.type = keepInside ? PGICP_FILTER_MAX_DIST : PGICP_FILTER_MIN_DIST; f.p[0] = (double)limit; f.p[1] = (double)(dim + 1);
            return true;
        }
        void inPlaceFilter(DataPoints &c) override
        {
            const T al = limit < (T)0 ? -limit : limit;
            compactColumns(c, [&](int j) {
                if (dim >= 0) return keepInside ? c.features(dim, j) < limit : c.features(dim, j) > limit;
                const T x = c.features(0, j), y = c.features(1, j), z = c.features(2, j);
                const T r = std::sqrt((x * x + y * y) + z * z);          // Eigen's .norm() of the three coordinates, in T
                return keepInside ? r < al : r > al;
            });
        }
    };
    //! [EXT] BoundingBoxDataPointsFilter{xMin..zMax, removeInside}: a point is inside when every coordinate lies
    //! strictly between its bounds; removeInside = 1 (the default) drops the inside, 0 keeps only the inside
    struct BoundingBoxDataPointsFilter : DataPointsFilter {
        T lo[3], hi[3]; bool removeInside;
        BoundingBoxDataPointsFilter(const T l[3], const T h[3], bool rm) : removeInside(rm) { for (int a = 0; a < 3; a++) { lo[a] = l[a]; hi[a] = h[a]; } }
        bool deviceSpec(pgicp_filter &f) const override
        {
            std::memset(&f, 0, sizeof f);
            f.type = PGICP_FILTER_BOUNDING_BOX;
            for (int a = 0; a < 3; a++) { f.p[a] = (double)lo[a]; f.p[3 + a] = (double)hi[a]; }
            f.p[6] = removeInside ? 1.0 : 0.0;
            return true;
        }
        void inPlaceFilter(DataPoints &c) override
        {
            compactColumns(c, [&](int j) {
                bool in = true;
                for (int a = 0; a < 3; a++) in = in && lo[a] < c.features(a, j) && c.features(a, j) < hi[a];
                return in != removeInside;
            });
        }
    };
    //! [EXT] RemoveNaNDataPointsFilter: drops points with a NaN coordinate
    struct RemoveNaNDataPointsFilter : DataPointsFilter {
        bool deviceSpec(pgicp_filter &f) const override { std::memset(&f, 0, sizeof f); f.type = PGICP_FILTER_REMOVE_NAN; return true; }
        void inPlaceFilter(DataPoints &c) override
        {
            compactColumns(c, [&](int j) {
                for (int i = 0; i < c.features.rows(); i++) if (c.features(i, j) != c.features(i, j)) return false;
                return true;
            });
        }
    };
    //! [EXT] ObservationDirectionDataPointsFilter{x, y, z}: descriptor observationDirections = sensor position - point
    struct ObservationDirectionDataPointsFilter : DataPointsFilter {
        T c0[3];
        ObservationDirectionDataPointsFilter(T x, T y, T z) { c0[0] = x; c0[1] = y; c0[2] = z; }
        void inPlaceFilter(DataPoints &c) override
        {
            const int n = (int)c.features.cols();
            Matrix d(3, n);
            for (int j = 0; j < n; j++)
                for (int a = 0; a < 3; a++) d(a, j) = c0[a] - c.features(a, j);
            c.addDescriptor("observationDirections", d);
        }
    };
    //! [EXT] OrientNormalsDataPointsFilter{towardCenter}: flips every normal that points away from (towardCenter = 1)
    //! or toward (0) the sensor; needs the normals and observationDirections descriptors
    struct OrientNormalsDataPointsFilter : DataPointsFilter {
        bool towardCenter;
        explicit OrientNormalsDataPointsFilter(bool t) : towardCenter(t) {}
        void inPlaceFilter(DataPoints &c) override
        {
            if (!c.descriptorExists("normals")) throw std::runtime_error("OrientNormalsDataPointsFilter: cannot find normals in descriptors");
            if (!c.descriptorExists("observationDirections")) throw std::runtime_error("OrientNormalsDataPointsFilter: cannot find observation directions in descriptors");
            const int rn = c.getDescriptorStartingRow("normals"), ro = c.getDescriptorStartingRow("observationDirections");
            const int n = (int)c.features.cols();
            for (int j = 0; j < n; j++) {
                T dot = 0;
                for (int a = 0; a < 3; a++) dot += c.descriptors(rn + a, j) * c.descriptors(ro + a, j);
                if (towardCenter ? dot < 0 : dot > 0)
                    for (int a = 0; a < 3; a++) c.descriptors(rn + a, j) = -c.descriptors(rn + a, j);
            }
        }
    };
    //! [EXT] ShadowDataPointsFilter{eps} (DataPointsFilters/Shadow.cpp as restated in oracle/icp_oracle.c: orc_shadow_keep): keeps a
    //! point while |normal.normalized() . position.normalized()| > sin(eps) -- drops surfaces seen at a grazing angle from the sensor's
    //! origin.  Needs the normals descriptor; host side (it reads a descriptor the device input stage does not carry).
    struct ShadowDataPointsFilter : DataPointsFilter {
        T epsAngle, eps;
        explicit ShadowDataPointsFilter(T e) : epsAngle(e), eps(std::sin(e)) {}
        void inPlaceFilter(DataPoints &c) override
        {
            if (!c.descriptorExists("normals")) throw std::runtime_error("ShadowDataPointsFilter: cannot find normals in descriptors");
            const int rn = c.getDescriptorStartingRow("normals");
            auto normalized = [](T &x, T &y, T &z) { const T zz = (x * x + y * y) + z * z; if (zz > T(0)) { const T s = std::sqrt(zz); x = x / s; y = y / s; z = z / s; } };
            compactColumns(c, [&](int j) {
                T ax = c.descriptors(rn, j), ay = c.descriptors(rn + 1, j), az = c.descriptors(rn + 2, j);
                T bx = c.features(0, j), by = c.features(1, j), bz = c.features(2, j);
                normalized(ax, ay, az);
                normalized(bx, by, bz);
                const T d = (ax * bx + ay * by) + az * bz;
                return (d < T(0) ? -d : d) > eps;
            });
        }
    };
    //! [EXT] FixStepSamplingDataPointsFilter{startStep, endStep, stepMult} (DataPointsFilters/FixStepSampling.cpp): keeps points
    //! phase, phase + step, ...; after every cloud step *= stepMult, clamped at endStep in the direction of travel; init() goes back
    //! to startStep.  Upstream draws phase = rand() % step from the C library's global generator -- like RandomSampling there is
    //! nothing to be bit-exact against; here the phase is 0 (documented as NOT parity with upstream's sample, same density).
    struct FixStepSamplingDataPointsFilter : DataPointsFilter {
        double startStep, endStep, stepMult, step;
        explicit FixStepSamplingDataPointsFilter(double s, double e = -1, double m = 1) : startStep(s < 1 ? 1 : s), endStep(e < 1 ? startStep : e), stepMult(m), step(startStep) {}
        void init() override { step = startStep; }
        bool deviceSpec(pgicp_filter &f) const override { std::memset(&f, 0, sizeof f); f.type = PGICP_FILTER_FIX_STEP; f.p[0] = (double)(int)step; return true; }
        void afterDevicePass() override { advance(); }
        void advance()
        {
            const double delta = startStep * stepMult - startStep;
            step *= stepMult;
            if (delta < 0 && step < endStep) step = endStep;
            if (delta > 0 && step > endStep) step = endStep;
        }
        void inPlaceFilter(DataPoints &c) override
        {
            const int iStep = (int)step;
            compactColumns(c, [&](int j) { return j % iStep == 0; });
            advance();
        }
    };
    //! [EXT] RandomSamplingDataPointsFilter{prob}: upstream keeps a point when rand() / RAND_MAX < prob -- the C library's
    //! global generator, so no two runs (or platforms) agree and there is nothing to be bit-exact against.  Here the draw is
    //! a build-owned counter-based generator (SplitMix64 of seed and point index): the sample is reproducible, has the same
    //! distribution, and is documented as NOT bit-parity with upstream.  `seed` is an extra, optional parameter.
    struct RandomSamplingDataPointsFilter : DataPointsFilter {
        T prob; unsigned long long seed;
        RandomSamplingDataPointsFilter(T p, unsigned long long s) : prob(p), seed(s) {}
        bool deviceSpec(pgicp_filter &f) const override
        {
            if (seed >= (1ULL << 53)) return false;              // (the seed crosses the ABI as a double)
            std::memset(&f, 0, sizeof f);
            f.type = PGICP_FILTER_RANDOM_SAMPLING; f.p[0] = (double)prob; f.p[1] = (double)seed;
            return true;
        }
        static unsigned long long mix(unsigned long long z)
        {
            z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            return z ^ (z >> 31);
        }
        void inPlaceFilter(DataPoints &c) override
        {
            compactColumns(c, [&](int j) { return (double)(mix(seed * 0x100000001B3ULL + (unsigned long long)j) >> 11) / 9007199254740992.0 < (double)prob; });
        }
    };
    //! [EXT] MaxPointCountDataPointsFilter{maxCount, seed} (DataPointsFilters/MaxPointCount.cpp as of the libpointmatcher pgslam was
    //! written against): when more than maxCount points arrive, a random sample with prob = T(maxCount) / T(N) -- about maxCount
    //! points survive.  The draw is RandomSampling's build-owned one (not rand()-parity, see there).
    struct MaxPointCountDataPointsFilter : DataPointsFilter {
        unsigned maxCount; unsigned long long seed;
        MaxPointCountDataPointsFilter(unsigned m, unsigned long long s) : maxCount(m < 1 ? 1 : m), seed(s) {}
        bool deviceSpec(pgicp_filter &f) const override
        {
            if (seed >= (1ULL << 53)) return false;
            std::memset(&f, 0, sizeof f);
            f.type = PGICP_FILTER_MAX_POINT_COUNT; f.p[0] = (double)maxCount; f.p[1] = (double)seed;
            return true;
        }
        void inPlaceFilter(DataPoints &c) override
        {
            const int n = (int)c.features.cols();
            if (!((double)n > (double)maxCount)) return;
            const double prob = (double)((T)maxCount / (T)n);
            compactColumns(c, [&](int j) { return (double)(RandomSamplingDataPointsFilter::mix(seed * 0x100000001B3ULL + (unsigned long long)j) >> 11) / 9007199254740992.0 < prob; });
        }
    };
    //! [EXT] SurfaceNormalDataPointsFilter{knn, maxDist, epsilon, keepNormals, keepEigenValues}: normals (and,
    //! on request, eigenvalues) of every point from its knn neighbours, computed on the device by
    //! pgicp_surface_normals_*; the descriptors are appended as libpointmatcher appends them.
    struct SurfaceNormalDataPointsFilter : DataPointsFilter {
        int knn = 5; T maxDist = std::numeric_limits<T>::infinity(); bool keepNormals = true, keepEigenValues = false, keepDensities = false;
        pgslam_amd::LazyContext ctx;                 // made when the filter first runs, not when a YAML file is read
        SurfaceNormalDataPointsFilter(int k, T md, bool kn, bool ke, bool kd = false) : knn(k), maxDist(md), keepNormals(kn), keepEigenValues(ke), keepDensities(kd) {}
        SurfaceNormalDataPointsFilter(const SurfaceNormalDataPointsFilter &) = delete;
        SurfaceNormalDataPointsFilter &operator=(const SurfaceNormalDataPointsFilter &) = delete;
        void inPlaceFilter(DataPoints &c) override
        {
            const int n = (int)c.features.cols();
            if (n == 0 || (!keepNormals && !keepEigenValues && !keepDensities)) return;
            Matrix nrm(3, n), eig(3, n);
            std::vector<int32_t> ids(keepDensities ? (size_t)n * knn : 0);
            const double md = std::isfinite((double)maxDist) ? (double)maxDist : 1e300;
            check(ctx, pgslam_amd::Abi<T>::normals_ids(ctx, c.features.data(), (int)c.features.rows(), n, knn, md, nrm.data(), 3,
                                                       keepEigenValues ? eig.data() : nullptr, keepDensities ? ids.data() : nullptr));
            if (keepNormals) c.setDescriptor("normals", nrm);
            if (keepDensities) {
                // [EXT] computeDensity: neighbours found / volume of the sphere that holds them around their MEAN, in T (the oracle's
                // orc_densities); from the neighbour ids of the device search
                Matrix dens(1, n);
                for (int i = 0; i < n; i++) {
                    const int32_t *nb = ids.data() + (size_t)i * knn;
                    int cnt = 0;
                    T sx = 0, sy = 0, sz = 0;
                    for (int j = 0; j < knn; j++) if (nb[j] >= 0) { sx += c.features(0, nb[j]); sy += c.features(1, nb[j]); sz += c.features(2, nb[j]); cnt++; }
                    T r2 = 0;
                    if (cnt > 0) {
                        const T mx = sx / (T)cnt, my = sy / (T)cnt, mz = sz / (T)cnt;
                        for (int j = 0; j < knn; j++) if (nb[j] >= 0) {
                            const T dx = c.features(0, nb[j]) - mx, dy = c.features(1, nb[j]) - my, dz = c.features(2, nb[j]) - mz;
                            const T q = (dx * dx + dy * dy) + dz * dz;
                            if (q > r2) r2 = q;
                        }
                    }
                    const T r = std::sqrt(r2);
                    dens(0, i) = (T)cnt / ((T)((4.0 / 3.0) * 3.14159265358979323846) * ((r * r) * r));
                }
                c.setDescriptor("densities", dens);
            }
            if (keepEigenValues) c.setDescriptor("eigValues", eig);
        }
    };
    //! eigen-decomposition of a symmetric 3x3 (cyclic Jacobi in double): ev[k], columns V[.][k]
    static void symEigen3(double A[3][3], double V[3][3], double ev[3])
    {
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V[i][j] = i == j ? 1.0 : 0.0;
        for (int sweep = 0; sweep < 50; sweep++) {
            const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
            if (off == 0.0) break;
            for (int p = 0; p < 2; p++)
                for (int q = p + 1; q < 3; q++) {
                    if (A[p][q] == 0.0) continue;
                    const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                    const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                    for (int r = 0; r < 3; r++) { const double a = A[r][p], b = A[r][q]; A[r][p] = cs * a - sn * b; A[r][q] = sn * a + cs * b; }
                    for (int r = 0; r < 3; r++) { const double a = A[p][r], b = A[q][r]; A[p][r] = cs * a - sn * b; A[q][r] = sn * a + cs * b; }
                    for (int r = 0; r < 3; r++) { const double a = V[r][p], b = V[r][q]; V[r][p] = cs * a - sn * b; V[r][q] = sn * a + cs * b; }
                }
        }
        for (int i = 0; i < 3; i++) ev[i] = A[i][i];
    }
    //! [EXT] SamplingSurfaceNormalDataPointsFilter{ratio, knn, samplingMethod, maxBoxDim, averageExistingDescriptors, keepNormals}
    //! (DataPointsFilters/SamplingSurfaceNormal.cpp; upstream's setDefault() reference filter): the cloud is cut at the median of
    //! its widest dimension until a range holds at most knn points; each such box gets ONE normal (PCA of its points) and keeps
    //! every point with probability `ratio` (samplingMethod 0) or one point moved to its mean, the existing descriptors
    //! averaged (samplingMethod 1); boxes larger than maxBoxDim or of rank < 2 are dropped; the kept points stay in cloud order.
    //! The statement is the oracle's (oracle/icp_oracle.c orc_sampling_surface_normal, which says what is upstream's and what
    //! cannot be: nth_element's arrangement -> a sort on (coordinate, index); rand() -> the seeded draw; EigenSolver -> Jacobi).
    //! Host side: it runs once per keyframe / map, on the reference (Localizer.hpp:314-315).
    struct SamplingSurfaceNormalDataPointsFilter : DataPointsFilter {
        T ratio; int knn, samplingMethod; T maxBoxDim; bool averageExistingDescriptors, keepNormals; unsigned long long seed;
        SamplingSurfaceNormalDataPointsFilter(T r = T(0.5), int k = 7, int method = 0, T box = std::numeric_limits<T>::infinity(), bool avg = true,
                                              bool kn = true, unsigned long long sd = 1)
            : ratio(r), knn(k), samplingMethod(method), maxBoxDim(box), averageExistingDescriptors(avg), keepNormals(kn), seed(sd) {}
        void inPlaceFilter(DataPoints &c) override
        {
            const int n = (int)c.features.cols();
            if (n == 0) return;
            const int drows = (int)c.descriptors.rows();
            std::vector<int> idx(n);
            for (int i = 0; i < n; i++) idx[i] = i;
            std::vector<char> keep(n, 0);
            Matrix nrm(3, n);
            struct Range { int first, last; T lo[3], hi[3]; };
            std::vector<Range> stack;
            Range all{0, n, {c.features(0, 0), c.features(1, 0), c.features(2, 0)}, {c.features(0, 0), c.features(1, 0), c.features(2, 0)}};
            for (int i = 0; i < n; i++)
                for (int a = 0; a < 3; a++) { const T v = c.features(a, i); if (v < all.lo[a]) all.lo[a] = v; if (v > all.hi[a]) all.hi[a] = v; }
            stack.push_back(all);
            Matrix merged(drows > 0 ? drows : 1, 1);
            while (!stack.empty()) {
                const Range R = stack.back();
                stack.pop_back();
                const int count = R.last - R.first;
                if (count > knn) {
                    int cut = 0;
                    for (int a = 1; a < 3; a++) if (R.hi[a] - R.lo[a] > R.hi[cut] - R.lo[cut]) cut = a;
                    std::sort(idx.begin() + R.first, idx.begin() + R.last, [&](int i, int j) {
                        const T x = c.features(cut, i), y = c.features(cut, j);
                        return x < y || (x == y && i < j);
                    });
                    const int right = count / 2, left = count - right;
                    const T cv = c.features(cut, idx[R.first + left]);
                    Range L = R, G = R;
                    L.last = R.first + left; L.hi[cut] = cv;
                    G.first = R.first + left; G.lo[cut] = cv;
                    stack.push_back(G);                  // (the left half is taken first; boxes are independent of one another)
                    stack.push_back(L);
                    continue;
                }
                // ---- a box
                T lo[3], hi[3], sum[3] = {0, 0, 0};
                for (int a = 0; a < 3; a++) { lo[a] = c.features(a, idx[R.first]); hi[a] = lo[a]; }
                for (int k = R.first; k < R.last; k++)
                    for (int a = 0; a < 3; a++) { const T v = c.features(a, idx[k]); if (v < lo[a]) lo[a] = v; if (v > hi[a]) hi[a] = v; sum[a] += v; }
                T box = hi[0] - lo[0];
                if (hi[1] - lo[1] > box) box = hi[1] - lo[1];
                if (hi[2] - lo[2] > box) box = hi[2] - lo[2];
                if (box > maxBoxDim) continue;
                const T mx = sum[0] / (T)count, my = sum[1] / (T)count, mz = sum[2] / (T)count;
                T c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
                for (int k = R.first; k < R.last; k++) {
                    const int i = idx[k];
                    const T dx = c.features(0, i) - mx, dy = c.features(1, i) - my, dz = c.features(2, i) - mz;
                    c00 += dx * dx; c01 += dx * dy; c02 += dx * dz; c11 += dy * dy; c12 += dy * dz; c22 += dz * dz;
                }
                double A[3][3] = {{(double)c00, (double)c01, (double)c02}, {(double)c01, (double)c11, (double)c12}, {(double)c02, (double)c12, (double)c22}}, V[3][3], ev[3];
                symEigen3(A, V, ev);
                int l = 0, h = 0;
                for (int k = 1; k < 3; k++) { if (ev[k] < ev[l]) l = k; if (ev[k] > ev[h]) h = k; }
                const int mid = 3 - l - h;
                if (l == h || !(ev[h] > 0.0) || !(ev[mid] > 3.0 * (double)std::numeric_limits<T>::epsilon() * ev[h])) continue;   // rank < 2
                if (samplingMethod == 0) {
                    for (int k = R.first; k < R.last; k++) {
                        const int i = idx[k];
                        if (!((double)(RandomSamplingDataPointsFilter::mix(seed * 0x100000001B3ULL + (unsigned long long)i) >> 11) / 9007199254740992.0 < (double)ratio)) continue;
                        keep[i] = 1;
                        nrm(0, i) = (T)V[0][l]; nrm(1, i) = (T)V[1][l]; nrm(2, i) = (T)V[2][l];
                    }
                } else {
                    const int i = idx[R.first];
                    keep[i] = 1;
                    nrm(0, i) = (T)V[0][l]; nrm(1, i) = (T)V[1][l]; nrm(2, i) = (T)V[2][l];
                    if (drows > 0 && averageExistingDescriptors) {
                        for (int r = 0; r < drows; r++) merged(r, 0) = 0;
                        for (int k = R.first; k < R.last; k++) for (int r = 0; r < drows; r++) merged(r, 0) += c.descriptors(r, idx[k]);
                        for (int r = 0; r < drows; r++) c.descriptors(r, i) = merged(r, 0) / (T)count;
                    }
                    c.features(0, i) = mx; c.features(1, i) = my; c.features(2, i) = mz;
                    if (c.features.rows() > 3) c.features(3, i) = T(1);
                }
            }
            if (keepNormals) c.setDescriptor("normals", nrm);
            compactColumns(c, [&](int j) { return keep[j] != 0; });
        }
    };
    //! [EXT] MaxDensityDataPointsFilter{maxDensity} (DataPointsFilters/MaxDensity.cpp): needs the `densities` descriptor
    //! (SurfaceNormalDataPointsFilter{keepDensities: 1}); keeps a point at or below maxDensity, a denser one with probability
    //! maxDensity / density -- times (1 - nbSaturatedPts / nbPointsIn) in INTEGER arithmetic for points at the cloud's largest
    //! density, as upstream writes it.  The draw is the build's seeded one (not rand()-parity, see RandomSampling).
    struct MaxDensityDataPointsFilter : DataPointsFilter {
        T maxDensity; unsigned long long seed;
        explicit MaxDensityDataPointsFilter(T d = T(10), unsigned long long s = 1) : maxDensity(d), seed(s) {}
        void inPlaceFilter(DataPoints &c) override
        {
            if (!c.descriptorExists("densities")) throw std::runtime_error("MaxDensityDataPointsFilter: Error, no densities found in descriptors.");
            const int n = (int)c.features.cols();
            if (n == 0) return;
            const int rd = c.getDescriptorStartingRow("densities");
            T last = c.descriptors(rd, 0);
            for (int i = 1; i < n; i++) if (c.descriptors(rd, i) > last) last = c.descriptors(rd, i);
            int saturated = 0;
            for (int i = 0; i < n; i++) saturated += c.descriptors(rd, i) == last;
            compactColumns(c, [&](int j) {
                const T density = c.descriptors(rd, j);
                if (!(density > maxDensity)) return true;
                float accept = (float)(maxDensity / density);
                if (density == last) accept = accept * (float)(1 - saturated / n);
                return (double)(RandomSamplingDataPointsFilter::mix(seed * 0x100000001B3ULL + (unsigned long long)j) >> 11) / 9007199254740992.0 < (double)accept;
            });
        }
    };
    //! [EXT] SimpleSensorNoiseDataPointsFilter{sensorType, gain} (DataPointsFilters/SimpleSensorNoise.cpp, restated as recalled):
    //! adds the one-row descriptor `simpleSensorNoise` = gain * max(minRadius, beamAngle * |p| + beamConst) -- sensorType 0 Sick
    //! LMS-1xx, 1 Hokuyo URG-04LX, 2 Hokuyo UTM-30LX, 4 Sick Tim3xx -- or gain * |p|^2 * 0.5 * 0.00285 for 3 (Kinect / Xtion);
    //! ErrorMinimizer::getOverlap() reads it (Localizer.hpp:278, LoopCloser.hpp:331).  oracle: orc_simple_sensor_noise
    struct SimpleSensorNoiseDataPointsFilter : DataPointsFilter {
        int sensorType; T gain;
        explicit SimpleSensorNoiseDataPointsFilter(int t = 0, T g = T(1)) : sensorType(t), gain(g)
        {
            if (t < 0 || t > 4) throw std::runtime_error("SimpleSensorNoiseDataPointsFilter: sensorType must be 0 ... 4");
        }
        void inPlaceFilter(DataPoints &c) override
        {
            static const T prm[5][3] = {{T(0.012), T(0.0068), T(0.0008)}, {T(0.028), T(0.0013), T(0.0001)}, {T(0.018), T(0.0006), T(0.0015)},
                                        {T(0), T(0), T(0)}, {T(0.004), T(0.0053), T(-0.0092)}};
            const int n = (int)c.features.cols();
            Matrix noise(1, n);
            for (int i = 0; i < n; i++) {
                const T x = c.features(0, i), y = c.features(1, i), z = c.features(2, i);
                const T r = std::sqrt((x * x + y * y) + z * z);
                T v;
                if (sensorType == 3) v = (r * r) * (T)(0.5 * 0.00285);
                else { v = prm[sensorType][1] * r + prm[sensorType][2]; if (v < prm[sensorType][0]) v = prm[sensorType][0]; }
                noise(0, i) = gain * v;
            }
            c.setDescriptor("simpleSensorNoise", noise);
        }
    };
    struct DataPointsFilters : std::vector<std::shared_ptr<DataPointsFilter>> {
        DataPointsFilters() {}
        //! Localizer.hpp:77 -- a YAML list of filters
        explicit DataPointsFilters(std::istream &in)
        {
            std::stringstream ss;
            ss << "filters:\n" << in.rdbuf();          // a bare sequence: give it a section name
            const auto chain = pgslam_amd::yaml_lite::parse(ss);
            if (chain.has("filters")) load(chain.sections.at("filters"));
        }
        void load(const std::vector<pgslam_amd::yaml_lite::Module> &mods)
        {
            using pgslam_amd::yaml_lite::to_double;
            for (auto &m : mods) {
                if (m.name == "IdentityDataPointsFilter") this->push_back(std::make_shared<IdentityDataPointsFilter>());
                else if (m.name == "MaxDistDataPointsFilter" || m.name == "MinDistDataPointsFilter") {
                    const int dim = m.params.count("dim") ? (int)to_double(m.params.at("dim"), m.name) : -1;
                    if (dim < -1 || dim > 2) throw std::runtime_error(m.name + ": dim must be -1 (radius), 0, 1 or 2");
                    const T lim = (T)to_double(m.params.count("maxDist") ? m.params.at("maxDist") : m.params.count("minDist") ? m.params.at("minDist") : "1", m.name);
                    this->push_back(std::make_shared<DistLimitDataPointsFilter>(lim, m.name == "MaxDistDataPointsFilter", dim));
                } else if (m.name == "SurfaceNormalDataPointsFilter") {
                    auto get = [&](const char *k, const char *def) { return m.params.count(k) ? m.params.at(k) : std::string(def); };
                    // (epsilon > 0 allows an approximate neighbour search upstream; the exact search here meets every allowance)
                    if (!(to_double(get("epsilon", "0"), m.name) >= 0.0)) throw std::runtime_error(m.name + ": epsilon must be >= 0");
                    for (const char *k : {"keepEigenVectors", "keepMatchedIds", "keepMeanDist", "smoothNormals"})
                        if (to_double(get(k, "0"), m.name) != 0.0) throw std::runtime_error(m.name + ": " + k + " is not supported");
                    const int knn = (int)to_double(get("knn", "5"), m.name);
                    if (knn < 3 || knn > 32) throw std::runtime_error(m.name + ": knn must be in [3, 32]");
                    this->push_back(std::make_shared<SurfaceNormalDataPointsFilter>(knn, (T)to_double(get("maxDist", "inf"), m.name),
                                                                                     to_double(get("keepNormals", "1"), m.name) != 0.0,
                                                                                     to_double(get("keepEigenValues", "0"), m.name) != 0.0,
                                                                                     to_double(get("keepDensities", "0"), m.name) != 0.0));
                } else if (m.name == "SamplingSurfaceNormalDataPointsFilter") {
                    auto get = [&](const char *k, const char *def) { return m.params.count(k) ? m.params.at(k) : std::string(def); };
                    for (const char *k : {"keepDensities", "keepEigenValues", "keepEigenVectors"})
                        if (to_double(get(k, "0"), m.name) != 0.0) throw std::runtime_error(m.name + ": " + k + " is not supported");
                    const double ratio = to_double(get("ratio", "0.5"), m.name), seed = to_double(get("seed", "1"), m.name);
                    const int knn = (int)to_double(get("knn", "7"), m.name), method = (int)to_double(get("samplingMethod", "0"), m.name);
                    if (!(ratio > 0.0 && ratio < 1.0) || knn < 3 || (method != 0 && method != 1) || !(seed >= 0.0 && seed < 9007199254740992.0))
                        throw std::runtime_error(m.name + ": ratio must be in (0, 1), knn >= 3, samplingMethod 0 or 1, seed in [0, 2^53)");
                    this->push_back(std::make_shared<SamplingSurfaceNormalDataPointsFilter>((T)ratio, knn, method, (T)to_double(get("maxBoxDim", "inf"), m.name),
                                                                                             to_double(get("averageExistingDescriptors", "1"), m.name) != 0.0,
                                                                                             to_double(get("keepNormals", "1"), m.name) != 0.0, (unsigned long long)seed));
                } else if (m.name == "MaxDensityDataPointsFilter") {
                    auto get = [&](const char *k, const char *def) { return to_double(m.params.count(k) ? m.params.at(k) : std::string(def), m.name); };
                    const double md = get("maxDensity", "10"), seed = get("seed", "1");
                    if (!(md > 0.0) || !(seed >= 0.0 && seed < 9007199254740992.0)) throw std::runtime_error(m.name + ": maxDensity must be positive, seed in [0, 2^53)");
                    this->push_back(std::make_shared<MaxDensityDataPointsFilter>((T)md, (unsigned long long)seed));
                } else if (m.name == "SimpleSensorNoiseDataPointsFilter") {
                    auto get = [&](const char *k, const char *def) { return to_double(m.params.count(k) ? m.params.at(k) : std::string(def), m.name); };
                    const double st = get("sensorType", "0"), gain = get("gain", "1");
                    if (st != std::floor(st) || st < 0.0 || st > 4.0 || !(gain >= 1.0)) throw std::runtime_error(m.name + ": sensorType must be 0 ... 4, gain >= 1");
                    this->push_back(std::make_shared<SimpleSensorNoiseDataPointsFilter>((int)st, (T)gain));
                } else if (m.name == "BoundingBoxDataPointsFilter") {
                    auto get = [&](const char *k, const char *def) { return (T)to_double(m.params.count(k) ? m.params.at(k) : std::string(def), m.name); };
                    const T lo[3] = {get("xMin", "-1"), get("yMin", "-1"), get("zMin", "-1")}, hi[3] = {get("xMax", "1"), get("yMax", "1"), get("zMax", "1")};
                    this->push_back(std::make_shared<BoundingBoxDataPointsFilter>(lo, hi, get("removeInside", "1") != (T)0));
                } else if (m.name == "RemoveNaNDataPointsFilter") {
                    this->push_back(std::make_shared<RemoveNaNDataPointsFilter>());
                } else if (m.name == "ObservationDirectionDataPointsFilter") {
                    auto get = [&](const char *k) { return (T)to_double(m.params.count(k) ? m.params.at(k) : std::string("0"), m.name); };
                    this->push_back(std::make_shared<ObservationDirectionDataPointsFilter>(get("x"), get("y"), get("z")));
                } else if (m.name == "OrientNormalsDataPointsFilter") {
                    this->push_back(std::make_shared<OrientNormalsDataPointsFilter>(
                        to_double(m.params.count("towardCenter") ? m.params.at("towardCenter") : std::string("1"), m.name) != 0.0));
                } else if (m.name == "ShadowDataPointsFilter") {
                    const double e = to_double(m.params.count("eps") ? m.params.at("eps") : std::string("0.1"), m.name);
                    if (!(e >= 0.0 && e <= 3.14159265358979323846)) throw std::runtime_error(m.name + ": eps must be in [0, pi]");
                    this->push_back(std::make_shared<ShadowDataPointsFilter>((T)e));
                } else if (m.name == "FixStepSamplingDataPointsFilter") {
                    auto get = [&](const char *k, const char *def) { return to_double(m.params.count(k) ? m.params.at(k) : std::string(def), m.name); };
                    const double start = get("startStep", "10"), end = get("endStep", "10"), mult = get("stepMult", "1");
                    if (!(start >= 1.0 && start <= 2147483647.0) || !(end >= 1.0 && end <= 2147483647.0) || !(mult > 0.0))
                        throw std::runtime_error(m.name + ": startStep and endStep must be in [1, INT_MAX], stepMult positive");
                    this->push_back(std::make_shared<FixStepSamplingDataPointsFilter>(start, m.params.count("endStep") ? end : start, mult));
                } else if (m.name == "RandomSamplingDataPointsFilter") {
                    auto get = [&](const char *k, const char *def) { return to_double(m.params.count(k) ? m.params.at(k) : std::string(def), m.name); };
                    const double prob = get("prob", "0.75"), seed = get("seed", "1");
                    if (!(prob >= 0.0 && prob <= 1.0) || !(seed >= 0.0 && seed < 9007199254740992.0)) throw std::runtime_error(m.name + ": prob must be in [0, 1], seed in [0, 2^53)");
                    this->push_back(std::make_shared<RandomSamplingDataPointsFilter>((T)prob, (unsigned long long)seed));
                } else if (m.name == "MaxPointCountDataPointsFilter") {
                    auto get = [&](const char *k, const char *def) { return to_double(m.params.count(k) ? m.params.at(k) : std::string(def), m.name); };
                    const double mc = get("maxCount", "1000"), seed = get("seed", "1");
                    if (!(mc >= 1.0 && mc <= 2147483647.0) || !(seed >= 0.0 && seed < 9007199254740992.0)) throw std::runtime_error(m.name + ": maxCount must be in [1, INT_MAX], seed in [0, 2^53)");
                    this->push_back(std::make_shared<MaxPointCountDataPointsFilter>((unsigned)mc, (unsigned long long)seed));
                } else
                    throw std::runtime_error("DataPointsFilters: unsupported filter '" + m.name +
                                             "' (supported: Identity, MinDist, MaxDist, BoundingBox, RemoveNaN, SurfaceNormal, "
                                             "SamplingSurfaceNormal, MaxDensity, ObservationDirection, OrientNormals, Shadow, FixStepSampling, RandomSampling, MaxPointCount "
                                             "(seeded samplers, not rand()-parity))");
            }
        }
        void init() { for (auto &f : *this) f->init(); }
        void apply(DataPoints &cloud) { for (auto &f : *this) f->inPlaceFilter(cloud); }
        //! every filter of the list has a device form (and the list fits pgicp_filter_cloud): the list as its argument
        bool deviceSpecs(std::vector<pgicp_filter> &out) const
        {
            out.clear();
            if (this->size() > PGICP_MAX_FILTERS) return false;
            for (auto &f : *this) { pgicp_filter s; if (!f->deviceSpec(s)) return false; out.push_back(s); }
            return true;
        }
        bool allIdentity() const
        {
            for (auto &f : *this) if (!std::dynamic_pointer_cast<IdentityDataPointsFilter>(f)) return false;
            return true;
        }
    };
    //! Localizer.hpp:103-106 in ONE device pass (pgicp_filter_cloud): filters.apply(cloud), in place, then
    //! cloud = RigidTransformation::compute(cloud, T).  `dev` receives the device address of the resulting features (a reading
    //! that needs no second upload; valid for the next three calls on `c`).  false: some filter has no device form, nothing
    //! was done (the caller applies the host filters and the transformation one after the other).
    static bool filterAndTransformOnDevice(pgicp_ctx *c, const DataPointsFilters &filters, DataPoints &cloud, const TransformationParameters &Tm,
                                           const T **dev = nullptr)
    {
        std::vector<pgicp_filter> specs;
        const int n = (int)cloud.getNbPoints();
        if (n == 0 || cloud.features.rows() < 3 || !filters.deviceSpecs(specs)) return false;
        RigidTransformation check_t;
        if (!check_t.checkParameters(Tm)) throw std::runtime_error("RigidTransformation: the transformation is not rigid");
        double T16[16];
        pgslam_amd::to_row_major16(Tm, T16);
        const int drows = (int)cloud.descriptors.rows();
        const int r0 = cloud.descriptorExists("normals") && cloud.getDescriptorDimension("normals") == 3 ? cloud.getDescriptorStartingRow("normals") : -1;
        const int r1 = cloud.descriptorExists("observationDirections") && cloud.getDescriptorDimension("observationDirections") == 3
                           ? cloud.getDescriptorStartingRow("observationDirections") : -1;
        int kept = 0;
        const T *d = nullptr;
        const int rc = sizeof(T) == 4
            ? pgicp_filter_cloud_f32(c, (int)specs.size(), specs.data(), (const float *)cloud.features.data(), (int)cloud.features.rows(),
                                     drows ? (const float *)cloud.descriptors.data() : nullptr, drows, n, T16, r0, r1, (float *)cloud.features.data(),
                                     drows ? (float *)cloud.descriptors.data() : nullptr, nullptr, &kept, (const float **)&d)
            : pgicp_filter_cloud_f64(c, (int)specs.size(), specs.data(), (const double *)cloud.features.data(), (int)cloud.features.rows(),
                                     drows ? (const double *)cloud.descriptors.data() : nullptr, drows, n, T16, r0, r1, (double *)cloud.features.data(),
                                     drows ? (double *)cloud.descriptors.data() : nullptr, nullptr, &kept, (const double **)&d);
        check(c, rc);
        for (auto &f : filters) f->afterDevicePass();               // (a FixStep filter's step moves on, as after inPlaceFilter)
        cloud.features.conservativeResize(cloud.features.rows(), kept);
        if (drows) cloud.descriptors.conservativeResize(drows, kept);
        if (dev) *dev = d;
        return true;
    }

    //! closes the gaps of the dropped columns (ascending indices) in features and descriptors: what the filters' compactColumns
    //! leaves, as a few block moves
    static void compactByDropped(DataPoints &c, const std::vector<int32_t> &dropped)
    {
        const int n = (int)c.features.cols();
        if (dropped.empty()) return;
        const size_t fr = (size_t)c.features.rows(), dr = (size_t)c.descriptors.rows();
        T *f = c.features.data(), *d = dr ? c.descriptors.data() : nullptr;
        int dst = dropped[0];
        for (size_t k = 0; k < dropped.size(); k++) {
            const int from = dropped[k] + 1, to = k + 1 < dropped.size() ? dropped[k + 1] : n;      // the kept run [from, to)
            if (to > from) {
                std::memmove(f + (size_t)dst * fr, f + (size_t)from * fr, sizeof(T) * fr * (size_t)(to - from));
                if (d) std::memmove(d + (size_t)dst * dr, d + (size_t)from * dr, sizeof(T) * dr * (size_t)(to - from));
                dst += to - from;
            }
        }
        c.features.conservativeResize(c.features.rows(), dst);
        if (dr) c.descriptors.conservativeResize((int)dr, dst);
    }
    //! The device half of filterAndTransformOnDevice for a cloud that needs NO transformation (`Tm` the identity): the filters run
    //! on the uploaded features, `dev` / `kept` describe the filtered device copy, `dropped` lists what the HOST cloud still holds
    //! too much (ascending) -- the caller applies compactByDropped(cloud, dropped) when it suits it, e.g. on another thread while
    //! the ICP runs on the device copy.  false: not applicable (a transformation, a filter without a device form, more than 4 096
    //! points dropped): nothing was done, the caller takes filterAndTransformOnDevice.
    static bool filterOnDeviceDeferred(pgicp_ctx *c, const DataPointsFilters &filters, const DataPoints &cloud, const TransformationParameters &Tm,
                                       const T **dev, int *kept, std::vector<int32_t> &dropped)
    {
        std::vector<pgicp_filter> specs;
        const int n = (int)cloud.getNbPoints();
        if (n == 0 || cloud.features.rows() < 3 || !filters.deviceSpecs(specs)) return false;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) if (Tm(i, j) != (i == j ? T(1) : T(0))) return false;
        dropped.resize(4096);
        int nd = 0;
        const int rc = sizeof(T) == 4
            ? pgicp_filter_cloud_dev_f32(c, (int)specs.size(), specs.data(), (const float *)cloud.features.data(), (int)cloud.features.rows(), n, dropped.data(),
                                         (int)dropped.size(), &nd, kept, (const float **)dev)
            : pgicp_filter_cloud_dev_f64(c, (int)specs.size(), specs.data(), (const double *)cloud.features.data(), (int)cloud.features.rows(), n, dropped.data(),
                                         (int)dropped.size(), &nd, kept, (const double **)dev);
        check(c, rc);
        if (nd > (int)dropped.size()) { dropped.clear(); return false; }       // (the filters' state has not moved: afterDevicePass was not called)
        dropped.resize((size_t)nd);
        for (auto &f : filters) f->afterDevicePass();
        return true;
    }

    struct ICPChainBase;

    // ------------------------------------------------------------------ Matcher
    struct Matcher {
        ICPChainBase *chain;
        int knn = 1; T epsilon = 0; T maxDist = std::numeric_limits<T>::infinity();
        int mapId = -1; int mapSize = 0;
        explicit Matcher(ICPChainBase *c) : chain(c) {}
        virtual ~Matcher() { release(); }
        void release() { if (mapId >= 0 && chain && chain->ctx.made()) { pgicp_map_destroy(chain->ctx, mapId); mapId = -1; } }
        //! Localizer.hpp:317, LoopCloser.hpp:356 -- index over the cloud AS GIVEN (not centred)
        virtual void init(const DataPoints &filteredReference) { initImpl(filteredReference, 0); }
        void initImpl(const DataPoints &ref, int center)
        {
            release();
            chain->pushParams();
            check(chain->ctx, A::map_create(chain->ctx, ref.xyzPtr(), ref.xyzStride(), ref.normalsPtr(), ref.normalsStride(), (int)ref.getNbPoints(), center, &mapId));
            mapSize = (int)ref.getNbPoints();
        }
        //! the same over a cloud that is in device memory already (DeviceCloud): no upload
        void initDevice(const pgslam_amd::DeviceCloud<T> &ref, int center)
        {
            release();
            chain->pushParams();
            check(chain->ctx, A::map_create_dev(chain->ctx, ref.xyz, ref.xs, ref.hasNormals() ? ref.nrm : nullptr, ref.ns, ref.n, center, &mapId));
            mapSize = ref.n;
        }
        //! Localizer.hpp:328, LoopCloser.hpp:358
        virtual Matches findClosests(const DataPoints &filteredReading)
        {
            if (mapId < 0) throw std::runtime_error("Matcher::findClosests: init() was not called");
            const int n = (int)filteredReading.getNbPoints();
            Matches m(knn < 1 ? 1 : knn, n);                  // knn x N, column-major: [point][neighbour], as the ABI writes them
            chain->pushParams();
            check(chain->ctx, A::match(chain->ctx, mapId, filteredReading.xyzPtr(), filteredReading.xyzStride(), n, m.ids.data(), m.dists.data()));
            return m;
        }
    };

    // ------------------------------------------------------------------ OutlierFilters
    struct OutlierFilter {
        virtual ~OutlierFilter() {}
        virtual OutlierWeights compute(const DataPoints &reading, const DataPoints &reference, const Matches &input) = 0;
    };
    struct TrimmedDistOutlierFilter : OutlierFilter {
        ICPChainBase *chain; T ratio;
        TrimmedDistOutlierFilter(ICPChainBase *c, T r) : chain(c), ratio(r) {}
        OutlierWeights compute(const DataPoints &, const DataPoints &, const Matches &input) override
        {
            OutlierWeights w(input.dists.rows(), input.dists.cols());
            chain->pushParams();
            T lim; int nf;
            check(chain->ctx, A::weights(chain->ctx, input.dists.data(), (int)input.dists.size(), w.data(), &lim, &nf));
            return w;
        }
    };
    //! [EXT] MedianDistOutlierFilter{factor}: limit = factor * getDistsQuantile(0.5) on the SQUARED match distances, weight 1
    //! while dist <= limit (OutlierFiltersImpl.cpp as restated in oracle/icp_oracle.c: orc_median_weights).  On the device
    //! it is the trimmed filter's exact order statistic at ratio 0.5, scaled (pgicp_params.quantile_scale).
    struct MedianDistOutlierFilter : OutlierFilter {
        ICPChainBase *chain; T factor;
        MedianDistOutlierFilter(ICPChainBase *c, T f) : chain(c), factor(f) {}
        OutlierWeights compute(const DataPoints &, const DataPoints &, const Matches &input) override
        {
            OutlierWeights w(input.dists.rows(), input.dists.cols());
            chain->pushParams();
            T lim; int nf;
            check(chain->ctx, A::weights(chain->ctx, input.dists.data(), (int)input.dists.size(), w.data(), &lim, &nf));
            return w;
        }
    };
    //! [EXT] RobustOutlierFilter{robustFct, tuning, scaleEstimator, nbIterationForScale, distanceType, approximation}: an
    //! M-estimator weight per pair from e2 = dist / scale^2 (OutlierFiltersImpl.cpp as restated in oracle/icp_oracle.c:
    //! orc_robust_weights), scale = sqrt(median absolute deviation of the finite squared distances) or 1.  Supported as
    //! upstream's defaults have it: distanceType point2point and nbIterationForScale 0 (the scale re-estimated at every
    //! call); no TrimmedDist / MedianDist filter beside it, knn 1.  Inside an ICP run the device applies it
    //! (pgicp_params.robust_*); this stage-level call is pgicp_outlier_weights under the same parameters.
    struct RobustOutlierFilter : OutlierFilter {
        ICPChainBase *chain; std::string robustFct, scaleEstimator; T tuning, approximation;
        RobustOutlierFilter(ICPChainBase *c, const std::string &fct = "cauchy", T tun = T(1), const std::string &scale = "mad",
                            T approx = std::numeric_limits<T>::infinity())
            : chain(c), robustFct(fct), scaleEstimator(scale), tuning(tun), approximation(approx)
        {
            fctCode(); scaleCode();
            if (!(tuning > T(0))) throw std::runtime_error("RobustOutlierFilter: tuning must be positive");
        }
        int fctCode() const
        {
            static const char *names[] = {"cauchy", "welsch", "sc", "gm", "tukey", "huber", "L1"};
            for (int k = 0; k < 7; k++) if (robustFct == names[k]) return PGICP_ROBUST_CAUCHY + k;
            throw std::runtime_error("RobustOutlierFilter: unknown robustFct " + robustFct + " (cauchy, welsch, sc, gm, tukey, huber, L1)");
        }
        int scaleCode() const
        {
            if (scaleEstimator == "mad") return PGICP_ROBUST_SCALE_MAD;
            if (scaleEstimator == "none") return PGICP_ROBUST_SCALE_NONE;
            throw std::runtime_error("RobustOutlierFilter: scaleEstimator " + scaleEstimator + " is not supported (mad, none)");
        }
        OutlierWeights compute(const DataPoints &, const DataPoints &, const Matches &input) override
        {
            OutlierWeights w(input.dists.rows(), input.dists.cols());
            chain->pushParams();
            T lim; int nf;
            check(chain->ctx, A::weights(chain->ctx, input.dists.data(), (int)input.dists.size(), w.data(), &lim, &nf));
            return w;
        }
    };
    //! [EXT] MaxDistOutlierFilter{maxDist}: weight 1 while the squared match distance is <= maxDist^2, else 0
    struct MaxDistOutlierFilter : OutlierFilter {
        T maxDist;
        explicit MaxDistOutlierFilter(T d) : maxDist(d) {}
        OutlierWeights compute(const DataPoints &, const DataPoints &, const Matches &input) override
        {
            OutlierWeights w(input.dists.rows(), input.dists.cols());
            const T lim = maxDist * maxDist;
            for (int j = 0; j < w.cols(); j++) for (int i = 0; i < w.rows(); i++) w(i, j) = input.dists(i, j) <= lim ? T(1) : T(0);
            return w;
        }
    };
    //! [EXT] SurfaceNormalOutlierFilter{maxAngle}: weight 0 when the angle between the (normalised) normals of a reading
    //! point and of its matched reference point exceeds maxAngle, or the match is invalid; ones when either cloud has no
    //! `normals` descriptor.  Inside an ICP run the device applies it (pgicp_params.normal_max_angle); this
    //! host version serves the stage-level call of the partial chains (Localizer.hpp:330, LoopCloser.hpp:360).
    struct SurfaceNormalOutlierFilter : OutlierFilter {
        T maxAngle;
        explicit SurfaceNormalOutlierFilter(T a) : maxAngle(a) {}
        OutlierWeights compute(const DataPoints &reading, const DataPoints &reference, const Matches &input) override
        {
            OutlierWeights w = OutlierWeights::Constant(input.ids.rows(), input.ids.cols(), T(1));
            if (!reading.descriptorExists("normals") || !reference.descriptorExists("normals")) return w;
            const int rr = reading.getDescriptorStartingRow("normals"), rf = reference.getDescriptorStartingRow("normals");
            const T eps = std::cos(maxAngle);
            auto normalized = [](T &x, T &y, T &z) { const T zz = (x * x + y * y) + z * z; if (zz > T(0)) { const T s = std::sqrt(zz); x = x / s; y = y / s; z = z / s; } };
            for (int j = 0; j < w.cols(); j++) {
                T ax = reading.descriptors(rr, j), ay = reading.descriptors(rr + 1, j), az = reading.descriptors(rr + 2, j);
                normalized(ax, ay, az);
                for (int i = 0; i < w.rows(); i++) {
                    const int id = input.ids(i, j);
                    if (id < 0) { w(i, j) = T(0); continue; }
                    T bx = reference.descriptors(rf, id), by = reference.descriptors(rf + 1, id), bz = reference.descriptors(rf + 2, id);
                    normalized(bx, by, bz);
                    if (((ax * bx + ay * by) + az * bz) < eps) w(i, j) = T(0);
                }
            }
            return w;
        }
    };
    struct OutlierFilters : std::vector<std::shared_ptr<OutlierFilter>> {
        //! Localizer.hpp:330, LoopCloser.hpp:360 -- element-wise product of the filters' weights
        OutlierWeights compute(const DataPoints &reading, const DataPoints &reference, const Matches &input)
        {
            if (this->empty()) return OutlierWeights::Constant(input.ids.rows(), input.ids.cols(), T(1));
            OutlierWeights w = (*this)[0]->compute(reading, reference, input);
            for (size_t k = 1; k < this->size(); k++) {
                const OutlierWeights o = (*this)[k]->compute(reading, reference, input);
                for (int j = 0; j < w.cols(); j++) for (int i = 0; i < w.rows(); i++) w(i, j) *= o(i, j);
            }
            return w;
        }
    };

    // ------------------------------------------------------------------ ErrorMinimizer
    struct ErrorMinimizer {
        //! the matched, weighted pairs (Localizer.hpp:332): compaction of the non-zero-weight matches
        struct ErrorElements {
            DataPoints reading, reference;
            OutlierWeights weights;
            Matches matches;
            int nbRejectedMatches = 0, nbRejectedPoints = 0;
            T pointUsedRatio = 0, weightedPointUsedRatio = 0;
            ErrorElements() {}
            ErrorElements(const DataPoints &requestedPts, const DataPoints &sourcePts, const OutlierWeights &outlierWeights, const Matches &m)
            {
                const int knn = outlierWeights.rows(), n = outlierWeights.cols();
                int kept = 0; double wsum = 0;
                for (int k = 0; k < knn; k++) for (int i = 0; i < n; i++) if (outlierWeights(k, i) != T(0)) { kept++; wsum += (double)outlierWeights(k, i); }
                if (kept == 0) throw ConvergenceError("ErrorMnimizer: no point to minimize");
                reading.features = Matrix(requestedPts.features.rows(), kept); reading.featureLabels = requestedPts.featureLabels;
                reading.descriptors = Matrix(requestedPts.descriptors.rows(), kept); reading.descriptorLabels = requestedPts.descriptorLabels;
                reference.features = Matrix(sourcePts.features.rows(), kept); reference.featureLabels = sourcePts.featureLabels;
                reference.descriptors = Matrix(sourcePts.descriptors.rows(), kept); reference.descriptorLabels = sourcePts.descriptorLabels;
                weights = OutlierWeights(1, kept); matches = Matches(1, kept);
                int j = 0;
                for (int k = 0; k < knn; k++)
                    for (int i = 0; i < n; i++) {
                        if (outlierWeights(k, i) == T(0)) continue;
                        const int id = m.ids(k, i);
                        for (int r = 0; r < reading.features.rows(); r++) reading.features(r, j) = requestedPts.features(r, i);
                        for (int r = 0; r < reading.descriptors.rows(); r++) reading.descriptors(r, j) = requestedPts.descriptors(r, i);
                        for (int r = 0; r < reference.features.rows(); r++) reference.features(r, j) = sourcePts.features(r, id);
                        for (int r = 0; r < reference.descriptors.rows(); r++) reference.descriptors(r, j) = sourcePts.descriptors(r, id);
                        weights(0, j) = outlierWeights(k, i); matches.ids(0, j) = id; matches.dists(0, j) = m.dists(k, i);
                        j++;
                    }
                nbRejectedMatches = knn * n - kept;
                pointUsedRatio = (T)((double)kept / (double)(knn * n));
                weightedPointUsedRatio = (T)(wsum / (double)(knn * n));
            }
        };
        ICPChainBase *chain;
        T sensorStdDev = T(0.01);
        bool withCov = false;
        bool pointToPoint = false;       //!< PointToPointErrorMinimizer (Kabsch); else PointToPlane(WithCov)
        bool force4DOF = false;          //!< PointToPlane[WithCov]ErrorMinimizer{force4DOF: 1}: rotation about z + translation only
        // state of the last ICP run / compute()
        T lastOverlap = 0; T lastResidual = 0; Matrix lastCov = Matrix::Zero(6, 6);
        explicit ErrorMinimizer(ICPChainBase *c) : chain(c) {}
        virtual ~ErrorMinimizer() {}
        //! Localizer.hpp:278, LoopCloser.hpp:331: weightedPointUsedRatio of the last error elements (pgslam's own clouds carry no
        //! simpleSensorNoise descriptor -- SURVEY.md A.7); for a reading that does (and, with the point-to-plane minimizer, `normals`
        //! too: ErrorMinimizers/PointToPlane.cpp) the share of the last error elements whose distance lies below mean + noise
        //! (ICPChainBase::sensorNoiseOverlap; oracle: orc_sensor_noise_overlap)
        T getOverlap() const { return lastOverlap; }
        T getWeightedPointUsedRatio() const { return lastRatio; }
        T lastRatio = 0;
        //! Localizer.hpp:238, LoopCloser.hpp:108: 6x6 Censi covariance, order [x y z rx ry rz]; zeros without "WithCov"
        Matrix getCovariance() const { return withCov ? lastCov : Matrix::Zero(6, 6); }
        //! LoopCloser.hpp:362: sum w (n.(p-q))^2 -- `reference` must be the cloud given to matcher->init
        T getResidualError(const DataPoints &filteredReading, const DataPoints &, const OutlierWeights &w, const Matches &m) const
        {
            double ratio, res;
            if (!chain->matcher || chain->matcher->mapId < 0) throw std::runtime_error("getResidualError: matcher->init() was not called");
            // pgicp_error_stats reads n x knn ids and weights with the CONTEXT's knn: push the chain's parameters and refuse
            // matches made under another knn (an out-of-bounds host read otherwise -- ADVICE round 4)
            chain->pushParams();
            const int K = chain->matcher->knn < 1 ? 1 : chain->matcher->knn, n = (int)filteredReading.getNbPoints();
            if (m.ids.rows() != K || w.rows() != K || m.ids.cols() != n || w.cols() != n)
                throw std::runtime_error("getResidualError: matches / weights are not knn x N of this chain's matcher");
            check(chain->ctx, A::stats(chain->ctx, chain->matcher->mapId, filteredReading.xyzPtr(), filteredReading.xyzStride(),
                                       (int)filteredReading.getNbPoints(), m.ids.data(), w.data(), &ratio, &res, nullptr));
            return (T)res;
        }
    };

    // ------------------------------------------------------------------ TransformationCheckers
    struct TransformationChecker { virtual ~TransformationChecker() {} std::string name; };
    struct CounterTransformationChecker : TransformationChecker { int maxIterationCount = 40; };
    struct DifferentialTransformationChecker : TransformationChecker { T minDiffRotErr = T(0.001), minDiffTransErr = T(0.001); int smoothLength = 3; };
    //! [EXT] BoundTransformationChecker{maxRotationNorm, maxTranslationNorm}: ConvergenceError when the accumulated correction
    //! leaves the bound
    struct BoundTransformationChecker : TransformationChecker { T maxRotationNorm = T(1), maxTranslationNorm = T(1); };
    struct TransformationCheckers : std::vector<std::shared_ptr<TransformationChecker>> {};

    // ------------------------------------------------------------------ ICP chain
    struct ICPChainBase {
        DataPointsFilters readingDataPointsFilters, readingStepDataPointsFilters, referenceDataPointsFilters;
        Transformations transformations;
        std::shared_ptr<Matcher> matcher;
        OutlierFilters outlierFilters;
        std::shared_ptr<ErrorMinimizer> errorMinimizer;
        TransformationCheckers transformationCheckers;
        pgslam_amd::LazyContext ctx;                 // the chain's device context and stream, made at first use
        pgicp_stats lastStats;
        //! Observer of every completed alignment (not part of libpointmatcher; its `inspector` slot sees iterations, not
        //! calls): the filtered reading, the filtered reference the index was built on, the initial guess, the result and
        //! the device statistics.  Used to record ICP calls of a SLAM run for replay through the CPU oracle.
        std::function<void(const DataPoints &, const DataPoints &, const TransformationParameters &, const TransformationParameters &,
                           const pgicp_stats &)> onAlign;
        const DataPoints *currentReference = nullptr;
        //! optional gate of the observer: asked once per completed alignment BEFORE the observer's arguments are made; false
        //! skips the observer for that call (a map that lives only in device memory is downloaded for the observer's sake)
        std::function<bool()> onAlignWanted;
        //! the host copy of a device-resident reference, made when an observer asks for it (ICPSequence::setMap(DeviceCloud))
        std::function<DataPoints()> lazyReference;
        DataPoints lazyReferenceCloud;
        const DataPoints *observedReference()
        {
            if (currentReference) return currentReference;
            if (!lazyReference) return nullptr;
            lazyReferenceCloud = lazyReference();
            lazyReference = nullptr;
            currentReference = &lazyReferenceCloud;
            return currentReference;
        }

        explicit ICPChainBase(int device = 0) : ctx(device)
        {
            std::memset(&lastStats, 0, sizeof lastStats);
            // (upstream's constructor leaves the chain EMPTY -- an ICP object used without setDefault() / loadFromYaml() has no
            // matcher there; here it starts with the default matcher / outlier filter / minimiser / checkers and no data-point
            // filters, so that members can be set one by one; setDefault() is upstream's, data-point filters included)
            setDefaultWithoutDataPointsFilters();
        }
        ICPChainBase(const ICPChainBase &) = delete;
        ICPChainBase &operator=(const ICPChainBase &) = delete;
        virtual ~ICPChainBase()
        {
            matcher.reset();
            ctx.destroy();
        }
        void cleanup()
        {
            readingDataPointsFilters.clear(); readingStepDataPointsFilters.clear(); referenceDataPointsFilters.clear();
            transformations.clear(); outlierFilters.clear(); transformationCheckers.clear();
            matcher.reset(); errorMinimizer.reset();
        }
        //! libpointmatcher defaults ([EXT] ICP.cpp ICPChaineBase::setDefault): RandomSamplingDataPointsFilter on the
        //! reading, SamplingSurfaceNormalDataPointsFilter on the reference (where the default point-to-plane minimiser gets its
        //! normals from), KDTreeMatcher, TrimmedDist 0.85, PointToPlane, Counter(40) + Differential
        virtual void setDefault()
        {
            setDefaultWithoutDataPointsFilters();
            readingDataPointsFilters.push_back(std::make_shared<RandomSamplingDataPointsFilter>(T(0.75), 1ULL));
            referenceDataPointsFilters.push_back(std::make_shared<SamplingSurfaceNormalDataPointsFilter>());
        }
        void setDefaultWithoutDataPointsFilters()
        {
            cleanup();
            transformations.push_back(std::make_shared<RigidTransformation>(&ctx));
            matcher = std::make_shared<Matcher>(this);
            outlierFilters.push_back(std::make_shared<TrimmedDistOutlierFilter>(this, T(0.85)));
            errorMinimizer = std::make_shared<ErrorMinimizer>(this);
            transformationCheckers.push_back(std::make_shared<CounterTransformationChecker>());
            transformationCheckers.push_back(std::make_shared<DifferentialTransformationChecker>());
        }
        //! Localizer.hpp:70,311; LoopCloser.hpp:73,348
        virtual void loadFromYaml(std::istream &in)
        {
            using pgslam_amd::yaml_lite::to_double;
            const auto y = pgslam_amd::yaml_lite::parse(in);
            cleanup();
            transformations.push_back(std::make_shared<RigidTransformation>(&ctx));
            if (y.has("readingDataPointsFilters")) readingDataPointsFilters.load(y.sections.at("readingDataPointsFilters"));
            if (y.has("readingStepDataPointsFilters")) {
                // step filters are applied ONCE here (they are pure functions of the cloud); upstream's random sampler draws
                // a fresh sample in every iteration, which one fixed subsample does not reproduce (overlap and covariance
                // would be those of the subsample): refused in this slot, never silently approximated
                for (auto &m : y.sections.at("readingStepDataPointsFilters"))
                    if (m.name == "RandomSamplingDataPointsFilter")
                        throw std::runtime_error("loadFromYaml: RandomSamplingDataPointsFilter is not supported in readingStepDataPointsFilters "
                                                 "(per-iteration resampling); use it in readingDataPointsFilters");
                readingStepDataPointsFilters.load(y.sections.at("readingStepDataPointsFilters"));
            }
            if (y.has("referenceDataPointsFilters")) referenceDataPointsFilters.load(y.sections.at("referenceDataPointsFilters"));
            matcher = std::make_shared<Matcher>(this);
            if (y.has("matcher") && !y.sections.at("matcher").empty()) {
                const auto &m = y.sections.at("matcher")[0];
                if (m.name != "KDTreeMatcher") throw std::runtime_error("loadFromYaml: unsupported matcher " + m.name);
                for (auto &kv : m.params) {
                    if (kv.first == "knn") matcher->knn = (int)to_double(kv.second, "knn");
                    else if (kv.first == "epsilon") matcher->epsilon = (T)to_double(kv.second, "epsilon");
                    else if (kv.first == "maxDist") matcher->maxDist = (T)to_double(kv.second, "maxDist");
                    else if (kv.first == "searchType") {}      // both search types return the same neighbours here
                    else throw std::runtime_error("KDTreeMatcher: unknown parameter " + kv.first);
                }
            }
            // the chain multiplies its filters' weights (A.4): one QUANTILE filter (TrimmedDist or MedianDist) and / or one
            // MaxDist filter, in any order
            if (y.has("outlierFilters")) {
                int n_trim = 0, n_max = 0, n_nrm = 0;
                for (auto &m : y.sections.at("outlierFilters")) {
                    if (m.name == "SurfaceNormalOutlierFilter" && n_nrm++ == 0) {
                        outlierFilters.push_back(std::make_shared<SurfaceNormalOutlierFilter>(m.params.count("maxAngle") ? (T)to_double(m.params.at("maxAngle"), "maxAngle") : T(1.57)));
                        continue;
                    }
                    if (m.name == "TrimmedDistOutlierFilter" && n_trim++ == 0)
                        outlierFilters.push_back(std::make_shared<TrimmedDistOutlierFilter>(this, m.params.count("ratio") ? (T)to_double(m.params.at("ratio"), "ratio") : T(0.85)));
                    else if (m.name == "MedianDistOutlierFilter" && n_trim++ == 0) {
                        const T f = m.params.count("factor") ? (T)to_double(m.params.at("factor"), "factor") : T(3);
                        if (!(f > T(0))) throw std::runtime_error("MedianDistOutlierFilter: factor must be positive");
                        outlierFilters.push_back(std::make_shared<MedianDistOutlierFilter>(this, f));
                    } else if (m.name == "RobustOutlierFilter" && n_trim++ == 0) {
                        std::string fct = "cauchy", scale = "mad", dtype = "point2point";
                        T tun = T(1), approx = std::numeric_limits<T>::infinity();
                        int nbIter = 0;
                        for (auto &kv : m.params) {
                            if (kv.first == "robustFct") fct = kv.second;
                            else if (kv.first == "scaleEstimator") scale = kv.second;
                            else if (kv.first == "distanceType") dtype = kv.second;
                            else if (kv.first == "tuning") tun = (T)to_double(kv.second, "tuning");
                            else if (kv.first == "approximation") approx = (T)to_double(kv.second, "approximation");
                            else if (kv.first == "nbIterationForScale") nbIter = (int)to_double(kv.second, "nbIterationForScale");
                            else throw std::runtime_error("RobustOutlierFilter: unknown parameter " + kv.first);
                        }
                        if (dtype != "point2point") throw std::runtime_error("RobustOutlierFilter: distanceType " + dtype + " is not supported (point2point)");
                        if (nbIter != 0) throw std::runtime_error("RobustOutlierFilter: nbIterationForScale must be 0 (the scale is estimated at every iteration)");
                        outlierFilters.push_back(std::make_shared<RobustOutlierFilter>(this, fct, tun, scale, approx));
                    } else if (m.name == "MaxDistOutlierFilter" && n_max++ == 0)
                        outlierFilters.push_back(std::make_shared<MaxDistOutlierFilter>(m.params.count("maxDist") ? (T)to_double(m.params.at("maxDist"), "maxDist") : T(1)));
                    else
                        throw std::runtime_error("loadFromYaml: unsupported outlier filter chain at " + m.name +
                                                 " (supported: one of TrimmedDistOutlierFilter, MedianDistOutlierFilter or RobustOutlierFilter, and / or one MaxDistOutlierFilter, "
                                                 "and / or one SurfaceNormalOutlierFilter)");
                }
            } else outlierFilters.push_back(std::make_shared<TrimmedDistOutlierFilter>(this, T(0.85)));
            errorMinimizer = std::make_shared<ErrorMinimizer>(this);
            if (y.has("errorMinimizer") && !y.sections.at("errorMinimizer").empty()) {
                const auto &m = y.sections.at("errorMinimizer")[0];
                if (m.name == "PointToPlaneWithCovErrorMinimizer") errorMinimizer->withCov = true;
                else if (m.name == "PointToPointErrorMinimizer") errorMinimizer->pointToPoint = true;
                // [EXT] PointToPointWithCovErrorMinimizer{sensorStdDev}: the point-to-point solve + the covariance estimate (the
                // body of PointToPlaneWithCov's, read from the reference's normals) -- what a point-to-point pgslam configuration
                // needs, its optimiser refuses the base class's zero covariance (Optimizer.hpp:94,109)
                else if (m.name == "PointToPointWithCovErrorMinimizer") { errorMinimizer->pointToPoint = true; errorMinimizer->withCov = true; }
                else if (m.name != "PointToPlaneErrorMinimizer") throw std::runtime_error("loadFromYaml: unsupported error minimizer " + m.name);
                for (auto &kv : m.params) {
                    if (kv.first == "sensorStdDev") errorMinimizer->sensorStdDev = (T)to_double(kv.second, "sensorStdDev");
                    else if (kv.first == "force4DOF" && !errorMinimizer->pointToPoint) errorMinimizer->force4DOF = !(kv.second == "0" || kv.second == "false");
                    else if ((kv.first == "force2D" || kv.first == "force4DOF") && (kv.second == "0" || kv.second == "false")) {}
                    else throw std::runtime_error(m.name + ": unsupported parameter " + kv.first);
                }
            }
            if (y.has("transformationCheckers"))
                for (auto &m : y.sections.at("transformationCheckers")) {
                    if (m.name == "CounterTransformationChecker") {
                        auto c = std::make_shared<CounterTransformationChecker>();
                        if (m.params.count("maxIterationCount")) c->maxIterationCount = (int)to_double(m.params.at("maxIterationCount"), "maxIterationCount");
                        transformationCheckers.push_back(c);
                    } else if (m.name == "DifferentialTransformationChecker") {
                        auto c = std::make_shared<DifferentialTransformationChecker>();
                        if (m.params.count("minDiffRotErr")) c->minDiffRotErr = (T)to_double(m.params.at("minDiffRotErr"), "minDiffRotErr");
                        if (m.params.count("minDiffTransErr")) c->minDiffTransErr = (T)to_double(m.params.at("minDiffTransErr"), "minDiffTransErr");
                        if (m.params.count("smoothLength")) c->smoothLength = (int)to_double(m.params.at("smoothLength"), "smoothLength");
                        transformationCheckers.push_back(c);
                    } else if (m.name == "BoundTransformationChecker") {
                        auto c = std::make_shared<BoundTransformationChecker>();
                        if (m.params.count("maxRotationNorm")) c->maxRotationNorm = (T)to_double(m.params.at("maxRotationNorm"), "maxRotationNorm");
                        if (m.params.count("maxTranslationNorm")) c->maxTranslationNorm = (T)to_double(m.params.at("maxTranslationNorm"), "maxTranslationNorm");
                        transformationCheckers.push_back(c);
                    } else
                        throw std::runtime_error("loadFromYaml: unsupported transformation checker " + m.name);
                }
            else {
                transformationCheckers.push_back(std::make_shared<CounterTransformationChecker>());
                transformationCheckers.push_back(std::make_shared<DifferentialTransformationChecker>());
            }
        }
        //! chain objects -> pgicp_params.  `readingHasNormals`: a SurfaceNormalOutlierFilter only acts when both clouds carry
        //! the `normals` descriptor (else its weights are all ones: it is left out of the device chain for that call)
        void pushParams(bool readingHasNormals = false)
        {
            pgicp_params p;
            pgicp_default_params(&p);
            pgicp_params cur;
            pgicp_get_params(ctx, &cur);
            p.matcher = cur.matcher; p.grid_cell = cur.grid_cell; p.check_every = cur.check_every;
            if (matcher) { p.knn = matcher->knn; p.epsilon = (double)matcher->epsilon; p.max_dist = (double)matcher->maxDist; }
            p.trim_ratio = 1.0;                    // no TrimmedDist filter in the chain: every finite pair passes it
            p.outlier_max_dist = 0.0;
            for (auto &f : outlierFilters) {
                if (auto t = std::dynamic_pointer_cast<TrimmedDistOutlierFilter>(f)) { p.trim_ratio = (double)t->ratio; p.quantile_scale = 1.0; }
                if (auto md = std::dynamic_pointer_cast<MedianDistOutlierFilter>(f)) { p.trim_ratio = 0.5; p.quantile_scale = (double)md->factor; }
                if (auto m = std::dynamic_pointer_cast<MaxDistOutlierFilter>(f)) p.outlier_max_dist = (double)m->maxDist;
                if (auto rb = std::dynamic_pointer_cast<RobustOutlierFilter>(f)) {
                    if (p.robust_fct != PGICP_ROBUST_NONE) throw std::runtime_error("ICP chain: more than one RobustOutlierFilter");
                    p.robust_fct = rb->fctCode(); p.robust_scale = rb->scaleCode(); p.robust_tuning = (double)rb->tuning;
                    p.robust_approx = std::isfinite((double)rb->approximation) ? (double)rb->approximation : 0.0;
                }
            }
            if (errorMinimizer) {
                p.sensor_std_dev = (double)errorMinimizer->sensorStdDev;
                p.error_minimizer = errorMinimizer->pointToPoint ? (errorMinimizer->withCov ? PGICP_MINIMIZER_POINT_TO_POINT_WITH_COV : PGICP_MINIMIZER_POINT_TO_POINT)
                                    : errorMinimizer->force4DOF ? PGICP_MINIMIZER_POINT_TO_PLANE_4DOF : PGICP_MINIMIZER_POINT_TO_PLANE;
            }
            for (auto &f : outlierFilters)
                if (auto sn = std::dynamic_pointer_cast<SurfaceNormalOutlierFilter>(f))
                    p.normal_max_angle = readingHasNormals ? (double)sn->maxAngle : 0.0;
            bool hasCounter = false, hasDiff = false;
            for (auto &c : transformationCheckers) {
                if (auto bc = std::dynamic_pointer_cast<BoundTransformationChecker>(c)) {
                    p.bound_max_rot = (double)bc->maxRotationNorm; p.bound_max_trans = (double)bc->maxTranslationNorm;
                }
                if (auto cc = std::dynamic_pointer_cast<CounterTransformationChecker>(c)) { p.max_iters = cc->maxIterationCount; hasCounter = true; }
                if (auto dc = std::dynamic_pointer_cast<DifferentialTransformationChecker>(c)) {
                    p.min_diff_rot = (double)dc->minDiffRotErr; p.min_diff_trans = (double)dc->minDiffTransErr; p.smooth_length = dc->smoothLength; hasDiff = true;
                }
            }
            if (!hasCounter) p.max_iters = 1 << 20;
            if (!hasDiff) { p.min_diff_rot = 0; p.min_diff_trans = 0; }
            check(ctx, pgicp_set_params(ctx, &p));
        }
        //! LoopCloser.hpp:317
        bool getMaxNumIterationsReached() const { return lastStats.max_iter_reached != 0; }
        unsigned getPrefilteredReadingPtsCount() const { return prefilteredReadingPtsCount; }
        unsigned getPrefilteredReferencePtsCount() const { return prefilteredReferencePtsCount; }

    protected:
        unsigned prefilteredReadingPtsCount = 0, prefilteredReferencePtsCount = 0;
        void storeStats(const pgicp_stats &s)
        {
            lastStats = s;
            errorMinimizer->lastOverlap = (T)s.overlap;
            errorMinimizer->lastRatio = (T)s.overlap;
            errorMinimizer->lastResidual = (T)s.residual;
            Matrix cov(6, 6);
            for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) cov(i, j) = (T)s.cov[i * 6 + j];
            errorMinimizer->lastCov = cov;
        }
    public:
        //! does getOverlap() of this chain take the sensor-noise branch for `reading`? (upstream: the reading carries
        //! `simpleSensorNoise` -- and `normals` when the minimizer is point-to-plane)
        bool sensorNoiseApplies(const DataPoints &reading) const
        {
            return errorMinimizer && reading.descriptorExists("simpleSensorNoise") && (errorMinimizer->pointToPoint || reading.descriptorExists("normals"));
        }
    protected:
        //! getOverlap()'s sensor-noise branch over the last error elements of problem `problem` of the align call that has just
        //! returned: the pairs the outlier filters kept in its last iteration (exact ids and distances: pgicp_debug_last_matches),
        //! dists = |reading - reference| in T, mean over the kept pairs, the share below mean + noise(point).  Kept = a neighbour and
        //! a squared distance within the last threshold (Trimmed / Median / MaxDist filters: what pgicp_stats.trim_limit is);
        //! a chain whose weights are not that (Robust, SurfaceNormal outlier filters) is refused, not approximated.
        T sensorNoiseOverlap(const DataPoints &reading, int problem, const pgicp_stats &st) const
        {
            for (auto &f : outlierFilters)
                if (std::dynamic_pointer_cast<RobustOutlierFilter>(f) || std::dynamic_pointer_cast<SurfaceNormalOutlierFilter>(f))
                    throw std::runtime_error("getOverlap: the sensor-noise overlap (a reading with simpleSensorNoise) is not supported with a Robust / SurfaceNormal outlier filter");
            const int n = (int)reading.getNbPoints(), knn = std::max(1, matcher ? matcher->knn : 1);
            std::vector<int32_t> ids((size_t)n * knn);
            std::vector<T> d2((size_t)n * knn);
            check(ctx, A::last_matches(ctx, problem, ids.data(), d2.data()));
            const int rn = reading.getDescriptorStartingRow("simpleSensorNoise");
            const T limit = (T)st.trim_limit;
            int nb = 0;
            T sum = 0;
            for (int k = 0; k < knn; k++)
                for (int i = 0; i < n; i++) {
                    const size_t e = (size_t)i * knn + k;
                    if (ids[e] >= 0 && d2[e] <= limit) { sum += std::sqrt(d2[e]); nb++; }
                }
            if (nb == 0) throw ConvergenceError("PointToPlaneErrorMinimizer: no element to minimize");
            const T mean = sum / (T)nb;
            int count = 0;
            for (int k = 0; k < knn; k++)
                for (int i = 0; i < n; i++) {
                    const size_t e = (size_t)i * knn + k;
                    if (ids[e] >= 0 && d2[e] <= limit && std::sqrt(d2[e]) < mean + reading.descriptors(rn, i)) count++;
                }
            return (T)count / (T)nb;
        }
    public:
        //! A reading whose filtered copy is on its way to (or already in) device memory: pgicp_upload_* started the
        //! transfer on the context's COPY stream and returned at once.  Not part of libpointmatcher; it is how a caller
        //! that already holds the next scan (LocalizerMT.hpp:27-40: it is queued while the current one aligns) lets the
        //! transfer overlap the running ICP.  Valid until the second-next uploadReading() on the same chain.
        struct DeviceReading {
            const T *dev = nullptr;
            std::shared_ptr<DataPoints> filtered;         // the host copy after the chain's reading filters (what was uploaded)
            //! points and stride of the device copy when they are known without looking at `filtered` (-1: ask it) -- a host copy whose
            //! gaps are still being closed on another thread must not be read (Localizer's deferred host compaction)
            int n = -1, stride = -1;
            int points() const { return n >= 0 ? n : (int)filtered->getNbPoints(); }
            int xyzStride() const { return stride >= 0 ? stride : filtered->xyzStride(); }
            explicit operator bool() const { return dev != nullptr; }
        };
        //! a device copy of a cloud may stand for the cloud itself in operator(): the chain's reading filters change nothing
        //! and no outlier filter looks at the reading's descriptors
        bool hasNormalFilter() const
        {
            for (auto &f : outlierFilters) if (std::dynamic_pointer_cast<SurfaceNormalOutlierFilter>(f)) return true;
            return false;
        }
        bool deviceReadingEquivalent() const
        {
            return !hasNormalFilter() && readingDataPointsFilters.allIdentity() && readingStepDataPointsFilters.allIdentity();
        }
        DeviceReading uploadReading(const DataPoints &readingIn)
        {
            DeviceReading r;
            r.filtered = std::make_shared<DataPoints>(readingIn);
            readingDataPointsFilters.init();
            readingDataPointsFilters.apply(*r.filtered);
            readingStepDataPointsFilters.init();
            readingStepDataPointsFilters.apply(*r.filtered);          // (applied once: see alignOnMap)
            check(ctx, A::upload(ctx, r.filtered->xyzPtr(), r.filtered->xyzStride(), (int)r.filtered->getNbPoints(), &r.dev));
            return r;
        }
    protected:
        TransformationParameters alignOnMap(const DeviceReading &r, const TransformationParameters &T_init)
        {
            // a device reading is coordinates only: a chain whose SurfaceNormalOutlierFilter compares the READING's normals must be
            // given the host cloud (refused, never run without the filter -- ADVICE round 4)
            if (hasNormalFilter() && r.filtered && r.filtered->normalsPtr() != nullptr)
                throw std::logic_error("ICP: a SurfaceNormalOutlierFilter needs the reading's normals; pass the host cloud, not a device reading");
            prefilteredReadingPtsCount = (unsigned)r.points();
            double Ti[16], To[16];
            pgslam_amd::to_row_major16(T_init, Ti);
            pgicp_stats st;
            pushParams();
            // (the call makes its stream wait, on the device, for the upload; the host does not)
            const int rc = A::align_dev(ctx, matcher->mapId, r.dev, r.xyzStride(), r.points(), Ti, To, &st);
            storeStats(st);
            check(ctx, rc);
            if (r.filtered && sensorNoiseApplies(*r.filtered)) errorMinimizer->lastOverlap = sensorNoiseOverlap(*r.filtered, 0, st);
            const TransformationParameters T_out = pgslam_amd::from_row_major16<T>(To);
            if (onAlign && (currentReference || lazyReference) && (!onAlignWanted || onAlignWanted()))
                if (const DataPoints *ref = observedReference()) onAlign(*r.filtered, *ref, T_init, T_out, st);
            return T_out;
        }
        TransformationParameters alignOnMap(const DataPoints &readingIn, const TransformationParameters &T_init)
        {
            DataPoints reading(readingIn);
            readingDataPointsFilters.init();
            readingDataPointsFilters.apply(reading);
            prefilteredReadingPtsCount = reading.getNbPoints();
            // [EXT] readingStepDataPointsFilters (Localizer.hpp:325-326) act on the filtered reading in ITS OWN frame at every
            // iteration, before the iteration's transform is applied; every filter restated here is a pure function of the
            // cloud, so the step filters give the same cloud in every iteration: they are applied once, here (the random
            // sampler, whose upstream version draws a new sample per iteration, is refused in this slot by loadFromYaml)
            readingStepDataPointsFilters.init();
            readingStepDataPointsFilters.apply(reading);
            double Ti[16], To[16];
            pgslam_amd::to_row_major16(T_init, Ti);
            pgicp_stats st;
            std::memset(&st, 0, sizeof st);
            bool normalFilter = false;
            for (auto &f : outlierFilters) normalFilter = normalFilter || std::dynamic_pointer_cast<SurfaceNormalOutlierFilter>(f) != nullptr;
            const bool withNormals = normalFilter && reading.normalsPtr() != nullptr;
            pushParams(withNormals);
            const int rc = withNormals ? A::align_nrm(ctx, matcher->mapId, reading.xyzPtr(), reading.xyzStride(), (int)reading.getNbPoints(),
                                                      reading.normalsPtr(), reading.normalsStride(), Ti, To, &st)
                                       : A::align(ctx, matcher->mapId, reading.xyzPtr(), reading.xyzStride(), (int)reading.getNbPoints(), Ti, To, &st);
            storeStats(st);
            check(ctx, rc);
            if (sensorNoiseApplies(reading)) errorMinimizer->lastOverlap = sensorNoiseOverlap(reading, 0, st);
            const TransformationParameters T_out = pgslam_amd::from_row_major16<T>(To);
            if (onAlign && (currentReference || lazyReference) && (!onAlignWanted || onAlignWanted()))
                if (const DataPoints *ref = observedReference()) onAlign(reading, *ref, T_init, T_out, st);
            return T_out;
        }
    };

    struct ICP : ICPChainBase {
        explicit ICP(int device = 0) : ICPChainBase(device) {}
        TransformationParameters operator()(const DataPoints &readingIn, const DataPoints &referenceIn) { return (*this)(readingIn, referenceIn, Matrix::Identity(4, 4)); }
        //! LoopCloser.hpp:98 -- reference filters, mean-centring, index build, loop
        TransformationParameters operator()(const DataPoints &readingIn, const DataPoints &referenceIn, const TransformationParameters &initialTransformationParameters)
        {
            return compute(readingIn, referenceIn, initialTransformationParameters);
        }
        TransformationParameters compute(const DataPoints &readingIn, const DataPoints &referenceIn, const TransformationParameters &T_init)
        {
            DataPoints reference(referenceIn);
            this->referenceDataPointsFilters.init();
            this->referenceDataPointsFilters.apply(reference);
            this->prefilteredReferencePtsCount = reference.getNbPoints();
            if (!reference.descriptorExists("normals") && !(this->errorMinimizer && this->errorMinimizer->pointToPoint && !this->errorMinimizer->withCov))
                throw std::runtime_error("PointToPlaneErrorMinimizer: the reference has no 'normals' descriptor");
            this->matcher->initImpl(reference, 1);
            this->currentReference = &reference;
            struct Reset { const DataPoints *&p; ~Reset() { p = nullptr; } } reset{this->currentReference};
            return this->alignOnMap(readingIn, T_init);
        }
        //! ICP::operator() for clouds that are in device memory already: mean-centring + index build of `reference` in place,
        //! the loop on the device copy of the reading (`reading.filtered` = its host cloud: sizes, the observer's copy).  Only for
        //! a chain whose filters change nothing (deviceReadingEquivalent, identity reference filters); `hostReference` makes the
        //! host copy of the reference when an observer asks for it.
        TransformationParameters computeOnDevice(const typename ICPChainBase::DeviceReading &reading, const pgslam_amd::DeviceCloud<T> &reference,
                                                 const TransformationParameters &T_init, std::function<DataPoints()> hostReference)
        {
            if (!this->referenceDataPointsFilters.allIdentity() || !this->deviceReadingEquivalent())
                throw std::logic_error("ICP::computeOnDevice: the chain filters its clouds or reads their descriptors");
            if (!reference.hasNormals() && !(this->errorMinimizer && this->errorMinimizer->pointToPoint && !this->errorMinimizer->withCov))
                throw std::runtime_error("PointToPlaneErrorMinimizer: the reference has no 'normals' descriptor");
            this->prefilteredReferencePtsCount = reference.n;
            this->matcher->initDevice(reference, 1);
            this->currentReference = nullptr;
            this->lazyReference = hostReference;
            struct Reset { std::function<DataPoints()> &f; const DataPoints *&p; ~Reset() { f = nullptr; p = nullptr; } } reset{this->lazyReference, this->currentReference};
            return this->alignOnMap(reading, T_init);
        }
    };

    struct ICPSequence : ICP {
        explicit ICPSequence(int device = 0) : ICP(device) {}
        bool hasMap() const { return mapPointCloud.getNbPoints() != 0 || mapOnDevice; }
        //! a map that is in device memory already (the localizer assembles it there from resident keyframe clouds) may stand
        //! for setMap(cloud): the chain's reference filters change nothing
        bool deviceMapEquivalent() const { return this->referenceDataPointsFilters.allIdentity(); }
        //! setMap for such a map: mean-centre + index build, no cloud crosses PCIe.  `hostCopy` makes the host cloud if an
        //! observer (onAlign) or getPrefilteredMap() asks for it.
        bool setMap(const pgslam_amd::DeviceCloud<T> &deviceCloud, std::function<DataPoints()> hostCopy)
        {
            if (!deviceMapEquivalent()) throw std::logic_error("ICPSequence::setMap(DeviceCloud): the chain has reference filters");
            if (!deviceCloud.hasNormals() && !(this->errorMinimizer && this->errorMinimizer->pointToPoint && !this->errorMinimizer->withCov))
                throw std::runtime_error("PointToPlaneErrorMinimizer: the map has no 'normals' descriptor");
            mapPointCloud = DataPoints();
            this->prefilteredReferencePtsCount = deviceCloud.n;
            this->matcher->initDevice(deviceCloud, 1);
            mapOnDevice = true;
            this->currentReference = nullptr;
            this->lazyReference = hostCopy;
            return true;
        }
        //! Localizer.hpp:148,168,254 -- copy, reference filters, mean-centre, build the index; stays resident in HBM
        bool setMap(const DataPoints &inputCloud)
        {
            mapPointCloud = inputCloud;
            this->referenceDataPointsFilters.init();
            this->referenceDataPointsFilters.apply(mapPointCloud);
            this->prefilteredReferencePtsCount = mapPointCloud.getNbPoints();
            if (!mapPointCloud.descriptorExists("normals") && !(this->errorMinimizer && this->errorMinimizer->pointToPoint && !this->errorMinimizer->withCov))
                throw std::runtime_error("PointToPlaneErrorMinimizer: the map has no 'normals' descriptor");
            this->matcher->initImpl(mapPointCloud, 1);
            this->currentReference = &mapPointCloud;
            this->lazyReference = nullptr;
            mapOnDevice = false;
            return true;
        }
        void clearMap() { mapPointCloud = DataPoints(); this->currentReference = nullptr; this->lazyReference = nullptr; mapOnDevice = false; this->matcher->release(); }
        const DataPoints &getPrefilteredMap()
        {
            if (mapOnDevice) if (const DataPoints *ref = this->observedReference()) return *ref;
            return mapPointCloud;
        }
        TransformationParameters operator()(const DataPoints &cloudIn) { return (*this)(cloudIn, Matrix::Identity(4, 4)); }
        //! Localizer.hpp:126.  Without a map the first cloud becomes the map and identity is returned
        TransformationParameters operator()(const DataPoints &cloudIn, const TransformationParameters &initialTransformationParameters)
        {
            if (!hasMap()) { setMap(cloudIn); return Matrix::Identity(4, 4); }
            return this->alignOnMap(cloudIn, initialTransformationParameters);
        }
        //! the same against a reading whose upload was started earlier (ICPChainBase::uploadReading): needs a map
        TransformationParameters operator()(const typename ICPChainBase::DeviceReading &reading, const TransformationParameters &initialTransformationParameters)
        {
            if (!hasMap()) throw std::runtime_error("ICPSequence: a device reading needs a map (setMap)");
            return this->alignOnMap(reading, initialTransformationParameters);
        }
    private:
        DataPoints mapPointCloud;
        bool mapOnDevice = false;
    };

    // ------------------------------------------------------------------ registrar (PM::get().REG(Transformation).create(...))
    struct TransformationRegistrarT {
        std::shared_ptr<Transformation> create(const std::string &name) const
        {
            if (name == "RigidTransformation") return std::make_shared<RigidTransformation>();
            throw std::runtime_error("Registrar: unknown transformation " + name);
        }
    };
    TransformationRegistrarT TransformationRegistrar;
    static PointMatcher &get() { static PointMatcher pm; return pm; }
};

// libpointmatcher's own spelling (pointmatcher/Registrar.h defines this macro for its users: `PM::get().REG(Transformation)
// .create("RigidTransformation")`, Localizer.hpp:24, LoopCloser.hpp:27): a drop-in header has to provide it, short as it is.
#ifndef REG
#define REG(name) name##Registrar
#endif
