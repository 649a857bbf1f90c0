// slam.hpp -- the host side of pgslam around the GPU hot path: keyframe graph, map manager, pose-graph
// optimiser, candidate searches and the PoseGraphSlam facade.
//
// Everything here is CPU code by design (north_star keeps the pose-graph back end on the host); it
// restates, without Boost.Graph and GTSAM (neither exists in this image):
//   * Types<T>::Graph (undirected, Keyframe vertices, Constraint edges)       reference types.h:24-57
//   * MapManager                                                             MapManager.hpp:36-158
//   * Optimizer (BetweenFactor / PriorFactor least squares, LM)              Optimizer.hpp:25-157
//   * LoopCloser::FindLocalMapCandidate / ProcessVertex / CheckIcpResult     LoopCloser.hpp:83-110,193-305,308-340
//   * Localizer::ProcessData / UpdateAfterIcp / FindNeighborLocalMapComposition  Localizer.hpp:91-268,393-483
//   * PoseGraphSlam::AddData / SetIcpConfig / WriteGraphviz                   PoseGraphSlam.hpp:21-76
// [EXT] GTSAM's LevenbergMarquardtOptimizer is replaced by a dependency-free LM on SE(3) that minimises the
// same cost (whitened Logmap residuals of between factors, rotation-first tangent order as
// Optimizer::PmCovToGtsamCov prepares it); the prior of sigma 1e-6 on the fixed vertex is imposed as a hard
// constraint.  The optimum is the same to solver tolerance; the iteration path is not GTSAM's.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <condition_variable>
#include <chrono>
#include <deque>
#include <limits>
#include <list>
#include <mutex>
#include <queue>
#include <thread>
#include <set>

#include "pgslam.hpp"
#include <iostream>

namespace pgslam {

//! a YAML file as text (the setters that take a path: Localizer.hpp:54-78, LoopCloser.hpp:58-74)
inline std::string slurp_config_file(const std::string &p)
{
    std::ifstream ifs(p);
    if (!ifs) throw std::runtime_error("[PoseGraphSlam] cannot open " + p);
    std::stringstream ss;
    ss << ifs.rdbuf();
    return ss.str();
}


// ------------------------------------------------------------------ graph
template <typename T>
class PoseGraph {
public:
    IMPORT_PGSLAM_TYPES(T)
    struct EdgeRec { size_t from, to; Constraint c; };
    size_t AddVertex(const Keyframe &kf) { v_.push_back(kf); adj_.emplace_back(); return v_.size() - 1; }
    //! undirected, at most one edge per vertex pair (boost::add_edge on a setS-less list graph would allow
    //! parallel edges; pgslam treats a second one as a logic error, MapManager.hpp:97,127)
    size_t AddEdge(size_t from, size_t to, const Constraint &c)
    {
        for (size_t id : adj_[from]) if (Other(id, from) == to) throw std::logic_error("PoseGraph: edge already exists");
        e_.push_back(EdgeRec{from, to, c});
        adj_[from].push_back(e_.size() - 1);
        adj_[to].push_back(e_.size() - 1);
        return e_.size() - 1;
    }
    size_t NumVertices() const { return v_.size(); }
    size_t NumEdges() const { return e_.size(); }
    Keyframe &operator[](size_t v) { return v_[v]; }
    const Keyframe &operator[](size_t v) const { return v_[v]; }
    const EdgeRec &Edge(size_t id) const { return e_[id]; }
    const std::vector<size_t> &IncidentEdges(size_t v) const { return adj_[v]; }
    size_t Other(size_t edge, size_t v) const { return e_[edge].from == v ? e_[edge].to : e_[edge].from; }

    //! Dijkstra over Constraint::weight from `src` on the sub-graph the filters keep.  `examine` is called
    //! for every vertex when it is settled (boost's on_examine_vertex), in non-decreasing distance; it may
    //! return false to stop the search (the reference throws StopSearch, LoopCloser.hpp:160-175).
    std::vector<double> Dijkstra(size_t src, const std::function<bool(size_t)> &keep_vertex,
                                 const std::function<bool(size_t)> &keep_edge, const std::function<bool(size_t)> &examine) const
    {
        const double inf = std::numeric_limits<double>::infinity();
        std::vector<double> dist(v_.size(), inf);
        if (keep_vertex && !keep_vertex(src)) return dist;
        using QE = std::pair<double, size_t>;
        std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq;
        std::vector<char> done(v_.size(), 0);
        dist[src] = 0.0;
        pq.push({0.0, src});
        while (!pq.empty()) {
            const auto [d, u] = pq.top();
            pq.pop();
            if (done[u]) continue;
            done[u] = 1;
            if (examine && !examine(u)) break;
            for (size_t id : adj_[u]) {
                if (keep_edge && !keep_edge(id)) continue;
                const size_t w = Other(id, u);
                if (done[w] || (keep_vertex && !keep_vertex(w))) continue;
                const double nd = d + (double)e_[id].c.weight;
                if (nd < dist[w]) { dist[w] = nd; pq.push({nd, w}); }
            }
        }
        return dist;
    }

private:
    std::vector<Keyframe> v_;
    std::vector<EdgeRec> e_;
    std::vector<std::vector<size_t>> adj_;
};

inline double TranslationDistance(const double *A16, const double *B16)
{
    const double dx = B16[3] - A16[3], dy = B16[7] - A16[7], dz = B16[11] - A16[11];
    return std::sqrt(dx * dx + dy * dy + dz * dz);
}
template <typename M>
double PoseDistance(const M &A, const M &B)          // Metrics<T>::Distance, metrics.hpp:7-12
{
    double s = 0;
    for (int k = 0; k < 3; k++) { const double d = (double)B(k, 3) - (double)A(k, 3); s += d * d; }
    return std::sqrt(s);
}
template <typename M>
double PoseWeight(const M &T_meas)                   // Metrics<T>::Weight, metrics.hpp:21-24
{
    return std::sqrt((double)T_meas(0, 3) * T_meas(0, 3) + (double)T_meas(1, 3) * T_meas(1, 3) + (double)T_meas(2, 3) * T_meas(2, 3));
}

// ------------------------------------------------------------------ SE(3) helpers (double, row-major 4x4)
namespace se3 {
struct Pose { double R[9]; double t[3]; };
inline Pose identity() { return Pose{{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}}; }
inline Pose mul(const Pose &a, const Pose &b)
{
    Pose o;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) o.R[3 * i + j] = a.R[3 * i] * b.R[j] + a.R[3 * i + 1] * b.R[3 + j] + a.R[3 * i + 2] * b.R[6 + j];
        o.t[i] = a.R[3 * i] * b.t[0] + a.R[3 * i + 1] * b.t[1] + a.R[3 * i + 2] * b.t[2] + a.t[i];
    }
    return o;
}
inline Pose inv(const Pose &a)
{
    Pose o;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.R[3 * i + j] = a.R[3 * j + i];
    for (int i = 0; i < 3; i++) o.t[i] = -(o.R[3 * i] * a.t[0] + o.R[3 * i + 1] * a.t[1] + o.R[3 * i + 2] * a.t[2]);
    return o;
}
//! Expmap of the twist [w (rotation), v (translation)] -- GTSAM's Pose3 tangent order
inline Pose exp(const double xi[6])
{
    const double wx = xi[0], wy = xi[1], wz = xi[2];
    const double th2 = wx * wx + wy * wy + wz * wz, th = std::sqrt(th2);
    double A, B, C;                               // sin(th)/th, (1-cos)/th^2, (th-sin)/th^3
    if (th < 1e-5) { A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; C = 1.0 / 6.0 - th2 / 120.0; }
    else { A = std::sin(th) / th; B = (1.0 - std::cos(th)) / th2; C = (th - std::sin(th)) / (th2 * th); }
    const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double K2[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
    Pose o;
    double V[9];
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        o.R[i] = I + A * K[i] + B * K2[i];
        V[i] = I + B * K[i] + C * K2[i];
    }
    for (int i = 0; i < 3; i++) o.t[i] = V[3 * i] * xi[3] + V[3 * i + 1] * xi[4] + V[3 * i + 2] * xi[5];
    return o;
}
//! Logmap, inverse of exp above
inline void log(const Pose &p, double xi[6])
{
    const double tr = p.R[0] + p.R[4] + p.R[8];
    double c = 0.5 * (tr - 1.0);
    c = std::max(-1.0, std::min(1.0, c));
    const double th = std::acos(c);
    double w[3] = {p.R[7] - p.R[5], p.R[2] - p.R[6], p.R[3] - p.R[1]};
    double k;
    if (th < 1e-5) k = 0.5 + th * th / 12.0;
    else if (M_PI - th < 1e-4) {                  // near pi: axis from the symmetric part
        double ax[3];
        for (int i = 0; i < 3; i++) ax[i] = std::sqrt(std::max(0.0, (p.R[4 * i] - c) / (1.0 - c)));
        if (w[0] < 0) ax[0] = -ax[0];
        if (w[1] < 0) ax[1] = -ax[1];
        if (w[2] < 0) ax[2] = -ax[2];
        for (int i = 0; i < 3; i++) xi[i] = th * ax[i];
        k = 0;
    } else k = th / (2.0 * std::sin(th));
    if (k != 0) for (int i = 0; i < 3; i++) xi[i] = k * w[i];
    const double wx = xi[0], wy = xi[1], wz = xi[2], th2 = wx * wx + wy * wy + wz * wz;
    const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double K2[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
    // V^-1 = I - K/2 + D K^2,  D = (1 - th*sin/(2(1-cos)))/th^2
    double D;
    const double t = std::sqrt(th2);
    if (t < 1e-5) D = 1.0 / 12.0 + th2 / 720.0;
    else D = (1.0 - 0.5 * t * std::sin(t) / (1.0 - std::cos(t))) / th2;
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int j = 0; j < 3; j++) {
            const double Vi = ((i == j) ? 1.0 : 0.0) - 0.5 * K[3 * i + j] + D * K2[3 * i + j];
            s += Vi * p.t[j];
        }
        xi[3 + i] = s;
    }
}
template <typename M>
Pose from_matrix(const M &m)
{
    Pose p;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) p.R[3 * i + j] = (double)m(i, j); p.t[i] = (double)m(i, 3); }
    return p;
}
template <typename M>
M to_matrix(const Pose &p)
{
    M m = M::Identity(4, 4);
    using S = typename std::decay<decltype(m(0, 0))>::type;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) m(i, j) = (S)p.R[3 * i + j]; m(i, 3) = (S)p.t[i]; }
    return m;
}
}  // namespace se3

//! Gauss-Jordan inverse of a 6x6 with partial pivoting; false if singular
inline bool invert6(const double A[36], double Ai[36])
{
    double M[6][12];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { M[i][j] = A[6 * i + j]; M[i][6 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 6; c++) {
        int piv = c;
        for (int r = c + 1; r < 6; r++) if (std::fabs(M[r][c]) > std::fabs(M[piv][c])) piv = r;
        if (!(std::fabs(M[piv][c]) > 0)) return false;
        if (piv != c) for (int j = 0; j < 12; j++) std::swap(M[c][j], M[piv][j]);
        const double d = 1.0 / M[c][c];
        for (int j = 0; j < 12; j++) M[c][j] *= d;
        for (int r = 0; r < 6; r++) if (r != c) { const double f = M[r][c]; if (f != 0) for (int j = 0; j < 12; j++) M[r][j] -= f * M[c][j]; }
    }
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Ai[6 * i + j] = M[i][6 + j];
    return true;
}

// ------------------------------------------------------------------ pose-graph least squares
//! min sum_k || L_k * Log( Z_k^-1 * X_from^-1 * X_to ) ||^2  with one pose held fixed; LM with GTSAM's default
//! parameters (lambda 1e-5, factor 10, upper bound 1e5, 100 iterations, relative / absolute error tolerance 1e-5).
class PoseGraphLeastSquares {
public:
    struct Between { size_t from, to; se3::Pose Z; double L[36]; };       // L: upper Cholesky factor of the information
    std::vector<se3::Pose> X;
    std::vector<Between> factors;
    size_t fixed = 0;
    int iterations = 0;
    double initial_error = 0, final_error = 0;

    //! cov6 row-major, order [rot, trans]; returns false if it is not positive definite
    static bool InformationFactor(const double cov6[36], double L[36])
    {
        double C[36], Ci[36];
        std::copy(cov6, cov6 + 36, C);
        // Cholesky C = G G^T, invert, then information = C^-1 = U^T U with U = G^-1 ... computed via explicit inverse
        double G[36] = {0};
        for (int i = 0; i < 6; i++)
            for (int j = 0; j <= i; j++) {
                double s = C[6 * i + j];
                for (int k = 0; k < j; k++) s -= G[6 * i + k] * G[6 * j + k];
                if (i == j) { if (!(s > 0)) return false; G[6 * i + i] = std::sqrt(s); }
                else G[6 * i + j] = s / G[6 * j + j];
            }
        // W = G^-1 (lower triangular); then r^T C^-1 r = |W r|^2
        double W[36] = {0};
        for (int c = 0; c < 6; c++) {
            for (int i = 0; i < 6; i++) {
                double s = (i == c) ? 1.0 : 0.0;
                for (int k = 0; k < i; k++) s -= G[6 * i + k] * W[6 * k + c];
                W[6 * i + c] = s / G[6 * i + i];
            }
        }
        std::copy(W, W + 36, L);
        (void)Ci;
        return true;
    }
    void Residual(const Between &f, const se3::Pose &Xi, const se3::Pose &Xj, double r[6]) const
    {
        double xi[6];
        se3::log(se3::mul(se3::inv(f.Z), se3::mul(se3::inv(Xi), Xj)), xi);
        for (int a = 0; a < 6; a++) { double s = 0; for (int b = 0; b < 6; b++) s += f.L[6 * a + b] * xi[b]; r[a] = s; }
    }
    double Error(const std::vector<se3::Pose> &P) const
    {
        double e = 0;
        for (const auto &f : factors) { double r[6]; Residual(f, P[f.from], P[f.to], r); for (int a = 0; a < 6; a++) e += r[a] * r[a]; }
        return 0.5 * e;
    }
    void Optimize()
    {
        const size_t N = X.size();
        double lambda = 1e-5;
        double err = Error(X);
        initial_error = err;
        iterations = 0;
        for (int it = 0; it < 100; it++) {
            Linearize();
            bool improved = false;
            double new_err = err;
            std::vector<se3::Pose> cand;
            while (lambda <= 1e5) {
                std::vector<double> delta = SolveDamped(lambda);
                cand = X;
                for (size_t v = 0; v < N; v++) if (v != fixed) cand[v] = se3::mul(X[v], se3::exp(delta.data() + 6 * v));
                new_err = Error(cand);
                if (new_err < err) { improved = true; lambda = std::max(lambda / 10.0, 1e-20); break; }
                lambda *= 10.0;
            }
            iterations = it + 1;
            if (!improved) break;
            const double dec = err - new_err;
            X.swap(cand);
            const double prev = err;
            err = new_err;
            if (dec < 1e-5 || dec / std::max(prev, 1e-300) < 1e-5) break;
        }
        final_error = err;
    }

private:
    std::vector<double> J_;        // per factor: 6 x 12 (d r / d delta_from | d r / d delta_to)
    std::vector<double> r_;        // per factor: 6
    void Linearize()
    {
        const size_t F = factors.size();
        J_.assign(F * 72, 0.0);
        r_.assign(F * 6, 0.0);
        const double h = 1e-6;
        for (size_t k = 0; k < F; k++) {
            const Between &f = factors[k];
            Residual(f, X[f.from], X[f.to], &r_[6 * k]);
            for (int side = 0; side < 2; side++)
                for (int a = 0; a < 6; a++) {
                    double d[6] = {0, 0, 0, 0, 0, 0}, rp[6], rm[6];
                    d[a] = h;
                    const se3::Pose Pp = se3::mul(side == 0 ? X[f.from] : X[f.to], se3::exp(d));
                    d[a] = -h;
                    const se3::Pose Pm = se3::mul(side == 0 ? X[f.from] : X[f.to], se3::exp(d));
                    if (side == 0) { Residual(f, Pp, X[f.to], rp); Residual(f, Pm, X[f.to], rm); }
                    else { Residual(f, X[f.from], Pp, rp); Residual(f, X[f.from], Pm, rm); }
                    for (int b = 0; b < 6; b++) J_[72 * k + 12 * b + 6 * side + a] = (rp[b] - rm[b]) / (2 * h);
                }
        }
    }
    // y = (J^T J + lambda I) x on the free variables
    void Apply(const std::vector<double> &x, double lambda, std::vector<double> &y) const
    {
        std::fill(y.begin(), y.end(), 0.0);
        for (size_t k = 0; k < factors.size(); k++) {
            const Between &f = factors[k];
            double Jx[6];
            for (int b = 0; b < 6; b++) {
                double s = 0;
                for (int a = 0; a < 6; a++) s += J_[72 * k + 12 * b + a] * x[6 * f.from + a] + J_[72 * k + 12 * b + 6 + a] * x[6 * f.to + a];
                Jx[b] = s;
            }
            for (int a = 0; a < 6; a++) {
                double s0 = 0, s1 = 0;
                for (int b = 0; b < 6; b++) { s0 += J_[72 * k + 12 * b + a] * Jx[b]; s1 += J_[72 * k + 12 * b + 6 + a] * Jx[b]; }
                y[6 * f.from + a] += s0;
                y[6 * f.to + a] += s1;
            }
        }
        for (size_t i = 0; i < x.size(); i++) y[i] += lambda * x[i];
        for (int a = 0; a < 6; a++) y[6 * fixed + a] = 0.0;
    }
    //! Preconditioned conjugate gradients.  The preconditioner is the system restricted to a SPANNING TREE of the graph
    //! (the odometry edges: every keyframe hangs on the reference keyframe it was created from) plus the diagonal blocks
    //! of the other edges -- solved exactly, leaves to root and back, in O(vertices).  What CG has to add are the loop
    //! edges' off-diagonal blocks, a perturbation of rank 6 per loop edge: it converges in a few steps per loop closure,
    //! where block-Jacobi needed tens of thousands on a 250-keyframe chain (0.8 s per solve; now milliseconds).
    std::vector<double> SolveDamped(double lambda) const
    {
        const size_t N = X.size(), n = 6 * N;
        std::vector<double> b(n, 0.0), D(N * 36, 0.0);
        for (size_t k = 0; k < factors.size(); k++) {
            const Between &f = factors[k];
            for (int a = 0; a < 6; a++) {
                double s0 = 0, s1 = 0;
                for (int c = 0; c < 6; c++) { s0 += J_[72 * k + 12 * c + a] * r_[6 * k + c]; s1 += J_[72 * k + 12 * c + 6 + a] * r_[6 * k + c]; }
                b[6 * f.from + a] -= s0;
                b[6 * f.to + a] -= s1;
                for (int a2 = 0; a2 < 6; a2++) {
                    double d0 = 0, d1 = 0;
                    for (int c = 0; c < 6; c++) { d0 += J_[72 * k + 12 * c + a] * J_[72 * k + 12 * c + a2]; d1 += J_[72 * k + 12 * c + 6 + a] * J_[72 * k + 12 * c + 6 + a2]; }
                    D[36 * f.from + 6 * a + a2] += d0;
                    D[36 * f.to + 6 * a + a2] += d1;
                }
            }
        }
        for (int a = 0; a < 6; a++) b[6 * fixed + a] = 0.0;
        // ---- spanning tree from the fixed vertex (breadth first over the factors; the first factor that reaches a vertex) ----
        std::vector<int> parent(N, -1), tree_factor(N, -1), order;
        {
            std::vector<std::vector<int>> inc(N);
            for (size_t k = 0; k < factors.size(); k++) { inc[factors[k].from].push_back((int)k); inc[factors[k].to].push_back((int)k); }
            std::vector<char> seen(N, 0);
            order.push_back((int)fixed);
            seen[fixed] = 1;
            for (size_t h = 0; h < order.size(); h++) {
                const int v = order[h];
                for (int k : inc[v]) {
                    const int w = (int)(factors[k].from == (size_t)v ? factors[k].to : factors[k].from);
                    if (!seen[w]) { seen[w] = 1; parent[w] = v; tree_factor[w] = k; order.push_back(w); }
                }
            }
            for (size_t v = 0; v < N; v++) if (!seen[v]) order.push_back((int)v);     // (not connected to the fixed vertex: diagonal only)
        }
        // off-diagonal block A(parent, v) = Jp^T Jv of the tree factor, row-major 6x6
        std::vector<double> O(N * 36, 0.0), Sinv(N * 36, 0.0), K(N * 36, 0.0), Dt = D;
        for (size_t v = 0; v < N; v++) {
            if (tree_factor[v] < 0) continue;
            const int k = tree_factor[v];
            const int sp = factors[k].from == (size_t)parent[v] ? 0 : 6, sv = 6 - sp;
            for (int a = 0; a < 6; a++)
                for (int c = 0; c < 6; c++) {
                    double t = 0;
                    for (int e = 0; e < 6; e++) t += J_[72 * k + 12 * e + sp + a] * J_[72 * k + 12 * e + sv + c];
                    O[36 * v + 6 * a + c] = t;
                }
        }
        for (size_t v = 0; v < N; v++) for (int i = 0; i < 6; i++) Dt[36 * v + 7 * i] += lambda + 1e-12;
        // eliminate leaves first: S_v = D_v - sum over children; K_v = O_v S_v^-1; D_parent -= K_v O_v^T
        for (size_t h = order.size(); h-- > 0;) {
            const int v = order[h];
            if (v == (int)fixed) continue;
            double A[36], Ai[36];
            for (int i = 0; i < 36; i++) A[i] = Dt[36 * v + i];
            if (!invert6(A, Ai)) { for (int i = 0; i < 36; i++) Ai[i] = (i % 7 == 0) ? 1.0 / std::max(A[i], 1e-12) : 0.0; }
            for (int i = 0; i < 36; i++) Sinv[36 * v + i] = Ai[i];
            const int pv = parent[v];
            if (pv < 0 || pv == (int)fixed) continue;                          // (the fixed vertex's unknowns are zero: nothing to update)
            for (int a = 0; a < 6; a++)
                for (int c = 0; c < 6; c++) { double t = 0; for (int e = 0; e < 6; e++) t += O[36 * v + 6 * a + e] * Ai[6 * e + c]; K[36 * v + 6 * a + c] = t; }
            for (int a = 0; a < 6; a++)
                for (int c = 0; c < 6; c++) { double t = 0; for (int e = 0; e < 6; e++) t += K[36 * v + 6 * a + e] * O[36 * v + 6 * c + e]; Dt[36 * pv + 6 * a + c] -= t; }
        }
        auto precond = [&](const std::vector<double> &r, std::vector<double> &z) {
            std::vector<double> t = r;
            for (size_t h = order.size(); h-- > 0;) {                          // leaves to root: fold every vertex into its parent
                const int v = order[h], pv = parent[v];
                if (v == (int)fixed || pv < 0 || pv == (int)fixed) continue;
                for (int a = 0; a < 6; a++) { double s2 = 0; for (int c = 0; c < 6; c++) s2 += K[36 * v + 6 * a + c] * t[6 * v + c]; t[6 * pv + a] -= s2; }
            }
            for (size_t h = 0; h < order.size(); h++) {                         // root to leaves: back-substitute
                const int v = order[h], pv = parent[v];
                if (v == (int)fixed) { for (int a = 0; a < 6; a++) z[6 * v + a] = 0.0; continue; }
                double rhs[6];
                for (int a = 0; a < 6; a++) {
                    double s2 = t[6 * v + a];
                    if (pv >= 0 && pv != (int)fixed) for (int c = 0; c < 6; c++) s2 -= O[36 * v + 6 * c + a] * z[6 * pv + c];
                    rhs[a] = s2;
                }
                for (int a = 0; a < 6; a++) { double s2 = 0; for (int c = 0; c < 6; c++) s2 += Sinv[36 * v + 6 * a + c] * rhs[c]; z[6 * v + a] = s2; }
            }
        };
        std::vector<double> x(n, 0.0), r = b, z(n), p(n), Ap(n);
        precond(r, z);
        p = z;
        double rz = 0, b2 = 0;
        for (size_t i = 0; i < n; i++) { rz += r[i] * z[i]; b2 += b[i] * b[i]; }
        if (b2 == 0) return x;
        const int max_it = (int)std::min<size_t>(20 * n + 100, 200000);
        for (int it = 0; it < max_it; it++) {
            Apply(p, lambda, Ap);
            double pAp = 0;
            for (size_t i = 0; i < n; i++) pAp += p[i] * Ap[i];
            if (!(pAp > 0)) break;
            const double alpha = rz / pAp;
            double r2 = 0;
            for (size_t i = 0; i < n; i++) { x[i] += alpha * p[i]; r[i] -= alpha * Ap[i]; r2 += r[i] * r[i]; }
            if (r2 <= 1e-24 * b2) break;
            precond(r, z);
            double rz_new = 0;
            for (size_t i = 0; i < n; i++) rz_new += r[i] * z[i];
            const double beta = rz_new / rz;
            rz = rz_new;
            for (size_t i = 0; i < n; i++) p[i] = z[i] + beta * p[i];
        }
        return x;
    }
};

template <typename T> class Localizer;
template <typename T> class LoopCloser;

// ------------------------------------------------------------------ MapManager (MapManager.hpp)
template <typename T>
class MapManager {
public:
    IMPORT_PGSLAM_TYPES(T)
    using Ptr = std::shared_ptr<MapManager<T>>;
    PoseGraph<T> &GetGraph() { return graph_; }
    //! MapManagerMT::GetGraphLock (MapManagerMT.hpp:17-21).  Taken by the single-thread flavour too (uncontended there);
    //! recursive, because in that flavour the loop closer and the optimiser run nested inside the localizer's update.
    std::unique_lock<std::recursive_mutex> GetGraphLock() { return std::unique_lock<std::recursive_mutex>(graph_mutex_); }
    size_t GetFixedVertex() const { return fixed_vertex_; }
    void SetLocalizer(std::shared_ptr<Localizer<T>> p) { localizer_ = p; }
    void SetLoopCloser(std::shared_ptr<LoopCloser<T>> p) { loop_closer_ = p; }
    size_t AddFirstKeyframe(DPPtr cloud, const Matrix &T_world_kf)
    {
        const size_t v = graph_.AddVertex(MakeKeyframe(cloud, T_world_kf));
        fixed_vertex_ = v;                                  // kept fixed during optimisation (MapManager.hpp:56-58)
        return v;
    }
    size_t AddNewKeyframe(size_t from, const Matrix &T_world_newkf, const Matrix &meas_T_from_newkf, const Matrix &meas_cov, DPPtr cloud);
    void AddLoopClosingConstraint(size_t from, size_t to, const Matrix &T_from_to, const Matrix &cov)
    {
        Constraint c;
        c.type = Constraint::kLoopConstraint; c.T_from_to = T_from_to; c.cov_from_to = cov; c.weight = (T)PoseWeight(T_from_to);
        graph_.AddEdge(from, to, c);
    }
    void UpdateKeyframeTransform(size_t v, const Matrix &updated, typename Types<T>::Time t) { graph_[v].optimized_T_world_kf = updated; graph_[v].update_time = t; version_++; }
    //! bumped whenever an optimisation rewrites keyframe poses: a localizer that has seen this version is up to date
    unsigned long long Version() const { return version_; }
    //! Keyframe clouds in device memory (graph locked): the copy is made on first use, through a context of the device that
    //! will read it, and the least recently used copies are dropped once the budget is exceeded -- a long drive does not fill
    //! HBM with clouds of places left behind (the host cloud stays; a dropped copy is uploaded again when a composition names
    //! the keyframe once more).  Budget: PGSLAM_DEVICE_KEYFRAMES_MB (default 8192); the `keep` newest uses are never dropped.
    void EnsureKeyframeResident(size_t v, pgicp_ctx *ctx)
    {
        Keyframe &kf = graph_[v];
        if (!kf.device_cloud) { EnsureKeyframeOnDevice<T>(kf, ctx); resident_bytes_ += DeviceKeyframeBytes<T>(kf); device_uploads_++; }
        resident_lru_.remove(v);
        resident_lru_.push_front(v);
        while (resident_bytes_ > device_budget_ && resident_lru_.size() > 16) {
            const size_t old = resident_lru_.back();
            resident_lru_.pop_back();
            resident_bytes_ -= std::min(resident_bytes_, DeviceKeyframeBytes<T>(graph_[old]));
            graph_[old].device_cloud.reset();                  // (copies of the keyframe still in use keep theirs until they go)
            device_evictions_++;
        }
    }
    void SetDeviceKeyframeBudgetMB(double mb) { device_budget_ = (size_t)(mb * 1048576.0); }
    size_t resident_keyframes() const { return resident_lru_.size(); }
    size_t resident_bytes() const { return resident_bytes_; }
    size_t device_uploads() const { return device_uploads_; }
    size_t device_evictions() const { return device_evictions_; }
    void NotifyKeyframeUpdate();
    void WriteGraphviz(const std::string &path)
    {
        std::ofstream ofs(path);
        ofs << "graph G {\n";
        for (size_t v = 0; v < graph_.NumVertices(); v++) ofs << graph_[v].id << " [label=" << graph_[v].id << "];\n";
        for (size_t e = 0; e < graph_.NumEdges(); e++) ofs << graph_[graph_.Edge(e).from].id << "--" << graph_[graph_.Edge(e).to].id << " ;\n";
        ofs << "}\n";
    }

private:
    Keyframe MakeKeyframe(DPPtr cloud, const Matrix &T_world_kf)
    {
        Keyframe kf;
        kf.id = graph_.NumVertices(); kf.cloud_ptr = cloud; kf.T_world_kf = T_world_kf; kf.optimized_T_world_kf = T_world_kf;
        kf.update_time = std::chrono::high_resolution_clock::now();
        return kf;
    }
    PoseGraph<T> graph_;
    std::recursive_mutex graph_mutex_;
    std::list<size_t> resident_lru_;                 // vertices with a device copy, most recently used first
    size_t resident_bytes_ = 0, device_uploads_ = 0, device_evictions_ = 0;
    size_t device_budget_ = (size_t)((std::getenv("PGSLAM_DEVICE_KEYFRAMES_MB") ? std::atof(std::getenv("PGSLAM_DEVICE_KEYFRAMES_MB")) : 8192.0) * 1048576.0);
    unsigned long long version_ = 0;
    size_t fixed_vertex_ = 0;
    std::weak_ptr<Localizer<T>> localizer_;
    std::weak_ptr<LoopCloser<T>> loop_closer_;
};

// ------------------------------------------------------------------ Optimizer (Optimizer.hpp)
template <typename T>
class Optimizer {
public:
    IMPORT_PGSLAM_TYPES(T)
    using Ptr = std::shared_ptr<Optimizer<T>>;
    using InputData = std::tuple<size_t, size_t, Matrix, Matrix>;       // from, to, T_from_to, COV_from_to (Optimizer.h:22)
    explicit Optimizer(typename MapManager<T>::Ptr mm) : map_manager_(mm) {}
    virtual ~Optimizer() {}
    virtual void AddNewData(size_t from, size_t to, const Matrix &T_from_to, const Matrix &cov)
    {
        data_buffer_.clear();
        data_buffer_.push_back(std::make_tuple(from, to, T_from_to, cov));
        ProcessData();
    }
    //! OptimizerMT::Main drains its whole buffer into one solve (OptimizerMT.hpp:59-65); the batched loop-closing
    //! dispatcher feeds the gathered edges through this
    void AddNewDataBatch(const std::vector<InputData> &batch) { data_buffer_ = batch; if (!batch.empty()) ProcessData(); }
    int last_iterations() const { return last_iterations_; }
    double last_initial_error() const { return e0_; }
    double last_final_error() const { return e1_; }
    int runs() const { return runs_; }                                  // solves so far and the host seconds they took
    double total_seconds() const { return seconds_; }
    int total_iterations() const { return iterations_; }

protected:
    //! [x y z rx ry rz] -> [rx ry rz x y z] (Optimizer.hpp:32-42), row-major double
    static void PmCovToRotFirst(const Matrix &m, double out[36])
    {
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) out[6 * i + j] = (double)m((i + 3) % 6, (j + 3) % 6);
    }
    void AddFactor(PoseGraphLeastSquares &ls, size_t from, size_t to, const Matrix &Tm, const Matrix &cov)
    {
        PoseGraphLeastSquares::Between f;
        f.from = from; f.to = to; f.Z = se3::from_matrix(Tm);
        double c[36];
        PmCovToRotFirst(cov, c);
        if (!PoseGraphLeastSquares::InformationFactor(c, f.L)) throw std::runtime_error("[Optimizer] constraint covariance is not positive definite");
        ls.factors.push_back(f);
    }
    //! PrepareForOptimization (graph locked) / the solve (unlocked) / UpdateAfterOptimization (graph locked): OptimizerMT.hpp:70-84
    void ProcessData()
    {
        auto &g = map_manager_->GetGraph();
        PoseGraphLeastSquares ls;
        {
            auto lock = map_manager_->GetGraphLock();
            for (size_t e = 0; e < g.NumEdges(); e++) AddFactor(ls, g.Edge(e).from, g.Edge(e).to, g.Edge(e).c.T_from_to, g.Edge(e).c.cov_from_to);
            for (auto &d : data_buffer_) AddFactor(ls, std::get<0>(d), std::get<1>(d), std::get<2>(d), std::get<3>(d));
            for (size_t v = 0; v < g.NumVertices(); v++) ls.X.push_back(se3::from_matrix(g[v].optimized_T_world_kf));
            ls.fixed = map_manager_->GetFixedVertex();                     // prior with sigma 1e-6 (Optimizer.hpp:122-130)
        }
        const auto t0 = std::chrono::steady_clock::now();
        ls.Optimize();
        seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        runs_++; iterations_ += ls.iterations;
        last_iterations_ = ls.iterations; e0_ = ls.initial_error; e1_ = ls.final_error;
        {
            auto lock = map_manager_->GetGraphLock();
            const auto now = std::chrono::high_resolution_clock::now();
            // (keyframes added while the solve ran keep their poses: they were not variables of it)
            for (size_t v = 0; v < ls.X.size(); v++) map_manager_->UpdateKeyframeTransform(v, se3::to_matrix<Matrix>(ls.X[v]), now);
            for (auto &d : data_buffer_) map_manager_->AddLoopClosingConstraint(std::get<0>(d), std::get<1>(d), std::get<2>(d), std::get<3>(d));
        }
        map_manager_->NotifyKeyframeUpdate();
    }
    typename MapManager<T>::Ptr map_manager_;
    std::vector<InputData> data_buffer_;
    int last_iterations_ = 0, runs_ = 0, iterations_ = 0;
    double e0_ = 0, e1_ = 0, seconds_ = 0;
};

// ------------------------------------------------------------------ loop closer with its candidate search
template <typename T>
class LoopCloser {
public:
    IMPORT_PGSLAM_TYPES(T)
    using Ptr = std::shared_ptr<LoopCloser<T>>;
    LoopCloser(typename MapManager<T>::Ptr mm, typename Optimizer<T>::Ptr opt) : map_manager_(mm), optimizer_(opt) {}
    void SetTopologicalDistanceThreshold(T v) { topo_dist_threshold_ = v; }
    void SetGeometricalDistanceThreshold(T v) { geom_dist_threshold_ = v; }
    void SetOverlapThreshold(T v) { overlap_threshold_ = v; if (closer_) closer_->SetOverlapThreshold(v); }
    void SetResidualErrorThreshold(T v) { residual_error_threshold_ = v; if (closer_) closer_->SetResidualErrorThreshold(v); }
    virtual void SetIcpConfigFromString(const std::string &yaml) { closer().SetIcpConfigFromString(yaml); }
    //! LoopCloser.hpp:58-74: the chain from a YAML FILE
    void SetIcpConfig(const std::string &config_path) { SetIcpConfigFromString(slurp_config_file(config_path)); }
    //! LoopCloser.hpp:53-56 -- the reference IGNORES its argument (`candidate_local_map_ = LocalMap(3)`): kept, so that a
    //! configuration written against it gives the same candidates here.  SetCandidateLocalMapCapacity is the setter that means it.
    void SetCandidateLocalMapMaxSize(size_t /*size*/) { capacity_ = 3; }
    void SetCandidateLocalMapCapacity(size_t size) { capacity_ = size < 1 ? 1 : size; }
    virtual ~LoopCloser() {}
    virtual void AddNewVertex(size_t v) { ProcessVertex(v); }
    int loops_closed() const { return loops_closed_; }
    int candidates_tried() const { return candidates_tried_; }
    typename PM::ICP &icp() { return closer().icp(); }

    //! LoopCloser.hpp:193-305: vertices geometrically close (<= geom threshold) and topologically far (> topo
    //! threshold) from input_v, nearest first; around the first one that admits it, the `capacity` vertices a
    //! Dijkstra on the graph without loop edges and without the topologically close vertices settles first.
    bool FindLocalMapCandidate(size_t input_v, std::vector<size_t> &composition) const
    {
        auto &g = map_manager_->GetGraph();
        const auto topo = g.Dijkstra(input_v, nullptr, nullptr, nullptr);
        std::vector<double> geom(g.NumVertices());
        for (size_t v = 0; v < g.NumVertices(); v++) geom[v] = PoseDistance(g[v].optimized_T_world_kf, g[input_v].optimized_T_world_kf);
        std::vector<size_t> cand;
        for (size_t v = 0; v < g.NumVertices(); v++)
            if (geom[v] <= (double)geom_dist_threshold_ && topo[v] > (double)topo_dist_threshold_) cand.push_back(v);
        std::stable_sort(cand.begin(), cand.end(), [&](size_t a, size_t b) { return geom[a] < geom[b]; });
        auto keep_v = [&](size_t v) { return topo[v] > (double)topo_dist_threshold_; };
        auto keep_e = [&](size_t e) { return g.Edge(e).c.type != Constraint::kLoopConstraint; };
        for (size_t cv : cand) {
            std::deque<size_t> comp;
            g.Dijkstra(cv, keep_v, keep_e, [&](size_t v) { comp.push_front(v); return comp.size() < capacity_; });   // record_n_and_stop
            if (comp.size() == capacity_) { composition.assign(comp.begin(), comp.end()); return true; }
        }
        return false;
    }
    //! what ProcessVertex hands to the ICP: LoopCloser.hpp:86-95 (graph locked while it is put together, LoopCloserMT.hpp:72-76)
    //! `reading_dev` / `reference_dev`: the same two clouds in device memory when the candidate was put together there (`device_ctx`
    //! given and the keyframes resident): the candidate map is assembled in HBM from the resident keyframe clouds, as the
    //! localizer's maps are -- `reference` (the host cloud) is then made only on demand (`host_reference`)
    struct PreparedCandidate {
        size_t ref_v, input_v; DPPtr reading; DPPtr reference; Matrix guess;
        std::shared_ptr<pgslam_amd::DeviceCloud<T>> reading_dev, reference_dev;
        std::function<DP()> host_reference;
    };
    void SetDeviceCandidates(bool on) { device_candidates_ = on; }
    size_t device_candidates() const { return device_candidates_made_; }
    bool PrepareCandidate(size_t input_v, PreparedCandidate &out, pgicp_ctx *device_ctx = nullptr)
    {
        auto lock = map_manager_->GetGraphLock();
        auto &g = map_manager_->GetGraph();
        if (g.NumVertices() < 2) return false;
        std::vector<size_t> comp;
        if (!FindLocalMapCandidate(input_v, comp)) return false;
        candidates_tried_++;
        // candidate local map: composition order, reference = back (LocalMap::UpdateToNewComposition)
        LocalMap<T> lm(capacity_);
        out.ref_v = comp.back(); out.input_v = input_v;
        out.reading = g[input_v].cloud_ptr;
        out.guess = g[out.ref_v].optimized_T_world_kf.inverse() * g[input_v].optimized_T_world_kf;           // LoopCloser.hpp:95
        if (device_ctx && device_candidates_) {
            for (size_t v : comp) { map_manager_->EnsureKeyframeResident(v, device_ctx); lm.PushKeyframe(g[v]); }
            map_manager_->EnsureKeyframeResident(input_v, device_ctx);
            const std::vector<Keyframe> order = lm.AssemblyOrder();
            out.reference_dev = std::make_shared<pgslam_amd::DeviceCloud<T>>();
            BuildLocalMapOnDevice<T>(device_ctx, order, *out.reference_dev);
            out.reading_dev = g[input_v].device_cloud;
            out.host_reference = [order]() { return BuildLocalMapCloud<T>(order); };
            out.reference.reset();
            device_candidates_made_++;
            return true;
        }
        for (size_t v : comp) lm.PushKeyframe(g[v]);
        lm.BuildCloudFromData();
        out.reference = std::make_shared<DP>(lm.Cloud());
        return true;
    }
    void ProcessVertex(size_t input_v)
    {
        PreparedCandidate c;
        PairLoopCloser<T> &lc = closer();
        if (!PrepareCandidate(input_v, c, lc.DeviceCandidateEquivalent() ? (pgicp_ctx *)lc.icp().ctx : nullptr)) return;
        auto r = c.reference_dev ? lc.ProcessCandidateOnDevice(c.reading, *c.reading_dev, *c.reference_dev, c.guess, c.host_reference)
                                 : lc.ProcessCandidate(*c.reading, *c.reference, c.guess);                     // :98 + CheckIcpResult
        if (r.accepted) {
            loops_closed_++;
            optimizer_->AddNewData(c.ref_v, input_v, r.T_refkf_kf, r.cov);                            // :104-108
        }
    }

protected:
    typename MapManager<T>::Ptr map_manager_;
    //! the ICP object (and with it the device context) is created on first use: the graph searches need no GPU
    PairLoopCloser<T> &closer()
    {
        if (!closer_) {
            closer_.reset(new PairLoopCloser<T>());
            closer_->SetOverlapThreshold(overlap_threshold_);
            closer_->SetResidualErrorThreshold(residual_error_threshold_);
        }
        return *closer_;
    }
    typename Optimizer<T>::Ptr optimizer_;
    std::unique_ptr<PairLoopCloser<T>> closer_;
    T topo_dist_threshold_ = T(3), geom_dist_threshold_ = T(3);               // LoopCloser.hpp:16-17
    T overlap_threshold_ = T(0.8), residual_error_threshold_ = T(5000);       // :18-19
    size_t capacity_ = 3;                                                     // :20
    int loops_closed_ = 0, candidates_tried_ = 0;
    bool device_candidates_ = std::getenv("PGSLAM_HOST_LOOP_CANDIDATES") == nullptr;     // candidate maps assembled in HBM (default)
    size_t device_candidates_made_ = 0;
};

//! one background job at a time on a thread of its own (made at first use): start(fn) ... wait()
class HostWorker {
public:
    ~HostWorker()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        if (t_.joinable()) t_.join();
    }
    void start(std::function<void()> fn)
    {
        wait();
        { std::lock_guard<std::mutex> l(m_); job_ = std::move(fn); busy_ = true; if (!t_.joinable()) t_ = std::thread(&HostWorker::Main, this); }
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [this] { return !busy_; });
        if (err_) { std::exception_ptr e = err_; err_ = nullptr; std::rethrow_exception(e); }
    }
private:
    void Main()
    {
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return stop_ || (busy_ && job_); });
                if (stop_) return;
                fn = std::move(job_);
                job_ = nullptr;
            }
            std::exception_ptr e;
            try { fn(); } catch (...) { e = std::current_exception(); }
            { std::lock_guard<std::mutex> l(m_); busy_ = false; err_ = e; }
            cv_.notify_all();
        }
    }
    std::thread t_;
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool busy_ = false, stop_ = false;
    std::exception_ptr err_;
};

// ------------------------------------------------------------------ localizer on the graph
template <typename T>
class Localizer {
public:
    IMPORT_PGSLAM_TYPES(T)
    using Ptr = std::shared_ptr<Localizer<T>>;
    explicit Localizer(typename MapManager<T>::Ptr mm, size_t capacity = 3)
        : map_manager_(mm), capacity_(capacity), rigid_(PM::get().REG(Transformation).create("RigidTransformation")),
          T_refkf_robot_(Matrix::Identity(4, 4)), T_world_robot_(Matrix::Identity(4, 4)), last_input_(Matrix::Identity(4, 4)) {}
    void SetOverlapThreshold(T v) { overlap_threshold_ = v; }
    void SetIcpConfigFromString(const std::string &yaml) { icp_yaml_ = yaml; probe_.reset(); probe_comp_.clear(); std::istringstream iss(yaml); icp_sequence_.loadFromYaml(iss); }
    void SetInputFiltersConfigFromString(const std::string &yaml) { std::istringstream iss(yaml); input_filters_ = DataPointsFilters(iss); }
    //! Localizer.hpp:54-78: the chain / the input filters from a YAML FILE
    void SetIcpConfig(const std::string &config_path) { SetIcpConfigFromString(slurp_config_file(config_path)); }
    void SetInputFiltersConfig(const std::string &config_path) { SetInputFiltersConfigFromString(slurp_config_file(config_path)); }
    //! Localizer.hpp:36-41: keyframes of the local map (and of the candidate compositions).  The reference replaces its LocalMap
    //! object; here the current composition is cut down to its newest `size` keyframes and the map re-assembled on the next scan.
    void SetLocalMapMaxSize(size_t size)
    {
        capacity_ = size < 1 ? 1 : size;
        if (comp_.size() > capacity_) { comp_.erase(comp_.begin(), comp_.end() - (std::ptrdiff_t)capacity_); probe_.reset(); probe_comp_.clear(); if (!comp_.empty()) Rebuild(); }
    }
    //! Localizer.hpp:49-52
    void SetMinimalOverlapThreshold(T v) { minimal_overlap_ = v; }
    const Matrix &T_world_robot() const { return T_world_robot_; }
    const std::vector<size_t> &composition() const { return comp_; }
    int rebuilds() const { return rebuilds_; }
    ICPSequence &icp() { return icp_sequence_; }

    virtual ~Localizer() {}
    virtual void AddNewData(unsigned long long, const std::string &, const Matrix &T_world_robot, const Matrix &T_robot_sensor, DPPtr cloud)
    {
        ProcessData(T_world_robot, T_robot_sensor, cloud);
    }
    //! host seconds spent in: [0] input filters + sensor transform, [1] the ICP call, [2] everything after it
    //! (neighbour composition, overlap probe, keyframe insertion, local-map rebuilds)
    const double *phase_seconds() const { return phase_s_; }
    //! inside [2]: [0] neighbour-composition search, [1] overlap probe, [2] local-map rebuilds, [3] keyframe insertion (the
    //! single-thread flavour closes loops and optimises inside it)
    const double *after_icp_seconds() const { return sub_s_; }
    void ProcessData(const Matrix &input_T_world_robot, const Matrix &input_T_robot_sensor, DPPtr cloud)
    {
        using clk = std::chrono::steady_clock;
        const auto t0 = clk::now();
        input_cloud_ = cloud;
        // (single-thread flavour, sensor at the robot's origin: the input stage leaves the handful of dropped points in the HOST
        // cloud for the moment -- the device copy is complete -- and their gaps are closed on the worker thread while the ICP runs)
        Deferred defer;
        typename PM::ICPChainBase::DeviceReading direct = PreProcess(input_T_robot_sensor, cloud, &defer);
        struct Join { HostWorker &w; bool on; ~Join() { if (on) { try { w.wait(); } catch (...) {} } } } join{host_worker_, false};
        if (defer.cloud) {
            if (comp_.empty()) PM::compactByDropped(*defer.cloud, defer.dropped);      // (the first cloud becomes a keyframe at once)
            else { host_worker_.start([defer]() { PM::compactByDropped(*defer.cloud, defer.dropped); }); join.on = true; deferred_compactions_++; }
        }
        const auto t1 = clk::now();
        phase_s_[0] += std::chrono::duration<double>(t1 - t0).count();
        auto &g = map_manager_->GetGraph();
        if (comp_.empty()) {                                                 // ProcessFirstCloud (graph locked, LocalizerMT.hpp:104-108)
            auto lock = map_manager_->GetGraphLock();
            comp_.push_back(map_manager_->AddFirstKeyframe(cloud, input_T_world_robot));
            Rebuild();
            T_refkf_robot_ = Matrix::Identity(4, 4);
            T_world_robot_ = input_T_world_robot;
            last_input_ = input_T_world_robot;
            return;
        }
        const Matrix d = last_input_.inverse() * input_T_world_robot;
        // the ICP runs outside the graph lock (LocalizerMT.hpp:95-96); on the device copy of the cloud if its upload was
        // started while the previous scan aligned (Prefetch), from the host cloud otherwise
        // (only THIS cloud's slot is consumed: the slot of the scan after it, whose transfer is under way, stays)
        typename PM::ICPChainBase::DeviceReading ahead;
        if (Prefetched *slot = FindPrefetched(cloud.get())) { ahead = slot->reading; *slot = Prefetched(); }   // (freed even if the ICP throws)
        if (!ahead && direct) ahead = direct;                    // the device copy the input stage left behind (no second upload)
        input_device_ = ahead && ahead.filtered.get() == cloud.get() ? ahead : typename PM::ICPChainBase::DeviceReading();
        // (whichever way the reading travels the ICP aligns the same points in the same order -- a device reading is only taken when
        //  the chain's reading filters change nothing -- so its correspondences can seed the overlap probe of this scan)
        last_icp_on_device_reading_ = false;
        if (ahead) { T_refkf_robot_ = icp_sequence_(ahead, T_refkf_robot_ * d); device_readings_used_++; last_icp_on_device_reading_ = (bool)input_device_; }
        else T_refkf_robot_ = icp_sequence_(*cloud, T_refkf_robot_ * d);
        if (join.on) { join.on = false; host_worker_.wait(); }       // the host cloud is whole again before anything below looks at it
        const auto t2 = clk::now();
        phase_s_[1] += std::chrono::duration<double>(t2 - t1).count();
        {
            auto lock = map_manager_->GetGraphLock();                        // LocalizerMT::UpdateAfterIcp, LocalizerMT.hpp:110-121
            // the graph may have been optimised while the ICP ran (upstream re-reads it unconditionally; here only when
            // an optimisation has actually rewritten poses since the last look)
            if (resync_before_update_ && synced_version_ != map_manager_->Version()) UpdateFromGraphNow();
            T_world_robot_ = g[comp_.back()].optimized_T_world_kf * T_refkf_robot_;
            UpdateAfterIcp();
        }
        last_input_ = input_T_world_robot;
        phase_s_[2] += std::chrono::duration<double>(clk::now() - t2).count();
    }
    //! Localizer.hpp:103-106: the input filters in place, then sensor frame -> robot frame.  Once per cloud (a cloud whose
    //! upload was prefetched has been through it already).
    //! Returns the device copy of the pre-processed cloud when the input stage ran on the device (pgicp_filter_cloud: filters
    //! and transform in one pass over the uploaded scan) and that copy can stand for the cloud in the ICP; empty otherwise.
    //! what the host cloud still holds too much after a deferred device pass (PM::filterOnDeviceDeferred)
    struct Deferred { DPPtr cloud; std::vector<int32_t> dropped; };
    size_t deferred_compactions() const { return deferred_compactions_; }
    typename PM::ICPChainBase::DeviceReading PreProcess(const Matrix &input_T_robot_sensor, DPPtr cloud, Deferred *defer = nullptr)
    {
        // per-cloud state: while scan k aligns, scan k + 1 has been through here already (Prefetch) -- one "last cloud"
        // memo would alternate between the two and send every prefetched cloud through the filters and the sensor
        // transform a second time, in place
        typename PM::ICPChainBase::DeviceReading direct;
        Prefetched *slot = FindPrefetched(cloud.get());
        if (slot && slot->preprocessed) return direct;
        direct = PreProcessOn(icp_sequence_.ctx, input_T_robot_sensor, cloud, defer);
        if (slot) slot->preprocessed = true;
        return direct;
    }
    //! The input stage of one cloud on a context of the caller's choosing (the MT flavour's pre-processing thread runs it on its
    //! own, while this object's thread aligns the scan before): touches the cloud, the input filters and `rigid_` only.
    typename PM::ICPChainBase::DeviceReading PreProcessOn(pgicp_ctx *ctx, const Matrix &input_T_robot_sensor, DPPtr cloud, Deferred *defer = nullptr)
    {
        typename PM::ICPChainBase::DeviceReading direct;
        const T *dev = nullptr;
        // deferred form: only when the ICP will run on the device copy and nobody observes the host cloud during it
        if (defer && device_input_stage_ && deferred_host_compaction_ && icp_sequence_.deviceReadingEquivalent() && !icp_sequence_.onAlign) {
            int kept = 0;
            if (PM::filterOnDeviceDeferred(ctx, input_filters_, *cloud, input_T_robot_sensor, &dev, &kept, defer->dropped)) {
                device_input_stages_++;
                direct.dev = dev; direct.filtered = cloud; direct.n = kept; direct.stride = cloud->xyzStride();
                if (!defer->dropped.empty()) defer->cloud = cloud;
                return direct;
            }
        }
        if (device_input_stage_ && PM::filterAndTransformOnDevice(ctx, input_filters_, *cloud, input_T_robot_sensor, &dev)) {
            device_input_stages_++;
            if (icp_sequence_.deviceReadingEquivalent()) { direct.dev = dev; direct.filtered = cloud; }
        } else {
            input_filters_.apply(*cloud);
            (*cloud) = rigid_->compute(*cloud, input_T_robot_sensor);
        }
        return direct;
    }
    //! a cloud whose input stage has run elsewhere (PreProcessOn): ProcessData takes it as pre-processed, with `reading` as its
    //! device copy (empty: the host cloud is aligned)
    void AdoptPreprocessed(DPPtr cloud, const typename PM::ICPChainBase::DeviceReading &reading)
    {
        Prefetched *slot = FindPrefetched(cloud.get());
        if (!slot) slot = FindPrefetched(nullptr);
        if (!slot) { prefetched_[0] = Prefetched(); slot = &prefetched_[0]; }
        slot->cloud = cloud.get(); slot->preprocessed = true; slot->reading = reading;
        prefetches_++;
    }
    //! scans whose input filters + sensor transform ran as one device pass
    size_t device_input_stages() const { return device_input_stages_; }
    void SetDeviceInputStage(bool on) { device_input_stage_ = on; }
    //! local maps assembled from device-resident keyframe clouds (default) or through the host, as upstream does
    void SetDeviceLocalMap(bool on) { device_local_map_ = on; }
    size_t device_rebuilds() const { return device_rebuilds_; }
    size_t probes_seeded() const { return probe_seeded_; }
    //! The NEXT scan, already queued (LocalizerMT.hpp:27-40): pre-process it and start its transfer to the device on the ICP
    //! context's copy stream (pgicp_upload_*), so that it travels while the current scan aligns.  Needs a map (the chain
    //! aligns device readings only against one) -- before the first keyframe nothing is prefetched.
    void Prefetch(const Matrix &input_T_robot_sensor, DPPtr cloud)
    {
        if (comp_.empty() || !icp_sequence_.hasMap() || FindPrefetched(cloud.get())) return;
        // two slots, as the context keeps two upload sets (pgicp_upload_*): the scan being aligned and the one after it
        Prefetched *slot = FindPrefetched(nullptr);
        if (!slot) return;
        slot->cloud = cloud.get();
        const auto direct = PreProcess(input_T_robot_sensor, cloud);
        // (a chain that looks at the reading's descriptors -- SurfaceNormalOutlierFilter -- or filters the reading takes the host
        // cloud in ProcessData: nothing is uploaded ahead for it, the slot only remembers that the cloud is pre-processed)
        if (direct) slot->reading = direct;
        else if (icp_sequence_.deviceReadingEquivalent()) slot->reading = icp_sequence_.uploadReading(*cloud);
        prefetches_++;
    }
    size_t prefetches() const { return prefetches_; }
    //! scans whose ICP ran on a device copy uploaded ahead of time
    size_t device_readings_used() const { return device_readings_used_; }
    //! MapManager::NotifyKeyframeUpdate -> Localizer::UpdateFromGraph (Localizer.hpp:155-176): after an
    //! optimisation the local map is rebuilt from the corrected poses and the world pose follows the reference
    virtual void UpdateFromGraph() { UpdateFromGraphNow(); }
    void UpdateFromGraphNow()
    {
        auto lock = map_manager_->GetGraphLock();
        synced_version_ = map_manager_->Version();
        if (comp_.empty()) return;
        Rebuild();
        T_world_robot_ = map_manager_->GetGraph()[comp_.back()].optimized_T_world_kf * T_refkf_robot_;
    }

protected:
    struct Prefetched {
        const DP *cloud = nullptr;                              // null: the slot is free
        bool preprocessed = false;
        typename PM::ICPChainBase::DeviceReading reading;
    };
    Prefetched prefetched_[2];
    Prefetched *FindPrefetched(const DP *cloud)
    {
        for (auto &p : prefetched_) if (p.cloud == cloud) return &p;
        return nullptr;
    }
    size_t prefetches_ = 0, device_readings_used_ = 0, device_input_stages_ = 0, deferred_compactions_ = 0;
    bool deferred_host_compaction_ = std::getenv("PGSLAM_SYNC_HOST_COMPACTION") == nullptr;
    HostWorker host_worker_;
    typename PM::ICPChainBase::DeviceReading input_device_;      // the current scan's device copy (while it is valid: this ProcessData)
    bool device_input_stage_ = std::getenv("PGSLAM_HOST_INPUT_STAGE") == nullptr;
    bool device_local_map_ = std::getenv("PGSLAM_HOST_LOCAL_MAP") == nullptr;      // keyframe clouds resident, maps assembled in HBM
    pgslam_amd::DeviceCloud<T> map_assembly_, probe_assembly_;                   // assembly buffers (grow-only), one per chain
    // the two maps as concatenations of keyframe clouds (identity of a cloud: its device copy; size: its points), in assembly order --
    // what the overlap probe's seeds are mapped through (pgicp_partial_chain_seeded); empty: that map was not assembled on the device
    std::vector<std::pair<const void *, int>> map_segs_, probe_segs_;
    static std::vector<std::pair<const void *, int>> SegmentsOf(const std::vector<Keyframe> &order)
    {
        std::vector<std::pair<const void *, int>> out;
        for (auto &kf : order) out.emplace_back((const void *)kf.device_cloud.get(), kf.device_cloud ? kf.device_cloud->n : 0);
        return out;
    }
    size_t device_rebuilds_ = 0, probe_seeded_ = 0;
    bool last_icp_on_device_reading_ = false;        // the last ICP aligned input_device_ (same points, same order as the probe's reading)
    bool resync_before_update_ = false;              // the MT flavour re-reads the graph before every update
    unsigned long long synced_version_ = 0;
    void Rebuild()
    {
        auto &g = map_manager_->GetGraph();
        LocalMap<T> lm(capacity_);
        // Keyframe clouds are immutable: each goes to device memory once, and a rebuild assembles the map THERE and indexes it
        // in place -- no cloud crosses PCIe (the host flow uploads the keyframes, downloads the assembled map and uploads it
        // again: 24 MB per rebuild at 100 k-pt scans).  Same values bit for bit; the host copy is made when somebody asks.
        if (device_local_map_ && icp_sequence_.deviceMapEquivalent()) {
            for (size_t v : comp_) { map_manager_->EnsureKeyframeResident(v, icp_sequence_.ctx); lm.PushKeyframe(g[v]); }
            const std::vector<Keyframe> order = lm.AssemblyOrder();
            BuildLocalMapOnDevice<T>(icp_sequence_.ctx, order, map_assembly_);
            icp_sequence_.setMap(map_assembly_, [order]() { return BuildLocalMapCloud<T>(order); });
            map_segs_ = SegmentsOf(order);
            device_rebuilds_++;
        } else {
            map_segs_.clear();
            for (size_t v : comp_) lm.PushKeyframe(g[v]);
            lm.BuildCloudFromData();
            icp_sequence_.setMap(lm.Cloud());
        }
        rebuilds_++;
    }
    T OverlapWith(const std::vector<size_t> &comp)                          // ComputeOverlapWith, Localizer.hpp:282-348
    {
        auto &g = map_manager_->GetGraph();
        if (!probe_) {                                  // kept between calls: its ICP objects own device contexts
            probe_.reset(new ScanLocalizer<T>());
            probe_->SetIcpConfigFromString(icp_yaml_);
            probe_comp_.clear();
        }
        // The neighbour composition is the same scan after scan until the vehicle moves on or the graph is optimised: its
        // world-frame map -- keyframe clouds (immutable) under their optimised poses (unchanged while the graph's version
        // is) -- is assembled, filtered and indexed once and asked again per scan (upstream rebuilds it, kd-tree included,
        // every time: Localizer.hpp:288-317; the answer is the same).
        static const bool always_rebuild = std::getenv("PGSLAM_PROBE_REBUILD") != nullptr;   // tests: the reference's own flow
        if (always_rebuild || probe_comp_ != comp || probe_version_ != map_manager_->Version()) {
            LocalMap<T> lm(capacity_);
            bool prepared = false;
            if (device_local_map_) {                    // (as Rebuild: assembled and moved to the world frame in device memory)
                for (size_t v : comp) { map_manager_->EnsureKeyframeResident(v, probe_->OverlapContext()); lm.PushKeyframe(g[v]); }
                const Matrix T_world_ref = g[comp.back()].optimized_T_world_kf;
                const std::vector<Keyframe> porder = lm.AssemblyOrder();
                BuildLocalMapOnDevice<T>(probe_->OverlapContext(), porder, probe_assembly_, &T_world_ref);
                prepared = probe_->PrepareOverlapReference(probe_assembly_);
                probe_segs_ = prepared ? SegmentsOf(porder) : std::vector<std::pair<const void *, int>>();
            } else {
                for (size_t v : comp) lm.PushKeyframe(g[v]);
            }
            if (!prepared) {
                probe_segs_.clear();
                lm.BuildCloudFromData();
                const DP world_map = rigid_->compute(lm.Cloud(), g[comp.back()].optimized_T_world_kf);
                probe_->PrepareOverlapReference(world_map);
            }
            probe_comp_ = comp;
            probe_version_ = map_manager_->Version();
        }
        // (the scan is on the device already when the input stage ran there: the probe reads that copy)
        if (input_device_ && input_device_.filtered.get() == input_cloud_.get()) {
            // Seeds for the probe's matcher (round 6): the ICP has just matched this very reading against a map that shares keyframes
            // with the candidate composition -- the same physical points, at known index offsets in the two assemblies.  Candidates
            // only: the probe's result is the unseeded one bit for bit (PGSLAM_PROBE_SEEDS=0 turns them off; run_probe_seeds).
            static const bool seeds_on = !(std::getenv("PGSLAM_PROBE_SEEDS") && std::atoi(std::getenv("PGSLAM_PROBE_SEEDS")) == 0);
            if (seeds_on && !map_segs_.empty() && !probe_segs_.empty() && map_segs_.size() <= 16 && last_icp_on_device_reading_) {
                std::vector<int32_t> src_start(map_segs_.size() + 1, 0), dst_start(map_segs_.size(), -1);
                for (size_t k = 0; k < map_segs_.size(); k++) {
                    src_start[k + 1] = src_start[k] + map_segs_[k].second;
                    int at = 0;
                    for (auto &ps : probe_segs_) { if (ps.first == map_segs_[k].first && ps.second == map_segs_[k].second) { dst_start[k] = at; break; } at += ps.second; }
                }
                probe_seeded_++;
                return probe_->ComputeOverlapAgainstPrepared(input_device_, T_world_robot_, icp_sequence_.ctx, src_start, dst_start);
            }
            return probe_->ComputeOverlapAgainstPrepared(input_device_, T_world_robot_);
        }
        return probe_->ComputeOverlapAgainstPrepared(*input_cloud_, T_world_robot_);
    }
    //! Localizer.hpp:393-483
    bool FindNeighborComposition(std::vector<size_t> &out)
    {
        auto &g = map_manager_->GetGraph();
        std::set<size_t> adj;
        for (size_t v : comp_)
            for (size_t e : g.IncidentEdges(v)) { const size_t w = g.Other(e, v); if (std::find(comp_.begin(), comp_.end(), w) == comp_.end()) adj.insert(w); }
        if (adj.empty()) return false;
        size_t closest = *adj.begin();
        double cd = PoseDistance(g[closest].optimized_T_world_kf, T_world_robot_);
        for (size_t v : adj) { const double d = PoseDistance(g[v].optimized_T_world_kf, T_world_robot_); if (d < cd) { cd = d; closest = v; } }
        std::vector<size_t> ext(comp_.begin(), comp_.end());
        ext.push_back(closest);
        auto keep_v = [&](size_t v) { return std::find(ext.begin(), ext.end(), v) != ext.end(); };
        const auto topo = g.Dijkstra(closest, keep_v, nullptr, nullptr);
        // decreasing topological distance from the new vertex (sort on reverse iterators, :455-457)
        std::stable_sort(ext.begin(), ext.end(), [&](size_t a, size_t b) { return topo[a] > topo[b]; });
        std::deque<size_t> comp;
        auto push = [&](size_t v) { comp.push_back(v); if (comp.size() > capacity_) comp.pop_front(); };
        for (size_t i = 0; i + 2 < ext.size(); i++) push(ext[i]);
        const size_t last = ext[ext.size() - 1], before = ext[ext.size() - 2];
        const double dl = PoseDistance(g[last].optimized_T_world_kf, T_world_robot_), db = PoseDistance(g[before].optimized_T_world_kf, T_world_robot_);
        if (db < dl) { push(last); push(before); } else { push(before); push(last); }
        out.assign(comp.begin(), comp.end());
        return true;
    }
public:
    //! The overlap half of IsBetterComposition (Localizer.hpp:363-372): a candidate composition is taken only
    //! when its overlap is itself sufficient (IsOverlapEnough, :355-361) AND larger than the current one --
    //! in the low-overlap branch a neighbour that is better but still insufficient must not stop the new
    //! keyframe (:228-247).  (The "same composition" half cannot occur here: FindNeighborComposition always
    //! brings a vertex from outside the composition.)
    static bool IsBetterOverlap(T current_overlap, T candidate_overlap, T threshold)
    {
        return candidate_overlap >= threshold && candidate_overlap > current_overlap;
    }
private:
    void UpdateAfterIcp()                                                   // Localizer.hpp:178-268
    {
        auto &g = map_manager_->GetGraph();
        const T overlap = icp_sequence_.errorMinimizer->getOverlap();
        std::vector<size_t> next = comp_, neigh;
        if (overlap < minimal_overlap_)                                     // Localizer.hpp:350-355: a warning, nothing else
            std::cerr << "[Localizer] WARNING: overlap below minimal overlap! (" << overlap << " < " << minimal_overlap_ << ")\n";
        const bool enough = overlap >= overlap_threshold_;
        bool took_neighbor = false;
        bool have_neigh;
        { SubTimer t(sub_s_[0]); have_neigh = FindNeighborComposition(neigh); }
        if (have_neigh) {
            T ov;
            { SubTimer t(sub_s_[1]); ov = OverlapWith(neigh); }
            if (IsBetterOverlap(overlap, ov, overlap_threshold_)) { next = neigh; took_neighbor = true; }
        }
        if (!took_neighbor) {
            if (enough) {
                size_t best = 0;
                double bd = PoseDistance(g[comp_[0]].optimized_T_world_kf, T_world_robot_);
                for (size_t i = 1; i < comp_.size(); i++) { const double d = PoseDistance(g[comp_[i]].optimized_T_world_kf, T_world_robot_); if (d < bd) { bd = d; best = i; } }
                if (best != comp_.size() - 1) std::swap(next[best], next[next.size() - 1]);
            } else {
                size_t v;
                { SubTimer t(sub_s_[3]); v = map_manager_->AddNewKeyframe(comp_.back(), T_world_robot_, T_refkf_robot_, icp_sequence_.errorMinimizer->getCovariance(), input_cloud_); }
                // the loop closer may have optimised the graph inside AddNewKeyframe: comp_ poses are re-read below
                next = comp_;
                next.push_back(v);
                if (next.size() > capacity_) next.erase(next.begin());
            }
        }
        const bool same_set = std::set<size_t>(next.begin(), next.end()) == std::set<size_t>(comp_.begin(), comp_.end());
        if (!(same_set && next.back() == comp_.back())) {
            const size_t old_ref = comp_.back();
            comp_ = next;
            { SubTimer t(sub_s_[2]); Rebuild(); }
            if (comp_.back() != old_ref) T_refkf_robot_ = g[comp_.back()].optimized_T_world_kf.inverse() * T_world_robot_;
        }
    }
    typename MapManager<T>::Ptr map_manager_;
    size_t capacity_;
    TransformationPtr rigid_;
    DataPointsFilters input_filters_;
    ICPSequence icp_sequence_;
    std::string icp_yaml_;
    std::unique_ptr<ScanLocalizer<T>> probe_;            // runs ComputeOverlapWith for candidate compositions
    std::vector<size_t> probe_comp_;                 // the composition whose world-frame map the probe holds indexed ...
    unsigned long long probe_version_ = 0;           // ... built at this version of the graph
    DPPtr input_cloud_;
    std::vector<size_t> comp_;                       // local map composition, reference keyframe last
    Matrix T_refkf_robot_, T_world_robot_, last_input_;
    T overlap_threshold_ = T(0.8);
    T minimal_overlap_ = T(0.5);                     // Localizer.hpp:28
    int rebuilds_ = 0;
    double phase_s_[3] = {0, 0, 0};
    double sub_s_[4] = {0, 0, 0, 0};
    struct SubTimer {
        double &acc; std::chrono::steady_clock::time_point t0;
        explicit SubTimer(double &a) : acc(a), t0(std::chrono::steady_clock::now()) {}
        ~SubTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    };
};

template <typename T>
size_t MapManager<T>::AddNewKeyframe(size_t from, const Matrix &T_world_newkf, const Matrix &meas_T_from_newkf, const Matrix &meas_cov, DPPtr cloud)
{
    if (from >= graph_.NumVertices()) throw std::logic_error("MapManager<T>::AddNewKeyframe(): Vertex 'from' must exist in the graph");
    const size_t v = graph_.AddVertex(MakeKeyframe(cloud, T_world_newkf));
    Constraint c;
    c.type = Constraint::kOdomConstraint; c.T_from_to = meas_T_from_newkf; c.cov_from_to = meas_cov; c.weight = (T)PoseWeight(meas_T_from_newkf);
    graph_.AddEdge(from, v, c);
    if (auto lc = loop_closer_.lock()) lc->AddNewVertex(v);                  // MapManager.hpp:107-111
    return v;
}
template <typename T>
void MapManager<T>::NotifyKeyframeUpdate()
{
    if (auto l = localizer_.lock()) l->UpdateFromGraph();
}

// ------------------------------------------------------------------ facade (PoseGraphSlam.h:17-68, single-thread flavour)

//! PoseGraphSlam.h:18-68: the base of both flavours, parametrised -- as upstream -- by the four worker class templates, so that
//! code that derives from it or names its workers (`typename Slam::Localizer`, `localizer_ptr_`) compiles unchanged.
template <typename T, template <typename> class MapManagerClass, template <typename> class LocalizerClass,
          template <typename> class LoopCloserClass, template <typename> class OptimizerClass>
class PoseGraphSlamBase {
public:
    using MapManager = MapManagerClass<T>;
    using MapManagerPtr = typename MapManager::Ptr;
    using Localizer = LocalizerClass<T>;
    using LocalizerPtr = typename Localizer::Ptr;
    using LoopCloser = LoopCloserClass<T>;
    using LoopCloserPtr = typename LoopCloser::Ptr;
    using Optimizer = OptimizerClass<T>;
    using OptimizerPtr = typename Optimizer::Ptr;
    IMPORT_PGSLAM_TYPES(T)

    PoseGraphSlamBase()
        : map_manager_ptr_(std::make_shared<MapManager>()), optimizer_ptr_(std::make_shared<Optimizer>(map_manager_ptr_)),
          loop_closer_ptr_(std::make_shared<LoopCloser>(map_manager_ptr_, optimizer_ptr_)),
          localizer_ptr_(std::make_shared<Localizer>(map_manager_ptr_))
    {
        map_manager_ptr_->SetLocalizer(localizer_ptr_);
        map_manager_ptr_->SetLoopCloser(loop_closer_ptr_);
    }
    //! PoseGraphSlam.hpp:29-36: construct and configure from the three YAML files
    PoseGraphSlamBase(const std::string &localizer_input_filters_config, const std::string &localizer_icp_config,
                      const std::string &loop_closer_icp_config)
        : PoseGraphSlamBase()
    {
        SetIcpConfig(localizer_input_filters_config, localizer_icp_config, loop_closer_icp_config);
    }
    virtual ~PoseGraphSlamBase() {}
    //! PoseGraphSlam.hpp:38-48: three file paths, handed to the workers' own setters
    void SetIcpConfig(const std::string &input_filters_path, const std::string &localizer_icp_path, const std::string &loop_closer_icp_path)
    {
        localizer_ptr_->SetInputFiltersConfig(input_filters_path);
        localizer_ptr_->SetIcpConfig(localizer_icp_path);
        loop_closer_ptr_->SetIcpConfig(loop_closer_icp_path);
    }
    //! (an addition: the YAML texts themselves)
    void SetIcpConfigFromStrings(const std::string &input_filters_yaml, const std::string &localizer_icp_yaml, const std::string &loop_closer_icp_yaml)
    {
        localizer_ptr_->SetInputFiltersConfigFromString(input_filters_yaml);
        localizer_ptr_->SetIcpConfigFromString(localizer_icp_yaml);
        loop_closer_ptr_->SetIcpConfigFromString(loop_closer_icp_yaml);
    }
    void AddData(unsigned long long timestamp, std::string world_frame_id, Matrix T_world_robot, Matrix T_robot_sensor, DPPtr cloud_ptr)
    {
        localizer_ptr_->AddNewData(timestamp, world_frame_id, T_world_robot, T_robot_sensor, cloud_ptr);
    }
    void WriteGraphviz(const std::string &path) { auto lock = map_manager_ptr_->GetGraphLock(); map_manager_ptr_->WriteGraphviz(path); }
    MapManager &map_manager() { return *map_manager_ptr_; }
    Localizer &localizer() { return *localizer_ptr_; }
    LoopCloser &loop_closer() { return *loop_closer_ptr_; }
    Optimizer &optimizer() { return *optimizer_ptr_; }

protected:
    MapManagerPtr map_manager_ptr_;
    OptimizerPtr optimizer_ptr_;
    LoopCloserPtr loop_closer_ptr_;
    LocalizerPtr localizer_ptr_;
};

//! PoseGraphSlam.h:63-66: the single-thread flavour is an alias of the base
template <typename T>
using PoseGraphSlam = PoseGraphSlamBase<T, MapManager, Localizer, LoopCloser, Optimizer>;

// ------------------------------------------------------------------ multi-thread flavour (PoseGraphSlamMT.hpp:21-26)
// Three workers with input queues, as in the reference: the localizer (LocalizerMT.hpp:27-99), the loop closer
// (LoopCloserMT.hpp:26-67) and the optimiser (OptimizerMT.hpp:26-68), sharing the graph under MapManager's lock.  Two ICP
// objects -- two device contexts, two HIP streams -- run concurrently, outside the lock, as upstream's two ICPs do.
// What differs, on purpose: the loop closer DRAINS its queue and runs all waiting candidates as ONE device batch
// (pgslam::LoopClosureBatch; upstream pops one vertex at a time, LoopCloserMT.hpp:49-62), and the optimiser -- as
// upstream -- drains its queue into one solve.  WaitIdle() is an addition for deterministic hosts (tests, benchmarks).

template <typename T>
class OptimizerMT : public Optimizer<T> {
public:
    IMPORT_PGSLAM_TYPES(T)
    using Base = Optimizer<T>;
    using Ptr = std::shared_ptr<OptimizerMT<T>>;
    explicit OptimizerMT(typename MapManager<T>::Ptr mm) : Base(mm) {}
    ~OptimizerMT() override { Stop(); }
    void AddNewData(size_t from, size_t to, const Matrix &T_from_to, const Matrix &cov) override
    {
        { std::lock_guard<std::mutex> l(m_); queue_.push_back(std::make_tuple(from, to, T_from_to, cov)); }
        cv_.notify_one();
    }
    void Run() { stop_ = false; thread_ = std::thread(&OptimizerMT::Main, this); }
    void Stop() { { std::lock_guard<std::mutex> l(m_); stop_ = true; } cv_.notify_all(); if (thread_.joinable()) thread_.join(); }
    bool Idle() { std::lock_guard<std::mutex> l(m_); return (queue_.empty() || paused_) && !busy_; }
    //! (for deterministic hosts: while paused the constraints queue up; Resume() lets one solve take them all)
    void Pause() { std::lock_guard<std::mutex> l(m_); paused_ = true; }
    void Resume() { { std::lock_guard<std::mutex> l(m_); paused_ = false; } cv_.notify_one(); }

private:
    void Main()
    {
        for (;;) {
            std::vector<typename Base::InputData> batch;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return (!queue_.empty() && !paused_) || stop_; });
                if (stop_) break;
                batch.assign(queue_.begin(), queue_.end());          // all of it: one solve (OptimizerMT.hpp:59-65)
                queue_.clear();
                busy_ = true;
            }
            std::exception_ptr err;
            try { this->AddNewDataBatch(batch); } catch (...) { err = std::current_exception(); }
            { std::lock_guard<std::mutex> l(m_); busy_ = false; if (err && !error_) error_ = err; }
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<typename Base::InputData> queue_;
    std::thread thread_;
    bool stop_ = false, busy_ = false, paused_ = false;
    std::exception_ptr error_;
public:
    std::exception_ptr TakeError() { std::lock_guard<std::mutex> l(m_); std::exception_ptr e = error_; error_ = nullptr; return e; }
};

template <typename T>
class LoopCloserMT : public LoopCloser<T> {
public:
    IMPORT_PGSLAM_TYPES(T)
    using Base = LoopCloser<T>;
    using Ptr = std::shared_ptr<LoopCloserMT<T>>;
    LoopCloserMT(typename MapManager<T>::Ptr mm, typename Optimizer<T>::Ptr opt) : Base(mm, opt), optimizer_(opt) {}
    ~LoopCloserMT() override { Stop(); }
    //! (the batch dispatcher on the worker thread owns the ICP object; the base class's one-at-a-time closer is not made)
    void SetIcpConfigFromString(const std::string &yaml) override { yaml_ = yaml; }
    void AddNewVertex(size_t v) override
    {
        { std::lock_guard<std::mutex> l(m_); queue_.push_back(v); }
        cv_.notify_one();
    }
    void Run() { stop_ = false; thread_ = std::thread(&LoopCloserMT::Main, this); }
    void Stop() { { std::lock_guard<std::mutex> l(m_); stop_ = true; } cv_.notify_all(); if (thread_.joinable()) thread_.join(); }
    bool Idle() { std::lock_guard<std::mutex> l(m_); return (queue_.empty() || paused_) && !busy_; }
    int batches() const { return batches_; }
    int largest_batch() const { return largest_batch_; }
    //! While paused the worker leaves its queue alone (vertices keep arriving); Resume() lets it drain what has piled up as ONE
    //! device batch.  SetMaxBatch(1) restores upstream's one vertex at a time (LoopCloserMT.hpp:49-62).  Both are for hosts
    //! that want a deterministic batch (tests, benchmarks): the dispatcher's results do not depend on how the queue is cut.
    void Pause() { std::lock_guard<std::mutex> l(m_); paused_ = true; }
    void Resume() { { std::lock_guard<std::mutex> l(m_); paused_ = false; } cv_.notify_one(); }
    void SetMaxBatch(size_t n) { std::lock_guard<std::mutex> l(m_); max_batch_ = n < 1 ? 1 : n; }
    size_t queued() { std::lock_guard<std::mutex> l(m_); return queue_.size(); }
    //! every edge the dispatcher has produced so far, accepted or not, in the order it was produced
    std::vector<pgicp_edge> edges() { std::lock_guard<std::mutex> l(m_); return edges_; }
    size_t device_batches() const { return device_batches_; }

private:
    void Main()
    {
        LoopClosureBatch<T> batch;                               // its ICP object (device context) lives on this thread
        bool configured = false;
        for (;;) {
            std::vector<size_t> vs;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return (!queue_.empty() && !paused_) || stop_; });
                if (stop_) break;
                const size_t take = std::min(queue_.size(), max_batch_);
                vs.assign(queue_.begin(), queue_.begin() + (long)take);     // every waiting vertex (up to max_batch_): one device batch
                queue_.erase(queue_.begin(), queue_.begin() + (long)take);
                busy_ = true;
            }
            std::exception_ptr err;
            try {
            if (!configured) { batch.SetIcpConfigFromString(yaml_); configured = true; }
            std::vector<typename Base::PreparedCandidate> cands;
            pgicp_ctx *dev_ctx = batch.DeviceCandidateEquivalent() ? batch.Context() : nullptr;
            for (size_t v : vs) {
                typename Base::PreparedCandidate c;
                if (this->PrepareCandidate(v, c, dev_ctx)) cands.push_back(c);
            }
            // a reading with `simpleSensorNoise` takes getOverlap()'s sensor-noise branch, which reads the ICP's LAST error elements:
            // the batch's fused residual pass replaces them, so such candidates go one at a time (the base class's ProcessCandidate)
            bool pairwise = false;
            for (auto &c : cands) pairwise = pairwise || (c.reading && c.reading->descriptorExists("simpleSensorNoise"));
            if (pairwise) {
                if (!pairwise_configured_) { this->closer().SetIcpConfigFromString(yaml_); pairwise_configured_ = true; }
                PairLoopCloser<T> &lc = this->closer();
                for (auto &c : cands) {
                    auto r = c.reference_dev && lc.DeviceCandidateEquivalent()
                                 ? lc.ProcessCandidateOnDevice(c.reading, *c.reading_dev, *c.reference_dev, c.guess, c.host_reference)
                                 : lc.ProcessCandidate(*c.reading, c.reference ? *c.reference : c.host_reference(), c.guess);
                    pgicp_edge e;
                    std::memset(&e, 0, sizeof e);
                    e.from_id = (long long)c.ref_v; e.to_id = (long long)c.input_v; e.max_iter_reached = r.max_iterations_reached ? 1 : 0;
                    e.overlap = (double)r.overlap; e.residual = (double)r.residual; e.accepted = r.accepted ? 1 : 0;
                    pgslam_amd::to_row_major16(r.T_refkf_kf, e.T_from_to);
                    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) e.cov[6 * i + j] = (double)r.cov(i, j);
                    { std::lock_guard<std::mutex> l(m_); edges_.push_back(e); }
                    if (r.accepted) { this->loops_closed_++; optimizer_->AddNewData(c.ref_v, c.input_v, r.T_refkf_kf, r.cov); }
                }
                largest_batch_ = std::max(largest_batch_, 1);
                batches_++;
                cands.clear();
            }
            if (!cands.empty()) {
                batch.Clear();
                for (auto &c : cands) {
                    typename LoopClosureBatch<T>::Candidate bc;
                    bc.from_id = (long long)c.ref_v; bc.to_id = (long long)c.input_v; bc.reading = c.reading; bc.reference = c.reference; bc.T_init = c.guess;
                    bc.reading_dev = c.reading_dev; bc.reference_dev = c.reference_dev;
                    batch.Add(bc);
                }
                const auto edges = batch.Run(batch.Shard(1, 0), this->overlap_threshold_, this->residual_error_threshold_);
                device_batches_ = batch.device_batches();
                { std::lock_guard<std::mutex> l(m_); edges_.insert(edges_.end(), edges.begin(), edges.end()); }
                batches_++;
                largest_batch_ = std::max(largest_batch_, (int)cands.size());
                for (const pgicp_edge &e : edges)
                    if (e.accepted) {
                        this->loops_closed_++;
                        Matrix cov(6, 6);
                        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) cov(i, j) = (T)e.cov[6 * i + j];
                        optimizer_->AddNewData((size_t)e.from_id, (size_t)e.to_id, pgslam_amd::from_row_major16<T>(e.T_from_to), cov);
                    }
            }
            } catch (...) { err = std::current_exception(); }
            { std::lock_guard<std::mutex> l(m_); busy_ = false; if (err && !error_) error_ = err; }
        }
    }
    typename Optimizer<T>::Ptr optimizer_;
    std::string yaml_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<size_t> queue_;
    std::thread thread_;
    bool stop_ = false, busy_ = false, paused_ = false, pairwise_configured_ = false;
    size_t max_batch_ = (size_t)1 << 30;
    int batches_ = 0, largest_batch_ = 0;
    size_t device_batches_ = 0;
    std::vector<pgicp_edge> edges_;
    std::exception_ptr error_;
public:
    std::exception_ptr TakeError() { std::lock_guard<std::mutex> l(m_); std::exception_ptr e = error_; error_ = nullptr; return e; }
};

template <typename T>
class LocalizerMT : public Localizer<T> {
public:
    IMPORT_PGSLAM_TYPES(T)
    using Base = Localizer<T>;
    using Ptr = std::shared_ptr<LocalizerMT<T>>;
    explicit LocalizerMT(typename MapManager<T>::Ptr mm, size_t capacity = 3) : Base(mm, capacity) { this->resync_before_update_ = true; }
    ~LocalizerMT() override { Stop(); }
    void AddNewData(unsigned long long, const std::string &, const Matrix &T_world_robot, const Matrix &T_robot_sensor, DPPtr cloud) override
    {
        auto it = std::make_shared<Item>();
        it->T_world_robot = T_world_robot; it->T_robot_sensor = T_robot_sensor; it->cloud = cloud;
        { std::lock_guard<std::mutex> l(m_); queue_.push_back(it); }
        cv_.notify_all();
    }
    //! MapManager::NotifyKeyframeUpdate lands here from the optimiser's thread: only the flag is set, the update itself
    //! happens on the localizer's own thread (LocalizerMT.hpp:123-136)
    void UpdateFromGraph() override
    {
        { std::lock_guard<std::mutex> l(m_); outdated_ = true; }
        cv_.notify_all();
    }
    //! The input stage on a thread (and context) of its own, up to two scans ahead of the ICP: ON by default -- measured warm
    //! (slam_run --mt --passes 4, 600 scans of 100 k points): 860-919 scans/s with it, 798-806 without, the localizer's thread
    //! waiting 0.01 ms per scan for the hand-off.  (Round 5 first measured the opposite from single cold passes, where the
    //! second context's first allocations and code-object loads land inside the timed run.)
    //! PGSLAM_MT_INPUT_THREAD=0 (or SetInputThread(false) before Run) turns it off.
    void SetInputThread(bool on) { pre_thread_on_ = on; }
    void Run()
    {
        stop_ = false;
        thread_ = std::thread(&LocalizerMT::Main, this);
        if (pre_thread_on_) pre_thread_ = std::thread(&LocalizerMT::PreMain, this);
    }
    void Stop()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        if (thread_.joinable()) thread_.join();
        if (pre_thread_.joinable()) pre_thread_.join();
    }
    bool Idle() { std::lock_guard<std::mutex> l(m_); return queue_.empty() && !busy_ && !outdated_; }
    size_t processed() { std::lock_guard<std::mutex> l(m_); return processed_; }
    //! seconds the localizer's thread has waited for a scan's input stage to finish on the pre-processing thread
    double waited_for_input_stage() { std::lock_guard<std::mutex> l(m_); return wait_pre_s_; }
    //! seconds the pre-processing thread has spent inside input stages
    double input_stage_seconds() { std::lock_guard<std::mutex> l(m_); return pre_busy_s_; }

private:
    // A queued scan.  Its input stage -- the input filters and the sensor->robot transform, Localizer.hpp:103-106, on the
    // device: upload, one pass, the filtered cloud back -- runs on the PRE-PROCESSING thread, on that thread's own context, while
    // the localizer's thread aligns the scan before it: upstream's queue (LocalizerMT.hpp:27-40) already holds the scan by then.
    // At most two scans ahead (their device copies live in the context's ring of result sets).  All pre-processing of this
    // flavour happens there, in arrival order: the filters' own state (a FixStep filter's step) sees the scans in order.
    struct Item {
        Matrix T_world_robot, T_robot_sensor;
        DPPtr cloud;
        int state = 0;                                              // 0 queued, 1 being pre-processed, 2 pre-processed
        typename PM::ICPChainBase::DeviceReading reading;
        std::exception_ptr err;
    };
    void PreMain()
    {
        for (;;) {
            std::shared_ptr<Item> it;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] {
                    if (stop_) return true;
                    // the scan being aligned is no longer in the queue: entries 0 and 1 are the two after it
                    for (size_t i = 0; i < queue_.size() && i < 2; i++) if (queue_[i]->state == 0) { it = queue_[i]; return true; }
                    return false;
                });
                if (stop_) break;
                it->state = 1;
            }
            const auto tp = std::chrono::steady_clock::now();
            try { it->reading = this->PreProcessOn(pre_ctx_, it->T_robot_sensor, it->cloud); } catch (...) { it->err = std::current_exception(); }
            { std::lock_guard<std::mutex> l(m_); it->state = 2; pre_busy_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count(); }
            cv_.notify_all();
        }
    }
    void Main()
    {
        for (;;) {
            bool outdated = false;
            std::shared_ptr<Item> it;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return !queue_.empty() || stop_ || outdated_; });
                if (stop_) break;
                outdated = outdated_;
                outdated_ = false;
                if (!queue_.empty()) {
                    if (pre_thread_on_) {
                        const auto tw = std::chrono::steady_clock::now();
                        cv_.wait(l, [this] { return queue_.front()->state == 2 || stop_; });      // (its input stage is done, or nearly)
                        wait_pre_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
                        if (stop_) break;
                    }
                    it = queue_.front();
                    queue_.pop_front();
                }
                busy_ = true;
            }
            cv_.notify_all();                                          // (the pre-processing thread may look one scan further)
            std::exception_ptr err;
            try {
                if (outdated) this->UpdateFromGraphNow();             // (takes the graph lock)
                if (it) {
                    if (it->err) std::rethrow_exception(it->err);
                    if (pre_thread_on_) this->AdoptPreprocessed(it->cloud, it->reading);
                    this->ProcessData(it->T_world_robot, it->T_robot_sensor, it->cloud);     // (without the input thread: the stage runs here, its
                                                                                             // host compaction on the worker while the ICP runs)
                }
            } catch (...) { err = std::current_exception(); }
            { std::lock_guard<std::mutex> l(m_); busy_ = false; if (it) processed_++; if (err && !error_) error_ = err; }
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Item>> queue_;
    std::thread thread_, pre_thread_;
    pgslam_amd::LazyContext pre_ctx_{0, std::getenv("PGSLAM_PRE_STAGE_NORMAL_PRIORITY") ? 0 : 1};     // (its short launches go ahead of the ICP's queued ones)
    bool stop_ = false, busy_ = false, outdated_ = false;
    bool pre_thread_on_ = !(std::getenv("PGSLAM_MT_INPUT_THREAD") && std::getenv("PGSLAM_MT_INPUT_THREAD")[0] == '0');
    size_t processed_ = 0;
    double wait_pre_s_ = 0, pre_busy_s_ = 0;
    std::exception_ptr error_;
public:
    std::exception_ptr TakeError() { std::lock_guard<std::mutex> l(m_); std::exception_ptr e = error_; error_ = nullptr; return e; }
};

//! MapManagerMT.h: upstream's lock-carrying map manager.  MapManager<T> here carries the lock in both flavours (it is
//! uncontended in the single-thread one), so the MT class only has to exist under its name.
template <typename T>
class MapManagerMT : public MapManager<T> {
public:
    using Ptr = std::shared_ptr<MapManagerMT<T>>;
};

//! PoseGraphSlamMT.h:17-30: derives from the base with the four MT workers and adds Run()
template <typename T>
class PoseGraphSlamMT : public PoseGraphSlamBase<T, MapManagerMT, LocalizerMT, LoopCloserMT, OptimizerMT> {
public:
    using Base = PoseGraphSlamBase<T, MapManagerMT, LocalizerMT, LoopCloserMT, OptimizerMT>;
    IMPORT_PGSLAM_TYPES(T)
    PoseGraphSlamMT() : Base() {}
    //! PoseGraphSlamMT.h:23-27
    PoseGraphSlamMT(const std::string &localizer_input_filters_config, const std::string &localizer_icp_config,
                    const std::string &loop_closer_icp_config)
        : Base(localizer_input_filters_config, localizer_icp_config, loop_closer_icp_config) {}
    ~PoseGraphSlamMT() override { this->localizer_ptr_->Stop(); this->loop_closer_ptr_->Stop(); this->optimizer_ptr_->Stop(); }
    //! PoseGraphSlamMT::Run (PoseGraphSlamMT.hpp:21-26)
    void Run() { this->localizer_ptr_->Run(); this->loop_closer_ptr_->Run(); this->optimizer_ptr_->Run(); }
    //! until every queue is empty and every worker rests (the stages feed one another: checked until stable)
    void WaitIdle()
    {
        for (int calm = 0; calm < 3;) {
            if (this->localizer_ptr_->Idle() && this->loop_closer_ptr_->Idle() && this->optimizer_ptr_->Idle()) calm++;
            else calm = 0;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        RethrowWorkerError();
    }
    //! An exception inside a worker (upstream lets it reach std::terminate) is kept; the worker goes on with its next item,
    //! and the first such exception is thrown here, on the caller's thread.
    void RethrowWorkerError()
    {
        for (std::exception_ptr e : {this->localizer_ptr_->TakeError(), this->loop_closer_ptr_->TakeError(), this->optimizer_ptr_->TakeError()})
            if (e) std::rethrow_exception(e);
    }
};

}  // namespace pgslam
