// icp_math.hpp -- small double-precision SE(3) / 6x6 routines shared by the
// device-side solve kernel and the host side of libpgicp.  Product code: it is
// written independently of oracle/icp_oracle.c, which restates the same
// published algorithms for checking.
#pragma once
#include <cmath>
#include <cfloat>

#if defined(__HIPCC__)
#define PGICP_HD __host__ __device__ inline
#else
#define PGICP_HD inline
#endif

namespace pgicp {

constexpr int kSys = 30;          // 21 (upper A) + 6 (b) + sum_w + kept + residual
constexpr int kHist = 16;         // Differential checker history capacity (smoothLength <= 15)

PGICP_HD void mat4_identity(double *a)
{
    for (int i = 0; i < 16; i++) a[i] = 0.0;
    a[0] = a[5] = a[10] = a[15] = 1.0;
}

PGICP_HD void mat4_mul(const double *a, const double *b, double *c)
{
    double t[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += a[i * 4 + k] * b[k * 4 + j];
            t[i * 4 + j] = s;
        }
    for (int i = 0; i < 16; i++) c[i] = t[i];
}

PGICP_HD void mat4_rigid_inverse(const double *t, double *o)
{
    double r[16];
    mat4_identity(r);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r[i * 4 + j] = t[j * 4 + i];
    for (int i = 0; i < 3; i++) r[i * 4 + 3] = -(r[i * 4 + 0] * t[3] + r[i * 4 + 1] * t[7] + r[i * 4 + 2] * t[11]);
    for (int i = 0; i < 16; i++) o[i] = r[i];
}

// Expand the packed upper triangle (row-major 00 01 .. 05 11 ..) to a full 6x6.
PGICP_HD void sys_to_full(const double *sys, double *A)
{
    int k = 0;
    for (int a = 0; a < 6; a++)
        for (int b = a; b < 6; b++) {
            A[a * 6 + b] = sys[k];
            A[b * 6 + a] = sys[k];
            k++;
        }
}

// LL^T with a relative pivot test.  Returns false when a pivot is not safely
// positive (numerically rank deficient => caller takes the minimal-norm path).
PGICP_HD bool chol_solve6(const double *A, const double *b, double rel_tol, double *x)
{
    double L[36];
    for (int i = 0; i < 36; i++) L[i] = 0.0;
    double dmax = 0.0;
    for (int i = 0; i < 6; i++) dmax = fmax(dmax, fabs(A[i * 6 + i]));
    const double tol = dmax * rel_tol;
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
        for (int m = 0; m < j; m++) d -= L[j * 6 + m] * L[j * 6 + m];
        if (!(d > tol)) return false;
        const double ljj = sqrt(d);
        L[j * 6 + j] = ljj;
        for (int i = j + 1; i < 6; i++) {
            double s = A[i * 6 + j];
            for (int m = 0; m < j; m++) s -= L[i * 6 + m] * L[j * 6 + m];
            L[i * 6 + j] = s / ljj;
        }
    }
    double y[6];
    for (int i = 0; i < 6; i++) {
        double s = b[i];
        for (int m = 0; m < i; m++) s -= L[i * 6 + m] * y[m];
        y[i] = s / L[i * 6 + i];
    }
    for (int i = 5; i >= 0; i--) {
        double s = y[i];
        for (int m = i + 1; m < 6; m++) s -= L[m * 6 + i] * x[m];
        x[i] = s / L[i * 6 + i];
    }
    return true;
}

// Minimal-norm solution of the symmetric PSD system through a cyclic Jacobi
// eigen-decomposition: x = sum_{lambda_e > tol} v_e (v_e.b)/lambda_e.
PGICP_HD int min_norm_solve6(const double *Ain, const double *b, double rel_tol, double *x)
{
    double a[36], v[36];
    for (int i = 0; i < 36; i++) { a[i] = Ain[i]; v[i] = 0.0; }
    for (int i = 0; i < 6; i++) v[i * 6 + i] = 1.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int i = 0; i < 6; i++)
            for (int j = i + 1; j < 6; j++) off += a[i * 6 + j] * a[i * 6 + j];
        if (off == 0.0) break;
        for (int p = 0; p < 5; p++)
            for (int q = p + 1; q < 6; q++) {
                const double apq = a[p * 6 + q];
                if (apq == 0.0) continue;
                const double theta = (a[q * 6 + q] - a[p * 6 + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 6; k++) {
                    const double akp = a[k * 6 + p], akq = a[k * 6 + q];
                    a[k * 6 + p] = c * akp - s * akq;
                    a[k * 6 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; k++) {
                    const double apk = a[p * 6 + k], aqk = a[q * 6 + k];
                    a[p * 6 + k] = c * apk - s * aqk;
                    a[q * 6 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 6; k++) {
                    const double vkp = v[k * 6 + p], vkq = v[k * 6 + q];
                    v[k * 6 + p] = c * vkp - s * vkq;
                    v[k * 6 + q] = s * vkp + c * vkq;
                }
            }
    }
    double emax = 0.0;
    for (int i = 0; i < 6; i++) emax = fmax(emax, fabs(a[i * 6 + i]));
    const double tol = emax * rel_tol;
    int rank = 0;
    for (int i = 0; i < 6; i++) x[i] = 0.0;
    for (int e = 0; e < 6; e++) {
        const double lam = a[e * 6 + e];
        if (!(lam > tol)) continue;
        rank++;
        double dot = 0.0;
        for (int i = 0; i < 6; i++) dot += v[i * 6 + e] * b[i];
        const double c = dot / lam;
        for (int i = 0; i < 6; i++) x[i] += c * v[i * 6 + e];
    }
    return rank;
}

// PointToPlaneErrorMinimizer{force4DOF} ([EXT] ErrorMinimizers/PointToPlane.cpp: F's first row is the z component of
// p x n alone, the system is 4 x 4 in [rz tx ty tz], the increment AngleAxis(x0, unitZ) + translation): the 4 x 4 system is
// the [2..5] block of the 6 x 6 one, so the 6 x 6 is made block diagonal -- rows and columns of rx, ry cleared, their
// diagonal set to the block's largest diagonal entry (the pivot test's scale stays the block's), b0 = b1 = 0 -- and solved
// as usual: x0 = x1 = 0 exactly, the rest is the 4 x 4 solve.
PGICP_HD void constrain_4dof(double *sys)
{
    // packed upper triangle, row by row: (0,0..5) = 0..5, (1,1..5) = 6..10, (2,2..5) = 11..14, (3,3..5) = 15..17, (4,4..5) = 18..19, (5,5) = 20
    const double dmax = fmax(fmax(fabs(sys[11]), fabs(sys[15])), fmax(fabs(sys[18]), fabs(sys[20])));
    for (int k = 0; k <= 10; k++) sys[k] = 0.0;
    sys[0] = dmax > 0.0 ? dmax : 1.0;
    sys[6] = sys[0];
    sys[21] = 0.0;
    sys[22] = 0.0;
}

// solvePossiblyUnderdeterminedLinearSystem.
PGICP_HD int solve6(const double *sys, double rel_tol, double *x)
{
    double A[36];
    sys_to_full(sys, A);
    if (chol_solve6(A, sys + 21, rel_tol, x)) return 6;
    return min_norm_solve6(A, sys + 21, rel_tol, x);
}

// x = [rx ry rz tx ty tz] -> 4x4 (AngleAxis(|r|, r/|r|)); degenerate => R = I.
PGICP_HD void delta_T(const double *x, double *T)
{
    mat4_identity(T);
    const double th = sqrt((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]);
    if (th > 0.0 && th < HUGE_VAL) {
        const double ux = x[0] / th, uy = x[1] / th, uz = x[2] / th;
        const double c = cos(th), s = sin(th), C = 1.0 - c;
        T[0] = c + ux * ux * C;      T[1] = ux * uy * C - uz * s; T[2] = ux * uz * C + uy * s;
        T[4] = uy * ux * C + uz * s; T[5] = c + uy * uy * C;      T[6] = uy * uz * C - ux * s;
        T[8] = uz * ux * C - uy * s; T[9] = uz * uy * C + ux * s; T[10] = c + uz * uz * C;
    }
    T[3] = x[3]; T[7] = x[4]; T[11] = x[5];
}

PGICP_HD void rot_to_quat(const double *T, double *q)
{
    const double m00 = T[0], m01 = T[1], m02 = T[2], m10 = T[4], m11 = T[5], m12 = T[6], m20 = T[8],
                 m21 = T[9], m22 = T[10];
    const double tr = m00 + m11 + m22;
    if (tr > 0.0) {
        const double s = sqrt(tr + 1.0) * 2.0;
        q[0] = 0.25 * s; q[1] = (m21 - m12) / s; q[2] = (m02 - m20) / s; q[3] = (m10 - m01) / s;
    } else if (m00 > m11 && m00 > m22) {
        const double s = sqrt(1.0 + m00 - m11 - m22) * 2.0;
        q[0] = (m21 - m12) / s; q[1] = 0.25 * s; q[2] = (m01 + m10) / s; q[3] = (m02 + m20) / s;
    } else if (m11 > m22) {
        const double s = sqrt(1.0 + m11 - m00 - m22) * 2.0;
        q[0] = (m02 - m20) / s; q[1] = (m01 + m10) / s; q[2] = 0.25 * s; q[3] = (m12 + m21) / s;
    } else {
        const double s = sqrt(1.0 + m22 - m00 - m11) * 2.0;
        q[0] = (m10 - m01) / s; q[1] = (m02 + m20) / s; q[2] = (m12 + m21) / s; q[3] = 0.25 * s;
    }
    const double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= nn;
}

PGICP_HD double quat_angular_distance(const double *a, const double *b)
{
    const double w = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    const double x = -a[0] * b[1] + a[1] * b[0] - a[2] * b[3] + a[3] * b[2];
    const double y = -a[0] * b[2] + a[1] * b[3] + a[2] * b[0] - a[3] * b[1];
    const double z = -a[0] * b[3] - a[1] * b[2] + a[2] * b[1] + a[3] * b[0];
    return 2.0 * atan2(sqrt(x * x + y * y + z * z), fabs(w));
}

// Counter + Differential transformation checkers.
struct Checker {
    int count;
    int n_hist;
    double quat[kHist][4];
    double trans[kHist][3];
};

PGICP_HD void checker_init(Checker &c)
{
    c.count = 0;
    c.n_hist = 1;
    for (int i = 0; i < kHist; i++) {
        for (int j = 0; j < 4; j++) c.quat[i][j] = 0.0;
        for (int j = 0; j < 3; j++) c.trans[i][j] = 0.0;
    }
    c.quat[0][0] = 1.0;   // checkers.init(T_iter = I)
}

// bit0 keep iterating, bit1 differential stop, bit2 counter stop, bit3 NaN, bit4 BoundTransformationChecker's limit exceeded
// (bound_rot / bound_trans <= 0: that checker is not in the chain).  The checkers run in the order Counter, Differential,
// Bound; the Counter's max-iterations condition leaves the check before the Bound is looked at.
PGICP_HD int checker_check(Checker &c, const double *T, int max_iters, double min_rot, double min_trans, int smooth,
                           double bound_rot = 0.0, double bound_trans = 0.0)
{
    int iterate = 1, flags = 0;
    c.count++;
    if (c.count >= max_iters) { iterate = 0; flags |= 4; }
    if (c.n_hist == kHist) {
        for (int i = 1; i < kHist; i++) {
            for (int j = 0; j < 4; j++) c.quat[i - 1][j] = c.quat[i][j];
            for (int j = 0; j < 3; j++) c.trans[i - 1][j] = c.trans[i][j];
        }
        c.n_hist--;
    }
    rot_to_quat(T, c.quat[c.n_hist]);
    c.trans[c.n_hist][0] = T[3]; c.trans[c.n_hist][1] = T[7]; c.trans[c.n_hist][2] = T[11];
    c.n_hist++;
    if (c.n_hist > smooth) {
        double rsum = 0.0, tsum = 0.0;
        for (int i = c.n_hist - 1; i >= c.n_hist - smooth; i--) {
            rsum += fabs(quat_angular_distance(c.quat[i], c.quat[i - 1]));
            const double dx = c.trans[i][0] - c.trans[i - 1][0], dy = c.trans[i][1] - c.trans[i - 1][1],
                         dz = c.trans[i][2] - c.trans[i - 1][2];
            tsum += sqrt(dx * dx + dy * dy + dz * dz);
        }
        rsum /= (double)smooth;
        tsum /= (double)smooth;
        if (rsum != rsum || tsum != tsum) return 8;
        if (rsum < min_rot && tsum < min_trans) { iterate = 0; flags |= 2; }
    }
    if (!(flags & 4) && (bound_rot > 0.0 || bound_trans > 0.0)) {
        // the checkers were initialised with T_iter = I: the bound is on the accumulated correction itself
        const double ident[4] = {1.0, 0.0, 0.0, 0.0};
        const double *q = c.quat[c.n_hist - 1], *t = c.trans[c.n_hist - 1];
        const double rot = quat_angular_distance(q, ident), tr = sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
        if ((bound_rot > 0.0 && rot > bound_rot) || (bound_trans > 0.0 && tr > bound_trans)) return 16;
    }
    return flags | iterate;
}

// ---- PointToPointErrorMinimizer: weighted Kabsch ------------------------------------------------------------------
// cyclic Jacobi on a symmetric 3x3 (columns of v: eigenvectors; a's diagonal: eigenvalues)
PGICP_HD void jacobi3_sym(double a[3][3], double v[3][3])
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 16; sweep++) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off == 0.0) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int r = 0; r < 3; r++) { const double arp = a[r][p], arq = a[r][q]; a[r][p] = c * arp - s * arq; a[r][q] = s * arp + c * arq; }
                for (int r = 0; r < 3; r++) { const double apr = a[p][r], aqr = a[q][r]; a[p][r] = c * apr - s * aqr; a[q][r] = s * apr + c * aqr; }
                for (int r = 0; r < 3; r++) { const double vrp = v[r][p], vrq = v[r][q]; v[r][p] = c * vrp - s * vrq; v[r][q] = s * vrp + c * vrq; }
            }
    }
}

// sums of the kept pairs (sys[0..2] sum w p, [3..5] sum w q, [6..14] sum w q_a p_b, [27] sum w) -> the increment: means,
// M = sum w (q - mq)(p - mp)^T, SVD through the eigen-decomposition of M^T M, R = U V^T (third singular pair negated when
// that is a reflection), t = mq - R mp.  Returns the rank of M (below 2 no rotation is determined: identity rotation).
PGICP_HD int solve_p2point(const double *sys, double *T)
{
    mat4_identity(T);
    const double sw = sys[27];
    if (!(sw > 0.0)) return 0;
    double mp[3], mq[3], M[3][3], B[3][3], W[3][3];
    for (int a = 0; a < 3; a++) { mp[a] = sys[a] / sw; mq[a] = sys[3 + a] / sw; }
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) M[a][b] = sys[6 + 3 * a + b] - sw * (mq[a] * mp[b]);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) B[i][j] = (M[0][i] * M[0][j] + M[1][i] * M[1][j]) + M[2][i] * M[2][j];
    jacobi3_sym(B, W);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; a++)
        for (int b = a + 1; b < 3; b++)
            if (B[ord[b]][ord[b]] > B[ord[a]][ord[a]]) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double S[3], U[3][3], V[3][3];
    for (int c = 0; c < 3; c++) {
        const double ev = B[ord[c]][ord[c]];
        S[c] = ev > 0.0 ? sqrt(ev) : 0.0;
        for (int r = 0; r < 3; r++) { V[r][c] = W[r][ord[c]]; U[r][c] = 0.0; }
    }
    const double tol = 1e-12 * S[0];
    bool have[3] = {false, false, false};
    for (int c = 0; c < 3; c++) {
        if (!(S[c] > tol)) continue;
        double u[3];
        for (int r = 0; r < 3; r++) u[r] = ((M[r][0] * V[0][c] + M[r][1] * V[1][c]) + M[r][2] * V[2][c]) / S[c];
        for (int b = 0; b < c; b++)
            if (have[b]) {
                const double d = (u[0] * U[0][b] + u[1] * U[1][b]) + u[2] * U[2][b];
                for (int r = 0; r < 3; r++) u[r] -= d * U[r][b];
            }
        const double nn = sqrt((u[0] * u[0] + u[1] * u[1]) + u[2] * u[2]);
        if (!(nn > 0.0)) continue;
        for (int r = 0; r < 3; r++) U[r][c] = u[r] / nn;
        have[c] = true;
    }
    if (have[0] && have[1] && !have[2]) {
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
        have[2] = true;
    }
    const int rank = S[0] > 0.0 ? (S[2] > tol ? 3 : (S[1] > tol ? 2 : 1)) : 0;
    double R[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};
    if (have[0] && have[1] && have[2]) {
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) R[a][b] = (U[a][0] * V[b][0] + U[a][1] * V[b][1]) + U[a][2] * V[b][2];
        const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                           R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
        if (det < 0.0)
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) R[a][b] = (U[a][0] * V[b][0] + U[a][1] * V[b][1]) - U[a][2] * V[b][2];
    }
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) T[4 * a + b] = R[a][b];
        T[4 * a + 3] = mq[a] - ((R[a][0] * mp[0] + R[a][1] * mp[1]) + R[a][2] * mp[2]);
    }
    return rank;
}

// inverse of a general 6x6 (Gauss-Jordan, partial pivoting); returns false if singular
PGICP_HD bool inverse6(const double *H, double *Hi)
{
    double M[6][12];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) { M[i][j] = H[i * 6 + j]; M[i][6 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 6; c++) {
        int piv = c;
        for (int r = c + 1; r < 6; r++) if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
        if (piv != c) for (int j = 0; j < 12; j++) { const double t = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = t; }
        const double d = M[c][c];
        if (d == 0.0) return false;
        for (int j = 0; j < 12; j++) M[c][j] /= d;
        for (int r = 0; r < 6; r++) {
            if (r == c) continue;
            const double f = M[r][c];
            for (int j = 0; j < 12; j++) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Hi[i * 6 + j] = M[i][6 + j];
    return true;
}

}  // namespace pgicp
