// matrix.hpp -- the minimal column-major dynamic matrix the drop-in layer needs.
//
// libpointmatcher's PM::Matrix is an Eigen::Matrix<T, Dynamic, Dynamic> (column
// major); neither Eigen nor libpointmatcher exists in this image,
// so this type offers the spelling pgslam actually uses on the hot path:
// Matrix::Identity(4,4), operator*, inverse(), (i,j), rows(), cols(), data()
// (reference src/pgslam/Localizer.hpp:23-25,119-127, LoopCloser.hpp:95,
// LocalMap.hpp:216-222).  It is NOT a linear-algebra library.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <stdexcept>
#include <vector>

namespace pgslam_amd {

template <typename T>
class Mat {
public:
    Mat() : r_(0), c_(0) {}
    Mat(int rows, int cols) : r_(rows), c_(cols), v_((size_t)rows * cols, T(0)) {}
    static Mat Zero(int rows, int cols) { return Mat(rows, cols); }
    static Mat Constant(int rows, int cols, T x) { Mat m(rows, cols); std::fill(m.v_.begin(), m.v_.end(), x); return m; }
    static Mat Identity(int rows, int cols)
    {
        Mat m(rows, cols);
        for (int i = 0; i < std::min(rows, cols); i++) m(i, i) = T(1);
        return m;
    }
    int rows() const { return r_; }
    int cols() const { return c_; }
    size_t size() const { return v_.size(); }
    T &operator()(int i, int j) { return v_[(size_t)j * r_ + i]; }
    const T &operator()(int i, int j) const { return v_[(size_t)j * r_ + i]; }
    T *data() { return v_.data(); }
    const T *data() const { return v_.data(); }
    void resize(int rows, int cols) { r_ = rows; c_ = cols; v_.assign((size_t)rows * cols, T(0)); }
    // keeps the leading block, like Eigen's conservativeResize
    void conservativeResize(int rows, int cols)
    {
        if (rows == r_) {                       // column-major: dropping or adding trailing columns keeps the leading ones in place
            if (cols != c_) { v_.resize((size_t)rows * cols, T(0)); c_ = cols; }
            return;
        }
        Mat m(rows, cols);
        for (int j = 0; j < std::min(cols, c_); j++)
            for (int i = 0; i < std::min(rows, r_); i++) m(i, j) = (*this)(i, j);
        *this = m;
    }
    Mat block(int i0, int j0, int nr, int nc) const
    {
        Mat m(nr, nc);
        for (int j = 0; j < nc; j++)
            for (int i = 0; i < nr; i++) m(i, j) = (*this)(i0 + i, j0 + j);
        return m;
    }
    Mat col(int j) const { return block(0, j, r_, 1); }
    Mat transpose() const
    {
        Mat m(c_, r_);
        for (int j = 0; j < c_; j++)
            for (int i = 0; i < r_; i++) m(j, i) = (*this)(i, j);
        return m;
    }
    Mat operator*(const Mat &o) const
    {
        if (c_ != o.r_) throw std::invalid_argument("Mat: shape mismatch in product");
        Mat m(r_, o.c_);
        for (int j = 0; j < o.c_; j++)
            for (int k = 0; k < c_; k++) {
                const T b = o(k, j);
                for (int i = 0; i < r_; i++) m(i, j) += (*this)(i, k) * b;
            }
        return m;
    }
    Mat operator+(const Mat &o) const { Mat m(*this); for (size_t i = 0; i < v_.size(); i++) m.v_[i] += o.v_[i]; return m; }
    Mat operator-(const Mat &o) const { Mat m(*this); for (size_t i = 0; i < v_.size(); i++) m.v_[i] -= o.v_[i]; return m; }
    Mat operator*(T s) const { Mat m(*this); for (auto &x : m.v_) x *= s; return m; }
    bool operator==(const Mat &o) const { return r_ == o.r_ && c_ == o.c_ && v_ == o.v_; }
    bool operator!=(const Mat &o) const { return !(*this == o); }
    // general inverse (Gauss-Jordan, partial pivoting) -- used on 4x4 poses and 6x6 covariances
    Mat inverse() const
    {
        if (r_ != c_) throw std::invalid_argument("Mat: inverse of a non-square matrix");
        const int n = r_;
        std::vector<double> a((size_t)n * 2 * n, 0.0);
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < n; j++) a[(size_t)i * 2 * n + j] = (double)(*this)(i, j);
            a[(size_t)i * 2 * n + n + i] = 1.0;
        }
        for (int c = 0; c < n; c++) {
            int p = c;
            for (int r = c + 1; r < n; r++)
                if (std::fabs(a[(size_t)r * 2 * n + c]) > std::fabs(a[(size_t)p * 2 * n + c])) p = r;
            if (a[(size_t)p * 2 * n + c] == 0.0) throw std::domain_error("Mat: singular matrix");
            if (p != c)
                for (int j = 0; j < 2 * n; j++) std::swap(a[(size_t)c * 2 * n + j], a[(size_t)p * 2 * n + j]);
            const double d = a[(size_t)c * 2 * n + c];
            for (int j = 0; j < 2 * n; j++) a[(size_t)c * 2 * n + j] /= d;
            for (int r = 0; r < n; r++) {
                if (r == c) continue;
                const double f = a[(size_t)r * 2 * n + c];
                if (f != 0.0)
                    for (int j = 0; j < 2 * n; j++) a[(size_t)r * 2 * n + j] -= f * a[(size_t)c * 2 * n + j];
            }
        }
        Mat m(n, n);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) m(i, j) = (T)a[(size_t)i * 2 * n + n + j];
        return m;
    }
    template <typename U> Mat<U> cast() const
    {
        Mat<U> m(r_, c_);
        for (int j = 0; j < c_; j++)
            for (int i = 0; i < r_; i++) m(i, j) = (U)(*this)(i, j);
        return m;
    }
    T norm() const { double s = 0; for (auto x : v_) s += (double)x * (double)x; return (T)std::sqrt(s); }
    T sum() const { double s = 0; for (auto x : v_) s += (double)x; return (T)s; }

private:
    int r_, c_;
    std::vector<T> v_;
};

// column-major 4x4 (PM::Matrix) <-> the ABI's 16 row-major doubles
template <typename T>
inline void to_row_major16(const Mat<T> &m, double out[16])
{
    if (m.rows() != 4 || m.cols() != 4) throw std::invalid_argument("expected a 4x4 transformation matrix");
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[i * 4 + j] = (double)m(i, j);
}
template <typename T>
inline Mat<T> from_row_major16(const double in[16])
{
    Mat<T> m(4, 4);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) m(i, j) = (T)in[i * 4 + j];
    return m;
}

}  // namespace pgslam_amd
