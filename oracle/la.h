// la.h — tiny dependency-free linear algebra for the CPU ORACLE (test infrastructure, not the product).
// Stands in for the Eigen types the reference uses (Vector3d, Matrix3d, Quaterniond, MatrixXd,
// SelfAdjointEigenSolver, JacobiSVD, LLT); semantics follow Eigen where it matters
// (Hamilton product, q*v, Quaterniond(Matrix3d), un-normalised deltaQ).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <vector>

namespace ola {

struct V3 {
    double x = 0, y = 0, z = 0;
    V3() = default;
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    double& operator[](int i) { return (&x)[i]; }
    double operator[](int i) const { return (&x)[i]; }
    V3 operator+(const V3& o) const { return { x + o.x, y + o.y, z + o.z }; }
    V3 operator-(const V3& o) const { return { x - o.x, y - o.y, z - o.z }; }
    V3 operator-() const { return { -x, -y, -z }; }
    V3 operator*(double s) const { return { x * s, y * s, z * s }; }
    V3 operator/(double s) const { return { x / s, y / s, z / s }; }
    V3& operator+=(const V3& o) { x += o.x; y += o.y; z += o.z; return *this; }
    double dot(const V3& o) const { return x * o.x + y * o.y + z * o.z; }
    V3 cross(const V3& o) const { return { y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x }; }
    double norm() const { return std::sqrt(dot(*this)); }
    V3 normalized() const { double n = norm(); return n > 0 ? *this / n : *this; }
};
inline V3 operator*(double s, const V3& v) { return v * s; }

struct M3 {
    double m[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
    static M3 identity() { M3 r; r.m[0][0] = r.m[1][1] = r.m[2][2] = 1; return r; }
    double& operator()(int i, int j) { return m[i][j]; }
    double operator()(int i, int j) const { return m[i][j]; }
    M3 operator*(const M3& o) const {
        M3 r;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += m[i][k] * o.m[k][j]; r.m[i][j] = s; }
        return r;
    }
    V3 operator*(const V3& v) const { return { m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z, m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z }; }
    M3 operator*(double s) const { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[i][j] * s; return r; }
    M3 operator+(const M3& o) const { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[i][j] + o.m[i][j]; return r; }
    M3 operator-(const M3& o) const { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[i][j] - o.m[i][j]; return r; }
    M3 operator-() const { return *this * -1.0; }
    M3 t() const { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[j][i]; return r; }
    V3 col(int j) const { return { m[0][j], m[1][j], m[2][j] }; }
    V3 row(int i) const { return { m[i][0], m[i][1], m[i][2] }; }
};
inline M3 skew(const V3& q) { M3 r; r.m[0][1] = -q.z; r.m[0][2] = q.y; r.m[1][0] = q.z; r.m[1][2] = -q.x; r.m[2][0] = -q.y; r.m[2][1] = q.x; return r; }

struct Q {   // Eigen::Quaterniond
    double w = 1, x = 0, y = 0, z = 0;
    Q() = default;
    Q(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
    V3 vec() const { return { x, y, z }; }
    Q operator*(const Q& b) const {
        return { w * b.w - x * b.x - y * b.y - z * b.z, w * b.x + x * b.w + y * b.z - z * b.y,
                 w * b.y + y * b.w + z * b.x - x * b.z, w * b.z + z * b.w + x * b.y - y * b.x };
    }
    double sqnorm() const { return w * w + x * x + y * y + z * z; }
    Q conj() const { return { w, -x, -y, -z }; }
    Q inverse() const { double n = sqnorm(); return { w / n, -x / n, -y / n, -z / n }; }
    Q normalized() const { double n = std::sqrt(sqnorm()); return { w / n, x / n, y / n, z / n }; }
    V3 operator*(const V3& v) const {   // Eigen _transformVector
        V3 u = vec();
        V3 uv = u.cross(v); uv = uv + uv;
        return v + uv * w + u.cross(uv);
    }
    M3 R() const {   // toRotationMatrix
        M3 r;
        const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
        const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
        r.m[0][0] = 1 - (tyy + tzz); r.m[0][1] = txy - twz; r.m[0][2] = txz + twy;
        r.m[1][0] = txy + twz; r.m[1][1] = 1 - (txx + tzz); r.m[1][2] = tyz - twx;
        r.m[2][0] = txz - twy; r.m[2][1] = tyz + twx; r.m[2][2] = 1 - (txx + tyy);
        return r;
    }
    static Q fromR(const M3& m) {   // Eigen quaternionbase_assign_impl<Matrix3>
        Q q;
        double t = m(0, 0) + m(1, 1) + m(2, 2);
        if (t > 0) {
            t = std::sqrt(t + 1.0); q.w = 0.5 * t; t = 0.5 / t;
            q.x = (m(2, 1) - m(1, 2)) * t; q.y = (m(0, 2) - m(2, 0)) * t; q.z = (m(1, 0) - m(0, 1)) * t;
        } else {
            int i = 0;
            if (m(1, 1) > m(0, 0)) i = 1;
            if (m(2, 2) > m(i, i)) i = 2;
            int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
            double v[3];
            v[i] = 0.5 * t; t = 0.5 / t;
            q.w = (m(k, j) - m(j, k)) * t;
            v[j] = (m(j, i) + m(i, j)) * t; v[k] = (m(k, i) + m(i, k)) * t;
            q.x = v[0]; q.y = v[1]; q.z = v[2];
        }
        return q;
    }
    static Q fromTwoVectors(const V3& a, const V3& b) {   // Eigen setFromTwoVectors (non-degenerate branch)
        V3 v0 = a.normalized(), v1 = b.normalized();
        double c = v1.dot(v0);
        if (c < -1.0 + 1e-12) {   // nearly opposite: any orthogonal axis (Eigen uses an SVD here)
            V3 axis = std::fabs(v0.x) < 0.9 ? V3(1, 0, 0).cross(v0).normalized() : V3(0, 1, 0).cross(v0).normalized();
            return { 0, axis.x, axis.y, axis.z };
        }
        V3 axis = v0.cross(v1);
        double s = std::sqrt((1.0 + c) * 2.0), invs = 1.0 / s;
        return { s * 0.5, axis.x * invs, axis.y * invs, axis.z * invs };
    }
};
inline Q deltaQ(const V3& theta) { return { 1.0, theta.x / 2.0, theta.y / 2.0, theta.z / 2.0 }; }   // Utility::deltaQ (utility.h:32-44)

// dense row-major matrix
struct Mat {
    int r = 0, c = 0; std::vector<double> d;
    Mat() = default;
    Mat(int r_, int c_) : r(r_), c(c_), d((size_t)r_ * c_, 0.0) {}
    double& operator()(int i, int j) { return d[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
    void zero() { std::fill(d.begin(), d.end(), 0.0); }
};
inline Mat matmul(const Mat& a, const Mat& b) {
    Mat o(a.r, b.c);
    for (int i = 0; i < a.r; ++i) for (int k = 0; k < a.c; ++k) { double v = a(i, k); if (v == 0) continue; for (int j = 0; j < b.c; ++j) o(i, j) += v * b(k, j); }
    return o;
}
inline Mat transpose(const Mat& a) { Mat o(a.c, a.r); for (int i = 0; i < a.r; ++i) for (int j = 0; j < a.c; ++j) o(j, i) = a(i, j); return o; }

// symmetric eigen-decomposition (cyclic Jacobi); eigenvalues ascending like SelfAdjointEigenSolver
inline void sym_eig(const Mat& A, std::vector<double>& ev, Mat& V) {
    const int n = A.r;
    Mat a = A; V = Mat(n, n);
    for (int i = 0; i < n; ++i) V(i, i) = 1;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i) { diag += a(i, i) * a(i, i); for (int j = i + 1; j < n; ++j) off += a(i, j) * a(i, j); }
        if (off <= 1e-30 * (diag + 1e-300)) break;
        for (int p = 0; p < n - 1; ++p) for (int q = p + 1; q < n; ++q) {
            double apq = a(p, q);
            if (std::fabs(apq) < 1e-300) continue;
            double theta = (a(q, q) - a(p, p)) / (2 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
            double cs = 1 / std::sqrt(t * t + 1), sn = t * cs;
            for (int k = 0; k < n; ++k) { double akp = a(k, p), akq = a(k, q); a(k, p) = cs * akp - sn * akq; a(k, q) = sn * akp + cs * akq; }
            for (int k = 0; k < n; ++k) { double apk = a(p, k), aqk = a(q, k); a(p, k) = cs * apk - sn * aqk; a(q, k) = sn * apk + cs * aqk; }
            for (int k = 0; k < n; ++k) { double vkp = V(k, p), vkq = V(k, q); V(k, p) = cs * vkp - sn * vkq; V(k, q) = sn * vkp + cs * vkq; }
        }
    }
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int x, int y) { return a(x, x) < a(y, y); });
    ev.resize(n);
    Mat Vs(n, n);
    for (int k = 0; k < n; ++k) { ev[k] = a(idx[k], idx[k]); for (int i = 0; i < n; ++i) Vs(i, k) = V(i, idx[k]); }
    V = Vs;
}

// right singular vector of the smallest singular value of an (m x 4) matrix == JacobiSVD(...).matrixV().rightCols<1>()
inline void smallest_right_singular4(const Mat& A, double out[4]) {
    Mat AtA(4, 4);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < A.r; ++k) s += A(k, i) * A(k, j); AtA(i, j) = s; }
    std::vector<double> ev; Mat V;
    sym_eig(AtA, ev, V);
    for (int i = 0; i < 4; ++i) out[i] = V(i, 0);
}

// Cholesky A = L L^T (lower); returns false if not positive definite
inline bool cholesky(const Mat& A, Mat& L) {
    const int n = A.r; L = Mat(n, n);
    for (int j = 0; j < n; ++j) {
        double s = A(j, j);
        for (int k = 0; k < j; ++k) s -= L(j, k) * L(j, k);
        if (!(s > 0)) return false;
        double d = std::sqrt(s); L(j, j) = d;
        for (int i = j + 1; i < n; ++i) { double t = A(i, j); for (int k = 0; k < j; ++k) t -= L(i, k) * L(j, k); L(i, j) = t / d; }
    }
    return true;
}
inline void chol_solve(const Mat& L, std::vector<double>& b) {   // in place
    const int n = L.r;
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L(i, k) * b[k]; b[i] = s / L(i, i); }
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= L(k, i) * b[k]; b[i] = s / L(i, i); }
}
// general inverse by Gauss-Jordan with partial pivoting (Matrix15d::inverse stand-in)
inline Mat inverse(const Mat& A) {
    const int n = A.r; Mat a = A, inv(n, n);
    for (int i = 0; i < n; ++i) inv(i, i) = 1;
    for (int c = 0; c < n; ++c) {
        int p = c; for (int i = c + 1; i < n; ++i) if (std::fabs(a(i, c)) > std::fabs(a(p, c))) p = i;
        if (p != c) for (int j = 0; j < n; ++j) { std::swap(a(c, j), a(p, j)); std::swap(inv(c, j), inv(p, j)); }
        double piv = a(c, c);
        for (int j = 0; j < n; ++j) { a(c, j) /= piv; inv(c, j) /= piv; }
        for (int i = 0; i < n; ++i) if (i != c) { double f = a(i, c); if (f == 0) continue; for (int j = 0; j < n; ++j) { a(i, j) -= f * a(c, j); inv(i, j) -= f * inv(c, j); } }
    }
    return inv;
}

}  // namespace ola
