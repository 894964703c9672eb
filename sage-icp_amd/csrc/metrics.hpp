// KITTI trajectory metrics (row f-4): the relative sequence error of the KITTI devkit and the
// absolute trajectory error after a rigid Umeyama alignment.  Host-only C++ (the reference's
// metrics are CPU code as well); no Eigen: the 4x4 products / rigid inverses, the 3x3 SVD behind
// the alignment and the rotation angle are written out.
//
// Reference: cpp/sage_icp/metrics/Metrics.cpp
//   SeqError                 :140-155 (CalcSequenceErrors :88-136, lengths 100..800 m :35, every
//                            10th frame :96, speed from 10 Hz :126, rot. error scaled by 180/3.14 :152)
//   AbsoluteTrajectoryError  :157-191 (Eigen::umeyama(source, target, false) :169, RMSE :186-190)
// Poses are row-major 4x4 doubles (Eigen::Matrix4d is column-major: the Python binding and the
// C ABI take row-major, the natural layout of a KITTI poses.txt row padded with 0 0 0 1).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace sageicp {
namespace metrics {

struct M4 {
    double m[16];
    double operator()(int r, int c) const { return m[4 * r + c]; }
    double &operator()(int r, int c) { return m[4 * r + c]; }
};

inline M4 mul(const M4 &a, const M4 &b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a(i, k) * b(k, j);
            r(i, j) = s;
        }
    return r;
}

// General 4x4 inverse by cofactors (what Eigen's fixed-size Matrix4d::inverse() evaluates); the
// inputs are rigid transforms, but the devkit code inverts them as plain matrices.
inline M4 inverse(const M4 &a) {
    const double *m = a.m;
    double inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
             m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
             m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
             m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
              m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
             m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
             m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
             m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
              m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
             m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
             m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
              m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
              m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
             m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
             m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
              m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
              m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    M4 r;
    for (int i = 0; i < 16; ++i) r.m[i] = inv[i] / det;
    return r;
}

// ---- KITTI devkit relative error (Metrics.cpp:33-136) ------------------------------------------
struct SegmentError {
    int32_t first_frame;
    double r_err, t_err, len, speed;
};

inline std::vector<double> trajectory_distances(const std::vector<M4> &poses) {   // :44-59
    std::vector<double> dist;
    dist.push_back(0.0);
    for (size_t i = 1; i < poses.size(); ++i) {
        const double dx = poses[i - 1](0, 3) - poses[i](0, 3);
        const double dy = poses[i - 1](1, 3) - poses[i](1, 3);
        const double dz = poses[i - 1](2, 3) - poses[i](2, 3);
        dist.push_back(dist[i - 1] + std::sqrt(dx * dx + dy * dy + dz * dz));
    }
    return dist;
}

inline int32_t last_frame_from_segment_length(const std::vector<double> &dist, int32_t first,
                                              double len) {   // :61-70
    for (size_t i = static_cast<size_t>(first); i < dist.size(); ++i)
        if (dist[i] > dist[first] + len) return static_cast<int32_t>(i);
    return -1;
}

inline std::vector<SegmentError> sequence_errors(const std::vector<M4> &gt,
                                                 const std::vector<M4> &res) {   // :88-136
    static const double lengths[8] = {100, 200, 300, 400, 500, 600, 700, 800};
    std::vector<SegmentError> err;
    const int32_t step_size = 10;
    const std::vector<double> dist = trajectory_distances(gt);
    for (size_t first = 0; first < gt.size(); first += step_size) {
        for (int i = 0; i < 8; ++i) {
            const double len = lengths[i];
            const int32_t last = last_frame_from_segment_length(dist, static_cast<int32_t>(first), len);
            if (last == -1) continue;
            const M4 d_gt = mul(inverse(gt[first]), gt[last]);
            const M4 d_res = mul(inverse(res[first]), res[last]);
            const M4 e = mul(inverse(d_res), d_gt);
            const double d = 0.5 * (e(0, 0) + e(1, 1) + e(2, 2) - 1.0);            // :72-78
            const double r_err = std::acos(std::max(std::min(d, 1.0), -1.0));
            const double t_err = std::sqrt(e(0, 3) * e(0, 3) + e(1, 3) * e(1, 3) + e(2, 3) * e(2, 3));
            const double num_frames = static_cast<double>(last - static_cast<int32_t>(first) + 1);
            err.push_back({static_cast<int32_t>(first), r_err / len, t_err / len, len,
                           len / (0.1 * num_frames)});
        }
    }
    return err;
}

// SeqError (:140-155): (average translational error in %, average rotational error in deg/100 m
// with the reference's 3.14).  With no segment long enough both are 0/0 = NaN, as in the reference.
inline void seq_error(const std::vector<M4> &gt, const std::vector<M4> &res, float *trans,
                      float *rot) {
    const std::vector<SegmentError> err = sequence_errors(gt, res);
    double t = 0.0, r = 0.0;
    for (const SegmentError &e : err) {
        t += e.t_err;
        r += e.r_err;
    }
    const double n = static_cast<double>(err.size());
    *trans = static_cast<float>(100.0 * (t / n));
    *rot = static_cast<float>(100.0 * (r / n) / 3.14 * 180.0);
}

// ---- 3x3 SVD (one-sided Jacobi) for the Umeyama alignment ------------------------------------------
// A = U diag(s) V^T, s sorted descending and non-negative (the convention of Eigen::JacobiSVD the
// reference relies on through Eigen::umeyama).
inline void svd3(const double A[9], double U[9], double s[3], double V[9]) {
    double B[9];
    for (int i = 0; i < 9; ++i) B[i] = A[i];
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 3; ++k) {
                    alpha += B[3 * k + p] * B[3 * k + p];
                    beta += B[3 * k + q] * B[3 * k + q];
                    gamma += B[3 * k + p] * B[3 * k + q];
                }
                off = std::max(off, std::fabs(gamma) / std::sqrt(std::max(alpha * beta, 1e-300)));
                if (gamma == 0.0) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double bp = B[3 * k + p], bq = B[3 * k + q];
                    B[3 * k + p] = c * bp - sn * bq;
                    B[3 * k + q] = sn * bp + c * bq;
                    const double vp = V[3 * k + p], vq = V[3 * k + q];
                    V[3 * k + p] = c * vp - sn * vq;
                    V[3 * k + q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    int order[3] = {0, 1, 2};
    double nrm[3];
    for (int j = 0; j < 3; ++j)
        nrm[j] = std::sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
    std::sort(order, order + 3, [&](int a, int b) { return nrm[a] > nrm[b]; });
    double Vs[9];
    for (int j = 0; j < 3; ++j) {
        const int o = order[j];
        s[j] = nrm[o];
        for (int k = 0; k < 3; ++k) {
            Vs[3 * k + j] = V[3 * k + o];
            U[3 * k + j] = nrm[o] > 0.0 ? B[3 * k + o] / nrm[o] : 0.0;
        }
    }
    for (int i = 0; i < 9; ++i) V[i] = Vs[i];
    // complete U to an orthonormal basis where singular values vanish (rank-deficient clouds)
    auto col = [&](int j, double v[3]) { v[0] = U[j]; v[1] = U[3 + j]; v[2] = U[6 + j]; };
    auto setcol = [&](int j, const double v[3]) { U[j] = v[0]; U[3 + j] = v[1]; U[6 + j] = v[2]; };
    const double tiny = 1e-300;
    if (!(s[0] > tiny)) {
        const double e0[3] = {1, 0, 0}, e1[3] = {0, 1, 0}, e2[3] = {0, 0, 1};
        setcol(0, e0); setcol(1, e1); setcol(2, e2);
        return;
    }
    if (!(s[1] > tiny * s[0]) || !(s[1] > 1e-14 * s[0])) {
        double u0[3];
        col(0, u0);
        double a[3] = {0, 0, 0};
        a[std::fabs(u0[0]) < 0.9 ? 0 : 1] = 1.0;
        const double d = a[0] * u0[0] + a[1] * u0[1] + a[2] * u0[2];
        double u1[3] = {a[0] - d * u0[0], a[1] - d * u0[1], a[2] - d * u0[2]};
        const double n1 = std::sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
        for (double &x : u1) x /= n1;
        setcol(1, u1);
    }
    if (!(s[2] > 1e-14 * s[0])) {
        double u0[3], u1[3];
        col(0, u0); col(1, u1);
        const double u2[3] = {u0[1] * u1[2] - u0[2] * u1[1], u0[2] * u1[0] - u0[0] * u1[2],
                              u0[0] * u1[1] - u0[1] * u1[0]};
        setcol(2, u2);
    }
}

inline double det3(const double M[9]) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// Eigen::umeyama(src, dst, with_scaling = false) for 3 x n clouds: the rigid T minimising
// sum |dst_i - T src_i|^2 (Umeyama 1991, eq. 40-43 with the reflection guard S).
inline M4 umeyama_rigid(const std::vector<double> &src, const std::vector<double> &dst, size_t n) {
    double ms[3] = {0, 0, 0}, md[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            ms[k] += src[3 * i + k];
            md[k] += dst[3 * i + k];
        }
    const double inv_n = 1.0 / static_cast<double>(n);
    for (int k = 0; k < 3; ++k) {
        ms[k] *= inv_n;
        md[k] *= inv_n;
    }
    double sigma[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // dst_demean * src_demean^T / n
    for (size_t i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                sigma[3 * r + c] += (dst[3 * i + r] - md[r]) * (src[3 * i + c] - ms[c]);
    for (double &x : sigma) x *= inv_n;
    double U[9], s[3], V[9];
    svd3(sigma, U, s, V);
    double S[3] = {1.0, 1.0, 1.0};
    if (det3(U) * det3(V) < 0.0) S[2] = -1.0;
    M4 T{};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
            for (int k = 0; k < 3; ++k) v += U[3 * r + k] * S[k] * V[3 * c + k];   // U S V^T
            T(r, c) = v;
        }
    for (int r = 0; r < 3; ++r)
        T(r, 3) = md[r] - (T(r, 0) * ms[0] + T(r, 1) * ms[1] + T(r, 2) * ms[2]);
    T(3, 0) = T(3, 1) = T(3, 2) = 0.0;
    T(3, 3) = 1.0;
    return T;
}

// angle of Eigen::AngleAxisd(R): through the quaternion of R, 2 atan2(|vec|, |w|)
inline double rotation_angle(const double R[9]) {
    // Eigen's matrix -> quaternion (Shepperd's branches)
    double w, x, y, z;
    const double t = R[0] + R[4] + R[8];
    if (t > 0.0) {
        double r = std::sqrt(t + 1.0);
        w = 0.5 * r;
        r = 0.5 / r;
        x = (R[7] - R[5]) * r;
        y = (R[2] - R[6]) * r;
        z = (R[3] - R[1]) * r;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double r = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        double q[3];
        q[i] = 0.5 * r;
        r = 0.5 / r;
        w = (R[3 * k + j] - R[3 * j + k]) * r;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * r;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * r;
        x = q[0]; y = q[1]; z = q[2];
    }
    const double n = std::sqrt(x * x + y * y + z * z);
    if (n == 0.0) return 0.0;
    return 2.0 * std::atan2(n, std::fabs(w));
}

// AbsoluteTrajectoryError (:157-191): (RMSE of the rotation angle [rad], RMSE of the translation
// [m]) after aligning the estimated positions to the ground truth.
inline void absolute_trajectory_error(const std::vector<M4> &gt, const std::vector<M4> &res,
                                      float *ate_rot, float *ate_trans) {
    const size_t n = gt.size();
    std::vector<double> src(3 * n), dst(3 * n);
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            src[3 * i + k] = res[i](k, 3);
            dst[3 * i + k] = gt[i](k, 3);
        }
    const M4 A = umeyama_rigid(src, dst, n);
    double rot = 0.0, trans = 0.0;
    for (size_t j = 0; j < n; ++j) {
        const M4 E = mul(A, res[j]);
        const M4 &G = gt[j];
        double dR[9];                                   // R_gt * R_est^T
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                dR[3 * r + c] = G(r, 0) * E(c, 0) + G(r, 1) * E(c, 1) + G(r, 2) * E(c, 2);
        double dt2 = 0.0;
        for (int r = 0; r < 3; ++r) {
            const double d = G(r, 3) - (dR[3 * r] * E(0, 3) + dR[3 * r + 1] * E(1, 3) + dR[3 * r + 2] * E(2, 3));
            dt2 += d * d;
        }
        const double th = rotation_angle(dR);
        rot += th * th;
        trans += dt2;
    }
    rot /= static_cast<double>(n);
    trans /= static_cast<double>(n);
    *ate_rot = static_cast<float>(std::sqrt(rot));
    *ate_trans = static_cast<float>(std::sqrt(trans));
}

}  // namespace metrics
}  // namespace sageicp
