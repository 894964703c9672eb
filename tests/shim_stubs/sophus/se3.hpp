// Minimal stand-in for <sophus/se3.hpp> (see Eigen/Core next to it): seven contiguous doubles
// (qx, qy, qz, qw, tx, ty, tz) behind data() — all the shim relies on — plus the three members the
// reference's caller (pipeline/sageICP.cpp:74-76,90,119) uses on poses: inverse(), operator*,
// translation().
#pragma once
#include <Eigen/Core>
namespace Sophus {
struct SE3d {
    double v[7] = {0, 0, 0, 1, 0, 0, 0};
    double *data() { return v; }
    const double *data() const { return v; }
    Eigen::Vector3d translation() const { return Eigen::Vector3d(v[4], v[5], v[6]); }
    static void rotate(const double q[4], const double p[3], double o[3]) {
        // o = p + 2 w (u x p) + 2 u x (u x p), u = q.vec, w = q.w
        const double cx = q[1] * p[2] - q[2] * p[1], cy = q[2] * p[0] - q[0] * p[2], cz = q[0] * p[1] - q[1] * p[0];
        const double dx = q[1] * cz - q[2] * cy, dy = q[2] * cx - q[0] * cz, dz = q[0] * cy - q[1] * cx;
        o[0] = p[0] + 2.0 * (q[3] * cx + dx);
        o[1] = p[1] + 2.0 * (q[3] * cy + dy);
        o[2] = p[2] + 2.0 * (q[3] * cz + dz);
    }
    SE3d operator*(const SE3d &b) const {
        SE3d r;
        const double *a = v, *c = b.v;
        r.v[0] = a[3] * c[0] + a[0] * c[3] + a[1] * c[2] - a[2] * c[1];
        r.v[1] = a[3] * c[1] - a[0] * c[2] + a[1] * c[3] + a[2] * c[0];
        r.v[2] = a[3] * c[2] + a[0] * c[1] - a[1] * c[0] + a[2] * c[3];
        r.v[3] = a[3] * c[3] - a[0] * c[0] - a[1] * c[1] - a[2] * c[2];
        double t[3];
        rotate(a, c + 4, t);
        r.v[4] = t[0] + a[4]; r.v[5] = t[1] + a[5]; r.v[6] = t[2] + a[6];
        return r;
    }
    SE3d inverse() const {
        SE3d r;
        r.v[0] = -v[0]; r.v[1] = -v[1]; r.v[2] = -v[2]; r.v[3] = v[3];
        double t[3];
        rotate(r.v, v + 4, t);
        r.v[4] = -t[0]; r.v[5] = -t[1]; r.v[6] = -t[2];
        return r;
    }
};
}  // namespace Sophus
