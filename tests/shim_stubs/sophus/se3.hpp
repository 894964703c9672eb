// Minimal stand-in for <sophus/se3.hpp> (see Eigen/Core next to it): seven contiguous doubles
// (qx, qy, qz, qw, tx, ty, tz) behind data(), which is all the shim relies on.
#pragma once
namespace Sophus {
struct SE3d {
    double v[7] = {0, 0, 0, 1, 0, 0, 0};
    double *data() { return v; }
    const double *data() const { return v; }
};
}  // namespace Sophus
