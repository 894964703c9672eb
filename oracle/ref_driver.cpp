// C driver around the REFERENCE's own hot path (test infrastructure; compiled only by oracle/ref_build.mk, and only
// where Eigen, Sophus, oneTBB and tsl::robin_map are installed — they are not in the build image, where this file is
// never compiled).  It includes the reference's headers from /root/reference and links against its two sources,
// compiled unmodified; nothing of the reference is copied here.  The functions mirror oracle/sage_oracle.cpp's C
// interface one for one so that tests/test_reference_build.py can run the same scenes through both:
//   core/VoxelHashMap.hpp:79-99   VoxelHashMap ctor / AddPoints / Update / GetCorrespondences / Pointcloud
//   core/Registration.hpp:32-39   RegisterFrame
#include <cstdint>
#include <cstring>
#include <vector>

#include "sage_icp/core/Registration.hpp"
#include "sage_icp/core/VoxelHashMap.hpp"

namespace {
std::vector<Eigen::Vector4d> to_vec(const double *xyzl, uint64_t n) {
    std::vector<Eigen::Vector4d> v(n);
    for (uint64_t i = 0; i < n; ++i) v[i] = Eigen::Vector4d(xyzl[4 * i], xyzl[4 * i + 1], xyzl[4 * i + 2], xyzl[4 * i + 3]);
    return v;
}
Sophus::SE3d to_se3(const double q[7]) {          // (qx, qy, qz, qw, tx, ty, tz): include/sageicp.h's pose layout
    return Sophus::SE3d(Eigen::Quaterniond(q[3], q[0], q[1], q[2]), Eigen::Vector3d(q[4], q[5], q[6]));
}
void from_se3(const Sophus::SE3d &T, double q[7]) {
    const Eigen::Quaterniond u = T.unit_quaternion();
    q[0] = u.x(); q[1] = u.y(); q[2] = u.z(); q[3] = u.w();
    q[4] = T.translation().x(); q[5] = T.translation().y(); q[6] = T.translation().z();
}
}  // namespace

extern "C" {

void *ref_map_create(double voxel_size, double max_distance, int basic, int critical, const int *labels, int n_labels) {
    return new sage_icp::VoxelHashMap(voxel_size, max_distance, basic, critical, std::vector<int>(labels, labels + n_labels));
}
void ref_map_destroy(void *m) { delete static_cast<sage_icp::VoxelHashMap *>(m); }
void ref_map_add_points(void *m, const double *xyzl, uint64_t n) {
    static_cast<sage_icp::VoxelHashMap *>(m)->AddPoints(to_vec(xyzl, n));
}
void ref_map_update(void *m, const double *xyzl, uint64_t n, const double origin[3]) {
    static_cast<sage_icp::VoxelHashMap *>(m)->Update(to_vec(xyzl, n), Eigen::Vector3d(origin[0], origin[1], origin[2]));
}
uint64_t ref_map_pointcloud(void *m, double *out, uint64_t cap) {
    const auto pc = static_cast<sage_icp::VoxelHashMap *>(m)->Pointcloud();
    for (uint64_t i = 0; i < pc.size() && i < cap; ++i) std::memcpy(out + 4 * i, pc[i].data(), 32);
    return pc.size();
}
// src / tgt: [n][4]; returns the number of correspondences (in query order, VoxelHashMap.cpp:119-129)
uint64_t ref_get_correspondences(void *m, const double *q, uint64_t n, double max_dist, double sem_th, double *src, double *tgt) {
    const auto [s, t] = static_cast<sage_icp::VoxelHashMap *>(m)->GetCorrespondences(to_vec(q, n), max_dist, sem_th);
    for (uint64_t i = 0; i < s.size(); ++i) {
        std::memcpy(src + 4 * i, s[i].data(), 32);
        std::memcpy(tgt + 4 * i, t[i].data(), 32);
    }
    return s.size();
}
void ref_register_frame(void *m, const double *frame, uint64_t n, const double init[7], double max_dist, double kernel,
                        double sem_th, double out[7]) {
    from_se3(sage_icp::RegisterFrame(to_vec(frame, n), *static_cast<sage_icp::VoxelHashMap *>(m), to_se3(init), max_dist, kernel,
                                     sem_th),
             out);
}

}  // extern "C"
