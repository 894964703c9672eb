// Drop-in replacement for the reference header cpp/sage_icp/core/Registration.hpp
// (NeSC-IV/sage-icp @ 2024_10_08, lines 32-39): the same two free functions, implemented over the
// C ABI of libsageicp_hip.so.  The reference's Registration.cpp is not compiled any more; the
// call site pipeline/sageICP.cpp:80-85 is unchanged.
#pragma once

#include <Eigen/Core>
#include <algorithm>
#include <sophus/se3.hpp>
#include <stdexcept>
#include <string>
#include <vector>

#include "VoxelHashMap.hpp"
#include "sageicp.h"

namespace sage_icp {

// core/Registration.cpp:103-111
inline void TransformPoints(const Sophus::SE3d &T, std::vector<Eigen::Vector4d> &points) {
    if (points.empty()) return;
    if (sageicp_transform_points(T.data(), points.front().data(), points.size(),
                                 VoxelHashMap::Device()) != SAGEICP_OK)
        throw std::runtime_error(std::string("sage_icp::TransformPoints: ") + sageicp_last_error());
}

// core/Registration.cpp:113-141
inline Sophus::SE3d RegisterFrame(const std::vector<Eigen::Vector4d> &frame,
                                  const VoxelHashMap &voxel_map,
                                  const Sophus::SE3d &initial_guess,
                                  double max_correspondence_distance,
                                  double kernel,
                                  double sem_th) {
    // Sophus::SE3d::data() is (qx, qy, qz, qw, tx, ty, tz): the ABI's pose layout
    double out[7];
    if (sageicp_register_frame(voxel_map.handle(), frame.empty() ? nullptr : frame.front().data(),
                               frame.size(), initial_guess.data(), max_correspondence_distance,
                               kernel, sem_th, out, nullptr) != SAGEICP_OK)
        throw std::runtime_error(std::string("sage_icp::RegisterFrame: ") + sageicp_last_error());
    Sophus::SE3d pose;
    std::copy(out, out + 7, pose.data());     // unit quaternion (x, y, z, w) then translation
    return pose;
}

}  // namespace sage_icp
