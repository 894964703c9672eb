// Drop-in replacement for the reference header cpp/sage_icp/core/Preprocessing.hpp
// (NeSC-IV/sage-icp @ 2024_10_08, lines 33-45): the same two free functions, running on the
// MI355X through the C ABI of libsageicp_hip.so (preprocess.hip).  The reference's
// Preprocessing.cpp — and with it the PCL dependency — is not compiled any more.
//
// Preprocess() with dynamic_vehicle_filter == true (PCL Euclidean clustering,
// Preprocessing.cpp:95-172) is not implemented: it throws.  Every pre-labelled configuration
// (ros/launch/odometry_gt.launch.py) runs with the filter off.
// VoxelDownsample() returns the survivors group by group in input order; the reference returns
// them in tsl::robin_map bucket order (same set of points).
#pragma once

#include <Eigen/Core>
#include <stdexcept>
#include <string>
#include <vector>

#include "VoxelHashMap.hpp"
#include "sageicp.h"

namespace sage_icp {

// core/Preprocessing.cpp:44-84
inline std::vector<Eigen::Vector4d> VoxelDownsample(const std::vector<Eigen::Vector4d> &frame,
                                                    const std::vector<std::vector<int>> &voxel_labels,
                                                    const std::vector<double> &voxel_size,
                                                    double vox_scale) {
    std::vector<int> counts, labels;
    for (const auto &g : voxel_labels) {
        counts.push_back(static_cast<int>(g.size()));
        labels.insert(labels.end(), g.begin(), g.end());
    }
    std::vector<Eigen::Vector4d> out(frame.size());
    uint64_t n = 0;
    if (sageicp_voxel_downsample(frame.empty() ? nullptr : frame.front().data(), frame.size(),
                                 static_cast<int>(voxel_size.size()), counts.data(), labels.data(),
                                 voxel_size.data(), vox_scale,
                                 out.empty() ? nullptr : out.front().data(), &n,
                                 VoxelHashMap::Device()) != SAGEICP_OK)
        throw std::runtime_error(std::string("sage_icp::VoxelDownsample: ") + sageicp_last_error());
    out.resize(n);
    return out;
}

// core/Preprocessing.cpp:86-187
inline std::vector<Eigen::Vector4d> Preprocess(const std::vector<Eigen::Vector4d> &frame,
                                               double max_range, double min_range,
                                               double label_max_range, bool dynamic_vehicle_filter,
                                               double /*dy_th*/,
                                               const std::vector<int> & /*dynamic_labels*/,
                                               const std::vector<int> & /*lankmark*/) {
    if (dynamic_vehicle_filter)
        throw std::runtime_error("sage_icp::Preprocess: dynamic_vehicle_filter needs PCL and is not "
                                 "available in the MI355X build");
    std::vector<Eigen::Vector4d> out(frame.size());
    uint64_t n = 0;
    if (sageicp_preprocess(frame.empty() ? nullptr : frame.front().data(), frame.size(), max_range,
                           min_range, label_max_range, out.empty() ? nullptr : out.front().data(), &n,
                           VoxelHashMap::Device()) != SAGEICP_OK)
        throw std::runtime_error(std::string("sage_icp::Preprocess: ") + sageicp_last_error());
    out.resize(n);
    return out;
}

}  // namespace sage_icp
