// OPTIONAL replacement for the reference header cpp/sage_icp/core/Preprocessing.hpp
// (NeSC-IV/sage-icp @ 2024_10_08, lines 33-45): the same two free functions, running on the
// MI355X through the C ABI of libsageicp_hip.so (preprocess.hip).
//
// Opt-in: this header lives in its own include root (sage-icp_amd/shim_preprocessing) and is NOT
// part of the drop-in for the registration hot path.  Add that root only for configurations that
// run with dynamic_vehicle_filter == false (ros/launch/odometry_gt.launch.py): the PCL Euclidean
// clustering of Preprocessing.cpp:95-172 is not reproduced here, and Preprocess() throws when it
// is asked for.  The reference's default SemanticKITTI launch (ros/launch/odometry.launch.py:50)
// sets the filter to true — keep the reference's own Preprocessing.{hpp,cpp} there (the default
// of INTEGRATION.md); registration still runs on the GPU.
// VoxelDownsample() returns the survivors in the reference's order — the bucket order of its
// tsl::robin_map, replayed by the library (csrc/robin_order.hpp) — unless
// sageicp_set_downsample_order(0) selects the faster group-by-group input order.
#pragma once

#include <Eigen/Core>
#include <stdexcept>
#include <string>
#include <vector>

#include "sage_icp/core/VoxelHashMap.hpp"
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
