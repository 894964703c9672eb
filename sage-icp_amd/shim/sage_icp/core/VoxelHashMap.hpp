// Drop-in replacement for the reference header cpp/sage_icp/core/VoxelHashMap.hpp
// (NeSC-IV/sage-icp @ 2024_10_08, lines 35-107): the same `sage_icp::VoxelHashMap` value type —
// constructor, member functions and public data members — implemented over the C ABI of
// libsageicp_hip.so (include/sageicp.h).  pipeline/sageICP.{hpp,cpp} and
// ros/ros2/OdometryServer.cpp compile against it unchanged; only the `map_` member (a
// tsl::robin_map in the reference, touched by nothing outside the struct) is replaced by an
// opaque handle to the host-authoritative map + HBM mirror.
//
// Value semantics are preserved: copy construction / copy assignment deep-copy the map
// (sageicp_map_clone) as OdometryServer.cpp:104 (`odometry_ = sageICP(config_)`) requires.
// A HIP failure has no analogue in the reference (its hot path has no error channel), so a
// non-zero ABI return is surfaced as std::runtime_error.
#pragma once

#include <Eigen/Core>
#include <sophus/se3.hpp>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "sageicp.h"

namespace sage_icp {

struct VoxelHashMap {
    using Vector4dVector = std::vector<Eigen::Vector4d>;
    using Vector4dVectorTuple = std::tuple<Vector4dVector, Vector4dVector>;
    using Voxel = Eigen::Vector3i;

    explicit VoxelHashMap(double voxel_size, double max_distance, int basic_points_per_voxel,
                          int critical_points_per_voxel, std::vector<int> basic_parts_labels)
        : voxel_size_(voxel_size),
          max_distance_(max_distance),
          basic_points_per_voxel_(basic_points_per_voxel),
          critical_points_per_voxel_{critical_points_per_voxel},
          basic_parts_labels_{std::move(basic_parts_labels)},
          map_(sageicp_map_create(voxel_size_, max_distance_, basic_points_per_voxel_,
                                  critical_points_per_voxel_, basic_parts_labels_.data(),
                                  static_cast<int>(basic_parts_labels_.size()), Device())) {
        if (!map_) throw std::runtime_error(std::string("sageicp_map_create: ") + sageicp_last_error());
#ifndef SAGE_ICP_SHIM_FAST_MAP
        // The drop-in node keeps the voxels the reference keeps: RemovePointsFarFromLocation erases while it
        // iterates its robin_map (VoxelHashMap.cpp:176-184), so the entry a deletion shifts into the bucket just
        // erased survives until a later frame, and Pointcloud() lists bucket order (:132-142).  A map in
        // reference-order mode reproduces both exactly (include/sageicp.h, sageicp_map_set_reference_order); since
        // round 6 its Update() runs on the GPU as well — the device inserts and finds the far voxels, the host replays
        // only the voxels concerned on its copy of the bucket array — at 0.38 ms per streamed frame against 0.25 ms
        // for -DSAGE_ICP_SHIM_FAST_MAP (every out-of-range voxel removed at once, block-pool order); poses are the
        // same on every stream measured.
        if (sageicp_map_set_reference_order(map_, 1) != 0) {
            const std::string why = sageicp_last_error();
            sageicp_map_destroy(map_);
            map_ = nullptr;
            throw std::runtime_error("sageicp_map_set_reference_order: " + why);
        }
#endif
    }

    VoxelHashMap(const VoxelHashMap &o)
        : voxel_size_(o.voxel_size_),
          max_distance_(o.max_distance_),
          basic_points_per_voxel_(o.basic_points_per_voxel_),
          critical_points_per_voxel_(o.critical_points_per_voxel_),
          basic_parts_labels_(o.basic_parts_labels_),
          map_(sageicp_map_clone(o.map_)) {
        // a failed clone (HIP out of memory, device error) must not pass for an empty map
        if (!map_) throw std::runtime_error(std::string("sageicp_map_clone: ") + sageicp_last_error());
    }
    VoxelHashMap(VoxelHashMap &&o) noexcept
        : voxel_size_(o.voxel_size_),
          max_distance_(o.max_distance_),
          basic_points_per_voxel_(o.basic_points_per_voxel_),
          critical_points_per_voxel_(o.critical_points_per_voxel_),
          basic_parts_labels_(std::move(o.basic_parts_labels_)),
          map_(o.map_) {
        o.map_ = nullptr;
    }
    VoxelHashMap &operator=(VoxelHashMap o) noexcept {   // copy-and-swap: copy and move assignment
        swap(o);
        return *this;
    }
    ~VoxelHashMap() { sageicp_map_destroy(map_); }

    // core/VoxelHashMap.cpp:48-130
    Vector4dVectorTuple GetCorrespondences(const Vector4dVector &points,
                                           double max_correspondance_distance, double th) const {
        Vector4dVector src(points.size()), tgt(points.size());
        uint64_t n = 0;
        Check(sageicp_get_correspondences(map_, Data(points), points.size(),
                                          max_correspondance_distance, th, Data(src), Data(tgt), &n,
                                          nullptr),
              "GetCorrespondences");
        src.resize(n);
        tgt.resize(n);
        return std::make_tuple(std::move(src), std::move(tgt));
    }
    inline void Clear() { Check(sageicp_map_clear(map_), "Clear"); }
    inline bool Empty() const { return sageicp_map_empty(map_) != 0; }
    void Update(const Vector4dVector &points, const Eigen::Vector3d &origin) {
        Check(sageicp_map_update(map_, Data(points), points.size(), origin.data()), "Update");
    }
    // the per-frame map update (pipeline/sageICP.cpp:89) runs on the GPU against the HBM-resident
    // map; same voxel blocks as the host entry sageicp_map_update_pose()
    void Update(const Vector4dVector &points, const Sophus::SE3d &pose) {
        Check(sageicp_map_update_pose_device(map_, Data(points), points.size(), pose.data()), "Update");
    }
    void AddPoints(const Vector4dVector &points) {
        Check(sageicp_map_add_points(map_, Data(points), points.size()), "AddPoints");
    }
    void RemovePointsFarFromLocation(const Eigen::Vector3d &origin) {
        Check(sageicp_map_remove_far(map_, origin.data()), "RemovePointsFarFromLocation");
    }
    Vector4dVector Pointcloud() const {
        Vector4dVector out(sageicp_map_size(map_));
        sageicp_map_pointcloud(map_, Data(out), out.size());
        return out;
    }

    double voxel_size_;
    double max_distance_;
    int basic_points_per_voxel_;
    int critical_points_per_voxel_;
    std::vector<int> basic_parts_labels_;

    // ---- not part of the reference surface -------------------------------------------------
    const sageicp_map *handle() const { return map_; }     // used by the Registration.hpp shim
    // HIP device ordinal for maps created by this process (one process per GPU); set it before
    // the first map is constructed, e.g. from LOCAL_RANK.
    static int &Device() {
        // the library this process loaded must speak the ABI this header was written against
        // (struct layouts are part of it): checked once, before the first map exists
        static const bool abi_ok = [] {
            if (sageicp_abi_version() != SAGEICP_ABI_VERSION)
                throw std::runtime_error("libsageicp_hip.so speaks ABI version " +
                                         std::to_string(sageicp_abi_version()) + ", the header shim " +
                                         std::to_string(SAGEICP_ABI_VERSION));
            return true;
        }();
        (void)abi_ok;
        static int device = 0;
        return device;
    }

private:
    // Eigen::Vector4d is four contiguous doubles, so a vector of them is the ABI's double[n][4]
    static const double *Data(const Vector4dVector &v) { return v.empty() ? nullptr : v.front().data(); }
    static double *Data(Vector4dVector &v) { return v.empty() ? nullptr : v.front().data(); }
    static void Check(int rc, const char *what) {
        if (rc != SAGEICP_OK)
            throw std::runtime_error(std::string("sage_icp::VoxelHashMap::") + what + ": " +
                                     sageicp_last_error());
    }
    void swap(VoxelHashMap &o) noexcept {
        std::swap(voxel_size_, o.voxel_size_);
        std::swap(max_distance_, o.max_distance_);
        std::swap(basic_points_per_voxel_, o.basic_points_per_voxel_);
        std::swap(critical_points_per_voxel_, o.critical_points_per_voxel_);
        basic_parts_labels_.swap(o.basic_parts_labels_);
        std::swap(map_, o.map_);
    }

    sageicp_map *map_;
};

}  // namespace sage_icp
