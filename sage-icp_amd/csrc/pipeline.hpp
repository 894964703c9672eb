// Per-frame pipeline counterpart: the host-side logic that decides the
// ARGUMENTS of the hot path for a stream of scans — adaptive threshold, constant-velocity guess,
// map update — around the device stages: range crop + two-level semantic voxel down-sampling
// (preprocess.hip, f-3) and the registration itself.  The map lives behind the same C ABI the
// shims use.
//
// Reference (cpp/sage_icp/):
//   pipeline/sageICP.cpp:54-95     sageICP::RegisterFrame           -> Pipeline::register_frame
//   pipeline/sageICP.cpp:97-121    Voxelize / GetAdaptiveThreshold / GetPredictionModel / HasMoved
//   core/Threshold.cpp:29-50       AdaptiveThreshold::ComputeThreshold, ComputeModelError
//   core/Preprocessing.cpp:44-84   VoxelDownsample (first point per voxel, per label group)
//   core/Preprocessing.cpp:173-187 Preprocess, dynamic_vehicle_filter == false branch
// Not reproduced: the PCL Euclidean-clustering "dynamic vehicle filter" (Preprocessing.cpp:95-172;
// PCL is not available and every pre-labelled configuration runs with it off) and deskewing
// (off in every launch file).  The down-sampled clouds come from the backend in the reference's
// emission order (the bucket order of its tsl::robin_map, Preprocessing.cpp:76-82, replayed by
// csrc/robin_order.hpp) unless sageicp_set_downsample_order(0) selected arrival order per label
// group.
#pragma once

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/sageicp.h"
#include "se3_math.h"

namespace sageicp {

struct Pose7 {
    double v[7] = {0, 0, 0, 1, 0, 0, 0};
};

class Pipeline {
public:
    explicit Pipeline(const sageicp_pipeline_config &c)
        : max_range(c.max_range), min_range(c.min_range), label_max_range(c.label_max_range),
          min_motion_th(c.min_motion_th), initial_threshold(c.initial_threshold), sem_th(c.sem_th),
          map_update_on_device(c.map_update_on_device != 0) {
        const int *gl = c.group_labels;
        for (int g = 0; g < c.n_groups; ++g) {
            groups.emplace_back(gl, gl + c.group_label_counts[g]);
            gl += c.group_label_counts[g];
            group_voxel.push_back(c.group_voxel_size[g]);
        }
        map = sageicp_map_create(c.voxel_size_map, c.local_map_range, c.basic_points_per_voxel,
                                 c.critical_points_per_voxel, c.basic_parts_labels,
                                 c.n_basic_parts_labels, c.device);
    }
    ~Pipeline() { sageicp_map_destroy(map); }
    Pipeline(const Pipeline &) = delete;
    Pipeline &operator=(const Pipeline &) = delete;

    bool ok() const { return map != nullptr; }

    // sageICP::reinitialize(), pipeline/sageICP.hpp:94-99
    void reinitialize() {
        poses.clear();
        model_error_sse2 = 0.0;
        num_samples = 0;
        model_deviation = Pose7();
        sageicp_map_clear(map);
    }

    // pipeline/sageICP.cpp:54-95.  The three device stages are supplied by the caller (capi.hip):
    //   be.voxelize(frame, n, n_source)      Preprocess() + Voxelize() (preprocess.hip;
    //                                        core/Preprocessing.cpp:173-187,44-84,
    //                                        pipeline/sageICP.cpp:57-67,97-101); both clouds stay
    //                                        on the device
    //   be.register_source(guess, max_corr, kernel, sem_th, pose, stats)    RegisterFrame(source, ...)
    //   be.update_map(pose)                  local_map_.Update(frame_downsample, pose)
    template <typename Backend>
    int register_frame(const double *frame, uint64_t n, double pose_out[7], double *icp_s,
                       double *total_s, uint64_t *n_source, sageicp_stats *stats, Backend &&be) {
        const auto t_pre = std::chrono::steady_clock::now();
        uint64_t n_src = 0;
        int rc = be.voxelize(frame, n, n_src);
        if (rc) return rc;
        const double sigma = adaptive_threshold();
        Pose7 prediction;                                     // GetPredictionModel()
        const size_t N = poses.size();
        if (N >= 2) {
            Pose7 inv;
            se3_inv(poses[N - 2].v, inv.v);
            se3_mul(inv.v, poses[N - 1].v, prediction.v);
        }
        const Pose7 last = N ? poses.back() : Pose7();
        Pose7 guess;
        se3_mul(last.v, prediction.v, guess.v);

        const auto t_icp = std::chrono::steady_clock::now();
        Pose7 new_pose;
        rc = be.register_source(guess.v, 3.0 * sigma, sigma / 3.0, sem_th, new_pose.v, stats);
        const auto t_end = std::chrono::steady_clock::now();
        if (rc) return rc;

        Pose7 ginv;
        se3_inv(guess.v, ginv.v);
        se3_mul(ginv.v, new_pose.v, model_deviation.v);       // UpdateModelDeviation
        rc = be.update_map(new_pose.v);
        if (rc) return rc;
        poses.push_back(new_pose);
        for (int i = 0; i < 7; ++i) pose_out[i] = new_pose.v[i];
        if (icp_s) *icp_s = std::chrono::duration<double>(t_end - t_icp).count();
        if (total_s) *total_s = std::chrono::duration<double>(t_end - t_pre).count();
        if (n_source) *n_source = n_src;
        return SAGEICP_OK;
    }

    std::vector<Pose7> poses;
    sageicp_map *map = nullptr;

    // label groups as flat tables for the device kernels
    void group_tables(std::vector<int> &counts, std::vector<int> &labels, std::vector<double> &vs) const {
        counts.clear(); labels.clear(); vs = group_voxel;
        for (const auto &g : groups) {
            counts.push_back(static_cast<int>(g.size()));
            labels.insert(labels.end(), g.begin(), g.end());
        }
    }
    bool map_update_on_device_() const { return map_update_on_device; }
    double max_range_() const { return max_range; }
    double min_range_() const { return min_range; }
    double label_max_range_() const { return label_max_range; }

private:
    // pipeline/sageICP.cpp:103-108,117-121 + core/Threshold.cpp:29-50
    double adaptive_threshold() {
        bool moved = false;
        if (!poses.empty()) {
            Pose7 inv, d;
            se3_inv(poses.front().v, inv.v);
            se3_mul(inv.v, poses.back().v, d.v);
            const double motion = std::sqrt(d.v[4] * d.v[4] + d.v[5] * d.v[5] + d.v[6] * d.v[6]);
            moved = motion > 5.0 * min_motion_th;
        }
        if (!moved) return initial_threshold;
        const double *q = model_deviation.v;
        // Eigen::AngleAxisd(R).angle(): 2 atan2(|q.vec|, |q.w|)
        const double theta = 2.0 * std::atan2(std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]),
                                              std::fabs(q[3]));
        const double delta_rot = 2.0 * max_range * std::sin(theta / 2.0);
        const double delta_trans = std::sqrt(q[4] * q[4] + q[5] * q[5] + q[6] * q[6]);
        const double model_error = delta_trans + delta_rot;
        if (model_error > min_motion_th) {
            model_error_sse2 += model_error * model_error;
            ++num_samples;
        }
        if (num_samples < 1) return initial_threshold;
        return std::sqrt(model_error_sse2 / num_samples);
    }

    double max_range, min_range, label_max_range;
    double min_motion_th, initial_threshold, sem_th;
    bool map_update_on_device;
    std::vector<std::vector<int>> groups;
    std::vector<double> group_voxel;
    double model_error_sse2 = 0.0;
    int num_samples = 0;
    Pose7 model_deviation;
};

}  // namespace sageicp
