// Per-frame pipeline counterpart: the host-side logic that decides the
// ARGUMENTS of the hot path for a stream of scans — range crop, two-level semantic voxel
// down-sampling, adaptive threshold, constant-velocity guess, map update.  Plain C++ on the
// host (it is scalar bookkeeping plus O(N) hash inserts once per frame); the registration and
// the map live behind the same C ABI the shims use.
//
// Reference (cpp/sage_icp/):
//   pipeline/sageICP.cpp:54-95     sageICP::RegisterFrame           -> Pipeline::register_frame
//   pipeline/sageICP.cpp:97-121    Voxelize / GetAdaptiveThreshold / GetPredictionModel / HasMoved
//   core/Threshold.cpp:29-50       AdaptiveThreshold::ComputeThreshold, ComputeModelError
//   core/Preprocessing.cpp:44-84   VoxelDownsample (first point per voxel, per label group)
//   core/Preprocessing.cpp:173-187 Preprocess, dynamic_vehicle_filter == false branch
// Not reproduced: the PCL Euclidean-clustering "dynamic vehicle filter" (Preprocessing.cpp:95-172;
// PCL is not available and every pre-labelled configuration runs with it off) and deskewing
// (off in every launch file).  Order of the down-sampled points is insertion order per label
// group; the reference emits tsl::robin_map bucket order (deviation D3 in DESIGN.md).
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
        : voxel_size_map(c.voxel_size_map), max_range(c.max_range), min_range(c.min_range),
          label_max_range(c.label_max_range), local_map_range(c.local_map_range),
          min_motion_th(c.min_motion_th), initial_threshold(c.initial_threshold), sem_th(c.sem_th) {
        const int *gl = c.group_labels;
        for (int g = 0; g < c.n_groups; ++g) {
            groups.emplace_back(gl, gl + c.group_label_counts[g]);
            gl += c.group_label_counts[g];
            group_voxel.push_back(c.group_voxel_size[g]);
        }
        for (int l = 0; l < 256; ++l) {
            lut[l] = -1;
            for (size_t g = 0; g < groups.size() && lut[l] < 0; ++g)
                if (std::find(groups[g].begin(), groups[g].end(), l) != groups[g].end())
                    lut[l] = static_cast<int>(g);
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

    // pipeline/sageICP.cpp:54-95
    int register_frame(const double *frame, uint64_t n, double pose_out[7], double *icp_s,
                       double *total_s, uint64_t *n_source, sageicp_stats *stats) {
        const auto t_pre = std::chrono::steady_clock::now();
        std::vector<double> cropped;
        preprocess(frame, n, cropped);
        std::vector<double> frame_downsample, source;
        voxel_downsample(cropped, 0.5, frame_downsample);     // Voxelize(), sageICP.cpp:97-101
        voxel_downsample(frame_downsample, 1.5, source);
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
        int rc = sageicp_register_frame(map, source.data(), source.size() / 4, guess.v, 3.0 * sigma,
                                        sigma / 3.0, sem_th, new_pose.v, stats);
        const auto t_end = std::chrono::steady_clock::now();
        if (rc) return rc;

        Pose7 ginv;
        se3_inv(guess.v, ginv.v);
        se3_mul(ginv.v, new_pose.v, model_deviation.v);       // UpdateModelDeviation
        rc = sageicp_map_update_pose(map, frame_downsample.data(), frame_downsample.size() / 4,
                                     new_pose.v);
        if (rc) return rc;
        poses.push_back(new_pose);
        for (int i = 0; i < 7; ++i) pose_out[i] = new_pose.v[i];
        if (icp_s) *icp_s = std::chrono::duration<double>(t_end - t_icp).count();
        if (total_s) *total_s = std::chrono::duration<double>(t_end - t_pre).count();
        if (n_source) *n_source = source.size() / 4;
        return SAGEICP_OK;
    }

    std::vector<Pose7> poses;
    sageicp_map *map = nullptr;

private:
    // core/Preprocessing.cpp:173-187 (dynamic_vehicle_filter == false)
    void preprocess(const double *f, uint64_t n, std::vector<double> &out) const {
        out.reserve(4 * n);
        for (uint64_t i = 0; i < n; ++i) {
            const double *p = f + 4 * i;
            const double norm = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
            if (norm < max_range && norm > min_range) {
                out.insert(out.end(), {p[0], p[1], p[2], norm > label_max_range ? 0.0 : p[3]});
            }
        }
    }

    // label -> first group that lists it (Preprocessing.cpp:57-64), -1 if none
    int group_of(int label) const {
        if (label >= 0 && label < 256) return lut[label];
        for (size_t g = 0; g < groups.size(); ++g)
            if (std::find(groups[g].begin(), groups[g].end(), label) != groups[g].end())
                return static_cast<int>(g);
        return -1;
    }

    // core/Preprocessing.cpp:44-84: first point per voxel wins, one grid per label group.
    // The grids are one flat open-addressed set keyed by (group, voxel): membership is all that
    // is needed, and the kept points are emitted group by group in insertion order.
    void voxel_downsample(const std::vector<double> &in, double scale, std::vector<double> &out) const {
        const size_t G = groups.size();
        const size_t n = in.size() / 4;
        size_t capacity = 64;
        while (capacity < 2 * n + 16) capacity <<= 1;
        struct Key { int g, x, y, z; };
        std::vector<Key> set(capacity, Key{-1, 0, 0, 0});
        const size_t mask = capacity - 1;
        std::vector<std::vector<double>> kept(G);
        for (size_t g = 0; g < G; ++g) kept[g].reserve(in.size() / (G ? G : 1) + 64);
        for (size_t i = 0; i < n; ++i) {
            const double *p = &in[4 * i];
            const int group = group_of(static_cast<int>(p[3]));
            if (group < 0) continue;
            const double vs = group_voxel[group] * scale;
            const Key key{group, static_cast<int>(p[0] / vs), static_cast<int>(p[1] / vs),
                          static_cast<int>(p[2] / vs)};
            size_t s = (voxel_hash(key.x, key.y, key.z) + 0x9E3779B9u * static_cast<uint32_t>(group)) & mask;
            bool present = false;
            for (;;) {
                const Key &e = set[s];
                if (e.g < 0) break;
                if (e.g == key.g && e.x == key.x && e.y == key.y && e.z == key.z) { present = true; break; }
                s = (s + 1) & mask;
            }
            if (present) continue;
            set[s] = key;
            kept[group].insert(kept[group].end(), p, p + 4);
        }
        out.clear();
        out.reserve(in.size());
        for (size_t g = 0; g < G; ++g) out.insert(out.end(), kept[g].begin(), kept[g].end());
    }

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

    double voxel_size_map, max_range, min_range, label_max_range, local_map_range;
    double min_motion_th, initial_threshold, sem_th;
    std::vector<std::vector<int>> groups;
    std::vector<double> group_voxel;
    int lut[256];
    double model_error_sse2 = 0.0;
    int num_samples = 0;
    Pose7 model_deviation;
};

}  // namespace sageicp
