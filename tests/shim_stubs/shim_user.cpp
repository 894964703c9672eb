// Exercises the header shim the way pipeline/sageICP.{hpp,cpp} and ros/ros2/OdometryServer.cpp use
// the reference headers: construction, copy assignment, Update, Clear/Empty, Pointcloud,
// GetCorrespondences, RegisterFrame, TransformPoints, public data members.
#include <cstdio>
#include <vector>

#include "sage_icp/core/Preprocessing.hpp"
#include "sage_icp/core/Registration.hpp"
#include "sage_icp/core/VoxelHashMap.hpp"

int main() {
    using sage_icp::VoxelHashMap;
    VoxelHashMap a(1.0, 100.0, 20, 20, {40, 44, 48, 49, 50, 70, 72});
    std::vector<Eigen::Vector4d> pts(100);
    for (int i = 0; i < 100; ++i) { pts[i][0] = 0.1 * i; pts[i][1] = 0.05 * i; pts[i][2] = 0.2; pts[i][3] = 40; }
    a.AddPoints(pts);
    Eigen::Vector3d origin;
    a.Update(pts, origin);
    Sophus::SE3d T;
    if (sageicp_device_count() > 0) a.Update(pts, T);     // the per-frame update runs on the GPU
    VoxelHashMap b(0.5, 50.0, 1, 1, {});
    b = a;                                   // OdometryServer.cpp:104 copy-assigns the pipeline
    VoxelHashMap c = std::move(b);
    if (c.Empty() || c.Pointcloud().size() != a.Pointcloud().size()) return 1;
    if (c.voxel_size_ != 1.0 || c.basic_points_per_voxel_ != 20 || c.basic_parts_labels_.size() != 7) return 2;
    a.Clear();
    if (!a.Empty() || c.Empty()) return 3;
    // empty map: RegisterFrame returns the initial guess without touching the device
    Sophus::SE3d guess;
    guess.data()[4] = 1.5;
    Sophus::SE3d out = sage_icp::RegisterFrame(pts, a, guess, 6.0, 0.6, 0.4);
    if (out.data()[4] != 1.5) return 4;
    if (sageicp_device_count() > 0) {
        auto [src, tgt] = c.GetCorrespondences(pts, 1.0, 0.4);
        if (src.size() != tgt.size() || src.empty()) return 5;
        Sophus::SE3d pose = sage_icp::RegisterFrame(pts, c, guess, 6.0, 0.6, 0.4);
        (void)pose;
        sage_icp::TransformPoints(T, pts);
        // Voxelize() as pipeline/sageICP.cpp:97-101 calls it
        const std::vector<std::vector<int>> labels = {{40, 44}, {50}};
        const std::vector<double> sizes = {0.5, 1.0};
        auto down = sage_icp::VoxelDownsample(pts, labels, sizes, 0.5);
        if (down.empty() || down.size() > pts.size()) return 6;
        auto crop = sage_icp::Preprocess(pts, 100.0, 0.05, 50.0, false, 0.5, {10}, {44, 48});
        if (crop.empty() || crop.size() > pts.size()) return 7;
    }
    std::puts("shim ok");
    return 0;
}
