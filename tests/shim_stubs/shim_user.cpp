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
        // known answer: a lattice map, the scan = a subset of its points moved by -t, so exact
        // correspondences exist and the registration must return the planted translation
        {
            VoxelHashMap m(1.0, 100.0, 20, 20, {40, 44, 48, 49, 50, 70, 72});
            std::vector<Eigen::Vector4d> lattice, scan;
            const double t[3] = {0.05, -0.03, 0.02};
            for (int i = 0; i < 24; ++i)
                for (int j = 0; j < 24; ++j)
                    for (int k = 0; k < 6; ++k) {
                        Eigen::Vector4d p;
                        p[0] = 0.43 * i + 0.011 * ((i * 7 + j * 3 + k) % 5) - 5.0;
                        p[1] = 0.39 * j + 0.013 * ((i + j * 5 + k * 2) % 7) - 4.5;
                        p[2] = 0.47 * k + 0.009 * ((i * 2 + j + k * 3) % 3) - 1.0;
                        p[3] = 40 + (i + j) % 2 * 10;
                        lattice.push_back(p);
                        if ((i + 2 * j + 3 * k) % 3 == 0) {
                            Eigen::Vector4d s = p;
                            s[0] -= t[0]; s[1] -= t[1]; s[2] -= t[2];
                            scan.push_back(s);
                        }
                    }
            m.AddPoints(lattice);
            Sophus::SE3d identity;
            const Sophus::SE3d est = sage_icp::RegisterFrame(scan, m, identity, 1.0, 0.1, 0.4);
            const double *d = est.data();         // (qx, qy, qz, qw, tx, ty, tz)
            // the loop stops once a step is shorter than 1e-4 (Registration.cpp:97,137), which
            // leaves the pose within millimetres of the planted one
            bool ok = d[3] > 0.999999;
            for (int a = 0; a < 3; ++a) {
                const double dt = d[4 + a] - t[a];
                ok = ok && dt < 3e-3 && dt > -3e-3 && d[a] < 2e-4 && d[a] > -2e-4;
            }
            if (!ok) {
                std::printf("pose %g %g %g %g | %g %g %g\n", d[0], d[1], d[2], d[3], d[4], d[5], d[6]);
                return 8;
            }
            std::puts("planted pose recovered");
        }
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
