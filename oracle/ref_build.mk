# oracle/_ref: the reference's OWN hot path, compiled from the sources where they lie under /root/reference —
#   core/Registration.cpp, core/VoxelHashMap.cpp, unmodified, plain g++ -O3 (the reference's flags: core/CMakeLists.txt:28),
#   not the reference's CMake — plus oracle/ref_driver.cpp (a C interface around them, ours).
# It needs Eigen 3.4, Sophus 1.22, oneTBB and tsl::robin_map (3rdparty/*/*.cmake fetch them from the network); NONE of
# them is in the build image, so this recipe is DORMANT there: tests/test_reference_build.py probes for the four headers
# and skips with that reason.  No stand-in headers are written to force a build.  On a machine that has them:
#     make -f oracle/ref_build.mk REF_INCLUDES="-I/usr/include/eigen3 -I/path/to/sophus -I/path/to/robin-map/include"
# builds oracle/_ref/libsage_ref.so (git-ignored; travels to the GPU box with the snapshot like every built .so), and the
# test then runs the golden scenes through it and through oracle/libsage_oracle.so: correspondences bit for bit, poses to
# 1e-12 — the one route from "parity unpinned" to pinned.
REF ?= /root/reference/cpp/sage_icp
CXX ?= g++
REF_INCLUDES ?=
REF_LIBS ?= -ltbb
OUT := $(dir $(lastword $(MAKEFILE_LIST)))_ref

$(OUT)/libsage_ref.so: $(REF)/core/Registration.cpp $(REF)/core/VoxelHashMap.cpp $(dir $(lastword $(MAKEFILE_LIST)))ref_driver.cpp
	mkdir -p $(OUT)
	$(CXX) -O3 -std=c++17 -fPIC -shared -I$(REF)/.. -I$(REF) $(REF_INCLUDES) -o $@ $^ $(REF_LIBS)

probe:
	@printf '#include <Eigen/Core>\n#include <sophus/se3.hpp>\n#include <tsl/robin_map.h>\n#include <tbb/parallel_reduce.h>\nint main(){return 0;}\n' | \
	    $(CXX) -std=c++17 -fsyntax-only $(REF_INCLUDES) -x c++ - 2>&1 | head -3
