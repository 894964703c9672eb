"""Generates tests/golden/c4_full.npz — the oracle's FULL registration of BASELINE configs[3]
(500,000-pt scan vs 10,000,000-pt map, 1 m voxels) with the steady and the cold parameter points.

The oracle needs minutes of CPU for this frame, so it is run once, here, in the build container,
and the GPU test (tests/test_gpu_parity.py::test_c4_full_size_properties) and bench.py's `parity`
block compare against the committed numbers instead of re-running it on the GPU box.

Data only: the seeded generator's parameters, and per parameter point the oracle's pose, iteration
count, first / last correspondence count, the exact sum of C_q over all iterations and the
correspondence count of every iteration.  Like tests/golden/make_golden.py these are pins made by
the repo's own CPU oracle (oracle/sage_oracle.cpp), NOT reference output (parity unpinned: the
reference holds no vectors for this path and cannot be built in this image).

Run from the repo root:  python tests/golden/make_c4_golden.py            (about 10 minutes on 8 cores)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import sage_icp_amd  # noqa: E402,F401  (the package of the generator; no device is touched)
from sage_icp_amd import synthetic as syn  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c4_full.npz")


class _OracleAsMap:
    """what synthetic.build_map_points asks of a map: AddPoints() and size()"""

    def __init__(self, voxel):
        self.m = oracle.Map(voxel, 100.0)

    def AddPoints(self, pts):
        self.m.add_points(pts)

    def size(self):
        return self.m.size()


def main(scale=1.0):
    name = "c4"
    t0 = time.time()
    w = syn.make_workload(name, lambda: _OracleAsMap(syn.WORKLOADS[name]["voxel"]), scale=scale)
    om = w["map"].m
    print("map %d pts in %d voxels, scan %d pts (%.0f s)" % (om.size(), om.num_voxels(), len(w["scan"]), time.time() - t0))
    out = dict(workload=np.array([syn.WORKLOADS[name]["seed"], syn.WORKLOADS[name]["map_points"] * scale,
                                  syn.WORKLOADS[name]["scan"] * scale, syn.WORKLOADS[name]["voxel"], scale]),
               map_size=np.array([om.size(), om.num_voxels()]),
               scan_checksum=np.array([float(np.sum(w["scan"][:, :3])), float(np.sum(w["scan"][:, 3]))]),
               T_gt=w["T_gt"])
    for params in ("steady", "cold"):
        p = syn.PARAMS[params]
        t1 = time.time()
        pose, st = om.register_frame(w["scan"], oracle.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
        print("%s: %d iterations, converged %d, n_corr %d -> %d, sum C_q %d, %.0f s" %
              (params, st.iterations, st.converged, st.n_corr_first, st.n_corr_last, st.sum_candidates_total,
               time.time() - t1))
        out[params + "_params"] = np.array([p["max_dist"], p["kernel"], p["sem_th"]])
        out[params + "_pose"] = pose
        out[params + "_counts"] = np.array([st.iterations, st.converged, st.n_corr_first, st.n_corr_last,
                                            st.sum_candidates_total, st.sum_corr_total], dtype=np.int64)
        out[params + "_last_step_norm"] = np.array([st.last_step_norm])
    np.savez_compressed(OUT if scale == 1.0 else OUT.replace(".npz", "_scale%g.npz" % scale), **out)
    print("wrote", OUT, "(%.0f s in all)" % (time.time() - t0))


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
