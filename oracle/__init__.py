"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see oracle/sage_oracle.cpp header).
Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from .oracle import (IDENTITY, SQNORM3_ORDER, Map, build_variants, Pipeline, Stats, align_clouds, build, ldlt_solve6, lib, num_threads,  # noqa: F401
                     robin_order_of, se3_apply, se3_exp, se3_inv, se3_log, se3_mul, set_robin_order,
                     transform_points, voxel_downsample)
