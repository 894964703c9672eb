"""The instrumented builds behind profiles/ (csrc/probes.h: SAGE_NN_TIMING, SAGE_LOOP_TIMING, SAGE_ICP_DELAY_PROBE; the
stamps beside the loops: SAGE_GN_TIMING; the counter-collection twin: SAGE_LOOP_INGRID) must keep compiling: each switch,
device code of kernels.hip only, compiled and thrown away.  CPU only (hipcc cross-compiles gfx950)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_probe_variant_compiles():
    spec = importlib.util.spec_from_file_location("_sageicp_build", os.path.join(ROOT, "sage-icp_amd", "build.py"))
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    failed = bm.compile_probe_variants()
    assert not failed, "\n".join("%s:\n%s" % (d, e[-1500:]) for d, e in failed.items())
