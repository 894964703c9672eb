"""Build k_nn variants (-D flags) into gpurun_out/ and time c2 / c1 frames with each, one process
per variant (a second copy of the library in one process would bind to the first one's symbols)."""
import os, sys, subprocess, time
sys.path.insert(0, os.getcwd())
if "VARIANT_LIB" in os.environ:
    import sage_icp_amd as sage
    from sage_icp_amd import synthetic as syn
    sage.LIB_PATH = os.environ["VARIANT_LIB"]
    res = []
    for name in ("c2", "c1"):
        w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
        f = sage.Frame(w["map"], w["scan"]); p = syn.PARAMS["cold"]
        for _ in range(3): sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
        t = time.perf_counter(); N = 20
        for _ in range(N): pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
        res.append("%s %.3f ms (%d it)" % (name, 1e3 * (time.perf_counter() - t) / N, st.iterations))
    print(os.environ.get("VARIANT_FLAGS") or "(default)", "->", "; ".join(res), flush=True)
    sys.exit(0)
src = ["sage-icp_amd/csrc/%s.hip" % n for n in ("kernels", "sort", "preprocess", "map_update", "capi")]
os.makedirs("gpurun_out", exist_ok=True)
for k, flags in enumerate(sys.argv[1:] or [""]):
    out = os.path.abspath("gpurun_out/libsageicp_var%d.so" % k)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-ffp-contract=off"] + flags.split() + src + ["-o", out, "-ldl"])
    subprocess.call([sys.executable, __file__], env=dict(os.environ, VARIANT_LIB=out, VARIANT_FLAGS=flags))
