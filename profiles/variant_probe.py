"""Build library variants (-D flags) into gpurun_out/ and time c2 frames (full and a 1/8 share)
with each, optionally under environment knobs; one process per variant.
usage: python profiles/variant_probe.py "<flags> [| ENV=.. ENV=..]" ...      ("" = the default build)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = [os.path.join(ROOT, "sage-icp_amd/csrc/%s.hip" % n) for n in ("kernels", "sort", "preprocess", "map_update", "capi")]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
built = {}
for k, spec in enumerate(sys.argv[1:] or [""]):
    flags, _, envs = spec.partition("|")
    flags = flags.strip()
    if flags not in built:
        out = os.path.join(ROOT, "gpurun_out/libsageicp_var%d.so" % len(built))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=off"] + flags.split() + src + ["-o", out, "-ldl", "-lpthread"])
        built[flags] = out
    env = dict(os.environ, KNOB_CHILD=spec.strip() or "(default)", KNOB_LIB=built[flags])
    for kv in envs.split():
        a, b = kv.split("=", 1)
        env[a] = b
    subprocess.call([sys.executable, os.path.join(ROOT, "profiles/knob_probe.py")], env=env, cwd=ROOT)
