"""ab_probe.py for the workloads of the launch-per-iteration form (c4 steady, c5 dense)."""
import os, subprocess, sys
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab_probe.py")).read()
CHILD = src.split("CHILD = r'''")[1].split("'''")[0]
for wl in (("c4", "steady", "6"), ("c5", "dense", "20")):
    for rep in range(3):
        for lib in sys.argv[1:]:
            env = dict(os.environ)
            if lib != "product":
                env["SAGEICP_VARIANT_LIB"] = lib
            subprocess.run([sys.executable, "-c", CHILD, *wl], env=env, timeout=900)
