"""Probe: does a small frame need its sort?  (SAGEICP_NO_SORT in a probe build of sort.hip: the caller's order.)"""
import os, subprocess, sys
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab_probe.py")).read()
CHILD = src.split("CHILD = r'''")[1].split("'''")[0]
lib = sys.argv[1]
for wl in (("c1", "cold", "200"), ("c2", "cold", "10")):
    for rep in range(3):
        for nosort in (None, "1"):
            env = dict(os.environ)
            env["SAGEICP_VARIANT_LIB"] = lib
            if nosort:
                env["SAGEICP_NO_SORT"] = "1"
            print("no sort" if nosort else "sorted ", end=" ", flush=True)
            subprocess.run([sys.executable, "-c", CHILD, *wl], env=env, timeout=900)
