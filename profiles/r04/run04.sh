# k_loop with three / four sets of candidates in flight, and its compact scan for every lanes-per-query variant
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_loop_kernel.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_loop_tests.txt
cat gpurun_out/r04_loop_tests.txt
(echo "== depth 3 (full records)"; timeout 300 python profiles/loop_times.py 8 cold | tail -3
 echo "== depth 4, compact scan"; SAGEICP_FILTER=1 timeout 300 python profiles/loop_times.py 8 cold | tail -3
 echo "== depth 2 (full records)"; LOOP_LIB=sage-icp_amd/_probe/libsageicp_looptiming_d2.so timeout 300 python profiles/loop_times.py 8 cold | tail -3
 echo "== depth 2, compact scan"; SAGEICP_FILTER=1 LOOP_LIB=sage-icp_amd/_probe/libsageicp_looptiming_d2.so timeout 300 python profiles/loop_times.py 8 cold | tail -3
 echo "== depth 3 (full records), 30k steady"; timeout 300 python profiles/loop_times.py 4 steady | tail -3
 echo "== depth 4, compact scan, 30k steady"; SAGEICP_FILTER=1 timeout 300 python profiles/loop_times.py 4 steady | tail -3
) > gpurun_out/r04_loop_times.txt 2>&1
cat gpurun_out/r04_loop_times.txt
timeout 900 python profiles/loop_probe.py quick > gpurun_out/r04_loop_probe.txt 2>&1
cat gpurun_out/r04_loop_probe.txt
SAGEICP_FILTER=1 timeout 900 python profiles/loop_probe.py quick > gpurun_out/r04_loop_probe_compact.txt 2>&1
grep -A4 "15000\|30000\|c1 cold" gpurun_out/r04_loop_probe_compact.txt
