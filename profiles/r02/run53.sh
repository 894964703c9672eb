# pipeline prefetch: parity + stream rate with / without
python -m pytest tests/test_pipeline.py tests/test_kitti_io.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -6
for pf in 0 1; do STREAM_PREFETCH=$pf python profiles/stream_probe.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; done > gpurun_out/stream_prefetch.txt 2>&1
cat gpurun_out/stream_prefetch.txt
python tools/run_sequence.py --synthetic 30 --out /tmp/t.tum 2>&1 | tail -3
