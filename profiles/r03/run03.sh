# r03 first full pass on the new code: GPU suite (incl. c2-cold / c5 full-size parity, device Pointcloud, the sqnorm3
# variant subset), rocprofv3 evidence for c2-cold and c4-steady, stream with LocalMap() every frame
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
bash profiles/run_profiles.sh r03 c2 cold > gpurun_out/prof_c2.log 2>&1; tail -40 gpurun_out/prof_c2.log
bash profiles/run_profiles.sh r03 c4 steady > gpurun_out/prof_c4.log 2>&1; tail -30 gpurun_out/prof_c4.log
STREAM_LOCALMAP=1 timeout 300 python profiles/stream_probe.py > gpurun_out/stream_localmap.txt 2>&1; cat gpurun_out/stream_localmap.txt
timeout 300 python profiles/stream_probe.py > gpurun_out/stream_plain.txt 2>&1; cat gpurun_out/stream_plain.txt
