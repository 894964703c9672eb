# four accumulator copies for small grids: loop tests, then the A/B against eight by frame size
timeout 1200 python -m pytest tests/test_loop_kernel.py tests/test_gpu_parity.py -m gpu -x -q -k "loop or small_frames or c1 or shapes or lanes" 2>&1 | tail -3
mkdir -p gpurun_out/r06
timeout 1200 python profiles/knob_ab.py "c1:cold:1:60 c1:steady:1:60 c2:cold:16:60 c2:cold:8:40 c2:cold:4:30" "SAGEICP_LOOP_COPIES=8" "" 2>&1 | tee gpurun_out/r06/copies_ab.txt
