#!/bin/bash
# timeline of the one-launch loop on the WHOLE c2 frame (two lanes per query: 3750 waves, every slot of the launch used)
mkdir -p gpurun_out
( SAGEICP_LW=1 timeout 600 python profiles/loop_times.py 1 cold
  echo ----; SAGEICP_LW=1 timeout 600 python profiles/loop_times.py 1 steady
  echo ----; timeout 600 python profiles/loop_times.py 2 cold ) > gpurun_out/r04_loop_times_c2full.txt 2>&1
tail -80 gpurun_out/r04_loop_times_c2full.txt
