cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
(timeout 400 python profiles/knob_probe.py "SAGEICP_LW=1" "SAGEICP_LW=2" "SAGEICP_LW=3" "SAGEICP_LW=4" 2>&1 | grep -v amdgpu.ids > gpurun_out/r02b/knob.txt); cat gpurun_out/r02b/knob.txt
bash profiles/r02/run5.sh
