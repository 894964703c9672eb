# the GPU suite on the build with the other association of the 3-term squared norms (oracle built the same way)
SAGE_SQNORM3_ORDER=1 timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_n1.txt 2>&1; tail -3 gpurun_out/gputests_n1.txt
