mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_rank_loses or direct_exchange or in_process or sharded or rccl or comm" 2>&1 | tail -25
