timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "property" --hypothesis-show-statistics 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -30
