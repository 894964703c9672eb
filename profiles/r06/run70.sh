mkdir -p gpurun_out/r06
timeout 1500 python profiles/unit_deal_probe.py 2>&1 | tee gpurun_out/r06/unit_deal_probe.txt
