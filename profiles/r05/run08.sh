mkdir -p gpurun_out
timeout 600 python profiles/resident_probe.py > gpurun_out/r05_run08_resident.txt 2>&1
grep -v "^sageicp" gpurun_out/r05_run08_resident.txt;  grep "^sageicp" gpurun_out/r05_run08_resident.txt | sort | uniq -c
