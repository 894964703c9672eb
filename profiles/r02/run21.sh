# lanes per query revisited with the current kernel: c2 (120k / 15k) and c4 (500k queries, multi-round)
for lw in 1 2 3; do SAGEICP_LW=$lw KNOB_CHILD="SAGEICP_LW=$lw" python profiles/knob_probe.py; done
for lw in 1 2 3; do echo "c4 LW=$lw"; SAGEICP_LW=$lw python bench.py --workload c4 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['iterations_per_frame'], d['roofline']['avg_launch_us'])"; done
