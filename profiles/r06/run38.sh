# full GPU suite + the driver's bench command on the tree of commit 4ac5ffa
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_run38.txt 2>&1; tail -5 gpurun_out/gputests_run38.txt
timeout 600 python bench.py > gpurun_out/bench_run38.json 2> gpurun_out/bench_run38.err; tail -c 3000 gpurun_out/bench_run38.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
