cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^$" | cut -c1-300 | tail -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/bench_r4c.json 2> gpurun_out/bench_r4c.err; tail -c 300 gpurun_out/bench_r4c.json; tail -3 gpurun_out/bench_r4c.err
