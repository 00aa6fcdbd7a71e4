cd /root/repo
python bench.py > gpurun_out/bench_r4e.json 2> gpurun_out/bench_r4e.err; tail -c 200 gpurun_out/bench_r4e.json
