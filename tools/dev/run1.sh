cd /root/repo
KEEP_DIST_DEVICE=0 timeout 1500 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r5p_bench_2ranks.json 2> gpurun_out/r5p_bench_2ranks.err; echo rc=$?
grep '^{' gpurun_out/r5p_bench_2ranks.json | tail -1 | cut -c1-900
tail -3 gpurun_out/r5p_bench_2ranks.err
