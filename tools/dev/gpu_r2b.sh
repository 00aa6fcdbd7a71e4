#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r2b_tests.log
timeout 900 python bench.py > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err
tail -c 3000 gpurun_out/r2b_tests.log
cat gpurun_out/r2b_bench.json
