mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_net.py -m gpu -q -rf 2>&1 | grep -v "^  File\|^Extension" | tail -60 > gpurun_out/r2d/all.log
tail -25 gpurun_out/r2d/all.log
for p in x3; do timeout 600 python tools/run_step.py $p 16 2 > gpurun_out/r2d/step_$p.log 2>&1; tail -2 gpurun_out/r2d/step_$p.log; done
cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2d/prof_x3 -o x3 -- python /root/repo/tools/run_step.py x3 16 2 > /root/repo/gpurun_out/r2d/prof.log 2>&1
ls -la gpurun_out/r2d/prof_x3 | head; python profiles/summarize_rocpd.py $(find gpurun_out/r2d/prof_x3 -name "*results.db" | head -1) 2 2>&1 | head -70 > gpurun_out/r2d/x3_kernel_stats.txt; find gpurun_out/r2d/prof_x3 -name "*.db" -size +60M -delete; head -60 gpurun_out/r2d/x3_kernel_stats.txt
