cd /root/repo
bash tools/profile_step.sh x3 16 r4p_x3_b16 > gpurun_out/prof_final.log 2>&1
rm -rf gpurun_out/r4p_x3_b16/trace_x3_b16 gpurun_out/r4p_x3_b16/pmc_x3_b16
du -sh gpurun_out/r4p_x3_b16; head -8 gpurun_out/r4p_x3_b16/x3_b16_kernel_stats.txt | cut -c1-160
