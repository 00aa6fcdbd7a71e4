cd /root/repo
PMC_COUNTERS= bash tools/profile_step.sh x3 16 r5r_x3_b16 2>&1 | grep -E "splitk|total kernel"
grep -E "splitk|total kernel" gpurun_out/r5r_x3_b16/x3_b16_kernel_stats.txt
