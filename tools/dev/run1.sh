prune() { find gpurun_out/$1 -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + ; find gpurun_out/$1 -name "*.log" -size +1M -delete; }
bash tools/profile_step.sh x3 16 r4p_x3_b16 > gpurun_out/r4p_x3_b16.out 2>&1; prune r4p_x3_b16
KEEP_AMD_GRAPH=0 bash tools/profile_step.sh x3 1 r4p_x3_b1 > gpurun_out/r4p_x3_b1.out 2>&1; prune r4p_x3_b1
bash tools/profile_step.sh fp32 16 r4p_fp32_b16 > gpurun_out/r4p_fp32_b16.out 2>&1; prune r4p_fp32_b16
bash tools/dev/pmc_conv.sh r4p_pmc_sq c64_512 c128_256 > gpurun_out/r4p_pmc_sq.txt 2>&1; rm -rf gpurun_out/r4p_pmc_sq
KEEP_X3_NO_STREAM=1 bash tools/dev/pmc_conv.sh r4p_pmc_sq_r3kernel c64_512 c128_256 > gpurun_out/r4p_pmc_sq_r3kernel.txt 2>&1; rm -rf gpurun_out/r4p_pmc_sq_r3kernel
du -sh gpurun_out; ls gpurun_out/r4p_x3_b16
head -12 gpurun_out/r4p_x3_b16/x3_b16_kernel_stats.txt | cut -c1-150
