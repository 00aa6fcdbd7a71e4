cd /root/repo
bash tools/profile_step.sh x3 48 r5p_x3_b48 > /dev/null 2>&1
python tools/dev/conv_census.py 48 conv_x3_kernel > gpurun_out/r5p_census_b48.txt 2>&1
python tools/dev/conv_census.py 48 halo > gpurun_out/r5p_census_halo_b48.txt 2>&1
head -8 gpurun_out/r5p_x3_b48/x3_b48_kernel_stats.txt | cut -c1-150
head -3 gpurun_out/r5p_census_halo_b48.txt | cut -c1-150
cat gpurun_out/r5p_x3_b48/x3_b48_pmc.json | head -c 600
find gpurun_out -name "*.db" -delete; du -sh gpurun_out
