cd /root/repo
bash tools/profile_step.sh x3 48 r5p_x3_b48 2>&1 | tail -3
python tools/dev/conv_census.py 48 halo > gpurun_out/r5p_census_halo_b48.txt 2>&1
python tools/dev/conv_census.py 48 conv_x3_kernel > gpurun_out/r5p_census_b48.txt 2>&1
ls gpurun_out/r5p_x3_b48
