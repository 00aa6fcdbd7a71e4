prune() { find gpurun_out -name "*.db" -delete; find gpurun_out -name "*.csv" -size +2M -delete; find gpurun_out -name "*.json" -size +8M -delete; }
bash tools/profile_step.sh x3 16 r5p_x3_b16 > gpurun_out/r5p_prof16.log 2>&1; tail -3 gpurun_out/r5p_prof16.log; prune
KEEP_AMD_GRAPH=0 KEEP_AMD_OVERLAP_MAX_CLIPS=0 PMC_COUNTERS= bash tools/profile_step.sh x3 1 r5p_x3_b1 > gpurun_out/r5p_prof1.log 2>&1; tail -3 gpurun_out/r5p_prof1.log; prune
python tools/dev/conv_census.py 16 conv_x3_kernel > gpurun_out/r5p_census_b16.txt 2>&1
python tools/dev/conv_census.py 16 halo > gpurun_out/r5p_census_halo_b16.txt 2>&1
python bench.py > gpurun_out/r5p_bench.json 2> gpurun_out/r5p_bench.err; tail -c 300 gpurun_out/r5p_bench.json; grep -v "it/s\]" gpurun_out/r5p_bench.err | tail -5
du -sh gpurun_out
