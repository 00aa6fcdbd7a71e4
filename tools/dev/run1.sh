set -x
cd /root/repo
bash tools/profile_step.sh x3 16 r5p_x3_b16 2>&1 | tail -3
KEEP_AMD_GRAPH=0 KEEP_AMD_OVERLAP_MAX_CLIPS=0 PMC_COUNTERS= bash tools/profile_step.sh x3 1 r5p_x3_b1 2>&1 | tail -3
python bench.py > gpurun_out/r5p_bench.json 2> gpurun_out/r5p_bench.err
python tools/dev/conv_census.py 16 conv_x3_kernel > gpurun_out/r5p_census_b16.txt 2>&1
python tools/dev/conv_census.py 16 halo > gpurun_out/r5p_census_halo_b16.txt 2>&1
python tools/dev/conv_census.py 1 "" > gpurun_out/r5p_census_b1.txt 2>&1
tail -c 600 gpurun_out/r5p_bench.json
