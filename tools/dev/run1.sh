cd /root/repo
export TMPDIR=/tmp
bash tools/profile_step.sh x3 48 r6p_x3_b48 > gpurun_out/r6p_step48.log 2>&1
cp gpurun_out/r6p_x3_b48/x3_b48_pmc.json profiles/r06_pmc_traffic.json
python bench.py > gpurun_out/r6p_bench.json 2> gpurun_out/r6p_bench.err
KEEP_AMD_GRAPH=0 KEEP_AMD_OVERLAP_MAX_CLIPS=0 PMC_COUNTERS= bash tools/profile_step.sh x3 1 r6p_x3_b1 > gpurun_out/r6p_step1.log 2>&1
python tools/dev/conv_census.py 48 conv_x3 > gpurun_out/r6p_census_b48.txt 2>/dev/null
python tools/dev/conv_census.py 48 halo > gpurun_out/r6p_census_halo_b48.txt 2>/dev/null
python tools/dev/conv_census.py 1 '' > gpurun_out/r6p_census_b1.txt 2>/dev/null
bash tools/dev/pmc_shapes.sh r6p_pmc_shapes > gpurun_out/r6p_pmc_shapes.txt 2>&1
python tools/dev/x3_ceiling.py > gpurun_out/r6p_ceiling.txt 2>/dev/null
tail -c 1500 gpurun_out/r6p_bench.json
