set -x
cd /root/repo
python bench.py > gpurun_out/r5p_bench.json 2> gpurun_out/r5p_bench.err
python tools/dev/conv_census.py 16 conv_x3_kernel > gpurun_out/r5p_census_b16.txt 2>&1
python tools/dev/conv_census.py 16 halo > gpurun_out/r5p_census_halo_b16.txt 2>&1
python tools/dev/conv_census.py 1 "" > gpurun_out/r5p_census_b1.txt 2>&1
python tools/dev/gemm_lat_bench.py > gpurun_out/r5p_gemm_forms.txt 2>&1
grep '^{' gpurun_out/r5p_bench.json | tail -1 | cut -c1-700
