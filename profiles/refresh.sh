#!/bin/bash
# Round 5 (the bench's default workload is 48 clips per call since the second half of the round: files *_b48*; the 16-clip files stay for continuity).
# Copy the summaries tools/profile_step.sh wrote under gpurun_out/ (on the GPU box, merged back by gpurun) into the
# tracked profiles/ directory.  Each source file is the unedited output of the command quoted in its header.
#   on the GPU box:   bash tools/profile_step.sh x3 16 r5p_x3_b16 ; KEEP_AMD_GRAPH=0 KEEP_AMD_OVERLAP_MAX_CLIPS=0 PMC_COUNTERS= bash tools/profile_step.sh x3 1 r5p_x3_b1 ;
#                     python bench.py > gpurun_out/r5p_bench.json ; python tools/dev/conv_census.py 16 conv > gpurun_out/r5p_census_b16.txt
#   here:             bash profiles/refresh.sh
set -e
cd "$(dirname "$0")/.."
G=gpurun_out
hdr() { echo "# rocprofv3 --kernel-trace --stats -- python tools/run_step.py $1 $2 2   ($1 policy, $2 clip(s) x T=20 per pass, 2 passes; pass 1 includes first-touch allocation and the x3 weight split; $3)"; }
[ -f $G/r5p_x3_b16/x3_b16_kernel_stats.txt ] && { hdr x3 16 "round-5 kernels"; cat $G/r5p_x3_b16/x3_b16_kernel_stats.txt; } > profiles/r05_x3_b16_kernel_stats.txt
[ -f $G/r5p_x3_b1/x3_b1_kernel_stats.txt ] && { hdr x3 1 "KEEP_AMD_GRAPH=0 KEEP_AMD_OVERLAP_MAX_CLIPS=0: eager launches on one stream, so every kernel is a trace record"; cat $G/r5p_x3_b1/x3_b1_kernel_stats.txt; } > profiles/r05_x3_b1_kernel_stats.txt
if [ -f $G/r5p_x3_b16/x3_b16_pmc.json ]; then
  cp $G/r5p_x3_b16/x3_b16_pmc.json profiles/r05_pmc_traffic_b16.json
  { echo "# rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace --output-format csv -- python tools/run_step.py x3 16 1"; echo "# KiB per launch as reported; FETCH_SIZE is doubled on gfx950 in r05_pmc_traffic.json (MI355X_MICROARCH.md, HBM section)"; cat $G/r5p_x3_b16/x3_b16_pmc.txt; } > profiles/r05_pmc_step_x3_b16.txt
fi
[ -f $G/r5p_bench.json ] && grep '^{' $G/r5p_bench.json | tail -1 > profiles/r05_bench_default.json
[ -f $G/r5p_x3_b48/x3_b48_kernel_stats.txt ] && { hdr x3 48 "the bench's default workload"; cat $G/r5p_x3_b48/x3_b48_kernel_stats.txt; } > profiles/r05_x3_b48_kernel_stats.txt
if [ -f $G/r5p_x3_b48/x3_b48_pmc.json ]; then
  cp $G/r5p_x3_b48/x3_b48_pmc.json profiles/r05_pmc_traffic.json
  { echo "# rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace --output-format csv -- python tools/run_step.py x3 48 1"; echo "# KiB per launch as reported; FETCH_SIZE is doubled on gfx950 in r05_pmc_traffic.json (MI355X_MICROARCH.md, HBM section)"; cat $G/r5p_x3_b48/x3_b48_pmc.txt; } > profiles/r05_pmc_step_x3_b48.txt
fi
[ -f $G/r5p_census_b48.txt ] && grep -v "^/opt" $G/r5p_census_b48.txt > profiles/r05_conv_census_x3_b48.txt
[ -f $G/r5p_census_halo_b48.txt ] && grep -v "^/opt" $G/r5p_census_halo_b48.txt > profiles/r05_conv_census_halo_x3_b48.txt
[ -f $G/r5p_census_b16.txt ] && grep -v "^/opt" $G/r5p_census_b16.txt > profiles/r05_conv_census_x3_b16.txt
[ -f $G/r5p_census_halo_b16.txt ] && grep -v "^/opt" $G/r5p_census_halo_b16.txt > profiles/r05_conv_census_halo_x3_b16.txt
[ -f $G/r5p_census_b1.txt ] && { echo "# kernel names are the PLAN family's (keep_conv_plan): at one clip the launcher of the halo family hands maps of <= 64 / 128 items of the 256-pixel kernels to"; echo "# conv3x3_x3q_kernel (un-split) / conv3x3_x3p_kernel (split-K partials) -- the bit-equal 64-pixel forms of DESIGN 5.6; the times are what ran (r05_x3_b1_kernel_stats.txt names them)"; grep -v "^/opt" $G/r5p_census_b1.txt; } > profiles/r05_conv_census_x3_b1.txt
[ -f $G/r5p_gemm_forms.txt ] && { echo "# python tools/dev/gemm_lat_bench.py: us per launch (hipGraph replay of 40 launches) of the token GEMMs by form and images per launch --"; echo "# seq: one sequential sum (KEEP_CONV_NO_GEMM_LAT); waves: gemm_x3l_kernel at every row count; tiles: conv_x3_kernel with canonical slices at every row count"; grep -v "^/opt" $G/r5p_gemm_forms.txt; } > profiles/r05_gemm_forms.txt
ls -la profiles/r05_* 2>/dev/null
