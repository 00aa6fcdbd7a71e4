#!/bin/bash
# Round 6: every r06_* file below comes from ONE gpurun call = one box (bench line, kernel tables, PMC traffic, census, per-shape PMC rows):
#   on the GPU box (tools/dev/run1.sh of that call):
#     bash tools/profile_step.sh x3 48 r6p_x3_b48 ; cp gpurun_out/r6p_x3_b48/x3_b48_pmc.json profiles/r06_pmc_traffic.json   (bench.py reads it: roofline.traffic)
#     python bench.py > gpurun_out/r6p_bench.json
#     KEEP_AMD_GRAPH=0 KEEP_AMD_OVERLAP_MAX_CLIPS=0 PMC_COUNTERS= bash tools/profile_step.sh x3 1 r6p_x3_b1
#     python tools/dev/conv_census.py 48 conv_x3 > gpurun_out/r6p_census_b48.txt ; ... 48 halo > r6p_census_halo_b48.txt ; ... 1 '' > r6p_census_b1.txt
#     bash tools/dev/pmc_shapes.sh r6p_pmc_shapes > gpurun_out/r6p_pmc_shapes.txt ; python tools/dev/x3_ceiling.py > gpurun_out/r6p_ceiling.txt
#   here:  bash profiles/refresh.sh
# Each source file is the unedited output of the command quoted in its header.  (Rounds 1-5: git log -- profiles/refresh.sh.)
set -e
cd "$(dirname "$0")/.."
G=gpurun_out
hdr() { echo "# rocprofv3 --kernel-trace --stats -- python tools/run_step.py $1 $2 2   ($1 policy, $2 clip(s) x T=20 per pass, 2 passes; pass 1 includes first-touch allocation and the x3 weight split; $3)"; }
[ -f $G/r6p_x3_b48/x3_b48_kernel_stats.txt ] && { hdr x3 48 "the bench's default workload; same box and call as profiles/r06_bench_default.json"; cat $G/r6p_x3_b48/x3_b48_kernel_stats.txt; } > profiles/r06_x3_b48_kernel_stats.txt
[ -f $G/r6p_x3_b1/x3_b1_kernel_stats.txt ] && { hdr x3 1 "KEEP_AMD_GRAPH=0 KEEP_AMD_OVERLAP_MAX_CLIPS=0: eager launches on one stream, so every kernel is a trace record; same box and call as the bench line"; cat $G/r6p_x3_b1/x3_b1_kernel_stats.txt; } > profiles/r06_x3_b1_kernel_stats.txt
if [ -f $G/r6p_x3_b48/x3_b48_pmc.json ]; then
  cp $G/r6p_x3_b48/x3_b48_pmc.json profiles/r06_pmc_traffic.json
  { echo "# rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace --output-format csv -- python tools/run_step.py x3 48 1   (same box and call as the bench line)"; echo "# KiB per launch as reported; FETCH_SIZE is doubled on gfx950 in r06_pmc_traffic.json (MI355X_MICROARCH.md, HBM section)"; cat $G/r6p_x3_b48/x3_b48_pmc.txt; } > profiles/r06_pmc_step_x3_b48.txt
fi
[ -f $G/r6p_bench.json ] && grep '^{' $G/r6p_bench.json | tail -1 > profiles/r06_bench_default.json
[ -f $G/r6p_census_b48.txt ] && grep -v "^/opt" $G/r6p_census_b48.txt > profiles/r06_conv_census_x3_b48.txt
[ -f $G/r6p_census_halo_b48.txt ] && grep -v "^/opt" $G/r6p_census_halo_b48.txt > profiles/r06_conv_census_halo_x3_b48.txt
[ -f $G/r6p_census_b1.txt ] && { echo "# kernel names are the PLAN family's (keep_conv_plan): at one clip the launcher of the halo family hands maps of <= 64 / 128 items of the 256-pixel kernels to"; echo "# conv3x3_x3q_kernel (un-split) / conv3x3_x3p_kernel (split-K partials) -- the bit-equal 64-pixel forms of DESIGN 5.6; the times are what ran (r06_x3_b1_kernel_stats.txt names them)"; grep -v "^/opt" $G/r6p_census_b1.txt; } > profiles/r06_conv_census_x3_b1.txt
[ -f $G/r6p_pmc_shapes.txt ] && { echo "# bash tools/dev/pmc_shapes.sh: per-shape PMC rows of the dominant kernel (conv3x3_halo_x3s_kernel, GroupNorm-swish prologue + residual + statistics) at the bench's 48 images per launch,"; echo "# one rocprofv3 --pmc pass per counter set and layer (tools/bench_conv.py c64_512_n48 / c128_256_n48, 6 launches each); same box and call as the bench line.  Launches under a PMC pass"; echo "# run at a lower clock (1.32 GHz effective here against 1.7-1.8 without counters): the traffic and the busy FRACTION are the evidence, not the time."; grep -v "^/opt" $G/r6p_pmc_shapes.txt; } > profiles/r06_pmc_conv_shapes.txt
[ -s $G/r6p_ceiling.txt ] && grep -v "^/opt" $G/r6p_ceiling.txt > profiles/r06_x3_ceiling_probe.txt      # python tools/dev/x3_ceiling.py, same box and call
ls -la profiles/r06_* 2>/dev/null
