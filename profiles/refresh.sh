#!/bin/bash
# Round 4: copy the summaries tools/profile_step.sh wrote under gpurun_out/ (on the GPU box, merged back by gpurun) into the
# tracked profiles/ directory.  Each source file is the unedited output of the command quoted in its header.
#   on the GPU box:   bash tools/profile_step.sh x3 16 r4p_x3_b16 ; bash tools/profile_step.sh fp32 16 r4p_fp32_b16 ;
#                     KEEP_AMD_GRAPH=0 bash tools/profile_step.sh x3 1 r4p_x3_b1 ; python bench.py > gpurun_out/r4p_bench.json
#   here:             bash profiles/refresh.sh
set -e
cd "$(dirname "$0")/.."
G=gpurun_out
hdr() { echo "# rocprofv3 --kernel-trace --stats -- python tools/run_step.py $1 $2 2   ($1 policy, $2 clip(s) x T=20 per pass, 2 passes; pass 1 includes first-touch allocation and the x3 weight split; $3)"; }
[ -f $G/r4p_x3_b16/x3_b16_kernel_stats.txt ] && { hdr x3 16 "round-4 final kernels"; cat $G/r4p_x3_b16/x3_b16_kernel_stats.txt; } > profiles/r04_x3_b16_kernel_stats.txt
[ -f $G/r4p_fp32_b16/fp32_b16_kernel_stats.txt ] && { hdr fp32 16 "exact-f32 MFMA policy"; cat $G/r4p_fp32_b16/fp32_b16_kernel_stats.txt; } > profiles/r04_fp32_b16_kernel_stats.txt
[ -f $G/r4p_x3_b1/x3_b1_kernel_stats.txt ] && { hdr x3 1 "KEEP_AMD_GRAPH=0: eager launches, so every kernel is a trace record"; cat $G/r4p_x3_b1/x3_b1_kernel_stats.txt; } > profiles/r04_x3_b1_kernel_stats.txt
if [ -f $G/r4p_x3_b16/x3_b16_pmc.json ]; then
  cp $G/r4p_x3_b16/x3_b16_pmc.json profiles/r04_pmc_traffic.json
  { echo "# rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace --output-format csv -- python tools/run_step.py x3 16 1"; echo "# KiB per launch as reported; FETCH_SIZE is doubled on gfx950 in r04_pmc_traffic.json (MI355X_MICROARCH.md, HBM section)"; cat $G/r4p_x3_b16/x3_b16_pmc.txt; } > profiles/r04_pmc_step_x3_b16.txt
fi
[ -f $G/r4p_bench.json ] && grep '^{' $G/r4p_bench.json | tail -1 > profiles/r04_bench_default.json
[ -f $G/r4p_bench_2ranks_1gpu.json ] && grep '^{' $G/r4p_bench_2ranks_1gpu.json | tail -1 > profiles/r04_bench_2ranks_on_1gpu_gloo.json
ls -la profiles/r04_* 2>/dev/null
