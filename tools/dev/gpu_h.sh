#!/bin/bash
cd /root/repo
bash tools/profile_step.sh x3 16 r2z > gpurun_out/r2z_x3.out 2>&1
REPO=$(pwd)
for POL in fp32 bf16; do
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/r2z/trace_${POL}" -o t -- python "$REPO/tools/run_step.py" $POL 16 2 > "$REPO/gpurun_out/r2z/trace_${POL}.log" 2>&1)
  DB=$(find gpurun_out/r2z/trace_${POL} -name "*results.db" | head -1)
  python profiles/summarize_rocpd.py "$DB" 2 > gpurun_out/r2z/${POL}_b16_kernel_stats.txt
done
find gpurun_out/r2z -name "*.db" -size +40M -delete
head -8 gpurun_out/r2z/x3_b16_kernel_stats.txt; head -6 gpurun_out/r2z/fp32_b16_kernel_stats.txt; head -6 gpurun_out/r2z/bf16_b16_kernel_stats.txt; ls gpurun_out/r2z
