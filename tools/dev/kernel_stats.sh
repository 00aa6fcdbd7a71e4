#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r2q
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/r2q/trace" -o t -- python "$REPO/tools/run_step.py" x3 16 2 > "$REPO/gpurun_out/r2q/trace.log" 2>&1
cd "$REPO"
DB=$(find gpurun_out/r2q/trace -name "*results.db" | head -1)
python profiles/summarize_rocpd.py "$DB" 2 > gpurun_out/r2q/x3_b16_kernel_stats.txt
find gpurun_out/r2q -name "*.db" -size +40M -delete
head -45 gpurun_out/r2q/x3_b16_kernel_stats.txt
