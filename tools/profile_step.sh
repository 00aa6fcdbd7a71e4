#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace statistics and HBM-traffic PMC passes of ONE policy's step.
#   tools/profile_step.sh <policy: x3|fp32|bf16> <clips per GPU> <out dir under gpurun_out/>
# Separate rocprofv3 runs: --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE (counters never share a run with
# each other or with tracing domains beyond the kernel trace; each under its own timeout -- PMC passes occasionally hang).
POL=${1:-x3}; B=${2:-16}; OUT=$(pwd)/gpurun_out/${3:-prof}
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace_${POL}_b$B" -o t -- python "$REPO/tools/run_step.py" $POL $B 2 > "$OUT/trace_${POL}_b$B.log" 2>&1
for C in ${PMC_COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_${POL}_b$B" -o ${POL}_$C -- python "$REPO/tools/run_step.py" $POL $B 1 > "$OUT/pmc_${POL}_${C}.log" 2>&1 || echo "PMC pass $C failed/timeout"
done
cd "$REPO"
DB=$(find "$OUT/trace_${POL}_b$B" -name "*results.db" | head -1)
python profiles/summarize_rocpd.py "$DB" 2 > "$OUT/${POL}_b${B}_kernel_stats.txt"
python profiles/summarize_pmc.py $(find "$OUT/pmc_${POL}_b$B" -name "*counter_collection.csv") > "$OUT/${POL}_b${B}_pmc.txt" 2>/dev/null
python profiles/pmc_to_json.py $POL $B "$OUT/pmc_${POL}_b$B" > "$OUT/${POL}_b${B}_pmc.json" 2>/dev/null
find "$OUT" -name "*.db" -delete
head -30 "$OUT/${POL}_b${B}_kernel_stats.txt"
