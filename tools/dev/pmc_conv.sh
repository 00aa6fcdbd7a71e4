#!/bin/bash
# PMC passes over tools/bench_conv.py layers (dev): stall attribution of the x3 halo kernels
#   tools/dev/pmc_conv.sh <out dir under gpurun_out/> <layers...>      env: CONV_FLAGS=512 (KEEP_CONV_NO_STREAM) for round 3's kernel
OUT=$(pwd)/gpurun_out/$1; shift
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAVES" ${PMC_EXTRA:+"$PMC_EXTRA"}; do
  i=$((i+1))
  X3=1 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/p$i" -o p$i -- python "$REPO/tools/bench_conv.py" "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed: $SET"
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(list)          # (kernel, grid) -> launch durations of the GRBM pass (us)
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        if os.environ.get('PMC_FILTER', 'halo_x3') not in k:
            continue
        a = acc[k][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            dur[k].append((us, float(r['Counter_Value'])))
for k, d in acc.items():
    print(k)
    for c in sorted(d):
        print(f"   {c:28s} {d[c][0] / d[c][1]:16.0f}  (n={d[c][1]})")
    if dur[k]:       # effective shader clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel wall time (MI355X_MICROARCH.md)
        us = sum(u for u, _ in dur[k]); ga = sum(g for _, g in dur[k])
        print(f"   effective clock (GRBM_GUI_ACTIVE / 8 / wall): {ga / 8 / us:8.0f} MHz over {len(dur[k])} launches, mean launch {us / len(dur[k]):.1f} us")
PY
