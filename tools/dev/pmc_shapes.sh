#!/bin/bash
# dev: per-shape PMC rows of the dominant kernel at the bench's launch sizes: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes),
# matrix-pipe busy cycles and the in-launch clock, for 64 ch @512^2 and 128 ch @256^2 with 48 images per launch, GroupNorm-swish prologue + residual + statistics.
#   tools/dev/pmc_shapes.sh <out dir under gpurun_out/>
OUT=$(pwd)/gpurun_out/${1:-pmc_shapes}
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for LAYER in c64_512_n48 c128_256_n48; do
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  X3=1 PRO_ONLY=1 RES=1 ITERS=6 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/$LAYER/p$i" -o p$i -- python "$REPO/tools/bench_conv.py" $LAYER > "$OUT/${LAYER}_p$i.log" 2>&1 || echo "pass $i of $LAYER failed: $SET"
done
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(list)
for f in glob.glob(out + '/c*/p*/**/*counter_collection.csv', recursive=True):
    shape = '64 ch @512^2 x 48 images' if '/c64_512_n48/' in f else '128 ch @256^2 x 48 images'
    for r in csv.DictReader(open(f)):
        if 'halo_x3' not in r['Kernel_Name']:
            continue
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        a = acc[shape][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            dur[shape].append((us, float(r['Counter_Value'])))
ALG = {'64 ch @512^2 x 48 images': 48 * 512 * 512 * 64 * 4 * 3, '128 ch @256^2 x 48 images': 48 * 256 * 256 * 128 * 4 * 3}      # read x + residual, write out (fp32)
FLOP = 48 * 2 * 512 * 512 * 64 * 64 * 9
for shape, d in sorted(acc.items()):
    print(shape)
    for c in sorted(d):
        print(f"   {c:28s} {d[c][0] / d[c][1]:18.0f}  per launch (n={d[c][1]})")
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:      # KiB as reported; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM section)
        hbm = (2 * d['FETCH_SIZE'][0] / d['FETCH_SIZE'][1] + d['WRITE_SIZE'][0] / d['WRITE_SIZE'][1]) * 1024
        print(f"   HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE): {hbm / 1e9:.3f} GB against {ALG[shape] / 1e9:.3f} GB algorithmic = {hbm / ALG[shape]:.2f} x")
    if dur[shape]:
        us = sum(u for u, _ in dur[shape]) / len(dur[shape]); ga = sum(g for _, g in dur[shape]) / len(dur[shape])
        clk = ga / 8 / us
        print(f"   mean launch {us:.1f} us = {FLOP / us / 1e6:.0f} TFLOP/s algorithmic; effective clock (GRBM_GUI_ACTIVE / 8 / wall) {clk:.0f} MHz")
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
            busy = d['SQ_VALU_MFMA_BUSY_CYCLES'][0] / d['SQ_VALU_MFMA_BUSY_CYCLES'][1]
            print(f"   matrix pipe busy: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x wall x clock) = {busy / (1024 * us * clk):.3f}")
PY
