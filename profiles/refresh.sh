#!/bin/bash
# Rebuild the committed summaries from the raw gpurun_out/ results of the commands quoted in each file header.
# usage: profiles/refresh.sh <bf16 kernel-trace db> <fp32 kernel-trace db> <bf16 pmc dir> <fp32 pmc dir> <bench json>
set -e
cd "$(dirname "$0")/.."
BF=$1; FP=$2; PMCB=$3; PMCF=$4; BJ=$5
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/run_step.py bf16 16 2   (bf16 policy, 16 clips x T=20 per pass, 2 passes; the first pass includes first-touch allocation)";
  python profiles/summarize_rocpd.py $BF 2; } > profiles/r01_bf16_b16_kernel_stats.txt
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/run_step.py fp32 16 2   (fp32 parity policy, 16 clips x T=20 per pass, 2 passes)";
  python profiles/summarize_rocpd.py $FP 2; } > profiles/r01_fp32_b16_kernel_stats.txt
for p in bf16 fp32; do B=16; PMC=$PMCB; [ $p = fp32 ] && PMC=$PMCF
  { echo "# rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace --output-format csv -- python tools/run_step.py $p $B 1   (one step = $B clips x T=20)";
    echo "# KiB per launch as reported; FETCH_SIZE must be doubled on gfx950 (MI355X_MICROARCH.md; calibrated on norm_act_bf16 in profiles/r01_pmc_halo_traffic.txt)";
    python profiles/summarize_pmc.py $PMC/${p}_*_counter_collection.csv | head -40; } > profiles/r01_pmc_step_${p}_b$B.txt
done
python - $PMCB $PMCF <<'PY'
import csv, collections, json, sys
csv.field_size_limit(1 << 30)
pmcs = {'bf16': sys.argv[1], 'fp32': sys.argv[2]}
out = {}
for pol, B in (('bf16', 16), ('fp32', 16)):
    pmc = pmcs[pol]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        for r in csv.DictReader(open(f'{pmc}/{pol}_{ctr}_counter_collection.csv')):
            n = r['Kernel_Name']
            if not n.startswith('void conv') and not n.startswith('conv'):
                continue
            n = n[5:n.index('(')] if n.startswith('void ') else n[:n.index('(')]
            a = agg[n][ctr]; a[0] += 1; a[1] += float(r['Counter_Value'])
    out[pol] = {'clips_per_gpu': B, 'kernels': {k: {'launches': v['FETCH_SIZE'][0],
                 'fetch_kib_raw_per_launch': round(v['FETCH_SIZE'][1] / v['FETCH_SIZE'][0], 1),
                 'write_kib_per_launch': round(v['WRITE_SIZE'][1] / v['WRITE_SIZE'][0], 1),
                 'hbm_bytes_per_launch': round((2 * v['FETCH_SIZE'][1] / v['FETCH_SIZE'][0] + v['WRITE_SIZE'][1] / v['WRITE_SIZE'][0]) * 1024)}
                for k, v in agg.items()}}
out['_note'] = ("rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs) over one step of tools/run_step.py; "
                "hbm_bytes = 2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE correction per MI355X_MICROARCH.md, calibrated on "
                "norm_act_bf16: profiles/r01_pmc_halo_traffic.txt)")
json.dump(out, open('profiles/r01_pmc_traffic.json', 'w'), indent=1)
PY
cp $BJ profiles/r01_bench_default.json
