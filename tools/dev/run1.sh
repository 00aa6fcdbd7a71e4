cd /root/repo
bash tools/profile_step.sh x3 48 r6p_x3_b48 > gpurun_out/r6p_profile_b48.log 2>&1
[ -s gpurun_out/r6p_x3_b48/x3_b48_pmc.json ] && cp gpurun_out/r6p_x3_b48/x3_b48_pmc.json profiles/r06_pmc_traffic.json && echo "pmc json refreshed on this box"
python bench.py > gpurun_out/r6p_bench.json 2> gpurun_out/r6p_bench.err; echo bench rc=$?
KEEP_AMD_GRAPH=0 KEEP_AMD_OVERLAP_MAX_CLIPS=0 PMC_COUNTERS= bash tools/profile_step.sh x3 1 r6p_x3_b1 > gpurun_out/r6p_profile_b1.log 2>&1
python tools/dev/conv_census.py 48 conv_x3 > gpurun_out/r6p_census_b48.txt 2>&1
python tools/dev/conv_census.py 48 halo > gpurun_out/r6p_census_halo_b48.txt 2>&1
python tools/dev/conv_census.py 1 '' > gpurun_out/r6p_census_b1.txt 2>&1
bash tools/dev/pmc_shapes.sh r6p_pmc_shapes > gpurun_out/r6p_pmc_shapes.txt 2>&1
python tools/dev/x3_ceiling.py > gpurun_out/r6p_ceiling.txt 2>&1
grep '^{' gpurun_out/r6p_bench.json | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(d['value'], d['ms_per_step'], d['config']['clips_per_gpu'], d['peak_hbm_gb'], r['frac'], r['conv_path_frac'], r.get('traffic'), r.get('algorithmic_bytes_per_launch'), r.get('avg_launch_ms'), r.get('practical_peak'), r.get('frac_of_practical'))
print(d.get('clips16',{}).get('value'), d['b1']['value'], d['b1'].get('latency_profile',{}).get('value'), d['cpu_baseline']['value'])
for leg in ('end_to_end', 'end_to_end_product'):
    for k, v in d.get(leg, {}).items():
        print(leg, k, v.get('value'), v.get('seconds'), (v.get('process_image_sequence') or {}).get('value'))
"
head -8 gpurun_out/r6p_x3_b48/x3_b48_kernel_stats.txt
