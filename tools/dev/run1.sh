cd /root/repo
python -m pytest tests/test_gpu_net.py -x -q -m gpu -s -k "cft or T3_vs_reference or T20_vs_reference or batched_clips or asian" 2>&1 | grep -v "^$" | cut -c1-400 | tail -25
python tools/dev/env_ab.py --b 1 split:KEEP_CFT_SPLIT=1 whole:KEEP_CFT_SPLIT=0 2>&1 | grep -v Warning | tail -6
python tools/dev/env_ab.py --b 16 split:KEEP_CFT_SPLIT=1 whole:KEEP_CFT_SPLIT=0 2>&1 | grep -v Warning | tail -6
python tools/dev/x3_ceiling.py > gpurun_out/r06_x3_ceiling_probe.txt 2>&1; cat gpurun_out/r06_x3_ceiling_probe.txt
