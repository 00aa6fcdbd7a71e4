#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/stab.log
cat gpurun_out/stab.log
