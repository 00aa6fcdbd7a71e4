#!/bin/bash
cd /root/repo
bash tools/profile_step.sh x3 16 r2p > gpurun_out/r2p_profile_x3_16.out 2>&1
bash tools/profile_step.sh x3 1 r2p > gpurun_out/r2p_profile_x3_1.out 2>&1
tail -30 gpurun_out/r2p_profile_x3_16.out
ls -la gpurun_out/r2p
