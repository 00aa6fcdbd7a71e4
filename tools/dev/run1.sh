python tools/dev/shape_prof.py x3 16 2>&1 | grep -v amdgpu.ids | head -22 | cut -c1-170 > gpurun_out/shape_new.txt
KEEP_X3_NO_STREAM=1 python tools/dev/shape_prof.py x3 16 2>&1 | grep -v amdgpu.ids | head -22 | cut -c1-170 > gpurun_out/shape_old.txt
cat gpurun_out/shape_new.txt; echo =====; cat gpurun_out/shape_old.txt
