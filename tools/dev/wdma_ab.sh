#!/bin/bash
# A/B of the LDS-DMA weight staging of the x3 halo kernel (default) against the VGPR-staged form (KEEP_X3_NO_WDMA=1)
for v in wdma vgpr; do
  if [ $v = vgpr ]; then export KEEP_X3_NO_WDMA=1; else unset KEEP_X3_NO_WDMA; fi
  echo "== $v"; X3=1 python tools/bench_conv.py c64_512 c128_256 up128_512 c256_64 c512_16 2>&1 | grep -v amdgpu
done
unset KEEP_X3_NO_WDMA
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "x3 or conv" --timeout 600 2>&1 | tail -5
