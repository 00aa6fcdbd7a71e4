cd /root/repo
for g in 512 256 384 640 768 1024; do echo "grid $g"; KEEP_HIP_LIB=comfyui-keep_amd/csrc/ab/lib_grid.so KEEP_X3_GRID=$g X3=1 PRO_ONLY=1 ITERS=30 python tools/bench_conv.py c64_512_n1 c128_256_n1 c64_512_n48 2>&1 | grep "TFLOP"; done
