cd /root/repo
for rep in 1 2; do
for lib in libkeep_hip.so libkeep_prio1.so libkeep_prio2.so; do
  echo "== $lib"
  ABL_LIB=$PWD/comfyui-keep_amd/csrc/$lib X3=1 python tools/bench_conv.py c64_512 c128_256 c256_64 2>&1 | grep "input=True" | cut -c1-160
done; done
