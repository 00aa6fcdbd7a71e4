cd /root/repo
python tools/dev/lib_ab.py --b 1 --rounds 2 base=default s4=comfyui-keep_amd/csrc/ab/lib_s2split4.so s8=comfyui-keep_amd/csrc/ab/lib_s2split8.so 2>&1 | grep -v Warning | tail -8
python tools/dev/lib_ab.py --b 48 --rounds 2 base=default s4=comfyui-keep_amd/csrc/ab/lib_s2split4.so s8=comfyui-keep_amd/csrc/ab/lib_s2split8.so 2>&1 | grep -v Warning | tail -8
