cd /root/repo
L=comfyui-keep_amd/csrc/ab
timeout 1200 python tools/dev/lib_ab.py --b 1 --rounds 2 r4k=default r16k=$L/lib_rows16k.so r64k=$L/lib_rows64k.so 2>&1 | grep "^round"
timeout 1200 python tools/dev/lib_ab.py --b 16 --rounds 2 r4k=default r16k=$L/lib_rows16k.so r64k=$L/lib_rows64k.so 2>&1 | grep "^round"
