python -m pytest tests -m gpu -x -q -s -k "streamed_sequence or gm_ffn_x3 or bgr_u8_to_comfy" 2>&1 | grep -v "it/s\]" | tail -30
echo "--- skew (dev lib)"; KEEP_HIP_LIB=$PWD/comfyui-keep_amd/csrc/ab/lib_ffn_dev.so python tools/dev/ffn_bench.py 2490368
echo "--- lockstep (dev lib)"; KEEP_FFN_NO_SKEW=1 KEEP_HIP_LIB=$PWD/comfyui-keep_amd/csrc/ab/lib_ffn_dev.so python tools/dev/ffn_bench.py 2490368
