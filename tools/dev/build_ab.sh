#!/bin/bash
# dev: build a variant of libkeep_hip.so with extra -D flags on the x3 convolution files:  build_ab.sh NAME "-DFLAG=.. ..."
set -e
cd "$(dirname "$0")/../../comfyui-keep_amd/csrc"
make -s >/dev/null
mkdir -p ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DKEEP_DEV_KNOBS $2"
/opt/rocm/bin/hipcc $F -c keep_conv_x3.hip -o ab/$1_x3.o &
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -c keep_conv_x3s.hip -o ab/$1_x3s.o &
ATT=keep_attn.o
if [ -n "$3" ]; then /opt/rocm/bin/hipcc $F -c keep_attn.hip -o ab/$1_attn.o & ATT=ab/$1_attn.o; fi      # third argument: rebuild the attention file too
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib_$1.so keep_abi.o keep_conv.o ab/$1_x3.o ab/$1_x3s.o keep_conv_x3p.o keep_gemm_x3l.o keep_ffn_x3.o $ATT keep_ops.o keep_yolo.o keep_paste.o
rm -f ab/$1_x3.o ab/$1_x3s.o ab/$1_attn.o
echo built ab/lib_$1.so
