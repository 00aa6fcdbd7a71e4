#!/bin/bash
# dev: build a variant of libkeep_hip.so with extra -D flags on the x3 convolution files:  build_ab.sh NAME "-DFLAG=.. ..."
set -e
cd "$(dirname "$0")/../../comfyui-keep_amd/csrc"
make -s >/dev/null
mkdir -p ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $2"
/opt/rocm/bin/hipcc $F -c keep_conv_x3.hip -o ab/$1_x3.o &
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -c keep_conv_x3s.hip -o ab/$1_x3s.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib_$1.so keep_abi.o keep_conv.o ab/$1_x3.o ab/$1_x3s.o keep_attn.o keep_ops.o keep_paste.o
rm -f ab/$1_x3.o ab/$1_x3s.o
echo built ab/lib_$1.so
