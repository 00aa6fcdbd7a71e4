timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "layernorm or small_ops or gm_mlp" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof20 -o bf16 -- python $GRAFT_REPO_ROOT/tools/run_step.py bf16 16 2 > /dev/null 2>&1
