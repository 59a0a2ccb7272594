#!/bin/bash
python -m pytest tests/test_pipe_gpu.py tests/test_model_gpu.py -q -x -k "pipe or dynamic or split_slices or sa_scale_pipe or sa_scale_fused" 2>&1 | tail -3
for cfg in "1 0" "0 8" "0 0"; do set -- $cfg
echo "dynamic $1 reserve $2"; CAPTRA_PIPE_DYNAMIC=$1 CAPTRA_PIPE_RESERVE=$2 python tools/exp_pipe_stages.py 2>&1 | grep -v amdgpu.ids
done
