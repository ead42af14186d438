mkdir -p gpurun_out/r3m
python -m pytest tests/test_backward_kernels.py -m gpu -q -s -k "mv_aggregate_backward or blend_backward" 2>&1 | grep -E "hip vs|passed|failed|^E  |Error" | cut -c1-250 > gpurun_out/r3m/t.txt
