mkdir -p gpurun_out/r3p
python -m pytest tests/test_backward_kernels.py -m gpu -q -s -k "ray_unet_backward" 2>&1 | grep -E "hip vs|passed|failed|^E  |Error" | cut -c1-250 > gpurun_out/r3p/t.txt
python tools/pose_refine_bench.py 2>&1 | grep "rays x" >> gpurun_out/r3p/t.txt
