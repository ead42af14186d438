mkdir -p gpurun_out/r3f
python -m pytest tests/test_backward_kernels.py tests/test_diff_render.py tests/test_gpu_configs.py tests/test_dropin_module.py -m gpu -q 2>&1 | tail -12 > gpurun_out/r3f/pytest.txt
python tools/pose_refine_bench.py > gpurun_out/r3f/pose.txt 2>&1
