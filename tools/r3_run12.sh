mkdir -p gpurun_out/r3l
python -m pytest tests/test_backward_kernels.py tests/test_diff_render.py -m gpu -q 2>&1 | tail -2 > gpurun_out/r3l/pytest.txt
python tools/pose_refine_bench.py > gpurun_out/r3l/pose.txt 2>&1
python tools/pose_step_profile.py > gpurun_out/r3l/prof.txt 2>&1
