mkdir -p gpurun_out/r3t
python -m pytest tests/test_backward_kernels.py tests/test_diff_render.py tests/test_dropin_module.py -m gpu -q 2>&1 | tail -3 > gpurun_out/r3t/t.txt
python tools/pose_refine_bench.py 2>&1 | grep "rays x" >> gpurun_out/r3t/t.txt
python tools/pose_step_profile.py > gpurun_out/r3t/prof.txt 2>&1
