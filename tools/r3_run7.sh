mkdir -p gpurun_out/r3g
python -m pytest tests/test_backward_kernels.py -m gpu -q -s -k point_branch 2>&1 | grep -E "hip vs|passed|failed|^E " | cut -c1-300 > gpurun_out/r3g/pb.txt
python -m pytest tests/test_diff_render.py tests/test_dropin_module.py -m gpu -q 2>&1 | tail -4 >> gpurun_out/r3g/pb.txt
python tools/pose_refine_bench.py >> gpurun_out/r3g/pb.txt 2>&1
