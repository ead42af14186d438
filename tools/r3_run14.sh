mkdir -p gpurun_out/r3n
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r3n/pytest.txt
python tools/pose_refine_bench.py > gpurun_out/r3n/pose.txt 2>&1
python tools/pose_step_profile.py > gpurun_out/r3n/prof.txt 2>&1
