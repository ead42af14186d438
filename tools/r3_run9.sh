mkdir -p gpurun_out/r3i
python tools/pose_step_profile.py > gpurun_out/r3i/prof.txt 2>&1
