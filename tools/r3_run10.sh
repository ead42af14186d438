mkdir -p gpurun_out/r3j
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r3j/pytest.txt
python tools/pose_refine_bench.py > gpurun_out/r3j/pose.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r3j/bench_c2.json 2> gpurun_out/r3j/bench_c2.err
