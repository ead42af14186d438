mkdir -p gpurun_out/r3d
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r3d/pytest.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3d/bench_c2.json 2> gpurun_out/r3d/bench_c2.err
python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3d/bench_c5.json 2> gpurun_out/r3d/bench_c5.err
python tools/race_check.py c2 bf16x3 10 > gpurun_out/r3d/race.txt 2>&1
