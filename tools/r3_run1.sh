set -x
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3a/pytest.txt
NERFLOC_BENCH_ONE_GPU=1 python3 bench.py --config c3 --gpus 2 --steps 5 --warmup 2 > gpurun_out/r3a/onegpu2.json 2> gpurun_out/r3a/onegpu2.err
for spec in "c3 8192" "c3 4096" "c3 2048" "c3 1024" "c4 16384" "c4 8192" "c4 4096" "c4 2048"; do
  set -- $spec
  python3 bench.py --config $1 --rays $2 --steps 10 --warmup 3 --no-cpu-baseline --also '' --force-gather 2>/dev/null | tail -1 >> gpurun_out/r3a/shard_sweep.jsonl
done
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
