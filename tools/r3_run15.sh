mkdir -p gpurun_out/r3o
python tools/pose_refine_bench.py 2>&1 | grep "HIP point" > gpurun_out/r3o/a.txt
NERFLOC_F32_BN128=1 python tools/pose_refine_bench.py 2>&1 | grep "HIP point" >> gpurun_out/r3o/a.txt
python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --also '' 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 bn256', d['ms_per_step'])" >> gpurun_out/r3o/a.txt
NERFLOC_F32_BN128=1 python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --also '' 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 bn128', d['ms_per_step'])" >> gpurun_out/r3o/a.txt
