mkdir -p gpurun_out/r3e
python -m pytest tests/test_gpu_configs.py -m gpu -q -k "sampled_rays and c5" -s 2>&1 | grep -E "^E  |c5 bf16|passed|failed" | cut -c1-1500 > gpurun_out/r3e/c5.txt
