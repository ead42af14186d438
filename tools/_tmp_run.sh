mkdir -p gpurun_out/r3u
python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "ill_conditioned" 2>&1 | grep -E "^\{|passed|failed|^E " | cut -c1-300 > gpurun_out/r3u/t.txt
