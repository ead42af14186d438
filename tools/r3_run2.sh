mkdir -p gpurun_out/r3b
python -m pytest tests/test_diff_render.py -m gpu -q -k "query_training" 2>&1 | grep -E "^E|passed|failed|worst" | head -20 > gpurun_out/r3b/q.txt
python tools/ray_order_bench.py c2 > gpurun_out/r3b/order.txt 2>&1
python tools/ray_order_bench.py c4 >> gpurun_out/r3b/order.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r3b/pytest.txt
