mkdir -p gpurun_out/r3k
for e in 00 10 01 11; do
echo "== NERFLOC_BWD_EXP=$e (first digit: forward recompute 1=cfg precision 0=fp32; second: backward 1=bf16x3 0=fp32)" >> gpurun_out/r3k/exp.txt
NERFLOC_BWD_EXP=$e python -m pytest tests/test_backward_kernels.py -m gpu -q -s -k "point_branch and bf16x3" 2>&1 | grep -E "hip vs|passed|failed" | cut -c1-200 >> gpurun_out/r3k/exp.txt
done
python tools/pose_refine_bench.py >> gpurun_out/r3k/exp.txt 2>&1
NERFLOC_BWD_EXP=10 python tools/pose_refine_bench.py >> gpurun_out/r3k/exp.txt 2>&1
