mkdir -p gpurun_out/r3c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/r3c/kt -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --also "" > gpurun_out/r3c/kt.log 2>&1
DB=$(find gpurun_out/r3c/kt -name "*.db" | head -1)
python tools/prof_timeline.py $DB gpurun_out/r3c/timeline.txt > /dev/null
python tools/prof_summary.py $DB gpurun_out/r3c/kstats.csv > /dev/null
rm -rf gpurun_out/r3c/kt
python -m pytest tests/test_diff_render.py -m gpu -q 2>&1 | tail -3 > gpurun_out/r3c/pytest.txt
