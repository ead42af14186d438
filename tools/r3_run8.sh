mkdir -p gpurun_out/r3h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in c4 c5; do
rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/r3h/kt_$c -- python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --also "" > gpurun_out/r3h/kt_$c.log 2>&1
DB=$(find gpurun_out/r3h/kt_$c -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/r3h/kstats_$c.csv > /dev/null
rm -rf gpurun_out/r3h/kt_$c
done
