OUT=gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-gradient-step --also "" --precision f16mx --no-side-stream $2 > $OUT/kt.log 2>&1
DB=$(find $OUT/kt -name "*.db" | head -1)
python tools/prof_summary.py $DB $OUT/serial_kernel_stats.csv > /dev/null
rm -rf $OUT/kt
cut -c1-150 $OUT/serial_kernel_stats.csv | head -30
