OUT=gpurun_out/r5j; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt -- python bench.py --rays 512 --steps 6 --warmup 2 --no-cpu-baseline --no-gradient-step --also "" > $OUT/kt.log 2>&1
DB=$(find $OUT/kt -name "*.db" | head -1)
python tools/prof_timeline.py $DB $OUT/c2_shard512_timeline.txt > /dev/null
rm -rf $OUT/kt
cut -c1-100 $OUT/c2_shard512_timeline.txt
python bench.py --rays 512 --steps 50 --warmup 10 --no-cpu-baseline --no-gradient-step --also "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512 rays ms', d['ms_per_step'])"
