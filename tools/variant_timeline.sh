#!/bin/bash
# variant_timeline.sh TAG NAME "bench args": rocprofv3 kernel trace of bench.py with the library variant NAME (nerf_loc_amd/csrc/variants/libnerfloc_NAME.so; "default" = the
# in-tree build) -> gpurun_out/TAG/NAME_timeline.txt + NAME_kernel_stats.csv (same-box A/B of several variants in one gpurun call)
TAG=$1; NAME=$2; ARGS=$3; LABEL=${4:-$2}   # LABEL: output name (default NAME) — the same variant under different environment switches
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$NAME" != "default" ]; then export NERFLOC_LIB=$GRAFT_REPO_ROOT/nerf_loc_amd/csrc/variants/libnerfloc_$NAME.so; fi
rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt_$LABEL -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-gradient-step --also "" --sustained-seconds 0 $ARGS > $OUT/kt_$LABEL.log 2>&1
DB=$(find $OUT/kt_$LABEL -name "*.db" | head -1)
python tools/prof_summary.py $DB $OUT/${LABEL}_kernel_stats.csv > /dev/null
python tools/prof_timeline.py $DB $OUT/${LABEL}_timeline.txt > /dev/null
grep '^{"metric"' $OUT/kt_$LABEL.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$LABEL', d['ms_per_step'], 'ms/step (under the profiler)', d['value'], 'rays/s')" >> $OUT/summary.txt 2>&1
# the same command without the profiler: the step time that counts
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gradient-step --also "" --sustained-seconds 0 $ARGS 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$LABEL', d['ms_per_step'], 'ms/step', d['value'], 'rays/s')" >> $OUT/summary.txt 2>&1
rm -rf $OUT/kt_$LABEL
