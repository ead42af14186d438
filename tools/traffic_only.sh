#!/bin/bash
# Re-collect only the HBM-traffic fingerprint (FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh) after a change of the kernel sources, so that
# profiles/rN_hbm_traffic.json carries the sources' hash bench.py compares against.  Usage (through gpurun): bash tools/traffic_only.sh <tag>
TAG=${1:-r3t}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in f w; do
  C=FETCH_SIZE; [ $n = w ] && C=WRITE_SIZE
  rocprofv3 --kernel-trace --pmc $C --output-format rocpd -d $OUT/pmc_$n -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gradient-step --also "" --sustained-seconds 0 > $OUT/pmc_$n.log 2>&1
  python tools/pmc_summary.py $(find $OUT/pmc_$n -name "*.db" | head -1) $OUT/pmc_$n.csv > /dev/null
done
python tools/hbm_traffic.py $(find $OUT/pmc_f -name "*.db" | head -1) $(find $OUT/pmc_w -name "*.db" | head -1) 7 $OUT/hbm_traffic.json f16mx > /dev/null
rm -rf $OUT/pmc_f $OUT/pmc_w
python bench.py --steps 20 --warmup 5 > $OUT/bench_1gpu.json 2> $OUT/bench_1gpu.err
