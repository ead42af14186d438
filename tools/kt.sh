#!/bin/bash
# Kernel trace (serial streams) of one bench run with a given library build: tools/kt.sh lib.so tag [grep-pattern]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NERFLOC_LIB=$PWD/$1 NERFLOC_SERIAL=1 rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/kt_$2 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also "" > /dev/null 2>&1
python tools/prof_summary.py $(find gpurun_out/kt_$2 -name "*.db" | head -1) gpurun_out/kt_$2.csv | grep -E "${3:-sample_chain}" | sed -E 's/\(anonymous namespace\):://g; s/\(.*\)"/"/' | cut -c1-120
