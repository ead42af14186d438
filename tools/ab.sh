#!/bin/bash
# Same-box A/B of two library builds: tools/ab.sh libA.so libB.so [bench args...]; prints ms_per_step alternately, 3 rounds each.
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in "$A" "$B"; do
    echo -n "$(basename $L): "
    NERFLOC_LIB=$PWD/$L python bench.py --steps 5 --warmup 2 --no-cpu-baseline --also "" "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"
  done
done
