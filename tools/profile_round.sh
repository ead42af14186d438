#!/bin/bash
# Collect the measurements a round's docs cite, on the GPU box (run through gpurun): bench lines of every BASELINE config, rocprofv3 kernel
# traces (with and without the side stream) + the timeline of one step, the separate PMC passes (SQ counters, FETCH_SIZE, WRITE_SIZE) and the HBM
# traffic summary, the pose-refinement step.  Usage: bash tools/profile_round.sh <tag>   -> gpurun_out/<tag>/ (copy what is cited into profiles/)
TAG=${1:-r6}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench_1gpu.json 2> $OUT/bench_1gpu.err
for c in c1 c3 c4 c5; do python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --sustained-seconds 3 2>/dev/null | grep '^{"metric"' | tail -1 >> $OUT/other_configs.jsonl; done
kt() {  # name, extra args
  rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt_$1 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-gradient-step --also "" --sustained-seconds 0 $2 > $OUT/kt_$1.log 2>&1
  DB=$(find $OUT/kt_$1 -name "*.db" | head -1)
  python tools/prof_summary.py $DB $OUT/${1}_kernel_stats.csv > /dev/null
  [ "$1" = "c2" ] && python tools/prof_timeline.py $DB $OUT/c2_timeline.txt > /dev/null
  rm -rf $OUT/kt_$1
}
kt c2 ""
kt c2_serial "--no-side-stream"
kt c2_x3_serial "--no-side-stream --precision bf16x3"
kt c5 "--config c5"
kt c4 "--config c4"
pmc() {  # name, counters
  rocprofv3 --kernel-trace --pmc $2 --output-format rocpd -d $OUT/pmc_$1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gradient-step --also "" --sustained-seconds 0 > $OUT/pmc_$1.log 2>&1
}
pmc sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
pmc sq2 "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
pmc f "FETCH_SIZE"
pmc w "WRITE_SIZE"
for n in sq sq2 f w; do DB=$(find $OUT/pmc_$n -name "*.db" | head -1); python tools/pmc_summary.py $DB $OUT/pmc_$n.csv > /dev/null; done
python tools/hbm_traffic.py $(find $OUT/pmc_f -name "*.db" | head -1) $(find $OUT/pmc_w -name "*.db" | head -1) 7 $OUT/hbm_traffic.json f16mx > /dev/null
rm -rf $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_f $OUT/pmc_w
python tools/pose_refine_bench.py 2>&1 | grep "rays x" > $OUT/pose_refine.txt
# rocprofv3 kernel traces of the two gradient benches (the whole processes: eager reference steps included; the library's kernels are the named ones)
for t in pose_refine train_step; do
  rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt_$t -- python tools/${t}_bench.py > $OUT/kt_$t.log 2>&1
  python tools/prof_summary.py $(find $OUT/kt_$t -name "*.db" | head -1) $OUT/${t}_kernel_stats.csv > /dev/null
  rm -rf $OUT/kt_$t
done
ROWS=40 python tools/pose_step_profile.py 2>&1 | grep "ms/step" > $OUT/pose_step_profile.txt
python tools/train_step_bench.py 2>&1 | grep "training step" > $OUT/train_step.txt
PROFILE=1 python tools/train_step_bench.py 2>&1 | grep "ms/step" > $OUT/train_step_profile.txt
python tools/setup_bench.py c2 > $OUT/setup_bench.txt 2>&1

# one GPU rendering a rank's shard of config 2 / 3 / 4 with the single-rank RCCL all-gather in the step: the projected strong-scaling curve
for spec in "c2 4096" "c2 2048" "c2 1024" "c2 512" "c3 8192" "c3 4096" "c3 2048" "c3 1024" "c4 16384" "c4 8192" "c4 4096" "c4 2048"; do
  set -- $spec
  python bench.py --config $1 --rays $2 --force-gather --steps 10 --warmup 3 --no-cpu-baseline --no-gradient-step --also "" --sustained-seconds 0 2>/dev/null | grep '^{"metric"' | tail -1 >> $OUT/shard_sweep.jsonl
done
python tools/multi_frame_bench.py 2>&1 | grep "frames" > $OUT/multi_frame.txt
NERFLOC_BENCH_ONE_GPU=1 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-gradient-step --also "" --sustained-seconds 0 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/bench_2rank_one_gpu_functional.json
