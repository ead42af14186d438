cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also ''"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format rocpd -d gpurun_out/pmc_sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also "" > gpurun_out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format rocpd -d gpurun_out/pmc_sq2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also "" > gpurun_out/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d gpurun_out/pmc_f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also "" > gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d gpurun_out/pmc_w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also "" > gpurun_out/pmc_w.log 2>&1
for n in sq sq2 f w; do DB=$(find gpurun_out/pmc_$n -name "*.db" | head -1); python tools/pmc_summary.py $DB gpurun_out/r2b_pmc_$n.csv > /dev/null; done
python tools/hbm_traffic.py $(find gpurun_out/pmc_f -name "*.db" | head -1) $(find gpurun_out/pmc_w -name "*.db" | head -1) 7 gpurun_out/r2_hbm_traffic.json
head -4 gpurun_out/r2b_pmc_sq.csv | cut -c1-400
