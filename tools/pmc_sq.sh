#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_sq1 $O/pmc_sq2
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O/pmc_sq1 -- python $R/bench.py --batch 3072 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $O/pmc_sq2 -- python $R/bench.py --batch 3072 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_sq2.log 2>&1
tail -2 $O/pmc_sq1.log | cut -c1-200
