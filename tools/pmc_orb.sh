#!/bin/bash
# SQ counters of the ORB kernels (bench.py workload c2, one step): tools/pmc_orb.sh [kernel-substring]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_orb; mkdir -p $O; K=${1:-k_fast_cells}
cd /tmp && export TMPDIR=/tmp
for pass in 1 2; do
  if [ $pass = 1 ]; then C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; else C="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT"; fi
  rm -rf $O/p$pass
  (cd $R && timeout 300 rocprofv3 --pmc $C -d $O/p$pass -- python bench.py --workload c2 --unique 8 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-profile > $O/p$pass.log 2>&1)
  (cd $R && python tools/rocpd_pmc_summary.py $O/p$pass $O/p$pass.txt | grep "$K" | awk '{printf "%s %s %s %.4g\n", $1, $2, $3, $4}')
  rm -rf $O/p$pass
done
