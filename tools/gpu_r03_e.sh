#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; grep -n "passed\|failed" $O/pytest.txt
export LSD_ONLY_TOP=14
for v in product nfaf64; do
  [ $v = product ] && unset SSLAM_LIB || export SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/$v.so
  SSLAM_PROF_STAGES=1 timeout 300 python tools/lsd_only.py 12288 64 2 > $O/lsd_only_$v.txt 2>&1; tail -n 1 $O/lsd_only_$v.txt
done
unset SSLAM_LIB
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_two_streams.json 2> $O/bench.err; cut -c1-300 $O/bench_two_streams.json
timeout 600 python bench.py --no-overlap --no-cpu-baseline --no-extras > $O/bench_one_stream.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/sq1 $O/sq2
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/sq1 -- python $R/bench.py --no-overlap --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-profile > $O/sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $O/sq2 -- python $R/bench.py --no-overlap --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-profile > $O/sq2.log 2>&1
cd $R; python tools/rocpd_pmc_summary.py $O/sq1 $O/sq1.txt > /dev/null 2>&1; python tools/rocpd_pmc_summary.py $O/sq2 $O/sq2.txt > /dev/null 2>&1; rm -rf $O/sq1 $O/sq2
head -30 $O/sq1.txt
