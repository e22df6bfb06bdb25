#!/bin/bash
# Round-3 call A (through gpurun): parity of the tiled-plane layout + integer rectangle counter, A/B against the round-2 library
# (lib/variants/r02.so), TCP / L2 counters of the sequential core in both layouts.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for fl in mw thr lat; do SSLAM_LSD_FLAVOUR=$fl timeout 400 python tools/fuzz_parity.py 250 $((31 + ${#fl})) > $O/fuzz_$fl.txt 2>&1; tail -3 $O/fuzz_$fl.txt; done
export LSD_ONLY_TOP=12
timeout 300 python tools/lsd_only.py 12288 64 2 > $O/lsd_only_new.txt 2>&1; tail -1 $O/lsd_only_new.txt
SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/r02.so timeout 300 python tools/lsd_only.py 12288 64 2 > $O/lsd_only_r02.txt 2>&1; tail -1 $O/lsd_only_r02.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_two_streams.json 2> $O/bench.err; cut -c1-600 $O/bench_two_streams.json
timeout 600 python bench.py --no-overlap --no-cpu-baseline --no-extras > $O/bench_one_stream.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
for v in new r02; do
  [ $v = r02 ] && export SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/r02.so || unset SSLAM_LIB
  i=0
  for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1)); rm -rf $O/pmc_${v}_$i
    (cd $R && timeout 300 rocprofv3 --pmc $set -d $O/pmc_${v}_$i -- python tools/lsd_only.py 6144 64 1 > $O/pmc_${v}_$i.log 2>&1)
    (cd $R && python tools/rocpd_pmc_summary.py $O/pmc_${v}_$i $O/pmc_${v}_$i.txt > /dev/null 2>&1); rm -rf $O/pmc_${v}_$i
    grep -h "k_lsd_regions\|k_lsd_grad\|k_nfa_count" $O/pmc_${v}_$i.txt | head -12
  done
done
