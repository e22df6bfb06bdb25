#!/bin/bash
# Round-5 measurement recipe on the final tree (through gpurun).  Counters in their own passes (no trace domains mixed in), through the C harness where the workload allows it
# (tools/step_check: the bench's step without Python, seconds per pass): FETCH_SIZE / WRITE_SIZE at 6144 frames -> profiles/pmc_traffic.json (regenerated on the box BEFORE the
# bench line is written, so that its roofline.traffic comes from counters taken on the kernels it times); three SQ passes at 3072 frames (issue, pipes, lane occupancy);
# the same two counters for BASELINE configs[3] through bench.py; kernel traces; then the GPU suite and the bench line.  Every step under its own timeout.
#     bash tools/build_c_harnesses.sh && gpurun --timeout 1500 -- 'bash tools/gpu_profile_r05_final.sh'
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
b=6144
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_c3_$c; (cd $R && timeout 200 rocprofv3 --pmc $c -d $O/pmc_c3_$c -- tools/step_check $b 1 1 1 > $O/pmc_c3_$c.log 2>&1)
done
(cd $R && python tools/rocpd_pmc_summary.py $O/pmc_c3_FETCH_SIZE $O/pmc_fetch_c3.txt > /dev/null; python tools/rocpd_pmc_summary.py $O/pmc_c3_WRITE_SIZE $O/pmc_write_c3.txt > /dev/null
 python tools/make_pmc_traffic.py $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE $b 3 2 $O/pmc_traffic_c3.json | head -30; cp $O/pmc_traffic_c3.json $R/profiles/pmc_traffic.json)
rm -rf $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
P3="SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf $O/sq$i
  (cd $R && timeout 200 rocprofv3 --pmc $P -d $O/sq$i -- tools/step_check 3072 1 0 1 > $O/sq$i.log 2>&1; python tools/rocpd_pmc_summary.py $O/sq$i $O/sq$i.txt > /dev/null; rm -rf $O/sq$i)
done
(cd $R && python tools/sq_table.py $O/sq1.txt $O/sq2.txt 3072 > $O/pmc_sq_table.txt 2>&1; python tools/sq_lanes.py $O/sq3.txt 3072 > $O/pmc_sq_lanes.txt 2>&1; head -30 $O/pmc_sq_lanes.txt)
# BASELINE configs[3] (1280x960 / 2000 / 400): the same two counters through bench.py at 1024 frames
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_c4_$c; timeout 400 rocprofv3 --pmc $c -d $O/pmc_c4_$c -- python $R/bench.py --workload c4 --batch 1024 --steps 1 --warmup 1 --no-overlap --no-cpu-baseline --no-extras --no-profile --no-other-workloads > $O/pmc_c4_$c.log 2>&1
done
(cd $R && python tools/rocpd_pmc_summary.py $O/pmc_c4_FETCH_SIZE $O/pmc_fetch_c4.txt > /dev/null; python tools/rocpd_pmc_summary.py $O/pmc_c4_WRITE_SIZE $O/pmc_write_c4.txt > /dev/null
 python tools/make_pmc_traffic.py $O/pmc_c4_FETCH_SIZE $O/pmc_c4_WRITE_SIZE 1024 3 2 $O/pmc_traffic_c4.json | head -8; cp $O/pmc_traffic_c4.json $R/profiles/pmc_traffic_c4.json)
rm -rf $O/pmc_c4_FETCH_SIZE $O/pmc_c4_WRITE_SIZE
# kernel traces: the step on two streams and on one, single frames
rm -rf $O/kt; (cd $R && timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -- tools/step_check 12288 5 2 > $O/kt.log 2>&1; python tools/rocpd_summary.py $O/kt $O/kernel_trace_two_streams.txt > /dev/null; rm -rf $O/kt)
rm -rf $O/kt1; (cd $R && timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt1 -- tools/step_check 12288 3 1 1 > $O/kt1.log 2>&1; python tools/rocpd_summary.py $O/kt1 $O/kernel_trace_one_stream.txt > /dev/null; rm -rf $O/kt1)
rm -rf $O/lat; (cd $R && timeout 200 rocprofv3 --kernel-trace --stats -d $O/lat -- tools/lat_check 1 "" > $O/lat.log 2>&1; python tools/rocpd_summary.py $O/lat $O/kernel_trace_single_frame.txt > /dev/null; rm -rf $O/lat)
head -14 $O/kernel_trace_one_stream.txt
cd $R
STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_two_streams.txt 2>&1; STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream.txt 2>&1; cat $O/step_two_streams.txt $O/step_one_stream.txt
LAT_PROFILE=1 timeout 80 tools/lat_check 2 "" "SSLAM_NFA_STREAM=0" > $O/lat_check.txt 2>&1; LAT_W=1280 LAT_H=960 LAT_NF=8 LAT_LINES=400 LAT_FRAMES=tools/lat_frames_1280x960.raw LAT_EXPECTED=tools/lat_expected_1280x960.bin LAT_PROFILE=1 timeout 80 tools/lat_check 2 "" > $O/lat_check_1280.txt 2>&1
timeout 900 python -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -14 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_r05.json 2> $O/bench_r05.err; tail -c 300 $O/bench_r05.err
timeout 400 python tools/fuzz_parity.py 900 20260928 > $O/fuzz_parity.txt 2>&1; tail -4 $O/fuzz_parity.txt
timeout 300 python tools/fuzz_matchers.py 1200 > $O/fuzz_matchers.txt 2>&1; tail -3 $O/fuzz_matchers.txt
timeout 300 python tools/fuzz_parity.py 500 20260930 > $O/fuzz_parity_b.txt 2>&1; tail -4 $O/fuzz_parity_b.txt      # (a second seed, added for the last run of the round)
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05z/bench_r05.json').read().strip().splitlines()[-1])
    print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['latency']['lines_extract_hipEvent'])
    print({k: v for k, v in d['pcie_inclusive'].items() if 'per_s' in k}); print(d.get('other_workloads')); print(d['cpu_baseline']['value'], d['cpu_baseline']['parity_vs_gpu']); print(d.get('latency_nfa_behind_core'))
except Exception as e: print('bench failed', e)
PY
