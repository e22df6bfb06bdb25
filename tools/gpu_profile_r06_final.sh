#!/bin/bash
# Round-6 measurement recipe on the final tree (through gpurun).  Counters in their own passes (no trace domains mixed in), through the C harness (tools/step_check: the bench's
# step without Python) in the SAME two-stream form the bench times, so that the counted kernels are the launch forms of the bench line (the guest form of the sequential core
# included): FETCH_SIZE / WRITE_SIZE -> profiles/pmc_traffic.json, three SQ passes -> profiles/sq_instr.json (what roofline.issue prices), two TCP passes (L2 reads per frame:
# roofline.random_sector); the same two traffic counters for BASELINE configs[3] through bench.py; kernel traces + the TIMELINE of one step; then the GPU suite, the bench line and
# the fuzzers.  Every step under its own timeout.
#     bash tools/build_c_harnesses.sh && gpurun --timeout 2400 -- 'bash tools/gpu_profile_r06_final.sh'
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
b=12288
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_c3_$c; (cd $R && timeout 300 rocprofv3 --pmc $c -d $O/pmc_c3_$c -- tools/step_check $b 1 1 > $O/pmc_c3_$c.log 2>&1)
done
(cd $R && python tools/rocpd_pmc_summary.py $O/pmc_c3_FETCH_SIZE $O/pmc_fetch_c3.txt > /dev/null; python tools/rocpd_pmc_summary.py $O/pmc_c3_WRITE_SIZE $O/pmc_write_c3.txt > /dev/null
 python tools/make_pmc_traffic.py $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE $b 3 2 $O/pmc_traffic_c3.json 1 | head -30; cp $O/pmc_traffic_c3.json $R/profiles/pmc_traffic.json)
rm -rf $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
P3="SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf $O/sq$i
  (cd $R && timeout 300 rocprofv3 --pmc $P -d $O/sq$i -- tools/step_check $b 1 0 > $O/sq$i.log 2>&1; python tools/rocpd_pmc_summary.py $O/sq$i $O/sq$i.txt > /dev/null; rm -rf $O/sq$i)
done
T1="TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
T2="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"
i=0
for P in "$T1" "$T2"; do
  i=$((i+1)); rm -rf $O/tcp$i
  (cd $R && timeout 300 rocprofv3 --pmc $P -d $O/tcp$i -- tools/step_check $b 1 0 > $O/tcp$i.log 2>&1; python tools/rocpd_pmc_summary.py $O/tcp$i $O/tcp$i.txt > /dev/null; rm -rf $O/tcp$i)
done
(cd $R && python tools/sq_table5.py $O/sq1.txt $O/sq2.txt $O/sq3.txt $b > $O/pmc_sq_table.txt 2>&1; python tools/tcp_table.py $O/tcp1.txt $O/tcp2.txt $b > $O/tcp_table.txt 2>&1; head -24 $O/pmc_sq_table.txt; head -8 $O/tcp_table.txt
 python tools/make_sq_instr.py $O/pmc_sq_table.txt $b $O/sq_instr.json "profiles/r06_pmc_sq_table.txt + profiles/r06_tcp_table.txt (round-6 final tree: rocprofv3 --pmc SQ / TCP passes over tools/step_check $b 1 0, the two-stream step: the launch forms the bench times)" $O/tcp_table.txt && cp $O/sq_instr.json $R/profiles/sq_instr.json)
# BASELINE configs[3] (1280x960 / 2000 / 400): the two traffic counters through bench.py at 1024 frames
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_c4_$c; timeout 400 rocprofv3 --pmc $c -d $O/pmc_c4_$c -- python $R/bench.py --workload c4 --batch 1024 --steps 1 --warmup 1 --no-overlap --no-cpu-baseline --no-extras --no-profile --no-other-workloads > $O/pmc_c4_$c.log 2>&1
done
(cd $R && python tools/rocpd_pmc_summary.py $O/pmc_c4_FETCH_SIZE $O/pmc_fetch_c4.txt > /dev/null; python tools/rocpd_pmc_summary.py $O/pmc_c4_WRITE_SIZE $O/pmc_write_c4.txt > /dev/null
 python tools/make_pmc_traffic.py $O/pmc_c4_FETCH_SIZE $O/pmc_c4_WRITE_SIZE 1024 3 2 $O/pmc_traffic_c4.json | head -8; cp $O/pmc_traffic_c4.json $R/profiles/pmc_traffic_c4.json)
rm -rf $O/pmc_c4_FETCH_SIZE $O/pmc_c4_WRITE_SIZE
# kernel traces: the step on two streams (+ its timeline) and on one, single frames
rm -rf $O/kt1; (cd $R && timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt1 -- tools/step_check 12288 3 1 1 > $O/kt1.log 2>&1; python tools/rocpd_summary.py $O/kt1 $O/kernel_trace_one_stream.txt > /dev/null; rm -rf $O/kt1)
rm -rf $O/kt; (cd $R && timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -- tools/step_check 12288 5 2 > $O/kt.log 2>&1; python tools/rocpd_summary.py $O/kt $O/kernel_trace_two_streams.txt > /dev/null; python tools/rocpd_timeline.py $O/kt $O/kernel_trace_one_stream.txt $O/timeline_two_streams.txt > /dev/null; rm -rf $O/kt)
rm -rf $O/lat; (cd $R && timeout 200 rocprofv3 --kernel-trace --stats -d $O/lat -- tools/lat_check 1 "" > $O/lat.log 2>&1; python tools/rocpd_summary.py $O/lat $O/kernel_trace_single_frame.txt > /dev/null; rm -rf $O/lat)
head -14 $O/kernel_trace_one_stream.txt; cat $O/timeline_two_streams.txt
cd $R
STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_two_streams.txt 2>&1; STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream.txt 2>&1; cat $O/step_two_streams.txt $O/step_one_stream.txt
STEP_GATE=core SSLAM_LSD_GUEST=0 STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_two_streams_round5_schedule.txt 2>&1; head -2 $O/step_two_streams_round5_schedule.txt
timeout 60 tools/gather_probe 16 6 > $O/gather_probe.json 2>&1; cat $O/gather_probe.json; cp $O/gather_probe.json $R/profiles/random_sector.json
LAT_PROFILE=1 timeout 80 tools/lat_check 2 "" "SSLAM_NFA_STREAM=0" > $O/lat_check.txt 2>&1; LAT_W=1280 LAT_H=960 LAT_NF=8 LAT_LINES=400 LAT_FRAMES=tools/lat_frames_1280x960.raw LAT_EXPECTED=tools/lat_expected_1280x960.bin LAT_PROFILE=1 timeout 80 tools/lat_check 2 "" > $O/lat_check_1280.txt 2>&1
timeout 1200 python -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -14 $O/pytest_gpu.txt
timeout 1200 python bench.py > $O/bench_r06.json 2> $O/bench_r06.err; tail -c 300 $O/bench_r06.err
timeout 400 python tools/fuzz_parity.py 900 20261001 > $O/fuzz_parity.txt 2>&1; tail -4 $O/fuzz_parity.txt
timeout 300 python tools/fuzz_matchers.py 1200 > $O/fuzz_matchers.txt 2>&1; tail -3 $O/fuzz_matchers.txt
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r06final/bench_r06.json').read().strip().splitlines()[-1])
    print(round(d['value']), d['ms_per_step'], d['roofline']['bound'], d['roofline']['frac'], d['roofline']['traffic'], d['latency']['lines_extract_hipEvent'])
    print(d['roofline']['bound_note']); print({k: v for k, v in d['pcie_inclusive'].items() if 'per_s' in k}); print({k: (v.get('value'), v.get('ms_per_step')) for k, v in d.get('other_workloads', {}).items()})
    print(d['cpu_baseline']['value'], d['cpu_baseline']['ms_per_frame'], d['cpu_baseline']['parity_vs_gpu']); print(d.get('latency_nfa_behind_core'))
except Exception as e: print('bench failed', e)
PY
