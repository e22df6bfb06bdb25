#!/bin/bash
# Copies what tools/gpu_profile_r06_final.sh left under gpurun_out/r06final/ (scratch) into profiles/ (tracked) under the names profiles/README.md lists.
#     bash tools/collect_r06_final.sh [previous-bench-name]     (the previous profiles/r06_final_bench_B12288_two_streams.json is kept under that name when given)
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r06final; P=profiles
[ -n "$1" ] && git mv $P/r06_final_bench_B12288_two_streams.json $P/$1
tail -1 $O/bench_r06.json > $P/r06_final_bench_B12288_two_streams.json
cp $O/pmc_traffic_c3.json $P/pmc_traffic.json; cp $O/pmc_traffic_c4.json $P/pmc_traffic_c4.json; cp $O/sq_instr.json $P/sq_instr.json; cp $O/gather_probe.json $P/random_sector.json
cp $O/pmc_fetch_c3.txt $P/r06_final_pmc_fetch_c3_B12288.txt; cp $O/pmc_write_c3.txt $P/r06_final_pmc_write_c3_B12288.txt
cp $O/pmc_fetch_c4.txt $P/r06_final_pmc_fetch_c4_1280x960_B1024.txt; cp $O/pmc_write_c4.txt $P/r06_final_pmc_write_c4_1280x960_B1024.txt
cp $O/sq1.txt $P/r06_final_pmc_sq_pass1.txt; cp $O/sq2.txt $P/r06_final_pmc_sq_pass2.txt; cp $O/sq3.txt $P/r06_final_pmc_sq_pass3_lanes.txt
cp $O/pmc_sq_table.txt $P/r06_final_pmc_sq_table.txt; cp $O/tcp_table.txt $P/r06_final_tcp_table.txt
for f in kernel_trace_one_stream kernel_trace_two_streams kernel_trace_single_frame timeline_two_streams step_one_stream step_two_streams step_two_streams_round5_schedule lat_check lat_check_1280 pytest_gpu fuzz_parity fuzz_matchers; do cp $O/$f.txt $P/r06_final_$f.txt; done
ls -la $P/r06_final_bench_B12288_two_streams.json $P/sq_instr.json
