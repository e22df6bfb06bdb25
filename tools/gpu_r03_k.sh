#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extras --no-profile > $O/$name.json 2>> $O/err.txt; python -c "
import json; d=json.load(open('$O/$name.json')); print('$name', round(d['value']), round(d['ms_per_step'],1))"; }
run default A=1
run prio_line SSLAM_LINE_STREAM_PRIORITY=-1
run prio_line_start SSLAM_LINE_STREAM_PRIORITY=-1 SSLAM_POINTS_AT_CORE=0
run start_together SSLAM_POINTS_AT_CORE=0
