#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value']), round(d['ms_per_step'],1))"; }
run default A=1
run defer SSLAM_DEFER_POINT_MATCH=1
run default2 A=1
run defer2 SSLAM_DEFER_POINT_MATCH=1
