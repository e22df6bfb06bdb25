#!/bin/bash
# round 4, call g: the Gaussian-variant tests + the suites their kernels touch
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_lines_gpu.py tests/test_configs_gpu.py tests/test_batch_gpu.py tests/test_shim_gpu.py tests/test_pin_gpu.py tests/test_group_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
SSLAM_ORB_BLUR_VARIANT=1 timeout 300 python -m pytest tests/test_shim_gpu.py -x -q -m gpu -k end_to_end > $O/pytest_shim_v1.txt 2>&1; tail -3 $O/pytest_shim_v1.txt
timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value']), d['ms_per_step'])"
