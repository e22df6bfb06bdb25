#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_match_gpu.py -x -q -m gpu -k "bow" 2>&1 | tail -2
timeout 300 python tools/bench_matchers.py 2>&1 | grep -i "bow" | head -4
echo "== two launches"; SSLAM_BOW_FORM=two timeout 300 python tools/bench_matchers.py 2>&1 | grep -i "SearchByBoW" | head -2
