#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_batch_gpu.py tests/test_configs_gpu.py tests/test_edge_gpu.py -x -q -m gpu 2>&1 | tail -2
