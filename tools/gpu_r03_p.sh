#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python tools/two_pipe_probe.py 1 12288 4 0 2>&1 | tail -1
timeout 300 python tools/two_pipe_probe.py 2 6144 4 1 2>&1 | tail -1
timeout 300 python tools/two_pipe_probe.py 2 6144 4 0 2>&1 | tail -1
timeout 300 python tools/two_pipe_probe.py 3 4096 4 1 2>&1 | tail -1
