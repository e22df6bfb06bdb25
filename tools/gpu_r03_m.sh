#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
