cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/lat -o lat -- python tools/latency_probe.py > gpurun_out/lat.log 2>&1
tail -2 gpurun_out/lat.log
python tools/rocpd_summary.py gpurun_out/lat 2>/dev/null | head -30 || ls -R gpurun_out/lat | head
