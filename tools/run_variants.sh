run() { timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],1), json.dumps({k:round(v,2) for k,v in d['roofline']['kernels_ms_per_step'].items() if 'nfa' in k}))"; }
for mw in 5 6; do
SSLAM_EXTRA_FLAGS="-DSSLAM_COUNT_MINWAVES=$mw" python structure-slam-pointline_amd/build.py --force > /dev/null 2>&1
run "count_minwaves=$mw"
done
