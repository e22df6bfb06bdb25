run() { timeout 900 python bench.py --batch $1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['config']['batch_per_gpu'], round(d['value']), round(d['ms_per_step'],1), json.dumps({k:round(v,1) for k,v in list(d['roofline']['kernels_ms_per_step'].items())[:2]}))"; }
SSLAM_EXTRA_FLAGS="-DSSLAM_LSD_MINWAVES=7" python structure-slam-pointline_amd/build.py --force > /dev/null 2>&1
run 7168 "minwaves7"
SSLAM_EXTRA_FLAGS="-DSSLAM_LSD_MINWAVES=8" python structure-slam-pointline_amd/build.py --force > /dev/null 2>&1
run 8192 "minwaves8"
