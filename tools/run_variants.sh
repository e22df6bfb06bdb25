run() { timeout 900 python bench.py --batch $1 --steps 3 --warmup 1 --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$3', d['config']['batch_per_gpu'], round(d['value']), round(d['ms_per_step'],1), json.dumps({k:round(v,1) for k,v in list(d['roofline']['kernels_ms_per_step'].items())[:4]}))"; }
run 3072 --no-overlap w3-serial
run 3072 "" w3-overlap
run 2048 "" w3-overlap
SSLAM_EXTRA_FLAGS="-DSSLAM_LSD_MINWAVES=4" python structure-slam-pointline_amd/build.py --force > /dev/null 2>&1
run 4096 --no-overlap w4-serial
run 4096 "" w4-overlap
run 3072 "" w4-overlap
