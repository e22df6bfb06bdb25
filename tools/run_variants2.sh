run() { timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['config']['batch_per_gpu'], round(d['value']), round(d['ms_per_step'],1), json.dumps({k:round(v,1) for k,v in list(d['roofline']['kernels_ms_per_step'].items())[:4]}))"; }
python structure-slam-pointline_amd/build.py > /dev/null 2>&1
run w3-serial --no-overlap
run w3-overlap ""
SSLAM_EXTRA_FLAGS="-DSSLAM_LSD_MINWAVES=4" python structure-slam-pointline_amd/build.py --force > /dev/null 2>&1
run w4-serial --no-overlap
run w4-overlap ""
export SSLAM_LSD_LDS_PAD=12288
run w4-pad12k-overlap ""
export SSLAM_LSD_LDS_PAD=16384
run w4-pad16k-overlap ""
run w4-pad16k-serial --no-overlap
