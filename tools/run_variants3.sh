run() { timeout 900 python bench.py --batch $1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['config']['batch_per_gpu'], round(d['value']), round(d['ms_per_step'],1), json.dumps({k:round(v,1) for k,v in list(d['roofline']['kernels_ms_per_step'].items())[:2]}))"; }
for v in "0 0" "0 1" "1 0" "1 1"; do set -- $v
  SSLAM_EXTRA_FLAGS="-DSSLAM_V_CANDRAD=$1 -DSSLAM_V_READLANE=$2" python structure-slam-pointline_amd/build.py --force > /dev/null 2>&1
  run 6144 "candrad=$1 readlane=$2"
done
SSLAM_EXTRA_FLAGS="-DSSLAM_LSD_MINWAVES=5" python structure-slam-pointline_amd/build.py --force > /dev/null 2>&1
run 5120 "minwaves5 B5120"
run 6144 "minwaves5 B6144"
SSLAM_EXTRA_FLAGS="-DSSLAM_LSD_MINWAVES=8" python structure-slam-pointline_amd/build.py --force > /dev/null 2>&1
run 8192 "minwaves8 B8192"
