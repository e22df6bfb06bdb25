run() { timeout 900 python bench.py --batch $2 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['config']['batch_per_gpu'], round(d['value']), round(d['ms_per_step'],1), json.dumps({k:round(v,1) for k,v in list(d['roofline']['kernels_ms_per_step'].items())[:4]}))"; }
b() { SSLAM_EXTRA_FLAGS="$1" python structure-slam-pointline_amd/build.py --force > /dev/null 2>&1; }
b "-DSSLAM_LSD_MINWAVES=5 -DSSLAM_LSD_QCAP=2048"; run mw5-q2048 4096; run mw5-q2048 5120
b "-DSSLAM_LSD_MINWAVES=6 -DSSLAM_LSD_QCAP=1024"; run mw6-q1024 6144
b "-DSSLAM_LSD_MINWAVES=8 -DSSLAM_LSD_QCAP=1024"; run mw8-q1024 8192
