run() { timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],1), json.dumps({k:round(v,2) for k,v in d['roofline']['kernels_ms_per_step'].items() if 'nfa' in k}))"; }
for cw in 1 4 8 16; do SSLAM_COUNT_WAVES=$cw run "count=$cw"; done
for ew in 2 4; do SSLAM_EVAL_WAVES=$ew run "eval=$ew"; done
