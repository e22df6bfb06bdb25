#!/bin/bash
# The individual gpurun calls of round 4 besides the main recipe (tools/gpu_profile_r04.sh), as sections of one script:
#     gpurun --timeout 2400 -- 'bash tools/gpu_r04_experiments.sh <section>'
# Outputs land in gpurun_out/r04<section>/; the tables and logs that matter were copied to profiles/ (profiles/README.md, round 4).
set -x
R=$GRAFT_REPO_ROOT; cd $R
case "$1" in
a)
# round 4, first GPU call: fused NFA launch + k_keylines LDS fix -- parity (fused form forced on the small test batches), then A/B bench lines
O=$R/gpurun_out/r04a; mkdir -p $O
SSLAM_NFA_FUSED=2 timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_configs_gpu.py tests/test_batch_gpu.py tests/test_pin_gpu.py -x -q -m gpu > $O/pytest_fused.txt 2>&1; tail -5 $O/pytest_fused.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_fused.json 2> $O/bench_fused.err; tail -c 300 $O/bench_fused.err
SSLAM_NFA_FUSED=0 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_unfused.json 2>/dev/null
SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/nfa4.so timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_fused_mw4.json 2>/dev/null
timeout 600 python bench.py --no-overlap --no-cpu-baseline --no-extras > $O/bench_fused_one_stream.json 2>/dev/null
SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/nfa4.so timeout 600 python bench.py --no-overlap --no-cpu-baseline --no-extras > $O/bench_fused_mw4_one_stream.json 2>/dev/null
python - <<'PY'
import json
for n in ('bench_fused','bench_unfused','bench_fused_mw4','bench_fused_one_stream','bench_fused_mw4_one_stream'):
    try:
        d=json.load(open('gpurun_out/r04a/%s.json'%n)); k=d['roofline']['kernels_ms_per_step']
        print(n, round(d['value']), round(d['ms_per_step'],1), {a: round(b,1) for a,b in k.items() if b>0.5})
    except Exception as e: print(n, 'failed', e)
PY
;;
b)
# round 4, call b: the in-process RCCL stand-in (N > 1 group paths), the stress / fuzz tests of the driver-run suite, bench line with other_workloads
O=$R/gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_group_gpu.py tests/test_pin_gpu.py -x -q -m gpu > $O/pytest_group.txt 2>&1; tail -15 $O/pytest_group.txt
timeout 900 python -m pytest tests/test_stress_gpu.py -x -q -m gpu -s > $O/pytest_stress.txt 2>&1; tail -8 $O/pytest_stress.txt
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04b/bench.json'))
print(round(d['value']), d['ms_per_step'], d.get('latency',{}).get('lines_extract_hipEvent'), d.get('other_workloads'))
print(d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('parity_vs_gpu'), d.get('pcie_inclusive',{}).get('pinned_frames_per_s'))
PY
;;
c)
# round 4, call c: stand-in RCCL tests (fixed), stress / fuzz tests, NFA launch forms, batch SearchForInitialization in LDS, single-frame latency A/B
O=$R/gpurun_out/r04c; mkdir -p $O
timeout 420 python -m pytest tests/test_group_gpu.py -x -q -m gpu > $O/pytest_group.txt 2>&1; tail -4 $O/pytest_group.txt
timeout 600 python -m pytest tests/test_lines_gpu.py tests/test_match_gpu.py tests/test_shim_gpu.py tests/test_batch_gpu.py -x -q -m gpu > $O/pytest_a.txt 2>&1; tail -4 $O/pytest_a.txt
timeout 400 python -m pytest tests/test_stress_gpu.py -x -q -m gpu -s > $O/pytest_stress.txt 2>&1; tail -4 $O/pytest_stress.txt
for v in "def:" "old:SSLAM_NFA_FUSED=0" "old256:SSLAM_NFA_FUSED=0 SSLAM_COUNT_WAVES=256 SSLAM_EVAL_WAVES=64" "old128:SSLAM_NFA_FUSED=0 SSLAM_COUNT_WAVES=128 SSLAM_EVAL_WAVES=32" "wg8:SSLAM_NFA_WAVES=8"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 200 python tools/latency_probe.py > $O/lat_$n.txt 2>&1; tail -1 $O/lat_$n.txt
done
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-overlap > $O/bench_one_stream.json 2>/dev/null
SSLAM_SFI_BATCH=global timeout 300 python bench.py --no-cpu-baseline --no-extras --no-overlap > $O/bench_one_stream_sfi_global.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_two_streams.json 2>/dev/null
python - <<'PY'
import json
for n in ('bench_one_stream','bench_one_stream_sfi_global','bench_two_streams'):
    try:
        d=json.load(open('gpurun_out/r04c/%s.json'%n)); k=d['roofline']['kernels_ms_per_step']
        print(n, round(d['value']), round(d['ms_per_step'],1), {a: round(b,1) for a,b in k.items() if b>0.5})
    except Exception as e: print(n, 'failed', e)
PY
;;
d)
# round 4, call d: the sequential core at 7 / 8 waves per SIMD (72 / 64 VGPRs) against 6, at the bench's batch and at batches that fill the larger slot counts; group test re-run
O=$R/gpurun_out/r04d; mkdir -p $O
timeout 420 python -m pytest tests/test_group_gpu.py -x -q -m gpu > $O/pytest_group.txt 2>&1; tail -3 $O/pytest_group.txt
V=$R/structure-slam-pointline_amd/lib/variants
for v in "w6:" "w7:SSLAM_LIB=$V/core7.so" "w8:SSLAM_LIB=$V/core8.so"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_$n.json 2>/dev/null
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extras --no-overlap > $O/bench_${n}_one.json 2>/dev/null
done
SSLAM_LIB=$V/core7.so timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 14336 > $O/bench_w7_B14336.json 2>/dev/null
SSLAM_LIB=$V/core8.so timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 16384 > $O/bench_w8_B16384.json 2>/dev/null
SSLAM_LIB=$V/core7.so timeout 300 python -m pytest tests/test_lines_gpu.py tests/test_batch_gpu.py -x -q -m gpu > $O/pytest_w7.txt 2>&1; tail -2 $O/pytest_w7.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04d/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['roofline']['kernels_ms_per_step']
        print(f.split('/')[-1], d['config']['batch_per_gpu'], round(d['value']), round(d['ms_per_step'],1), 'core', round(k.get('k_lsd_regions',0),1))
    except Exception as e: print(f, 'failed', e)
PY
;;
e)
# round 4, call e: the whole GPU suite on the final tree (log kept in profiles/), smoke
O=$R/gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=5 > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
;;
f)
# round 4, call f: the long randomised sweeps on the final kernels (logs -> profiles/r04_fuzz_*)
O=$R/gpurun_out/r04f; mkdir -p $O
timeout 900 python tools/fuzz_parity.py 1500 20260926 > $O/fuzz_parity_1500.txt 2>&1; tail -2 $O/fuzz_parity_1500.txt
timeout 600 python tools/fuzz_matchers.py > $O/fuzz_matchers.txt 2>&1; tail -2 $O/fuzz_matchers.txt
timeout 600 python tools/fuzz_reuse.py > $O/fuzz_reuse.txt 2>&1; tail -1 $O/fuzz_reuse.txt
timeout 300 python tools/cl_stress.py 300 6 > $O/cl_stress.txt 2>&1; tail -1 $O/cl_stress.txt
timeout 300 python tools/bench_matchers.py > $O/matchers.txt 2>&1; tail -14 $O/matchers.txt
;;
g)
# round 4, call g: the Gaussian-variant tests + the suites their kernels touch
O=$R/gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_lines_gpu.py tests/test_configs_gpu.py tests/test_batch_gpu.py tests/test_shim_gpu.py tests/test_pin_gpu.py tests/test_group_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
SSLAM_ORB_BLUR_VARIANT=1 timeout 300 python -m pytest tests/test_shim_gpu.py -x -q -m gpu -k end_to_end > $O/pytest_shim_v1.txt 2>&1; tail -3 $O/pytest_shim_v1.txt
timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value']), d['ms_per_step'])"
;;
h)
# round 4, call h: the counting sort on per-segment lists of defined pixels (k_lsd_grad / k_lsd_hist / k_lsd_scatter): parity, then the two bench lines
O=$R/gpurun_out/r04h; mkdir -p $O
timeout 700 python -m pytest tests/test_lines_gpu.py tests/test_configs_gpu.py tests/test_batch_gpu.py tests/test_edge_gpu.py tests/test_pin_gpu.py -x -q -m gpu > $O/pytest_lines.txt 2>&1; tail -6 $O/pytest_lines.txt
timeout 400 python bench.py --no-cpu-baseline --no-extras --no-other-workloads > $O/bench_two_streams.json 2> $O/bench.err; tail -c 300 $O/bench.err
timeout 400 python bench.py --no-cpu-baseline --no-extras --no-other-workloads --no-overlap > $O/bench_one_stream.json 2>/dev/null
python - <<'PY'
import json
for n in ('bench_two_streams','bench_one_stream'):
    try:
        d=json.load(open('gpurun_out/r04h/%s.json'%n)); k=d['roofline']['kernels_ms_per_step']
        print(n, round(d['value']), round(d['ms_per_step'],1), {a: round(b,2) for a,b in k.items() if b>0.3})
    except Exception as e: print(n, 'failed', e)
PY
;;
i)
# round 4, call i: k_lsd_grad variants (tools/build_variant.sh NAME FLAGS): whole-block stores, occupancy, rows per wave -- one-stream per-kernel times
O=$R/gpurun_out/r04i; mkdir -p $O
for n in base gw g8w g6w gr4 gr16 gw8; do
  L=$R/structure-slam-pointline_amd/lib/variants/$n.so; [ $n = base ] && L=$R/structure-slam-pointline_amd/lib/libsslam_frontend.so
  [ -f $L ] || continue
  SSLAM_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-extras --no-other-workloads --no-overlap --steps 3 --warmup 1 > $O/bench_$n.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open('$O/bench_$n.json')); k=d['roofline']['kernels_ms_per_step']
    print('$n', round(d['value']), round(d['ms_per_step'],1), {a: round(k[a],2) for a in ('k_blur7','k_lsd_grad','k_lsd_hist','k_lsd_scan','k_lsd_scatter','k_lsd_regions')})
except Exception as e: print('$n', 'failed', e)
PY
done 2>&1 | tee $O/summary.txt
;;
j)
# round 4, call j: FETCH_SIZE / WRITE_SIZE of the kernels after the segment-list prologue (regenerates profiles/pmc_traffic.json's numbers)
O=$R/gpurun_out/r04j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
b=6144
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_c3_$c && timeout 600 rocprofv3 --pmc $c -d $O/pmc_c3_$c -- python $R/bench.py --workload c3 --batch $b --steps 1 --warmup 1 --no-overlap --no-cpu-baseline --no-extras --no-profile --no-other-workloads > $O/pmc_c3_$c.log 2>&1
done
cd $R
python tools/rocpd_pmc_summary.py $O/pmc_c3_FETCH_SIZE $O/pmc_fetch_c3.txt > /dev/null
python tools/rocpd_pmc_summary.py $O/pmc_c3_WRITE_SIZE $O/pmc_write_c3.txt > /dev/null
python tools/make_pmc_traffic.py $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE $b 3 2 $O/pmc_traffic_c3.json | head -60
rm -rf $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE
;;
k)
# round 4, call k: the core capped at 5 / 4 waves per SIMD (LDS padding) so that point-branch waves find slots during its first round -- two-stream bench lines
O=$R/gpurun_out/r04k; mkdir -p $O
for pad in 0 8192 10240; do
  SSLAM_LSD_LDS_PAD=$pad timeout 300 python bench.py --no-cpu-baseline --no-extras --no-other-workloads --steps 3 --warmup 1 > $O/bench_pad$pad.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open('$O/bench_pad$pad.json')); k=d['roofline']['kernels_ms_per_step']
    print('pad $pad', round(d['value']), round(d['ms_per_step'],1), {a: round(b,1) for a,b in k.items() if b>3})
except Exception as e: print('pad $pad', 'failed', e)
PY
done 2>&1 | tee $O/summary.txt
;;
l)
# round 4, call l: persistent workgroups for the sequential core (SSLAM_LSD_PERSIST=G): G < wave slots leaves room for the point branch during the whole launch
O=$R/gpurun_out/r04l; mkdir -p $O
for g in ${GRIDS:-0 5120 4096 3072 2048}; do
  SSLAM_LSD_PERSIST=$g timeout 300 python bench.py --no-cpu-baseline --no-extras --no-other-workloads --steps 3 --warmup 1 > $O/bench_g$g.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open('$O/bench_g$g.json')); k=d['roofline']['kernels_ms_per_step']
    print('grid $g', round(d['value']), round(d['ms_per_step'],1), {a: round(b,1) for a,b in k.items() if b>3})
except Exception as e: print('grid $g', 'failed', e)
PY
done 2>&1 | tee $O/summary.txt
;;
m)
# round 4, call m: batch knn-2 on the matrix cores (k_knn2_expand + k_knn2_mfma) against the xor + popcount form: parity, then bench lines
O=$R/gpurun_out/r04m; mkdir -p $O
timeout 600 python -m pytest tests/test_match_gpu.py tests/test_batch_gpu.py -x -q -m gpu -k "knn2 or batch" > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
SSLAM_KNN2_BATCH=mfma2 timeout 600 python -m pytest tests/test_match_gpu.py tests/test_batch_gpu.py -x -q -m gpu -k "knn2 or batch" > $O/pytest_mfma2.txt 2>&1; tail -6 $O/pytest_mfma2.txt
for v in mfma mfma2 popc; do
  for ov in "" "--no-overlap"; do
    SSLAM_KNN2_BATCH=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --no-other-workloads --steps 3 --warmup 1 $ov > $O/bench_$v$ov.json 2>/dev/null
    python - <<PY
import json
try:
    d=json.load(open('$O/bench_$v$ov.json')); k=d['roofline']['kernels_ms_per_step']
    print('$v $ov', round(d['value']), round(d['ms_per_step'],1), {a: round(b,2) for a,b in k.items() if 'knn2' in a or b>20})
except Exception as e: print('$v $ov', 'failed', e)
PY
  done
done 2>&1 | tee $O/summary.txt
;;
n)
# round 4, call n: the horizontal blur pass of k_describe on the matrix cores: ORB parity (both blur variants, the blur stage tap), bench lines
O=$R/gpurun_out/r04n; mkdir -p $O
timeout 600 python -m pytest tests/test_orb_gpu.py tests/test_batch_gpu.py tests/test_pin_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
for v in mfma dnomfma; do
 L=$R/structure-slam-pointline_amd/lib/libsslam_frontend.so; [ $v = dnomfma ] && L=$R/structure-slam-pointline_amd/lib/variants/dnomfma.so
 for ov in "" "--no-overlap"; do
  SSLAM_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-extras --no-other-workloads --steps 3 --warmup 1 $ov > $O/bench_$v$ov.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open('$O/bench_$v$ov.json')); k=d['roofline']['kernels_ms_per_step']
    print('$v $ov', round(d['value']), round(d['ms_per_step'],1), {a: round(b,2) for a,b in k.items() if b>3})
except Exception as e: print('$v $ov', 'failed', e)
PY
 done
done 2>&1 | tee $O/summary.txt
;;
o)
# round 4, call o: where k_blur_sobel runs (prologue / behind the NFA stage) x where the point branch waits for the core event (before the pyramid / behind it)
O=$R/gpurun_out/r04o; mkdir -p $O
for sob in early late; do for pc in 1 pyr; do
  SSLAM_LBD_SOBEL=$sob SSLAM_POINTS_AT_CORE=$pc timeout 300 python bench.py --no-cpu-baseline --no-extras --no-other-workloads --steps 3 --warmup 1 > $O/bench_${sob}_$pc.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open('$O/bench_${sob}_$pc.json')); k=d['roofline']['kernels_ms_per_step']
    print('sobel $sob points $pc', round(d['value']), round(d['ms_per_step'],1), {a: round(b,1) for a,b in k.items() if b>3})
except Exception as e: print('sobel $sob points $pc', 'failed', e)
PY
done; done 2>&1 | tee $O/summary.txt
;;
p)
# round 4, call p: divisions out of k_fast_cells / k_resize, packed fp32 sums in k_lbd: parity of both extractors, bench lines
O=$R/gpurun_out/r04p; mkdir -p $O
timeout 700 python -m pytest tests/test_orb_gpu.py tests/test_lines_gpu.py tests/test_batch_gpu.py tests/test_configs_gpu.py tests/test_edge_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
for ov in "" "--no-overlap"; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-other-workloads --steps 3 --warmup 1 $ov > $O/bench$ov.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open('$O/bench$ov.json')); k=d['roofline']['kernels_ms_per_step']
    print('$ov', round(d['value']), round(d['ms_per_step'],1), {a: round(b,2) for a,b in k.items() if b>3})
except Exception as e: print('$ov', 'failed', e)
PY
done 2>&1 | tee $O/summary.txt
;;
q)
# round 4, call q: frames per step (rounds of the sequential core per launch): 12288 (two rounds of 6144 wave slots), 18432, 24576
O=$R/gpurun_out/r04q; mkdir -p $O
for b in ${BATCHES:-12288 18432 24576}; do
  timeout 400 python bench.py --no-cpu-baseline --no-extras --no-other-workloads --steps 3 --warmup 1 --batch $b > $O/bench_b$b.json 2> $O/bench_b$b.err
  python - <<PY
import json
try:
    d=json.load(open('$O/bench_b$b.json')); k=d['roofline']['kernels_ms_per_step']
    print('batch $b', round(d['value']), round(d['ms_per_step'],1), {a: round(v,1) for a,v in k.items() if v>8})
except Exception as e: print('batch $b', 'failed', e, open('$O/bench_b$b.err').read()[-300:])
PY
done 2>&1 | tee $O/summary.txt
rocm-smi --showmeminfo vram 2>/dev/null | tail -4
;;
r)
# round 4, call r: k_blur7 / k_blur_sobel with branch-free row loads (ROW_AHEAD rows requested ahead) and 24-bit multiplies: parity, one-stream kernel times per variant
O=$R/gpurun_out/r04r; mkdir -p $O
timeout 600 python -m pytest tests/test_lines_gpu.py tests/test_orb_gpu.py tests/test_edge_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for n in base ra1 ra2; do
  L=$R/structure-slam-pointline_amd/lib/variants/$n.so; [ $n = base ] && L=$R/structure-slam-pointline_amd/lib/libsslam_frontend.so
  [ -f $L ] || continue
  SSLAM_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-extras --no-other-workloads --no-overlap --steps 3 --warmup 1 > $O/bench_$n.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open('$O/bench_$n.json')); k=d['roofline']['kernels_ms_per_step']
    print('$n', round(d['value']), round(d['ms_per_step'],1), {a: round(k[a],2) for a in ('k_blur7','k_blur_sobel','k_lsd_grad','k_describe','k_lbd')})
except Exception as e: print('$n', 'failed', e)
PY
done 2>&1 | tee $O/summary.txt
;;
s)
# round 4, call s: lines.hip compiled with -mllvm -disable-machine-licm (constants rematerialised in loops instead of hoisted, spilled and reloaded from scratch):
# parity with that library, kernel times against the default build, single-frame latency
O=$R/gpurun_out/r04s; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants/nolicm.so
SSLAM_LIB=$V timeout 600 python -m pytest tests/test_lines_gpu.py tests/test_configs_gpu.py tests/test_batch_gpu.py tests/test_edge_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for n in base nolicm; do
  L=$V; [ $n = base ] && L=$R/structure-slam-pointline_amd/lib/libsslam_frontend.so
  for ov in "--no-overlap" ""; do
  SSLAM_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-extras --no-other-workloads $ov --steps 3 --warmup 1 > $O/bench_$n$ov.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open('$O/bench_$n$ov.json')); k=d['roofline']['kernels_ms_per_step']
    print('$n $ov', round(d['value']), round(d['ms_per_step'],1), {a: round(v,2) for a,v in k.items() if a in ('k_lsd_regions','k_nfa_all','k_lsd_grad','k_lbd','k_blur_sobel','k_blur7','k_lsd_scatter','k_keylines')})
except Exception as e: print('$n', 'failed', e)
PY
  done
  SSLAM_LIB=$L timeout 200 python tools/latency_probe.py 2>&1 | tail -1 | cut -c1-200
done 2>&1 | tee $O/summary.txt
;;
t)
# round 4, call t: Cs and S of a pixel in one 16-byte record (SSLAM_LSD_PACKED, default) against the two planes (variant `unpacked`): parity, kernel times, latency
O=$R/gpurun_out/r04t; mkdir -p $O
timeout 600 python -m pytest tests/test_lines_gpu.py tests/test_configs_gpu.py tests/test_batch_gpu.py tests/test_edge_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for n in base unpacked; do
  L=$R/structure-slam-pointline_amd/lib/variants/$n.so; [ $n = base ] && L=$R/structure-slam-pointline_amd/lib/libsslam_frontend.so
  for ov in "--no-overlap" ""; do
  SSLAM_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-extras --no-other-workloads $ov --steps 3 --warmup 1 > $O/bench_$n$ov.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open('$O/bench_$n$ov.json')); k=d['roofline']['kernels_ms_per_step']
    print('$n $ov', round(d['value']), round(d['ms_per_step'],1), {a: round(v,2) for a,v in k.items() if a in ('k_lsd_regions','k_nfa_all','k_lsd_grad','k_lsd_hist','k_lsd_scatter')})
except Exception as e: print('$n', 'failed', e)
PY
  done
  SSLAM_LIB=$L timeout 200 python tools/latency_probe.py 2>&1 | tail -1 | cut -c1-200
done 2>&1 | tee $O/summary.txt
;;
*) echo "usage: $0 {a|b|c|d|e|f|g|h|i|j|k|l|m|n|o|p|q|r|s|t}"; exit 2 ;;
esac
