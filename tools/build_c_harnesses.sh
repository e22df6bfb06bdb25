#!/bin/bash
# CPU side of the C-only GPU harnesses (tools/lat_check.c, mix_check.c, batch_check.c, nfa_stream_check.c, step_check.c): compile them against the in-tree library and prepare their inputs
# (the bench's frames + the CPU oracle's lines for them).  Binaries and inputs are git-ignored and travel to the GPU box with the gpurun snapshot; on the box they start in a
# second (no Python, no `import torch`), so a gpurun call that runs them costs 10-25 s of GPU time.
#     bash tools/build_c_harnesses.sh && gpurun --timeout 120 -- 'tools/lat_check 2 "" "SSLAM_NFA_STREAM=1"'
set -e
cd "$(dirname "$0")/.."
python structure-slam-pointline_amd/build.py > /dev/null
L=(-Iinclude -Lstructure-slam-pointline_amd/lib -lsslam_frontend '-Wl,-rpath,$ORIGIN/../structure-slam-pointline_amd/lib')
for t in lat_check mix_check nfa_stream_check; do gcc -O2 -Wall -Wno-misleading-indentation tools/$t.c "${L[@]}" -o tools/$t; done
gcc -O2 -Wall -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/batch_check.c "${L[@]}" -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,/opt/rocm/lib -o tools/batch_check
gcc -O2 -Wall -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/step_check.c "${L[@]}" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o tools/step_check
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe
[ -f tools/step_frames.raw ] && [ -f tools/step_expected_orb.bin ] || python tools/step_check_prepare.py
[ -f tools/lat_frames.raw ] && [ -f tools/lat_expected.bin ] || python tools/lat_check_prepare.py
[ -f tools/lat_frames_1280x960.raw ] || python tools/lat_check_prepare.py 1280 960 8 400
[ -f tools/mix_frames.bin ] || python tools/mix_check_prepare.py > /dev/null
ls -la tools/lat_check tools/mix_check tools/batch_check tools/nfa_stream_check tools/step_check tools/gather_probe tools/*.raw tools/*.bin
