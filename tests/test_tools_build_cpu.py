"""The C-only GPU harnesses under tools/ (no Python on the GPU box: a gpurun call that runs them costs 10-25 s) must keep compiling against include/sslam_frontend.h and
link against the in-tree library -- they are built here, on the CPU, and travel with the snapshot (tools/build_c_harnesses.sh)."""
import os, subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "structure-slam-pointline_amd", "lib")


@pytest.mark.parametrize("name,hip", [("lat_check", False), ("mix_check", False), ("nfa_stream_check", False), ("batch_check", True), ("step_check", True)])
def test_c_harness_compiles_and_links(tmp_path, name, hip):
    if not os.path.exists(os.path.join(LIBDIR, "libsslam_frontend.so")):
        import importlib.util
        spec = importlib.util.spec_from_file_location("sslam_build", os.path.join(ROOT, "structure-slam-pointline_amd", "build.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
        b.build(verbose=False)
    cmd = ["gcc", "-O1", "-Wall", "-Werror", "-Wno-misleading-indentation", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", name + ".c"), "-L" + LIBDIR, "-lsslam_frontend",
           "-Wl,-rpath," + LIBDIR, "-o", str(tmp_path / name)]
    if hip:
        if not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
            pytest.skip("no HIP headers")
        cmd[1:1] = ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]; cmd += ["-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    # without a GPU (or without its input files) it must say so and fail, not crash
    r = subprocess.run([str(tmp_path / name)], cwd=str(tmp_path), capture_output=True, text=True, timeout=60)
    assert r.returncode not in (0, -11, -6), (r.returncode, r.stderr[-300:])
    assert r.stderr.strip(), "no diagnostic"
