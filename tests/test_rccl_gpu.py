"""GPU: the exchange step of the sharded mode over RCCL.  The GPU box has one device, so the launch is the
driver's own (`python -m torch.distributed.run --nproc-per-node 1 ... bench.py`) with SSLAM_FORCE_COLLECTIVE=1: the process
group is nccl (= RCCL), every step's packed records go through dist.gather on the device, and bench.py reports (config.gather_check) whether what
rank 0 receives equals its own packed results.  World sizes > 1 are covered on gloo in test_dist_cpu.py."""
import json, os, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_bench_under_torchrun_with_rccl_gather():
    env = dict(os.environ, SSLAM_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "64",
           "--no-cpu-baseline", "--no-other-workloads"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
    assert out["config"]["mean_keypoints"] > 500 and out["config"]["mean_lines"] > 50
    assert out["config"]["gather_check"] is True          # rank 0 compared what RCCL delivered with its own packed records
