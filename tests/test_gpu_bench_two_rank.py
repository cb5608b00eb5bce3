"""bench.py --gpus 2 exactly as the driver launches it (python -m torch.distributed.run, one process per rank), on a
box with ONE GPU: both ranks share the device (VF_SHARE_GPU=1) and gloo carries the collectives (VF_DIST_BACKEND=gloo;
RCCL refuses two ranks on one device).  Exercises what a multi-GPU run needs besides RCCL itself: process-group init
before the first HIP call, per-rank synthetic shards, the gradient arena, the barrier + MAX-over-ranks timing, rank 0's
single JSON line, and a clean shutdown.  The launcher is started as a child process (never exec'd in place)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("reducer", ["arena", "ddp", "xgmi"])
def test_bench_two_ranks_on_one_gpu(reducer):
    env = dict(os.environ, VF_DIST_BACKEND="gloo", VF_SHARE_GPU="1", VF_REDUCER=reducer, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "2", "--batch", "2", "--views", "3"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    if reducer == "xgmi" and r.returncode != 0 and any(k in r.stderr for k in ("vf_xgmi_export failed", "vf_xgmi_open failed",
                                                                                "vf_xgmi_alloc failed")):
        pytest.skip("HIP IPC is not available on this box (a property of the box, not of the reducer)")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only, one line
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 2 and res["scaling"] == "weak"
    assert res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 4
    # whole-job throughput = views of BOTH ranks per step / max-over-ranks step time
    assert abs(res["value"] - 2 * 6 * 1e3 / res["ms_per_step"]) < 1e-6 * res["value"]
    assert res["loss"] == res["loss"] and 0 < res["loss"] < 10
    assert "roofline" not in res and "cpu_baseline" not in res      # N=1-only legs
