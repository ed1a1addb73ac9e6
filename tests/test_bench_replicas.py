"""N>1 path of bench.py rehearsed on CPU: two processes, torch.distributed `gloo`, the GPU step replaced by a
rank-dependent sleep.  Checks the protocol the driver relies on: barrier on both sides of the timed region, MAX over
ranks, whole-job aggregate value, exactly one JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_replicas_protocol():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
           "4", "--warmup", "1", "--backend", "gloo", "--simulate-step-ms", "50", "--res", "240"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["higher_is_better"] is True and r["unit"] == "Mpix/s" and r["vs_baseline"] is None
    # rank 1 sleeps 100 ms per step: the job's time is the slowest rank's, the value is the 2-rank aggregate
    assert 95.0 <= r["ms_per_step"] <= 160.0, r["ms_per_step"]
    expect = 2 * (135 * 240 / 1e6) / (r["ms_per_step"] * 1e-3)
    assert abs(r["value"] - expect) / expect < 1e-3
    assert "roofline" in r and "cpu_baseline" not in r
