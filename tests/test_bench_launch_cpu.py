"""`python bench.py --gpus N` without a launcher starts its own ranks: the driver's SCALE
command is the plain one, and it used to stop at "launch with torch.distributed.run"."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_self_launch_command(monkeypatch):
    import bench
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    rc = bench.self_launch(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"])
    cmd = seen["cmd"]
    assert rc == 7                                            # a dead rank's status is the command's status
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_plain_multi_gpu_command_starts_ranks():
    """no GPU here: both ranks die in `require_gpu` / the RCCL set-up -- but they were STARTED (the old message is
    gone) and the command's status is non-zero"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert "launch with torch.distributed.run" not in out
    assert "torch.distributed" in out or "ChildFailedError" in out or "exitcode" in out
    assert r.returncode != 0
