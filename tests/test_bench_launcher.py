"""bench.py --gpus N without a GPU: the launcher's command line, the refusal on a world-size mismatch, and the
per-workload defaults.  (The N-rank run itself is rehearsed on the GPU box with the gloo backend:
profiles/r2_bench_two_rank_gloo.json.)"""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", BENCH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launcher_command_starts_one_rank_per_gpu():
    b = _bench()
    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1", "--workload", "stress"]
    cmd = b.launcher_command(4, argv, 29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29555"
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == argv            # the ranks see the same arguments, --gpus included
    port = b.free_port()
    assert 1024 < port < 65536


def test_parent_launches_before_touching_torch(monkeypatch):
    """WORLD_SIZE unset and --gpus 2: main() goes to the launcher (and nothing else) and exits with its code."""
    b = _bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(b.subprocess, "run", fake_run)
    torch_loaded_before = "torch" in sys.modules
    with pytest.raises(SystemExit) as e:
        b.main(["--gpus", "2", "--chains", "4194304"])
    assert e.value.code == 7
    assert "--nproc-per-node=2" in seen["cmd"] and seen["cmd"][-4:] == ["--gpus", "2", "--chains", "4194304"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    if not torch_loaded_before:
        assert "torch" not in sys.modules, "the parent imported torch before starting its ranks"


def test_world_size_mismatch_is_refused():
    """A rank whose launcher world size differs from --gpus refuses before any GPU call."""
    env = {**os.environ, "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, env=env)
    assert r.returncode != 0
    assert "world size 2" in r.stderr and "--gpus 4" in r.stderr
    # the silent case of round 1: --gpus 8 launched as ONE process with no launcher env must not run one rank
    b = _bench()
    os.environ.pop("WORLD_SIZE", None)
    args = b.parse(["--gpus", "8"])
    os.environ["WORLD_SIZE"] = "1"
    try:
        with pytest.raises(SystemExit) as e:
            b.world_from_env(args)
        assert "does not equal --gpus 8" in str(e.value)
    finally:
        os.environ.pop("WORLD_SIZE", None)


def test_workload_defaults():
    b = _bench()
    a = b.parse([])
    assert (a.gpus, a.workload, a.chains, a.coords, a.sweeps, a.step_size) == (1, "chain", 1 << 24, 3, 256, 2.48)
    a = b.parse(["--workload", "nucleon"])
    assert (a.chains, a.sweeps, a.step_size, a.nucleus) == (1 << 22, 256, 2.2, "pb")
    a = b.parse(["--workload", "stress", "--gpus", "8"])
    assert (a.chains, a.sweeps, a.step_size, a.gpus) == (1 << 21, 64, 0.8, 8)
    assert b.NUCLEI["au"] == (197, 6.38, 0.535) and b.NUCLEI["pb"] == (208, 6.62, 0.546)
    with pytest.raises(SystemExit):
        b.parse(["--gpus", "0"])
