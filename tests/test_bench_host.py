"""CPU tests of bench.py's host logic: `python bench.py --gpus N` becomes the launcher of its own N ranks
(`self_launch`: torch.distributed.run on 127.0.0.1, a free port, dmabuf IPC in the environment, the children's exit status
relayed), the FLOP bookkeeping of SURVEY §8-d, and the per-rank spread the multi-GPU line reports. No GPU, no ranks are
started: `subprocess.run` is replaced by a recorder."""
import os
import socket
import subprocess
import sys
import types

import pytest

import bench


class _Recorder:
    def __init__(self, rc):
        self.rc, self.calls = rc, []

    def __call__(self, cmd, env=None, **kw):
        self.calls.append((list(cmd), dict(env or {}), kw))
        return types.SimpleNamespace(returncode=self.rc)


@pytest.mark.parametrize("n,rc", [(2, 0), (8, 0), (4, 3)])
def test_self_launch_command_env_and_exit_status(monkeypatch, n, rc):
    rec = _Recorder(rc)
    monkeypatch.setattr(subprocess, "run", rec)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", str(n), "--steps", "7", "--warmup", "2"])
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
    monkeypatch.setenv("OCM_CPU_THREADS", "16")
    assert bench.self_launch(n) == rc  # the children's status is ours
    (cmd, env, kw), = rec.calls
    # one torch.distributed.run of THIS script, single node, one process per GPU, rendezvous on the loopback address
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and f"--nproc-per-node={n}" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    port = int(cmd[cmd.index("--master-port") + 1])
    assert 1024 <= port <= 65535
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:  # the port was free when it was picked
        so.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        so.bind(("127.0.0.1", port))
    script = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[script + 1:] == ["--gpus", str(n), "--steps", "7", "--warmup", "2"]  # the caller's flags travel unchanged
    assert script > cmd.index("--master-port")  # launcher options come before the script
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"  # dmabuf IPC: RCCL needs it on this driver
    assert env["OMP_NUM_THREADS"] == str(max(1, 16 // n))  # the host cores are shared by the ranks
    assert not kw.get("shell")


def test_self_launch_keeps_the_callers_environment(monkeypatch):
    rec = _Recorder(0)
    monkeypatch.setattr(subprocess, "run", rec)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")  # an explicit setting is not overridden
    monkeypatch.setenv("OMP_NUM_THREADS", "3")
    monkeypatch.setenv("OCM_BENCH_BACKEND", "gloo")
    bench.self_launch(2)
    env = rec.calls[0][1]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "1" and env["OMP_NUM_THREADS"] == "3" and env["OCM_BENCH_BACKEND"] == "gloo"


def test_main_becomes_the_launcher_only_outside_a_rank(monkeypatch):
    """--gpus N without WORLD_SIZE: launcher (exit status relayed through SystemExit). With WORLD_SIZE set (a rank started by
    torch.distributed.run) main() must not launch again; a world that disagrees with --gpus is refused."""
    rec = _Recorder(5)
    monkeypatch.setattr(subprocess, "run", rec)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 5 and len(rec.calls) == 1
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert len(rec.calls) == 1 and "WORLD_SIZE=4" in str(e.value.code)


def test_flop_bookkeeping_matches_survey():
    """F_map = 8.644 GFLOP per ViT-S/16 224^2 tile, 104.3 (ViT-B/16 384^2), 186.0 (ViT-S/8 384^2)."""
    f, n = bench.flops_per_tile(384, 12, 16, 224)
    assert n == 197 and abs(f / 1e9 - 8.644) < 5e-3
    f, n = bench.flops_per_tile(768, 12, 16, 384)
    assert n == 577 and abs(f / 1e9 - 104.3) < 0.1
    f, n = bench.flops_per_tile(384, 12, 8, 384)
    assert n == 2305 and abs(f / 1e9 - 186.0) < 0.1
    cf = bench.class_flops(384, 1536, 197, 64, 16, 3)
    assert abs(cf["fc1_gemm"] / 1e9 - 14.873) < 1e-2 and cf["fc1_gemm"] == cf["fc2_gemm"]
    assert abs(cf["qkv_gemm"] / 1e9 - 11.155) < 1e-2 and abs(cf["proj_gemm"] / 1e9 - 3.718) < 1e-2


def test_spread_is_min_median_max():
    assert bench.spread([3.0]) == [3.0, 3.0, 3.0]
    assert bench.spread([2.0, 1.0]) == [1.0, 1.5, 2.0]
    assert bench.spread([5.0, 1.0, 2.0]) == [1.0, 2.0, 5.0]
    assert bench.spread([4.0, 1.0, 3.0, 2.0]) == [1.0, 2.5, 4.0]
