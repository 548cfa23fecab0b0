"""bench.py --gpus N without a torch.distributed.run environment starts its own ranks (the driver's invocation form).
CPU tests of the parent: the environment and command line it gives each child, and that a failing rank makes the parent
exit non-zero with that rank's stderr tail (here every rank fails: there is no GPU and the HIP path has no CPU fallback)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_child_environment_and_command_line():
    bench = _bench_module()
    base = {"PATH": "/usr/bin", "NDP_DIST_BACKEND": "gloo"}
    envs = [bench.child_rank_env(base, r, 4, 29999) for r in range(4)]
    for r, env in enumerate(envs):
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29999"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"                 # dmabuf IPC: RCCL / hipIpc need it on this platform
        assert env["NDP_DIST_BACKEND"] == "gloo" and env["PATH"] == "/usr/bin"   # the caller's environment is kept
    assert "RANK" not in base                                           # the parent's own environment is not touched
    argv = bench.child_rank_argv(["--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert argv[0] == sys.executable and argv[1] == os.path.join(ROOT, "bench.py")
    assert argv[2:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]


@pytest.mark.skipif(torch.cuda.is_available(), reason="the failure path is exercised where no GPU exists")
def test_parent_reports_a_failing_rank_and_exits_nonzero():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 1
    assert p.stdout.strip() == ""                                       # no JSON line from a failed run
    assert "2-rank run failed" in p.stderr and "stderr tail of rank" in p.stderr
    assert "needs an MI355X" in p.stderr                                # the child's own message came through
