"""The driver's multi-GPU invocation, `python3 bench.py --gpus N --steps 20 --warmup 5` with no torch.distributed.run
environment, rehearsed with two ranks on the one GPU of the test box (NDP_BENCH_ONE_GPU=1: every rank uses cuda:0;
gloo for the rendezvous / collectives because RCCL refuses two ranks on one device).  The figures it prints are not
measurements (the ranks time-slice one GPU); what is tested is that the parent starts the ranks, relays ONE JSON line,
exits 0, and that the data-parallel workloads all ran with replicas that stayed bit-identical."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("exchange", ["auto", "p2p"])
def test_bench_gpus_2_starts_its_own_ranks(exchange):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NDP_BENCH_ONE_GPU="1", NDP_DIST_BACKEND="gloo", NDP_DP_EXCHANGE=exchange, NDP_BENCH_LAUNCH_TIMEOUT_S="500")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]                            # ONE JSON line on stdout, nothing else
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["scaling"] == "weak"
    assert out["metric"] == "gan_train_steps_per_sec_traj8_batch64" and out["value"] > 0
    cfg = out["config"]
    assert cfg["global_batch"] == 128 and cfg["parallelism"] == "dp2"
    assert cfg["replicas_bit_identical"] is True
    # shared GPU: "auto" must have chosen the collective (the in-kernel hand-shake needs a GPU per rank), "p2p" forces it
    assert cfg["gradient_exchange"].startswith("rccl" if exchange == "auto" else ("p2p", "rccl (p2p"))
    assert "extras_failed" not in out, out.get("extras_failed")
    for name in ("strong_config3", "config5_shard", "forward_model_dp"):
        assert name in out, sorted(out)
        assert out[name]["replicas_bit_identical"] is True, (name, out[name])
    assert out["strong_config3"]["global_batch"] == 256 and out["strong_config3"]["scaling"] == "strong"
    assert "roofline" in out and out["roofline"]["kernel"].startswith("k_")
