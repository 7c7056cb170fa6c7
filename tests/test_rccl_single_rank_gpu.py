"""RCCL smoke on ONE GPU: bench.py under P3_FORCE_COLLECTIVES=1 runs the N > 1 code path (SyncBatchNorm statistic exchange, positional
gradient-bucket all-reduces issued from backward, barrier / MAX bracket) over a 1-rank "nccl" process group - the most a single
device allows, since RCCL refuses two ranks on one GPU.  Values must equal the plain N = 1 run: a 1-rank SUM is the identity."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *flags):
    env = dict(os.environ, **extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lean", "--steps", "3", "--warmup", "3", *flags], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("{"), f"the JSON line must be the last line of stdout, got: {last[:200]!r}"
    return json.loads(last)


@pytest.mark.gpu
def test_single_rank_rccl_runs_the_multi_gpu_step():
    forced = _bench({"P3_FORCE_COLLECTIVES": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29641"})
    plain = _bench({"P3_FORCE_COLLECTIVES": "0"}, "--graph", "0")
    c = forced["config"]["collectives"]
    assert c["backend"] == "nccl" and c["forced_single_rank"] and c["world"] == 1
    assert forced["config"]["sync_bn"] and not forced["config"]["hip_graph"]
    assert c["grad_buckets"] >= 2 and c["early_bucket_launches"] >= c["grad_buckets"] - 1      # buckets left backward before finish()
    assert c["syncbn_collectives"] > 0
    assert "collectives" not in plain["config"]
    assert abs(forced["final_loss"] - plain["final_loss"]) <= 2e-3 * abs(plain["final_loss"]), (forced["final_loss"], plain["final_loss"])
    # same kernels plus a handful of 1-rank collectives: the step must not get slower by more than the collectives' launch cost
    assert forced["ms_per_step"] < 1.15 * plain["ms_per_step"], (forced["ms_per_step"], plain["ms_per_step"])


@pytest.mark.gpu
def test_bench_gpus_2_self_launch_two_ranks_on_one_device():
    """`python bench.py --gpus 2` (no launcher): bench.py starts torch.distributed.run itself as a child before touching the GPU; on a 1-GPU box
    the two ranks share device 0 over gloo (test hooks - RCCL refuses two ranks on one device).  n_gpus = 2, whole-job value = 2 x tiles."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(P3_BENCH_BACKEND="gloo", P3_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--lean", "--steps", "3", "--warmup", "3", "--batch", "16"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stderr[-3000:], r.stdout[-1000:])
    last = r.stdout.strip().splitlines()[-1]
    d = json.loads(last)
    c = d["config"]["collectives"]
    assert d["n_gpus"] == 2 and c["world"] == 2 and c["backend"] == "gloo" and d["config"]["parallelism"] == "dp2" and d["config"]["sync_bn"]
    assert abs(d["value"] - 2 * 16 / d["ms_per_step"] * 1e3) < 0.01 * d["value"]
    assert c["syncbn_collectives"] > 0 and c["grad_buckets"] >= 2 and d["final_loss"] == d["final_loss"]
