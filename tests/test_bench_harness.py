"""bench.py's launch / rendezvous / reporting logic on CPU (`--dry-run`: gloo, no kernel): `python bench.py --gpus 2`
without a launcher starts two ranks itself and prints ONE well-formed JSON line; the torchrun form still works; the C5
strong split (`--scaling strong`: global batch 32 / N, BASELINE config 5) is reported as such."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env["GLOO_SOCKET_IFNAME"] = "lo"
    return env


def _one_line(cmd):
    r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for key in REQUIRED:
        assert key in d, key
    return d


def test_self_launch_two_ranks():
    d = _one_line([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["dry_run"] is True
    assert d["scaling"] == "weak" and d["config"]["batch_per_gpu"] == 4
    assert d["value"] > 0 and d["ms_per_step"] >= 1.0  # each dry step waits 1 ms


def test_torchrun_launch_two_ranks():
    d = _one_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                   "--master-addr", "127.0.0.1", "--master-port", "29611", "bench.py", "--gpus", "2", "--dry-run",
                   "--steps", "3", "--warmup", "0"])
    assert d["n_gpus"] == 2 and d["steps"] == 3


@pytest.mark.parametrize("n,per_gpu", [(1, 32), (2, 16)])
def test_c5_strong_split(n, per_gpu):
    d = _one_line([sys.executable, "bench.py", "--gpus", str(n), "--dry-run", "--steps", "2", "--workload", "c5",
                   "--scaling", "strong"])
    assert d["scaling"] == "strong" and d["n_gpus"] == n and d["config"]["batch_per_gpu"] == per_gpu


def test_strong_is_c5_only():
    r = subprocess.run([sys.executable, "bench.py", "--dry-run", "--scaling", "strong"], cwd=ROOT, env=_clean_env(),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "c5" in r.stderr


def test_rank_lost_before_rendezvous_ends_the_launch():
    """A rank that dies before init_process_group must not leave `bench.py --gpus N` hanging on the store timeout: the
    parent polls every child, ends the siblings and exits non-zero (bench.py spawn_ranks)."""
    import time
    env = dict(_clean_env(), FA_BENCH_DRY_FAIL_RANK="1", FA_BENCH_RENDEZVOUS_S="120")
    t0 = time.time()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--steps", "2"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=110)
    assert r.returncode != 0 and "rank exit codes" in r.stderr
    assert time.time() - t0 < 60  # well under the rendezvous timeout
    assert not r.stdout.strip()
