"""bench.py with more than one rank, on whatever devices the box has.

* one device: RCCL refuses two ranks on the same GPU (SMN_BENCH_SHARE_GPU=1 puts both there), which is exactly the
  "communicator could not be brought up" case: the run must exit NON-ZERO with one JSON line that says why -- and run
  independent replicas only when --allow-replica-fallback asks for them, labelled as such;
* `--sharded-path`: the whole P > 1 step (one build launch, column-first exchange, piece-wise consumption) through a real
  one-rank RCCL communicator, equal to the fused single-GPU step bit for bit;
* two or more devices (skipped on the one-GPU boxes): `bench.py --gpus 2` for C4's and C5's networks at a small N,
  log-pdf and logdet equal to the single-GPU line of the same workload.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--n", "2500", "--d", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-recursion-probe",
         "--no-exclusive-probe", "--no-other-workloads"]


def _run(extra, env_extra=None, timeout=420):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, capture_output=True, text=True,
                       timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r, lines


def _devices():
    from smnngp import _lib as L
    n = C.c_int(0)
    L._lib.smn_device_count(C.byref(n))
    return n.value


def test_no_communicator_is_a_failure_not_a_flat_scaling_curve():
    r, lines = _run(["--gpus", "2"], {"SMN_BENCH_SHARE_GPU": "1"})
    assert r.returncode != 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    got = json.loads(lines[0])
    assert got["value"] is None and got["n_gpus"] == 2 and got["comm"]["rccl_ranks"] == 0 and got["comm"]["fallback"]
    assert "no RCCL communicator" in got["error"]


def test_replica_fallback_only_on_request_and_labelled():
    r, lines = _run(["--gpus", "2", "--allow-replica-fallback"], {"SMN_BENCH_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    got = json.loads(lines[0])
    assert got["scaling"] == "replicas" and got["comm"]["fallback"] and got["comm"]["rccl_ranks"] == 0
    assert "independent replicas" in got["config"]["parallelism"] and got["value"] > 0


def test_sharded_path_on_a_one_rank_communicator_equals_the_fused_step():
    r1, l1 = _run([])
    r2, l2 = _run(["--sharded-path"])
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-1500:], r2.stderr[-1500:])
    a, b = json.loads(l1[0]), json.loads(l2[0])
    assert b["comm"] == {"rccl_ranks": 1, "fallback": None} and a["comm"]["rccl_ranks"] == 0
    assert a["result"] == b["result"]                                   # bit-identical log-pdf, logdet, info
    for key in ("kernel_build_ms", "exchange_ms", "scatter_ms", "exchange_exposed_ms", "exchange_stall_ms",
                "build_only_speedup", "build_plus_exposed_assembly_speedup", "exchange_ranges"):
        assert key in b, key
    assert b["exchange_ranges"] >= 2 and b["scaling"] == "strong"


def test_one_gpu_build_reference_of_the_sharded_line_is_one_launch():
    """From 112 tile rows on the fused single-GPU call builds the matrix in TWO launches (the corner beside the first panel chain);
    the sharded line's reference for `build_only_speedup` must be the one-launch build with the chip to itself, not the average of
    the two launches: with one rank the ratio is about 1."""
    big = ["--n", "14400", "--d", "256", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-recursion-probe",
           "--no-exclusive-probe", "--no-other-workloads", "--sharded-path"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + big, capture_output=True, text=True, timeout=420, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    b = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert 0.75 < b["build_only_speedup"] < 1.35, (b["build_only_speedup"], b["one_gpu_build_ms"], b["kernel_build_ms"])


@pytest.mark.parametrize("config", ["c4", "c5"])
def test_two_ranks_on_two_devices_equal_the_single_gpu_step(config):
    if _devices() < 2:
        pytest.skip("one visible device")
    r1, l1 = _run(["--config", config])
    r2, l2 = _run(["--config", config, "--gpus", "2"])
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-1500:], r2.stderr[-1500:])
    a, b = json.loads(l1[0]), json.loads(l2[0])
    assert b["n_gpus"] == 2 and b["comm"] == {"rccl_ranks": 2, "fallback": None}
    assert a["result"] == b["result"]                                   # same tiles, same arithmetic, same schedule
