"""bench.py host-side contract pieces that need no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stdout_carries_only_the_json_line():
    """RCCL prints a version banner and gloo a connection line to stdout; the driver parses stdout.  _JsonOut points
    fd 1 at stderr for the whole run and writes the line to a saved copy of the real stdout."""
    code = "\n".join([
        "import sys, ctypes", "sys.path.insert(0, %r)" % ROOT, "sys.argv = ['bench.py']", "import bench",
        "o = bench._JsonOut()",
        "print('noise from python')",
        "ctypes.CDLL(None).printf(b'noise from C stdio\\n')",
        "o.emit('{\"ok\": 1}')",
        "print('late noise')"])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"ok": 1}\n'
    assert "noise from python" in r.stderr and "noise from C stdio" in r.stderr and "late noise" in r.stderr


def test_launch_model_counts_the_flops_of_the_lower_triangle():
    """cholesky_launch_model: trailing + strip updates of the two-level schedule add up to the N^3/3 of the
    factorisation (plus the appended 128 rows), within the tile granularity."""
    sys.path.insert(0, ROOT)
    import bench
    n = 16384
    trail, n_trail, strip, n_strip = bench.cholesky_launch_model(n + 128, n)
    total = trail + strip
    assert 0.95 * n ** 3 / 3 < total < 1.12 * n ** 3 / 3
    assert n_trail == 64 and n_strip == 64      # 48 near + 16 far updates (the look-ahead splits the far ones again)


def test_committed_traffic_file_is_the_one_bench_reads():
    with open(os.path.join(ROOT, "profiles", "r01f_pmc_traffic.json")) as f:
        t = json.load(f)
    assert t["traffic_bytes_per_launch"] > 0 and t["launches"] > 0
