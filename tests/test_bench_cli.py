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


def test_gpus_n_starts_its_own_ranks_before_any_gpu_call(tmp_path):
    """`python bench.py --gpus 3` with no launcher: the parent imports nothing that touches a GPU and starts three fresh
    children with RANK / LOCAL_RANK / WORLD_SIZE set and one shared rendezvous file.  Rehearsed with a stand-in for the
    worker body (this container has no GPU): every child reports its environment, rank 0 publishes the id file the way
    exchange_rccl_id does, the others read it back."""
    sys.path.insert(0, ROOT)
    stub = tmp_path / "fake_bench.py"
    stub.write_text("\n".join([
        "import os, sys, json, time",
        "sys.path.insert(0, %r)" % ROOT,
        "import bench",
        "class FakeLib:",
        "    class _lib:",
        "        @staticmethod",
        "        def smn_comm_unique_id(buf):",
        "            buf.raw = bytes(range(128)); return 0",
        "if 'RANK' not in os.environ:",
        "    a = bench.parse(); bench.launch_ranks(a)",
        "rdv = sys.argv[sys.argv.index('--rendezvous-file') + 1]",
        "uid = bench.exchange_rccl_id(FakeLib, rdv, int(os.environ['RANK']), timeout_s=30)",
        "out = dict(rank=os.environ['RANK'], local=os.environ['LOCAL_RANK'], world=os.environ['WORLD_SIZE'], uid=uid.raw.hex())",
        "open(os.path.join(%r, 'rank%%s.json' %% os.environ['RANK']), 'w').write(json.dumps(out))" % str(tmp_path),
    ]))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, str(stub), "--gpus", "3", "--steps", "2"], capture_output=True, text=True,
                       timeout=180, env=env)
    assert r.returncode == 0, r.stderr
    got = [json.loads((tmp_path / ("rank%d.json" % i)).read_text()) for i in range(3)]
    assert [g["rank"] for g in got] == ["0", "1", "2"] and [g["local"] for g in got] == ["0", "1", "2"]
    assert all(g["world"] == "3" for g in got)
    assert all(g["uid"] == bytes(range(128)).hex() for g in got)           # the id rank 0 made reached every rank


def test_launcher_code_path_imports_no_gpu_module():
    """The parent of `--gpus N` must not initialise the GPU: nothing above launch_ranks() in main() may import the
    library (or torch)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    head = main[:main.index("launch_ranks(args)")]
    assert "import" not in head.replace("# never returns", "")
    import re
    top = src[src.index('"""', 10) + 3:src.index("def parse():")]           # the module level below the docstring
    assert not re.search(r"^\s*(import|from)\s+(torch|smnngp)", top, re.M)
    assert not re.search(r"^\s*(import|from)\s+torch", src, re.M)            # no torch anywhere in the bench


def test_committed_traffic_file_is_the_one_bench_reads():
    sys.path.insert(0, ROOT)
    import argparse

    import bench
    a = argparse.Namespace(config="c4", n=16384, d=3072, layers=4, act="relu", dtype="f32")
    for k in bench.SCHEDULE_KNOBS:
        os.environ.pop(k, None)
    v, src = bench.pmc_traffic(a, False)
    assert v > 0 and src.startswith("profiles/")
    assert bench.pmc_traffic(a, True)[0] is None
    os.environ["SMN_XCD_MAP"] = "0"                  # an A/B run must not carry the default schedule's counters
    try:
        assert bench.pmc_traffic(a, False)[0] is None
    finally:
        del os.environ["SMN_XCD_MAP"]
    a.n = 8192
    assert bench.pmc_traffic(a, False)[0] is None


def test_launcher_ends_the_other_ranks_when_one_fails(tmp_path):
    """A rank that dies before the rendezvous (no such device, say) must not leave the others waiting in it for ever."""
    stub = tmp_path / "fake_bench.py"
    stub.write_text("\n".join([
        "import os, sys, time",
        "sys.path.insert(0, %r)" % ROOT,
        "import bench",
        "if 'RANK' not in os.environ:",
        "    bench.launch_ranks(bench.parse())",
        "if os.environ['RANK'] == '1':",
        "    sys.exit(7)",
        "time.sleep(600)",
    ]))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, str(stub), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 7 and time.time() - t0 < 60


def test_ranks_agree_on_a_failed_communicator_and_fall_back_to_the_file_rendezvous(tmp_path):
    """If smn_comm_init fails on ANY rank (no RCCL, no peer access), every rank must learn it; with
    --allow-replica-fallback the barrier and the max over ranks then go through files: two processes, rank 1 reports a
    failed communicator."""
    worker = tmp_path / "w.py"
    worker.write_text("\n".join([
        "import sys, json", "sys.path.insert(0, %r)" % ROOT, "rank = int(sys.argv[1]); sys.argv = ['bench.py']", "import bench",
        "d = %r" % str(tmp_path / "rdv.run"),
        "ok = bench.agree_on_communicator(d, 2, rank, ok=(rank == 0), timeout_s=60)",
        "s = bench.FileSync(d, 2, rank)",
        "s.barrier()",
        "m = s.max(10.0 + rank)",
        "g = s.gather(float(rank))",
        "print(json.dumps(dict(ok=ok, max=m, gather=g)))"]))
    procs = [subprocess.Popen([sys.executable, str(worker), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1] for o in outs]
    for o in outs:
        got = json.loads(o[0].strip().splitlines()[-1])
        assert got == {"ok": False, "max": 11.0, "gather": [0.0, 1.0]}


def test_stale_rendezvous_files_of_an_earlier_run_are_not_read(tmp_path):
    """A file older than the process that started the ranks belongs to an earlier run (same port, recycled parent PID): the id,
    the status flags and the file-rendezvous values are all read through _read_fresh, which treats it as absent."""
    sys.path.insert(0, ROOT)
    import time

    import bench
    f = tmp_path / "init_1"
    f.write_text("1")
    assert bench._read_fresh(str(f)) == "1"
    old = bench._launcher_start_time() - 100.0
    os.utime(str(f), (old, old))
    assert bench._read_fresh(str(f)) is None
    assert bench._read_fresh(str(tmp_path / "missing")) is None
    # exchange_rccl_id: a non-zero rank does not take a stale id for the run's
    stale = tmp_path / "rccl_id"
    stale.write_bytes(bytes(128))
    os.utime(str(stale), (old, old))
    t0 = time.time()
    try:
        bench.exchange_rccl_id(None, str(stale), 1, timeout_s=0.3)
        raised = False
    except RuntimeError:
        raised = True
    assert raised and time.time() - t0 < 5


def test_watchdog_names_the_stuck_rank_and_its_phase_in_one_json_line(tmp_path):
    """Two ranks under the built-in launcher; rank 1 stops making progress in a named phase (as a rank stuck inside RCCL would:
    its main thread sits in a C call, the watchdog thread keeps running).  Within the stall limit the run ends with status
    3 and ONE JSON line on stdout that names rank 1 and the phase; nothing is left running."""
    stub = tmp_path / "fake_bench.py"
    stub.write_text("\n".join([
        "import os, sys, time",
        "sys.path.insert(0, %r)" % ROOT,
        "import bench",
        "if 'RANK' not in os.environ:",
        "    bench.launch_ranks(bench.parse())",
        "rank = int(os.environ['RANK'])",
        "rdv = sys.argv[sys.argv.index('--rendezvous-file') + 1]",
        "out = bench._JsonOut()",
        "wd = bench.Watchdog(rdv + '.run', rank, 2, out)",
        "if rank == 1:",
        "    wd.phase('step 2: gather 3/16')",
        "    time.sleep(600)",
        "for k in range(6000):",
        "    wd.phase('step %d: factor' % k)",
        "    time.sleep(0.1)",
    ]))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(SMN_BENCH_STALL_S="3", SMN_BENCH_RANK_TIMEOUT_S="60")
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, str(stub), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 3 and time.time() - t0 < 40, (r.returncode, r.stderr)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    got = json.loads(lines[0])
    assert got["stuck_rank"] == 1 and got["phase"] == "step 2: gather 3/16" and got["value"] is None and got["n_gpus"] == 2
    assert "no progress" in got["error"] and got["ranks"]["0"]["phase"].startswith("step ")


def test_watchdog_overall_limit_and_a_rank_that_never_started(tmp_path):
    """diagnose(): a rank without a heartbeat file is the one named; the overall limit ends a run whose ranks all keep moving."""
    sys.path.insert(0, ROOT)
    import time

    import bench
    d = tmp_path / "run"
    d.mkdir()
    (d / "hb_0").write_text(json.dumps({"phase": "smn_comm_init", "t_phase": time.time() - 5, "t": time.time()}))
    got = bench.diagnose(str(d), 2, "test")
    assert got["stuck_rank"] == 1 and "never started" in got["phase"] and got["ranks"]["0"]["phase"] == "smn_comm_init"
    (d / "hb_1").write_text(json.dumps({"phase": "done", "t_phase": time.time() - 9, "t": time.time()}))
    got = bench.diagnose(str(d), 2, "test")
    assert got["stuck_rank"] == 0 and got["phase"] == "smn_comm_init"        # a finished rank is never the stuck one
