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
    """If smn_comm_init fails on ANY rank (no RCCL, no peer access), every rank must learn it and run as a replica, with
    the barrier and the max over ranks going through files: two processes, rank 1 reports a failed communicator."""
    code = "\n".join([
        "import sys, json", "sys.path.insert(0, %r)" % ROOT, "sys.argv = ['bench.py']", "import bench",
        "rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0",
    ])
    worker = tmp_path / "w.py"
    worker.write_text("\n".join([
        "import sys, json", "sys.path.insert(0, %r)" % ROOT, "rank = int(sys.argv[1]); sys.argv = ['bench.py']", "import bench",
        "ok, d = bench.agree_on_communicator(%r, 2, rank, ok=(rank == 0), timeout_s=60)" % str(tmp_path / "rdv"),
        "s = bench.FileSync(d, 2, rank)",
        "s.barrier()",
        "m = s.max(10.0 + rank)",
        "g = s.gather(float(rank))",
        "print(json.dumps(dict(ok=ok, max=m, gather=g)))"]))
    del code
    procs = [subprocess.Popen([sys.executable, str(worker), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1] for o in outs]
    for o in outs:
        got = json.loads(o[0].strip().splitlines()[-1])
        assert got == {"ok": False, "max": 11.0, "gather": [0.0, 1.0]}
