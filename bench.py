#!/usr/bin/env python3
"""bench.py — the hot path of BASELINE.json on MI355X: NNGP kernel build + jittered Cholesky (+ LML heads).

One "step" = one SPR.loss evaluation (spax/models.py:93-98) on synthetic inputs already resident in HBM:
fused Gram + layer recursion -> K + eps I -> blocked Cholesky with y carried -> log-marginal likelihood.

    python bench.py                         the metric's configuration on one GPU: N=16384 d=3072 L=4 ReLU fp32 (C4)
    python bench.py --config c2|c5|c3       BASELINE.json's other configurations (shapes; c3 = conv-NNGP + Student-t, fp64)
    python bench.py --gpus P                P ranks, one per GPU: the parent starts P fresh child processes BEFORE anything
                                            touches a GPU (RANK / LOCAL_RANK / WORLD_SIZE in their environment); rank 0
                                            hands the 128-byte RCCL id to the others through a file.  No torch, no launcher.
    python -m torch.distributed.run --nproc-per-node P ... bench.py --gpus P    the same workers under an external launcher

With P > 1 the kernel build is sharded over the ranks by 128-row tile rows dealt cyclically (sharding.py, cyclic
column-first layout): every rank builds its share in ONE launch on all CUs, then the exchange goes out column range by
column range (an RCCL all-gather + a scatter into the factorisation workspace per range, on side streams) while every
rank is already factoring: the first panel chain waits for the first 1024 columns only (strong scaling: total work fixed;
the factorisation itself is replicated, so the whole step is Amdahl-bound -- the line states the kernel-build speed-up).

Multi-rank runs are self-diagnosing: every rank keeps a heartbeat (phase + step) in a file; when a rank makes no progress
for SMN_BENCH_STALL_S (120) seconds or the run exceeds SMN_BENCH_RANK_TIMEOUT_S (240), ONE JSON line names the stuck rank
and its phase and the run exits non-zero.  When the RCCL communicator cannot be brought up the run exits non-zero too
(`"comm": {"fallback": reason}`) unless --allow-replica-fallback asks for independent replicas.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 MFMA (16x16x4 and 32x32x2 forms alike), dense
PEAK_F64_MFMA_TFLOPS = 78.6    # datasheet: one v_mfma_f64_16x16x4_f64 per 64 cycles per SIMD at 2.4 GHz
# What a hand-written stream of independent MFMAs sustains on the whole chip (profiles/r04_mfma_forms.txt: accumulators in VGPRs
# or AGPRs alike, 1-4 waves per SIMD, clock 2.39 GHz): the datasheet figures are the hardware's rate.  (Round 3's "104 cycles per
# f64 MFMA" was the compiler copying the accumulators VGPR <-> AGPR every iteration of the microbench's loop.)
MEASURED_ISSUE_CEILING_TFLOPS = {"f32": 155.0, "f64": 77.4}
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s achievable)
TILE = 128

CONFIGS = {   # BASELINE.json `configs` (shapes); c4 is the one the metric is quoted on
    "c4": dict(n=16384, d=3072, layers=4, act="relu", dtype="f32"),
    "c2": dict(n=4096, d=512, layers=3, act="relu", dtype="f32"),
    "c5": dict(n=32768, d=1024, layers=6, act="erf", dtype="f32"),
    "c3": dict(n=10000, d=3072, layers=4, act="relu", dtype="f64"),      # 32x32x3 images, conv-NNGP, Student-t
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    p.add_argument("--n", type=int, default=None)
    p.add_argument("--d", type=int, default=None)
    p.add_argument("--layers", type=int, default=None)
    p.add_argument("--act", default=None)
    p.add_argument("--dtype", default=None, choices=["f32", "f64"])
    p.add_argument("--eps", type=float, default=None)
    p.add_argument("--pieces", type=int, default=0, help="column ranges of the exchange (0: sharding.default_col_pieces, at most 16)")
    p.add_argument("--allow-replica-fallback", action="store_true",
                   help="P > 1 and no RCCL communicator: run P independent replicas (scaling = \"replicas\") instead of exiting non-zero")
    p.add_argument("--cpu-sample-n", type=int, default=0, help="0: the benched N")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-recursion-probe", action="store_true")
    p.add_argument("--no-exclusive-probe", action="store_true", help="skip the look-ahead-off pass that fills frac_exclusive")
    p.add_argument("--with-ntk", action="store_true", help="sharded path: build NNGP + NTK jointly and pipeline both (default for --config c5)")
    p.add_argument("--no-other-workloads", action="store_true",
                   help="skip C2 / C5 / C3 / fp64 / predictive-path measurements appended to the default line")
    p.add_argument("--sharded-path", action="store_true",
                   help="run the P>1 step (sharded build + column-first exchange + LML) even with one rank (rehearsal on one GPU)")
    p.add_argument("--rendezvous-file", default=None, help=argparse.SUPPRESS)
    a = p.parse_args()
    for k, v in CONFIGS[a.config].items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    return a


# ----------------------------------------------------------------------------- launcher (no GPU call in this process)
RANK_TIMEOUT_S = float(os.environ.get("SMN_BENCH_RANK_TIMEOUT_S", "240"))   # whole multi-rank run, per rank, from its start
RANK_STALL_S = float(os.environ.get("SMN_BENCH_STALL_S", "120"))            # no change of phase / step for this long


def launch_ranks(args):
    """`python bench.py --gpus P` without a launcher: P fresh children, one per GPU.  This process never touches a GPU
    (a GPU-initialised process must not be re-executed), it only waits; the children rendezvous through files in a fresh
    directory.  Every child carries its own watchdog (Watchdog below) and rank 0 prints the diagnosis; this parent only ends
    the others when one has failed, and prints the diagnosis itself if the children are past their limit without one."""
    tmp = tempfile.mkdtemp(prefix="smnngp_bench_")
    rdv = os.path.join(tmp, "rccl_id")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus))
        cmd = [sys.executable, os.path.abspath(sys.argv[0])] + sys.argv[1:] + ["--rendezvous-file", rdv]   # the script as invoked
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    deadline = time.time() + RANK_TIMEOUT_S + 20.0     # the ranks' own watchdogs fire first
    while alive:
        if time.time() > deadline:
            if not os.path.exists(os.path.join(rdv + ".run", "line_emitted")):
                line = diagnose(rdv + ".run", args.gpus, "ranks still running %.0f s after the start" % (RANK_TIMEOUT_S + 20.0))
                sys.stdout.write(json.dumps(line) + "\n")
                sys.stdout.flush()
            for q in alive:
                q.terminate()
            for q in alive:
                try:
                    q.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    q.kill()
            rc = rc or 124
            break
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc = abs(r) or 1
                time.sleep(1.0)                       # (rank 0 may be writing the diagnosis)
                for q in alive:
                    q.terminate()
        if alive:
            time.sleep(0.05)
    shutil.rmtree(tmp, ignore_errors=True)
    sys.exit(rc)


def _launcher_start_time():
    """When the process that started the ranks began (our launcher or torchrun's agent): rendezvous files older than that
    belong to an earlier run that left them behind (same port, recycled parent PID) and are treated as absent."""
    try:
        return os.stat("/proc/%d" % os.getppid()).st_mtime - 2.0
    except OSError:
        return 0.0


def _read_fresh(path, mode="r"):
    """Contents of `path`, or None when it is missing or stale (older than the launcher)."""
    try:
        if os.stat(path).st_mtime < _launcher_start_time():
            return None
        with open(path, mode) as f:
            return f.read()
    except (FileNotFoundError, ValueError, OSError):
        return None


def _write_atomic(path, data, mode="w"):
    with open(path + ".tmp", mode) as f:
        f.write(data)
    os.replace(path + ".tmp", path)


def diagnose(run_dir, world, why):
    """The ONE JSON line of a multi-rank run that did not finish: every rank's last heartbeat, and as `stuck_rank` the rank
    that entered its current phase first (the ranks waiting for it inside a collective arrived later) -- or that never
    wrote one."""
    now = time.time()
    ranks, stuck, oldest = {}, None, None
    for r in range(world):
        raw = _read_fresh(os.path.join(run_dir, "hb_%d" % r))
        try:
            hb = json.loads(raw) if raw else None
        except ValueError:
            hb = None
        if hb is None:
            ranks[str(r)] = {"phase": "never started (no heartbeat)", "in_phase_s": None}
            if oldest is None or oldest > -1.0:
                stuck, oldest = r, -1.0
            continue
        ranks[str(r)] = {"phase": hb["phase"], "in_phase_s": round(now - hb["t_phase"], 1), "heartbeat_age_s": round(now - hb["t"], 1)}
        if hb["phase"] != "done" and (oldest is None or hb["t_phase"] < oldest):
            stuck, oldest = r, hb["t_phase"]
    phase = ranks[str(stuck)]["phase"] if stuck is not None else None
    return {"metric": "kernel-build + Cholesky wallclock (ms) and GFLOP/s", "value": None, "unit": "GFLOP/s", "n_gpus": world,
            "error": "multi-rank run did not finish: %s; rank %s stopped first, in phase '%s'" % (why, stuck, phase),
            "stuck_rank": stuck, "phase": phase, "ranks": ranks, "higher_is_better": True}


class Watchdog:
    """Heartbeat + limits of one rank.  The phase lives in memory (no file I/O in the timed loop); a daemon thread mirrors it
    into <run_dir>/hb_<rank> four times a second (ctypes releases the GIL, so it runs while the main thread sits in RCCL),
    reads the other ranks' files and, when a rank has not changed phase for RANK_STALL_S or this rank is past
    RANK_TIMEOUT_S, ends the run: rank 0 prints the diagnosis (ONE JSON line), every rank exits with status 3.  A
    GPU-initialised process is never re-executed; it just exits."""

    def __init__(self, run_dir, rank, world, out_fd, limit_s=None, stall_s=None):
        import threading
        self.dir, self.rank, self.world, self.out_fd = run_dir, rank, world, out_fd
        self.limit_s = RANK_TIMEOUT_S if limit_s is None else limit_s
        self.stall_s = RANK_STALL_S if stall_s is None else stall_s
        self.t0 = time.time()
        self._phase, self._t_phase = "init", self.t0
        self._stop = False
        os.makedirs(run_dir, exist_ok=True)
        self._beat()
        self.thread = threading.Thread(target=self._loop, daemon=True)
        self.thread.start()

    def phase(self, name):
        if name != self._phase:
            self._phase, self._t_phase = name, time.time()

    def _beat(self):
        _write_atomic(os.path.join(self.dir, "hb_%d" % self.rank),
                      json.dumps({"phase": self._phase, "t_phase": self._t_phase, "t": time.time()}))

    def _loop(self):
        while not self._stop:
            try:
                self._beat()
                why = self._expired()
                if why:
                    self._end(why)
            except Exception:   # noqa: BLE001  (a vanished directory at shutdown must not raise in a daemon thread)
                pass
            time.sleep(0.25)

    def _expired(self):
        now = time.time()
        if now - self.t0 > self.limit_s:
            return "rank %d past the %.0f s limit (SMN_BENCH_RANK_TIMEOUT_S)" % (self.rank, self.limit_s)
        other = _read_fresh(os.path.join(self.dir, "expired"))
        if other is not None:
            return other or "another rank's watchdog ended the run"       # (that rank's reason)
        for r in range(self.world):
            raw = _read_fresh(os.path.join(self.dir, "hb_%d" % r)) if r != self.rank else None
            t_phase, ph = self._t_phase, self._phase
            if r != self.rank:
                if not raw:
                    continue                          # not started yet: covered by the overall limit
                try:
                    hb = json.loads(raw)
                except ValueError:
                    continue
                t_phase, ph = hb["t_phase"], hb["phase"]
            if ph != "done" and now - t_phase > self.stall_s:
                return "rank %d made no progress for %.0f s (SMN_BENCH_STALL_S)" % (r, self.stall_s)
        return None

    def _end(self, why):
        try:
            if not os.path.exists(os.path.join(self.dir, "expired")):
                _write_atomic(os.path.join(self.dir, "expired"), why)
        except OSError:
            pass
        if self.rank == 0:
            time.sleep(0.6)                           # every rank's last heartbeat is on disk
            self.out_fd.emit(json.dumps(diagnose(self.dir, self.world, why)))
            try:
                _write_atomic(os.path.join(self.dir, "line_emitted"), "1")
            except OSError:
                pass
        else:
            time.sleep(2.0)                           # rank 0 reads the heartbeats first
        os._exit(3)

    def done(self):
        self.phase("done")
        self._beat()
        self._stop = True


def exchange_rccl_id(L, path, rank, timeout_s=120.0):
    """Rank 0 creates the 128-byte RCCL id and publishes it with an atomic rename; the others poll for the file (a stale
    one -- older than the launcher -- is not read)."""
    uid = C.create_string_buffer(128)
    if rank == 0:
        assert L._lib.smn_comm_unique_id(uid) == 0, "RCCL unavailable"
        _write_atomic(path, uid.raw, "wb")
        return uid
    t0 = time.time()
    while True:
        raw = _read_fresh(path, "rb")
        if raw is not None and len(raw) == 128:
            return C.create_string_buffer(raw, 128)
        if time.time() - t0 > timeout_s:
            raise RuntimeError("rank %d: no RCCL id at %s after %.0f s" % (rank, path, timeout_s))
        time.sleep(0.02)


class RankSync:
    """Host-side barrier and max-over-ranks through the one collective the library has (an RCCL all-gather of one
    double per rank on the context's stream): no torch, no MPI."""

    def __init__(self, ctx, world, rank):
        self.ctx, self.world, self.rank = ctx, world, rank
        self.buf = ctx.empty((max(world, 1),), np.float64) if world > 1 else None

    def gather(self, value):
        if self.world == 1:
            return [value]
        mine = np.array([value], np.float64)
        slot = C.c_void_p(self.buf.ptr.value + 8 * self.rank)
        self.ctx.call("smn_memcpy_h2d", slot, mine.ctypes.data_as(C.c_void_p), 8)
        self.ctx.call("smn_allgather", self.world, 1, slot, self.buf.ptr, 1)   # in place, dtype code 1 = f64
        return list(self.buf.numpy())

    def barrier(self):
        self.ctx.synchronize()
        self.gather(0.0)

    def max(self, value):
        return max(self.gather(value))


class FileSync:
    """The same three calls through files in a directory every rank of the node sees: the replica fall-back when the RCCL
    communicator cannot be brought up (--allow-replica-fallback; there is nothing to exchange then)."""

    def __init__(self, directory, world, rank):
        self.dir, self.world, self.rank, self.k = directory, world, rank, 0

    def gather(self, value, timeout_s=600.0):
        self.k += 1
        _write_atomic(os.path.join(self.dir, "g%d_%d" % (self.k, self.rank)), repr(float(value)))
        out, t0 = [], time.time()
        for r in range(self.world):
            path = os.path.join(self.dir, "g%d_%d" % (self.k, r))
            while True:
                raw = _read_fresh(path)
                try:
                    out.append(float(raw))
                    break
                except (TypeError, ValueError):
                    if time.time() - t0 > timeout_s:
                        raise RuntimeError("rank %d: rank %d never reached step %d of the file rendezvous" % (self.rank, r, self.k))
                    time.sleep(0.0005)
        return out

    def barrier(self):
        self.gather(0.0)

    def max(self, value):
        return max(self.gather(value))


def agree_on_communicator(run_dir, world, rank, ok, timeout_s=180.0):
    """Every rank publishes whether its smn_comm_init succeeded; all of them read all of them (stale files of an earlier run
    are not read).  Returns all_ok."""
    os.makedirs(run_dir, exist_ok=True)
    _write_atomic(os.path.join(run_dir, "init_%d" % rank), "1" if ok else "0")
    t0, flags = time.time(), []
    for r in range(world):
        while True:
            raw = _read_fresh(os.path.join(run_dir, "init_%d" % r))
            if raw is not None:
                flags.append(raw.strip() == "1")
                break
            if time.time() - t0 > timeout_s:
                raise RuntimeError("rank %d: rank %d did not report its communicator within %.0f s" % (rank, r, timeout_s))
            time.sleep(0.005)
    return all(flags)


# ----------------------------------------------------------------------------- helpers
SCHEDULE_KNOBS = ("SMN_XCD_MAP", "SMN_SUPER", "SMN_SUPER_WIDE_ROWS", "SMN_CHAIN_CUS", "SMN_CHAIN_MIN_N", "SMN_PANEL_LEAF")


def pmc_traffic(args, sharded):
    """(bytes per launch, source note): L2-fabric-side bytes per launch of the dominant kernel from the committed rocprofv3
    PMC pass of this exact workload AND schedule (bench.py cannot run the profiler on itself).  The counters were taken
    with every schedule knob at its default; a run that sets one (an A/B line) gets null, not another configuration's figure."""
    if sharded or args.config != "c4" or (args.n, args.d, args.layers, args.act, args.dtype) != (16384, 3072, 4, "relu", "f32"):
        return None, "no PMC pass committed for this workload"
    changed = [k for k in SCHEDULE_KNOBS if os.environ.get(k) is not None]
    if changed:
        return None, "schedule knobs set (%s): the committed PMC pass is of the default schedule" % ", ".join(changed)
    for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f)["traffic_bytes_per_launch"], "profiles/%s (rocprofv3 --pmc, default schedule)" % name
        except Exception:
            continue
    return None, "no PMC file found"


def cpu_baseline(args, np_dtype, eps, gpu_logpdf=None):
    """The CPU oracle (NumPy/SciPy port of the same math) on the benched workload, or a bounded sample of it
    (--cpu-sample-n), on this box's host cores.  Also the checker of the headline number: the oracle's log-pdf and
    its relative difference to the GPU's (same inputs, same dtype) go into the line when N is the benched N."""
    from oracle import host_parallel as HP       # test infrastructure: imported by this leg only
    ns = min(args.cpu_sample_n or min(args.n, 16384), args.n)     # bounded: LAPACK at N = 32768 alone is minutes of host time
    rng = np.random.default_rng(0)
    x = rng.standard_normal((args.n, args.d)).astype(np_dtype)[:ns]     # the GPU's inputs (same seed, same draw order)
    y = rng.standard_normal(args.n).astype(np_dtype)[:ns]
    cores = HP.host_cores()
    # The oracle's maps are NumPy ufunc chains (single-threaded); the reference's CPU path (JAX/XLA) spreads its
    # elementwise work over the host cores.  Same oracle functions, applied to row blocks on a thread pool (ufuncs
    # release the GIL); the Gram and the factorisation go to the multi-threaded BLAS/LAPACK as they stand.
    k, t_build = HP.mlp_kernel_rows_threaded(x, args.layers, args.act, 1.0, 1e-8, 1.0, dtype=np_dtype, cores=cores)
    lp, quad, logdet, t_chol = HP.gaussian_lml(k, y, eps)
    del k
    flops = 2.0 * ns * ns * args.d + ns ** 3 / 3.0
    out = {
        "value": flops / (t_build + t_chol) / 1e9, "unit": "GFLOP/s", "cores": int(cores), "kind": "port",
        "sample": "%s N=%d (d=%d, L=%d %s, %s): NumPy/SciPy oracle (layer maps on a %d-thread pool), build %.2f s + Cholesky/LML %.2f s"
                  % ("the benched workload," if ns == args.n else "same workload at", ns, args.d, args.layers, args.act,
                     np.dtype(np_dtype).name, cores, t_build, t_chol),
        "cpu_logpdf": lp, "cpu_logdet": logdet,
    }
    if ns == args.n and gpu_logpdf is not None:
        out["gpu_logpdf"] = gpu_logpdf
        out["rel_diff_vs_gpu"] = abs(gpu_logpdf - lp) / abs(lp)
    return out


class _JsonOut:
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner to stdout when a communicator comes up, so
    file descriptor 1 is pointed at stderr for the whole run and the JSON line is written to a saved copy of the real one."""

    def __init__(self):
        sys.stdout.flush()
        self.fd = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line):
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)          # whatever C stdio still buffers goes to stderr, not behind the JSON line
        except Exception:
            pass
        os.write(self.fd, (line + "\n").encode())


CATS = ["prep", "build", "recursion", "panel", "strip", "trail", "misc", "comm", "exposed", "stall"]


def read_profile(ctx, per_steps):
    prof = {}
    for cat, name in enumerate(CATS):
        ms, cnt = C.c_double(), C.c_int()
        ctx.call("smn_profile_read", cat, C.byref(ms), C.byref(cnt))
        prof[name] = (ms.value / per_steps, cnt.value // per_steps)
    return prof


# ----------------------------------------------------------------------------- C3: conv-NNGP + Student-t, fp64
def measure_conv(L, ctx, n, nl, act, eps, steps, warmup, cpu_sample_n=0):
    """BASELINE configs[2]: 4-layer conv-NNGP kernel of N CIFAR-shaped images (32x32x3) + Student-t (inverse-gamma scale
    mixture, alpha = beta = 2) log-marginal likelihood, fp64, one GPU.  SURVEY 8(d): VALU-bound, reported as
    pair-pixel-layers per second with the VALU-busy share of the committed PMC pass."""
    h = w = 32; c = 3
    rng = np.random.default_rng(0)
    xh = rng.standard_normal((n, h, w, c))
    xh /= np.sqrt((xh ** 2).mean(axis=(1, 2, 3), keepdims=True))
    yh = (rng.integers(0, 10, n) == 3).astype(np.float64) - 0.1
    x = ctx.to_device(xh); y = ctx.to_device(yh)
    k = ctx.empty((n, n), np.float64)
    lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
    tb = [0.0]

    def step():
        t0 = time.perf_counter()
        ctx.call("smn_kernel_cnn", L.F64, L.ACT[act], nl, 1.3, 0.2, 1.0, x.ptr, n, None, 0, h, w, c, L.FILL_LOWER, k.ptr, n)
        ctx.synchronize()
        tb[0] += time.perf_counter() - t0
        ctx.call("smn_lml", L.F64, k.ptr, n, n, y.ptr, eps, 4.0, 1.0, C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))

    for _ in range(warmup):
        step()
    ctx.synchronize()
    tb[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ms_per_step = dt / steps * 1e3
    build_ms = tb[0] / steps * 1e3
    ppl = n * (n + 1) / 2.0 * h * w * nl                       # pair-pixel-layers of the lower triangle (what is computed)
    valu = None
    for name in ("r03_pmc_cnn.json", "r02_pmc_cnn.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                valu = json.load(f)
            break
        except Exception:
            pass
    out = {
        "metric": "conv-NNGP kernel build + Student-t LML wallclock at N=%d 32x32x3 images, %d-layer %s, fp64" % (n, nl, act),
        "value": ppl / (build_ms * 1e-3), "unit": "pair-pixel-layers/s", "steps": steps,
        "warmup": warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C3: get_cnn_kernel(%d, %s) on %d images 32x32x3 (lower triangle) + smn_lml Student-t df=4" % (nl, act, n),
                   "N": n, "layers": nl, "act": act, "w_std": 1.3, "b_std": 0.2, "eps_abs": eps, "parallelism": "single GPU"},
        "phases_ms": {"kernel_build": round(build_ms, 3), "cholesky_lml": round(ms_per_step - build_ms, 3)},
        "result": {"logpdf": lp.value, "logdet": logdet.value, "info": info.value},
        "roofline": {"kernel": "conv_pair32_kernel<double> (one wave per image pair; 3x3 box sums in registers, one activation map per pixel and layer)",
                     "bound": "valu", "achieved": ppl / (build_ms * 1e-3), "unit": "pair-pixel-layers/s",
                     "valu_busy": None if valu is None else valu.get("valu_busy"),
                     "valu_busy_source": None if valu is None else valu.get("source"),
                     "peak": None, "frac": None if valu is None else valu.get("valu_busy"), "traffic": None},
        "cpu_baseline": None,
    }
    if cpu_sample_n:
        # the oracle's conv kernel (NumPy, one thread) on a bounded sample: the first cpu_sample_n images, full square
        from oracle import nngp_oracle as O          # test infrastructure: imported by this leg only
        ns = min(cpu_sample_n, n)
        t0 = time.perf_counter()
        kc = O.cnn_kernel(xh[:ns], None, nl, act, 1.3, 0.2, 1.0)
        tc = time.perf_counter() - t0
        got = k.numpy()[:ns, :ns]
        il = np.tril_indices(ns)
        out["cpu_baseline"] = {
            "value": ns * ns * float(h * w * nl) / tc, "unit": "pair-pixel-layers/s", "cores": 1, "kind": "port",
            "sample": "the first %d of the %d images, both triangles (the NumPy oracle computes the full square), %.1f s" % (ns, n, tc),
            "max_rel_diff_vs_gpu": float(np.max(np.abs(got[il] - kc[il])) / np.max(np.abs(kc[il])))}
    del x, y, k
    return out


def bench_conv(args, out_fd, L, ctx, sync, rank, world):
    out = measure_conv(L, ctx, args.n, args.layers, args.act, args.eps if args.eps is not None else 1e-4, args.steps, args.warmup,
                       cpu_sample_n=0 if args.no_cpu_baseline else (args.cpu_sample_n or 160))
    out.update({"n_gpus": world, "scaling": "strong", "vs_baseline": None})
    if rank == 0:
        out_fd.emit(json.dumps(out))


# ----------------------------------------------------------------------------- the other single-GPU workloads of BASELINE.json
def measure_mlp_loss(L, ctx, n, d, nl, act, dtype, eps, steps, warmup):
    """One SPR.loss workload (fused build + Cholesky + Gaussian LML) timed like the headline: ms per step, the trailing
    update's launches under hipEvents in the timed region, the other categories in an untimed detail pass."""
    np_dtype = np.float32 if dtype == "f32" else np.float64
    code = L.dtype_code(np_dtype)
    rng = np.random.default_rng(0)
    x = ctx.to_device(rng.standard_normal((n, d)).astype(np_dtype))
    y = ctx.to_device(rng.standard_normal(n).astype(np_dtype))
    lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()

    def step():
        ctx.call("smn_spr_loss", code, L.NET_MLP, L.ACT[act], nl, 1.0, 1e-8, 1.0, x.ptr, n, d, d, y.ptr, eps, 0.0, 1.0,
                 C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))

    for _ in range(warmup):
        step()
    ctx.synchronize()
    ctx.call("smn_profile_enable", 2 << 5)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.synchronize()
    ms_per_step = (time.perf_counter() - t0) / steps * 1e3
    ms, cnt = C.c_double(), C.c_int()
    ctx.call("smn_profile_read", 5, C.byref(ms), C.byref(cnt))
    trail_ms, trail_cnt = ms.value / steps, cnt.value // steps
    fl = {}
    for cat in (4, 5):
        v = C.c_double()
        ctx.call("smn_profile_flops", cat, C.byref(v))
        fl[cat] = v.value / steps
    ctx.call("smn_profile_enable", 1)
    step()
    ctx.synchronize()
    prof = read_profile(ctx, 1)
    ctx.call("smn_profile_enable", 0)
    peak = PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_F64_MFMA_TFLOPS
    flops_counted = 2.0 * n * n * d + n ** 3 / 3.0
    chol_wall = ms_per_step - prof["build"][0] - prof["prep"][0]
    out = {
        "workload": "SPR.loss: N=%d d=%d L=%d %s %s" % (n, d, nl, act, dtype),
        "ms_per_step": ms_per_step, "steps": steps, "warmup": warmup, "value": flops_counted / (ms_per_step * 1e-3) / 1e9,
        "unit": "GFLOP/s", "dtype": dtype, "eps_abs": eps,
        "phases_ms": {k: round(v[0], 4) for k, v in prof.items() if v[0] > 0 and k != "trail"},
        "result": {"logpdf": lp.value, "logdet": logdet.value, "info": info.value},
        "roofline": {"kernel": "Cholesky trailing update (update_kernel / trail_kernel)", "bound": "mfma",
                     "achieved": fl[5] / (trail_ms * 1e-3) / 1e12 if trail_ms > 0 else None, "peak": peak, "unit": "TFLOP/s",
                     "frac": fl[5] / (trail_ms * 1e-3) / 1e12 / peak if trail_ms > 0 else None,
                     "peak_source": "datasheet (%s MFMA, dense)" % dtype,
                     "frac_vs_measured_issue_ceiling": fl[5] / (trail_ms * 1e-3) / 1e12 / MEASURED_ISSUE_CEILING_TFLOPS[dtype] if trail_ms > 0 else None,
                     "launches_per_step": trail_cnt, "summed_launch_ms": trail_ms,
                     "cholesky_wall_ms": chol_wall,
                     "cholesky_mfma_frac": (fl[4] + fl[5]) / (chol_wall * 1e-3) / 1e12 / peak},
    }
    out["phases_ms"]["trail_summed"] = round(trail_ms, 4)
    del x, y
    return out


def measure_predict(L, ctx, n, d, nl, act, t, steps, warmup):
    """The predictive path at the headline size through the spax facade (spax/kernels.py:29-32, spax/models.py:100-120):
    SPR.test_nll with a Gaussian likelihood (ONE partial factorisation of the joint kernel: posterior mean, covariance,
    NLL) and with the Student-t likelihood (the same, plus the fp64 build + factorisation that y^T (b/a K + 1e-6 I)^-1 y
    costs: spax/likelihoods.py:60-61)."""
    from smnngp import nt_kernels
    from smnngp.spax.kernels import NNGPKernel
    from smnngp.spax.likelihoods import GaussianLikelihood, StudentTLikelihood
    from smnngp.spax.models import SPR
    rng = np.random.default_rng(0)
    xh = rng.standard_normal((n, d)).astype(np.float32)
    yh = rng.standard_normal(n).astype(np.float32)
    xt = ctx.to_device(rng.standard_normal((t, d)).astype(np.float32))
    yt = rng.standard_normal(t)
    xd = ctx.to_device(xh)
    kernel = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(nl, act=act, w_std=w, b_std=b, last_w_std=l), 1.0, 1e-8, 1.0)
    flops = n ** 3 / 3.0 + float(n) * n * t + float(n) * t * t          # factorisation + K_td L^-T + Schur complement
    out = {"workload": "SPR.test_nll: N=%d d=%d L=%d %s f32, T=%d test rows, C=1 (relative ridge 1e-3)" % (n, d, nl, act, t),
           "flops_counted": flops, "flops_note": "N^3/3 + N^2 T + N T^2 (kernel builds not counted)"}
    for name, lik in (("gaussian", GaussianLikelihood()), ("student_t", StudentTLikelihood(2.0, 2.0))):
        model = SPR(kernel, lik, xd, yh, 0.0, 1.0, eps=1e-3)
        val = None

        def call():
            model._quad64_cache = None        # every timed call pays for everything (the model keeps the fp64 quadratic form
            return model.test_nll(xt, yt)     # of the last hyper-parameter setting: timed separately below)
        for _ in range(warmup):
            val = call()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            val = call()
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out[name] = {"ms_per_call": ms, "test_nll": float(val),
                     "roofline": {"kernel": "partial Cholesky of the joint kernel [[K+ridge, .], [K_td, K_tt]] (trailing updates carry the test rows)",
                                  "bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                                  "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}}
        if name == "student_t":
            t0 = time.perf_counter()
            for _ in range(steps):
                val2 = model.test_nll(xt, yt)
            ctx.synchronize()
            out[name]["ms_repeat_call_same_hyperparameters"] = (time.perf_counter() - t0) / steps * 1e3
            out[name]["repeat_note"] = ("validation + test split of one check point (train.py:203-212): the fp64 quadratic form "
                                        "depends on the training data and hyper-parameters only and is kept; same value: %s" % (val2 == val))
        del model
    out["student_t_extra_ms"] = out["student_t"]["ms_per_call"] - out["gaussian"]["ms_per_call"]
    out["student_t_extra_note"] = ("the fp64 kernel build + factorisation behind y^T (b/a K + 1e-6 I)^-1 y (spax/models.py:107, likelihoods.py:60-61); "
                                   "bound by the f64 matrix rate: running it beside the fp32 posterior on a second context buys 3 % (profiles/r04_two_context_probe.txt)")
    return out


def measure_loss_grad(L, ctx, n, d, nl, act, steps, warmup):
    """One analytic loss + gradient call (SPR.loss_and_grad: what objax.GradValues(model.loss, vars) provides at
    experiments/regression/train.py:61-67) beside SPR.loss at the same size, Student-t head.  The gradient needs K~^-1 in
    full: N^3 flops of factorisation work against the loss's N^3/3."""
    from smnngp import nt_kernels
    from smnngp.spax.kernels import NNGPKernel
    from smnngp.spax.likelihoods import StudentTLikelihood
    from smnngp.spax.models import SPR
    rng = np.random.default_rng(0)
    xd = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32))
    yh = rng.standard_normal(n).astype(np.float32)
    kernel = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(nl, act=act, w_std=w, b_std=b, last_w_std=l), 1.0, 0.3, 1.0)
    model = SPR(kernel, StudentTLikelihood(2.0, 2.0), xd, yh, 0.0, 1.0, eps=1e-2)
    res = {}
    for name, fn in (("loss", model.loss), ("loss_and_grad", model.loss_and_grad)):
        for _ in range(warmup):
            fn()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            val = fn()
        ctx.synchronize()
        res[name] = ((time.perf_counter() - t0) / steps * 1e3, val)
    flops = 2.0 * n * n * d + float(n) ** 3                               # Gram + (factor, L^-1, L^-T L^-1)
    ms = res["loss_and_grad"][0]
    loss, grads = res["loss_and_grad"][1]
    return {"workload": "SPR.loss_and_grad: N=%d d=%d L=%d %s f32, Student-t head, gradients of w_std, b_std, last_w_std, eps, a, b" % (n, d, nl, act),
            "ms_per_step": ms, "loss_ms": res["loss"][0], "loss_evaluations": ms / res["loss"][0],
            "loss": float(loss), "grad_w_std": float([v for k, v in grads.items() if k.endswith("w_std") and "last" not in k][0]),
            "flops_counted": flops, "flops_note": "2 N^2 d + N^3 (factorisation N^3/3, L^-T N^3/3, L^-T L^-1 N^3/3; the contraction pass not counted)",
            "roofline": {"kernel": "augmented factorisation [[K~, .], [I, 0]] with the identity block's structural zeros left out of every launch",
                         "bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}}


def measure_sweep(L, ctx, n, t, d, nl, act, steps=2):
    """The reference's grid search (experiments/regression/find.py:134-199, its default grid: 3 w_std x 3 b_std x 11 eps x
    3 alpha x 3 beta) on one synthetic data set of N training and T test points: 99 problems, two factorisations each.
    Batched (sweeps.find_grid: smn_spr_predict_batch + smn_spr_loss_batch + smn_mixture_nll, grid.y = 99) beside the
    same 198 factorisations issued one by one (smn_spr_predict + smn_spr_loss per cell: what a loop over the serial
    entry points costs)."""
    from smnngp import sweeps
    rng = np.random.default_rng(0)
    xh = rng.standard_normal((n, d)).astype(np.float32)
    yh = np.sin(xh[:, 0]) + 0.1 * rng.standard_normal(n).astype(np.float32)
    xth = rng.standard_normal((t, d)).astype(np.float32)
    yth = np.sin(xth[:, 0])
    ws, bs = (1.0, 1.4, 2.0), (0.0, 0.3, 1.0)
    es = tuple(float("1e%d" % v) for v in range(-6, 5))
    kw = dict(network="mlp", num_hiddens=nl, activation=act, w_std_list=ws, b_std_list=bs, eps_list=es,
              alpha_list=(1.0, 2.0, 3.0), beta_list=(1.0, 2.0, 3.0), ctx=ctx)
    x, xt = ctx.to_device(xh), ctx.to_device(xth)
    y = ctx.to_device(yh.reshape(n, 1).astype(np.float32))
    got = sweeps.find_grid(x, yh, xth, yth, **kw)                       # warm-up (workspaces, first-launch costs)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        got = sweeps.find_grid(x, yh, xth, yth, **kw)
    ctx.synchronize()
    total_ms = (time.perf_counter() - t0) / steps * 1e3
    cells = [(w, b, e) for w in ws for b in bs for e in es]
    wv, bv, ev = (np.array(v) for v in zip(*cells))
    bkw = dict(network="mlp", num_hiddens=nl, activation=act, w_std=wv, b_std=bv, last_w_std=1.0)
    t0 = time.perf_counter()
    for _ in range(steps):
        sweeps.predict_batch(ctx, x, y, xt, diag_reg=ev, on_device=True, **bkw)
        sweeps.loss_batch(ctx, x, y, eps=ev, **bkw)
    ctx.synchronize()
    batch_ms = (time.perf_counter() - t0) / steps * 1e3
    # the same 198 factorisations through the serial entry points
    mean_d, cov_d = ctx.empty((t, 1), np.float32), ctx.empty((t, t), np.float32)
    lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()

    def serial():
        for w, b, e in cells:
            ctx.call("smn_spr_predict", L.F32, L.NET_MLP, L.ACT[act], nl, w, b, 1.0, x.ptr, n, d, xt.ptr, t, d, d, y.ptr, 1, e, 0.0,
                     mean_d.ptr, cov_d.ptr, t, None, None, C.byref(info))
            ctx.call("smn_spr_loss", L.F32, L.NET_MLP, L.ACT[act], nl, w, b, 1.0, x.ptr, n, d, d, y.ptr, e, 0.0, 1.0,
                     C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
    serial()
    ctx.synchronize()
    t0 = time.perf_counter()
    serial()
    ctx.synchronize()
    serial_ms = (time.perf_counter() - t0) * 1e3
    nprob = 2 * len(cells)
    flops = len(cells) * (2.0 * n ** 3 / 3.0 + float(n) * n * t + float(n) * t * t + 2.0 * (2 * n * n + 2 * n * t + t * t) * d)
    return {"workload": "find_grid (find.py:134-199 default grid 3x3x11 x 3x3): N=%d T=%d d=%d L=%d %s f32, %d factorisations" % (n, t, d, nl, act, nprob),
            "ms_per_sweep": total_ms, "device_batches_ms": batch_ms, "serial_calls_ms": serial_ms,
            "speedup_vs_serial_calls": serial_ms / batch_ms, "problems_per_s": nprob / (batch_ms * 1e-3),
            "value": flops / (batch_ms * 1e-3) / 1e9, "unit": "GFLOP/s",
            "flops_note": "per cell 2 N^3/3 + N^2 T + N T^2 + the two Gram builds; batches only (the mixture kernel and the host tables are in ms_per_sweep)",
            "finite_cells": int(np.isfinite(got["gnll"]).sum()), "best_gaussian": got["best_gaussian"], "best_student": got["best_student"]}


def measure_small_n(L, ctx, n, d, nl, act, g, steps=20):
    """The reference's training size (train.py:178-212: N = 245): G independent SPR.loss evaluations (a population of
    hyper-parameter settings) as one batched call beside G serial calls."""
    from smnngp import sweeps
    rng = np.random.default_rng(0)
    x = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32))
    y = ctx.to_device(rng.standard_normal((n, 1)).astype(np.float32))
    wv = np.linspace(0.8, 2.0, g); bv = np.linspace(0.0, 1.0, g); ev = np.full(g, 1e-2)
    kw = dict(network="mlp", num_hiddens=nl, activation=act, w_std=wv, b_std=bv, last_w_std=1.0, eps=ev)
    sweeps.loss_batch(ctx, x, y, **kw)
    t0 = time.perf_counter()
    for _ in range(steps):
        sweeps.loss_batch(ctx, x, y, **kw)
    batch_us = (time.perf_counter() - t0) / steps * 1e6
    lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()

    def serial():
        for w, b in zip(wv, bv):
            ctx.call("smn_spr_loss", L.F32, L.NET_MLP, L.ACT[act], nl, w, b, 1.0, x.ptr, n, d, d, y.ptr, 1e-2, 0.0, 1.0,
                     C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
    serial()
    t0 = time.perf_counter()
    for _ in range(3):
        serial()
    serial_us = (time.perf_counter() - t0) / 3 * 1e6
    return {"workload": "%d x SPR.loss: N=%d d=%d L=%d %s f32 (train.py-sized problems)" % (g, n, d, nl, act),
            "batched_us_per_problem": batch_us / g, "serial_us_per_problem": serial_us / g, "speedup": serial_us / batch_us,
            "problems_per_s": g / (batch_us * 1e-6)}


def other_workloads(L, ctx, budget_s=330.0):
    """BASELINE.json's other single-GPU configurations and the predictive path, measured in this process after the
    headline (each in its own try: a failing or skipped workload never breaks the bench line)."""
    t_start = time.perf_counter()
    out = {}
    plan = [
        ("c2", lambda: measure_mlp_loss(L, ctx, 4096, 512, 3, "relu", "f32", 1e-3, 20, 3)),
        ("sweep_n2048", lambda: measure_sweep(L, ctx, 2048, 256, 16, 4, "relu")),
        ("batch_n245", lambda: measure_small_n(L, ctx, 245, 6, 2, "relu", 256)),
        ("predict_c4", lambda: measure_predict(L, ctx, 16384, 3072, 4, "relu", 2048, 3, 1)),
        ("grad_c4", lambda: measure_loss_grad(L, ctx, 16384, 3072, 4, "relu", 3, 1)),
        ("f64_n8192", lambda: measure_mlp_loss(L, ctx, 8192, 3072, 4, "relu", "f64", 1e-6, 5, 1)),
        ("c5", lambda: measure_mlp_loss(L, ctx, 32768, 1024, 6, "erf", "f32", 1e-3, 4, 1)),
        ("c3", lambda: measure_conv(L, ctx, 10000, 4, "relu", 1e-4, 2, 1)),
    ]
    for name, fn in plan:
        if time.perf_counter() - t_start > budget_s:
            out[name] = {"skipped": "time budget of %.0f s for the extra workloads used up" % budget_s}
            continue
        try:
            t0 = time.perf_counter()
            out[name] = fn()
            out[name]["measure_wall_s"] = round(time.perf_counter() - t0, 2)
        except Exception as e:   # noqa: BLE001
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


# ----------------------------------------------------------------------------- main
def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args)                         # never returns; nothing below ran in the parent
    out_fd = _JsonOut()
    world = int(os.environ.get("WORLD_SIZE", "1")) if "RANK" in os.environ else 1
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    from smnngp import _lib as L

    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    code = L.dtype_code(np_dtype)
    eps = args.eps if args.eps is not None else (1e-3 if args.dtype == "f32" else 1e-6)
    act = L.ACT[args.act]
    n, d, nl = args.n, args.d, args.layers
    # SMN_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box only): every rank on device 0 -- RCCL refuses such a communicator,
    # which exercises the replica fall-back below
    ctx = L.Context(0 if os.environ.get("SMN_BENCH_SHARE_GPU") == "1" else local_rank)

    sharded = world > 1 or args.sharded_path
    comm = {"rccl_ranks": 0, "fallback": None}     # top level of the line: how many ranks the RCCL communicator has, or why none
    wd = None
    if world > 1:
        # under an external launcher the file name carries the launcher's own run id where it exports one, so that two
        # runs can never meet in one file; files older than the launcher itself are ignored (_read_fresh)
        path = args.rendezvous_file or os.path.join(
            tempfile.gettempdir(), "smnngp_uid_%s_%s_%d" % (os.environ.get("MASTER_PORT", "0"),
                                                            os.environ.get("TORCHELASTIC_RUN_ID", "run"), os.getppid()))
        run_dir = path + ".run"
        wd = Watchdog(run_dir, rank, world, out_fd)
        err = None
        wd.phase("rccl id")
        try:
            uid = exchange_rccl_id(L, path, rank)
            wd.phase("smn_comm_init")
            ctx.call("smn_comm_init", world, rank, uid)
        except Exception as e:   # noqa: BLE001  (no RCCL, no peer access, ...): agree with the other ranks on what to do
            err = "%s: %s" % (type(e).__name__, e)
        wd.phase("agree on the communicator")
        all_ok = agree_on_communicator(run_dir, world, rank, err is None)
        if all_ok:
            sync = RankSync(ctx, world, rank)
            comm["rccl_ranks"] = world
        else:
            if err is None:
                ctx.call("smn_comm_destroy")
            comm["fallback"] = err or "another rank could not join the RCCL communicator"
            if not args.allow_replica_fallback:
                # a P-GPU line measured on P replicas would read like a flat scaling curve: no line of that kind, a failure
                if rank == 0:
                    out_fd.emit(json.dumps({
                        "metric": "kernel-build + Cholesky wallclock (ms) and GFLOP/s", "value": None, "unit": "GFLOP/s",
                        "n_gpus": world, "higher_is_better": True, "comm": comm,
                        "error": "no RCCL communicator over %d ranks (%s); nothing was measured "
                                 "(--allow-replica-fallback runs independent replicas instead)" % (world, comm["fallback"])}))
                wd.done()
                sys.exit(4)
            # --allow-replica-fallback: every rank runs the whole single-GPU step, timed with the same barrier and max over
            # ranks through files; the line says "scaling": "replicas"
            sys.stderr.write("bench.py rank %d: no communicator (%s): running %d independent replicas\n" % (rank, comm["fallback"], world))
            sharded = False
            sync = FileSync(run_dir, world, rank)
    else:
        if args.sharded_path:                      # one-rank communicator: the P>1 code path on a one-GPU box
            uid = C.create_string_buffer(128)
            assert L._lib.smn_comm_unique_id(uid) == 0, "RCCL unavailable"
            ctx.call("smn_comm_init", 1, 0, uid)
            comm["rccl_ranks"] = 1
        sync = RankSync(ctx, world, rank)
    say = wd.phase if wd is not None else (lambda name: None)
    if world > 1:
        say("first barrier")
        sync.barrier()                             # every rank has read the id and every status file

    if args.config == "c3":
        bench_conv(args, out_fd, L, ctx, sync, rank, world)
        if sharded:
            sync.barrier()
            ctx.call("smn_comm_destroy")
        if wd is not None:
            wd.done()
        return

    rng = np.random.default_rng(0)                 # same seed on every rank: X is replicated
    x = ctx.to_device(rng.standard_normal((n, d)).astype(np_dtype))
    y = ctx.to_device(rng.standard_normal(n).astype(np_dtype))
    lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
    res = {}

    if not sharded:
        def step():
            ctx.call("smn_spr_loss", code, L.NET_MLP, act, nl, 1.0, 1e-8, 1.0, x.ptr, n, d, d, y.ptr, eps, 0.0, 1.0,
                     C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
            res["v"] = (lp.value, logdet.value, info.value)
        parts = 0
    else:
        # Cyclic column-first shard (sharding.py): rank r builds every lower tile of its tile rows in ONE launch into its own
        # chunk; the chunk goes out column range by column range (RCCL all-gather + scatter into the factorisation workspace
        # on side streams) while the factorisation, issued right behind, waits for each range only where it first needs it.
        from smnngp import sharding
        cols = sharding.default_col_pieces(n, world, max_pieces=args.pieces) if args.pieces else sharding.default_col_pieces(n, world)
        lay = sharding.col_layout(n, world, cols)
        parts = len(cols) - 1
        mine = ctx.empty((lay["elems"],), np_dtype)
        stage = ctx.empty((world * lay["elems"],), np_dtype)
        backend = sharding.DeviceBackend(ctx)
        spec = (L.NET_MLP, act, nl, 1.0, 1e-8, 1.0)
        # BASELINE config 5 ("erf NNGP + NTK ... shard + single-GPU Cholesky on assembled kernel"): the build launch is the
        # joint NNGP + NTK one and the NTK's pieces ride the same streams into a full matrix of the caller's
        ntk = None
        if args.config == "c5" or args.with_ntk:
            ntk_arrays = (ctx.empty((lay["elems"],), np_dtype), ctx.empty((world * lay["elems"],), np_dtype), ctx.empty((n, n), np_dtype))
            res["ntk_arrays"] = ntk_arrays                  # owners stay alive for the run
            ntk = (ntk_arrays[0].ptr, ntk_arrays[1].ptr, ntk_arrays[2].ptr, n)
        res["k"] = 0

        def step():
            res["k"] += 1
            k = res["k"]
            v = sharding.lml_sharded_cols(backend, code, spec, x.ptr, n, d, d, y.ptr, rank, world, mine.ptr, stage.ptr,
                                          eps, 0.0, 1.0, cols=cols, ntk=ntk, progress=lambda ph: say("step %d: %s" % (k, ph)))
            res["v"] = (v[0], v[2], v[3])

    say("warm-up")
    for _ in range(args.warmup):
        step()
    say("barrier before the timed region")
    sync.barrier()
    # Timed region: hipEvent pairs around the launches of the DOMINANT kernel only (category 5, the Cholesky trailing
    # update: 79 launches per step).  Pairs around all ~280 launches of a step cost ~2 ms of queue time per step
    # (profiles/r01e_event_overhead.txt), so the other categories are timed in a separate, untimed pass below.
    ctx.call("smn_profile_enable", 2 << 5)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    say("barrier after the timed region")
    sync.barrier()
    dt = time.perf_counter() - t0
    say("detail pass")
    ms, cnt = C.c_double(), C.c_int()
    ctx.call("smn_profile_read", 5, C.byref(ms), C.byref(cnt))
    trail_timed = (ms.value / max(args.steps, 1), cnt.value // max(args.steps, 1))
    # MFMA flops the library EXECUTED (whole tiles, counted as the launches are issued) per step, by category
    fl = {}
    for cat in (4, 5):
        v = C.c_double()
        ctx.call("smn_profile_flops", cat, C.byref(v))
        fl[cat] = v.value / max(args.steps, 1)
    DETAIL_STEPS = 2
    ctx.call("smn_profile_enable", 1)              # untimed detail pass: every category
    for _ in range(DETAIL_STEPS):
        step()
    sync.barrier()
    prof = read_profile(ctx, DETAIL_STEPS)
    ctx.call("smn_profile_enable", 0)
    prof["trail"] = trail_timed
    dt = sync.max(dt)
    ms_per_step = dt / args.steps * 1e3

    if rank == 0:
        n_total = n + TILE
        flops_counted = 2.0 * n * n * d + n ** 3 / 3.0                 # Gram 2N^2 d + Cholesky N^3/3
        trail_fl, strip_fl = fl[5], fl[4]
        peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_F64_MFMA_TFLOPS
        per = prof                                 # per step: `trail` from the timed region, the rest from the detail pass
        kp = ((d + 31) // 32 * 32) if args.dtype == "f32" else ((d + 15) // 16 * 16)
        t = n_total // TILE
        if not sharded:
            build_tiles = t * (t + 1) // 2
        else:
            from smnngp import sharding as S_
            build_tiles = sum(tt + 1 for tt in S_.rank_tile_rows(n, world, 0))   # lower tiles rank 0's tile rows hold
        build_fl = build_tiles * TILE * TILE * 2.0 * kp
        trail_ms = per["trail"][0]
        roof = {
            "kernel": ("update_kernel<float,1> + trail_kernel<float> (persistent form, launches over 512 tiles)"
                       if args.dtype == "f32" else "update_kernel<double,1>")
                      + ": Cholesky trailing update C -= P P^T on lower 128x128 tiles (two-level: K=256 inside a "
                        "super-panel, K=super-panel width beyond it; from N=8192 the far update is split and its bulk "
                        "runs on a CU-masked stream beside the next super-panel's panel chain)",
            "bound": "mfma",
            "achieved": trail_fl / (trail_ms * 1e-3) / 1e12 if trail_ms > 0 else None,
            "peak": peak, "unit": "TFLOP/s",
            "frac": (trail_fl / (trail_ms * 1e-3) / 1e12 / peak) if trail_ms > 0 else None,
            "peak_source": "datasheet (%s MFMA, dense): %.1f TFLOP/s" % (args.dtype, peak),
            "frac_vs_measured_issue_ceiling": (trail_fl / (trail_ms * 1e-3) / 1e12 / MEASURED_ISSUE_CEILING_TFLOPS[args.dtype]) if trail_ms > 0 else None,
            "measured_issue_ceiling": "%.1f TFLOP/s: independent v_mfma_%s_16x16x4 chains, whole chip (profiles/r04_mfma_forms.txt)"
                                      % (MEASURED_ISSUE_CEILING_TFLOPS[args.dtype], args.dtype),
            # fabric-side bytes per launch cannot be read without rocprofv3: taken from the committed PMC pass of this exact
            # workload, else null
            "traffic": pmc_traffic(args, sharded)[0], "traffic_source": pmc_traffic(args, sharded)[1],
            "launches_per_step": per["trail"][1], "avg_launch_ms": trail_ms / max(per["trail"][1], 1),
            "flops_per_step": trail_fl,
        }
        # Look-ahead (default from N=8192): the far updates run on a CU-masked stream BESIDE the next super-panel's panel
        # chain, so the launch durations above overlap with other kernels and `frac` (kept as the contract defines it)
        # understates the kernel.  Two more readings of the same kernel:
        min_n_env = os.environ.get("SMN_CHAIN_MIN_N")
        lookahead = n_total >= int(min_n_env or "8192") and int(os.environ.get("SMN_CHAIN_CUS", "32")) > 0
        # (sharded: the all-gathers and scatters ride under the factorisation on side streams; what the main stream waited for
        # before its first panel is `exposed` and is not the factorisation's time)
        build_wall_ms = per["build"][0]
        chol_wall_ms = ms_per_step - build_wall_ms - per["prep"][0] - (0.0 if sharded else per["misc"][0]) - per["exposed"][0]
        roof["cholesky_wall_ms"] = chol_wall_ms
        if not sharded and t >= 112 and lookahead:
            roof["cholesky_wall_note"] = ("split build: the kernel matrix's corner (a second build launch, ~1.2 ms of phases_ms.build at "
                                          "C4) runs beside the factorisation's first panel chain; cholesky_wall_ms = step - build launches - "
                                          "prep, i.e. that overlap is booked to the build")
        # every MFMA flop of the factorisation (trailing + strip updates) over its wall time, panel chain included
        roof["cholesky_mfma_frac"] = (trail_fl + strip_fl) / (chol_wall_ms * 1e-3) / 1e12 / peak
        roof["lookahead"] = bool(lookahead)
        if lookahead and not sharded and not args.no_exclusive_probe:
            # the same launches with the look-ahead off (one stream, nothing else on the GPU): an untimed extra pass on a
            # second context created with SMN_CHAIN_MIN_N out of reach
            try:
                os.environ["SMN_CHAIN_MIN_N"] = "1000000000"
                ctx2 = L.Context(local_rank)
                x2 = ctx2.to_device(x.numpy()); y2 = ctx2.to_device(y.numpy())

                def step2():
                    ctx2.call("smn_spr_loss", code, L.NET_MLP, act, nl, 1.0, 1e-8, 1.0, x2.ptr, n, d, d, y2.ptr, eps, 0.0,
                              1.0, C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
                step2(); ctx2.synchronize()
                ctx2.call("smn_profile_enable", 1)
                for _ in range(2):
                    step2()
                ctx2.synchronize()
                ms2, cnt2 = C.c_double(), C.c_int()
                ctx2.call("smn_profile_read", 5, C.byref(ms2), C.byref(cnt2))
                ctx2.call("smn_profile_enable", 0)
                ex_ms = ms2.value / 2
                roof["frac_exclusive"] = trail_fl / (ex_ms * 1e-3) / 1e12 / peak
                roof["exclusive_ms_per_step"] = ex_ms
                del x2, y2, ctx2
            except Exception as e:
                roof["frac_exclusive"] = None
                roof["exclusive_error"] = str(e)
            finally:
                if min_n_env is None:
                    del os.environ["SMN_CHAIN_MIN_N"]
                else:
                    os.environ["SMN_CHAIN_MIN_N"] = min_n_env
        others = {}
        if per["build"][0] > 0:
            others["build_kernel (fused Gram + %d-layer recursion, executed tiles)" % nl] = {
                "bound": "mfma", "achieved": build_fl / (per["build"][0] * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": build_fl / (per["build"][0] * 1e-3) / 1e12 / peak, "ms": per["build"][0]}
        # what the step EXECUTED on the MFMA (lower-triangle Gram tiles, whole update tiles) over its wall time, beside the
        # SURVEY-sanctioned counted rate in `value`
        executed = build_fl + trail_fl + strip_fl
        out = {
            "metric": "kernel-build + Cholesky wallclock (ms) and GFLOP/s at N=%d, %d-layer %s NNGP" % (n, nl, args.act),
            "value": flops_counted / (ms_per_step * 1e-3) / 1e9, "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if (sharded or world == 1) else "replicas", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic", "comm": comm,
            "config": {"workload": "%s: SPR.loss = NNGP kernel build + jittered Cholesky + Gaussian LML, N=%d d=%d L=%d %s"
                                   % (args.config.upper(), n, d, nl, args.act),
                       "N": n, "d": d, "layers": nl, "act": args.act, "w_std": 1.0, "b_std": 1e-8, "last_w_std": 1.0,
                       "eps_abs": eps, "flops_counted": flops_counted,
                       "parallelism": ("single GPU" if world == 1 else "%d independent replicas (--allow-replica-fallback; no RCCL communicator: %s)" % (world, comm["fallback"])) if not sharded else
                       "cyclic tile-row shards x%d (one build launch per rank), column-first exchange in %d ranges (RCCL all-gather + scatter per range on side streams, consumed by the factorisation range by range), replicated Cholesky" % (world, parts)},
            "executed_tflops": executed / (ms_per_step * 1e-3) / 1e12,
            "executed_note": "MFMA flops actually issued per step (lower-triangle Gram tiles + whole update tiles; rank 0's share of the build when sharded) / step time; `value` counts 2N^2 d + N^3/3",
            "phases_ms": {k: round(v[0], 4) for k, v in per.items()},
            "phases_ms_source": "trail: hipEvents in the timed region; others: separate untimed pass with events on every launch",
            "result": {"logpdf": res["v"][0], "logdet": res["v"][1], "info": res["v"][2]},
            "roofline": roof,
        }
        if sharded:
            # The north-star's "kernel-build speed-up at N GPUs", stated with and without the exchange.  The rank's build is ONE
            # launch on every CU (`kernel_build_ms`).  The all-gathers (`exchange_ms`, summed over the column ranges) and the
            # scatters (`scatter_ms`) run on side streams UNDER the factorisation; what the main stream waited for between the
            # end of its build and its first panel -- the first column range's gather + scatter -- is `exchange_exposed_ms`, and
            # any later wait of the factorisation for a range that had not landed yet is `exchange_stall_ms` (0 when the first
            # panel chain covers the rest of the exchange).  The whole step is Amdahl-bound: every rank repeats the
            # factorisation (north_star: "single-GPU Cholesky on the assembled kernel").
            out["kernel_build_ms"] = round(per["build"][0], 4)
            out["exchange_ms"] = round(per["comm"][0], 4)
            out["scatter_ms"] = round(per["misc"][0], 4)
            out["exchange_exposed_ms"] = round(per["exposed"][0], 4)
            out["exchange_stall_ms"] = round(per["stall"][0], 4)
            out["build_plus_exposed_assembly_ms"] = round(per["build"][0] + per["prep"][0] + per["exposed"][0], 4)
            out["exchange_ranges"] = parts
            out["exchange_bytes_in_per_rank"] = int((world - 1) * lay["elems"] * np.dtype(np_dtype).itemsize * (2 if ntk is not None else 1))
            out["whole_step_note"] = ("strong scaling of the kernel build only: the factorisation (%.1f ms of the step) is replicated on every rank, "
                                      "so ms_per_step is Amdahl-bound" % chol_wall_ms)
            # the same shape's fused lower build on ONE GPU, measured here and now on this rank (one un-sharded step), for the ratio
            try:
                ctx.call("smn_debug_split_build", 0)      # ONE build launch with the chip to itself (the fused call otherwise
                ctx.call("smn_profile_enable", 2 << 1)    # builds the matrix's corner beside the first panel chain)
                try:
                    for _ in range(2):
                        ctx.call("smn_spr_loss", code, L.NET_MLP, act, nl, 1.0, 1e-8, 1.0, x.ptr, n, d, d, y.ptr, eps, 0.0, 1.0,
                                 C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
                finally:
                    ctx.call("smn_debug_split_build", 1)
                ms1, cnt1 = C.c_double(), C.c_int()
                ctx.call("smn_profile_read", 1, C.byref(ms1), C.byref(cnt1))
                ctx.call("smn_profile_enable", 0)
                one = ms1.value / max(cnt1.value, 1)
                out["one_gpu_build_ms"] = round(one, 4)
                out["build_only_speedup"] = round(one / max(per["build"][0], 1e-9), 3)
                out["build_plus_exposed_assembly_speedup"] = round((one + per["prep"][0]) / max(out["build_plus_exposed_assembly_ms"], 1e-9), 3)
            except Exception as e:
                out["one_gpu_build_error"] = str(e)
        # stand-alone recursion (a3): HBM roofline probe on a stored K0, outside the timed region
        if world == 1 and not sharded and not args.no_recursion_probe:
            try:
                k0 = ctx.empty((n, n), np_dtype); kk = ctx.empty((n, n), np_dtype)
                q1 = ctx.empty((n,), np_dtype)
                ctx.call("smn_gram", code, x.ptr, n, d, None, 0, 0, d, k0.ptr, n, q1.ptr, None)
                for _ in range(2):
                    ctx.call("smn_recursion", code, L.NET_MLP, act, nl, 1.0, 1e-8, 1.0, k0.ptr, n, n, n, q1.ptr, q1.ptr,
                             1, L.GET_NNGP, kk.ptr, None, n)
                ctx.call("smn_profile_enable", 1)
                reps = 5
                for _ in range(reps):
                    ctx.call("smn_recursion", code, L.NET_MLP, act, nl, 1.0, 1e-8, 1.0, k0.ptr, n, n, n, q1.ptr, q1.ptr,
                             1, L.GET_NNGP, kk.ptr, None, n)
                ms, cnt = C.c_double(), C.c_int()
                ctx.call("smn_profile_read", 2, C.byref(ms), C.byref(cnt))
                ctx.call("smn_profile_enable", 0)
                rec_ms = ms.value / max(cnt.value, 1)
                nbytes = 2.0 * n * n * np.dtype(np_dtype).itemsize
                rec_traffic = None
                if (args.n, args.layers, args.act, args.dtype) == (16384, 4, "relu", "f32"):
                    try:
                        for name in ("r04_pmc_recursion.json", "r01e_pmc_recursion.json"):
                            fn = os.path.join(ROOT, "profiles", name)
                            if os.path.exists(fn):
                                with open(fn) as f:
                                    rec_traffic = json.load(f)["traffic_bytes_per_launch"]
                                break
                    except Exception:
                        rec_traffic = None
                # achieved = SURVEY 8(d) algorithmic bytes (read N^2 + write N^2) / time; the symmetric kernel reads
                # only the lower tiles, so the bytes it really moves (`traffic`, PMC) are ~0.77x of that
                others["recursion_sym_kernel (stand-alone %d-layer %s map over a stored symmetric K0)" % (nl, args.act)] = {
                    "bound": "hbm", "achieved": nbytes / (rec_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": nbytes / (rec_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": rec_traffic, "ms": rec_ms}
                del k0, kk
            except Exception as e:  # the probe must never break the bench line
                others["recursion_kernel"] = {"error": str(e)}
        out["roofline_other_kernels"] = others
        if world == 1 and not sharded and not args.no_other_workloads and args.config == "c4" and \
                (args.n, args.d, args.layers, args.act, args.dtype) == (16384, 3072, 4, "relu", "f32"):
            del x, y                                   # the headline's inputs: the extra workloads bring their own
            out["other_workloads"] = other_workloads(L, ctx)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, np_dtype, eps, res["v"][0])
        elif world == 1:
            out["cpu_baseline"] = None
        out_fd.emit(json.dumps(out))
    if sharded:
        say("final barrier")
        sync.barrier()
        ctx.call("smn_comm_destroy")
    if wd is not None:
        wd.done()


if __name__ == "__main__":
    main()
