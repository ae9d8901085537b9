#!/usr/bin/env python3
"""bench.py — the hot path of BASELINE.json on MI355X: NNGP kernel build + jittered Cholesky (+ LML heads).

One "step" = one SPR.loss evaluation (spax/models.py:93-98) on synthetic inputs already resident in HBM:
fused Gram + 4-layer ReLU recursion -> K + eps I -> blocked Cholesky with y carried -> log-marginal
likelihood.  Workload at N=1: BASELINE.json configs[3] shape on one GPU (N=16384, d=3072, L=4, fp32).
With --gpus P > 1 (launched by torch.distributed.run, one rank per GPU): the kernel build is row-sharded
over the ranks (paired lower-block layout, sharding.py), assembled with ONE RCCL all-gather, and every rank
factors the assembled kernel
(strong scaling: total work fixed).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F64_MFMA_TFLOPS = 78.6    # datasheet (not in the local guide)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s achievable)
TILE = 128


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--n", type=int, default=16384)
    p.add_argument("--d", type=int, default=3072)
    p.add_argument("--layers", type=int, default=4)
    p.add_argument("--act", default="relu")
    p.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    p.add_argument("--eps", type=float, default=None)
    p.add_argument("--cpu-sample-n", type=int, default=0, help="0: the benched N")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-recursion-probe", action="store_true")
    p.add_argument("--no-exclusive-probe", action="store_true", help="skip the look-ahead-off pass that fills frac_exclusive")
    p.add_argument("--sharded-path", action="store_true",
                   help="run the N>1 step (row shard + all-gather + LML) even with one rank (rehearsal on one GPU)")
    return p.parse_args()


def pmc_traffic(args, sharded):
    """Measured HBM-side bytes per launch of the dominant kernel (persistent launches only), from the committed
    rocprofv3 PMC pass; only valid for the default workload it was taken on."""
    if sharded or (args.n, args.d, args.layers, args.act, args.dtype) != (16384, 3072, 4, "relu", "f32"):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r01f_pmc_traffic.json")) as f:
            return json.load(f)["traffic_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(args, np_dtype, eps, gpu_logpdf=None):
    """The CPU oracle (NumPy/SciPy port of the same math) on the benched workload, or a bounded sample of it
    (--cpu-sample-n), on this box's host cores.  Also the checker of the headline number: the oracle's log-pdf and
    its relative difference to the GPU's (same inputs, same dtype) go into the line when N is the benched N."""
    from oracle import host_parallel as HP       # test infrastructure: imported by this leg only
    ns = min(args.cpu_sample_n or args.n, args.n)
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


def main():
    args = parse()
    out_fd = _JsonOut()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    multi = world > 1 or (args.sharded_path and "RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if multi:
        # torch (host-side gloo rendezvous only) goes in FIRST: the wheel carries its own ROCm runtime and RCCL, and
        # loaded in this order libsmnngp.so binds to that same runtime, so the process holds one HIP and one RCCL
        import torch                               # noqa: F401
        import torch.distributed                   # noqa: F401
    from smnngp import _lib as L

    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    code = L.dtype_code(np_dtype)
    eps = args.eps if args.eps is not None else (1e-3 if args.dtype == "f32" else 1e-6)
    act = L.ACT[args.act]
    n, d, nl = args.n, args.d, args.layers
    ctx = L.Context(local_rank)

    dist = None
    # `--sharded-path` under torch.distributed.run with ONE rank walks exactly the N>1 code (torch import, gloo
    # rendezvous, broadcast of the RCCL id, communicator, all-gather) on a one-GPU box
    if multi:
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("gloo")            # host-side rendezvous only; the data path is RCCL below
        uid = C.create_string_buffer(128)
        if rank == 0:
            assert L._lib.smn_comm_unique_id(uid) == 0, "RCCL unavailable"
        t = torch.tensor(list(uid.raw), dtype=torch.uint8)
        dist.broadcast(t, 0)
        uid = C.create_string_buffer(bytes(t.tolist()), 128)
        ctx.call("smn_comm_init", world, rank, uid)
    elif args.sharded_path:                        # plain `python bench.py --sharded-path`: one-rank communicator, no torch
        uid = C.create_string_buffer(128)
        assert L._lib.smn_comm_unique_id(uid) == 0, "RCCL unavailable"
        ctx.call("smn_comm_init", 1, 0, uid)

    rng = np.random.default_rng(0)                 # same seed on every rank: X is replicated
    x = ctx.to_device(rng.standard_normal((n, d)).astype(np_dtype))
    y = ctx.to_device(rng.standard_normal(n).astype(np_dtype))
    lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()

    sharded = world > 1 or args.sharded_path
    if not sharded:
        def step():
            ctx.call("smn_spr_loss", code, L.NET_MLP, act, nl, 1.0, 1e-8, 1.0, x.ptr, n, d, d, y.ptr, eps, 0.0, 1.0,
                     C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
    else:
        # Balanced symmetric shard (sharding.py, paired layout): rank r builds the lower trapezoids of row blocks
        # r and 2P-1-r packed into its chunk of `stage`, ONE in-place RCCL all-gather moves N^2/2-ish elements in
        # total, smn_lml_from_blocks scatters them straight into the factorisation workspace and factors (replicated).
        from smnngp import sharding
        es = np.dtype(np_dtype).itemsize
        stage = ctx.empty((world * sharding.paired_chunk_elems(n, world),), np_dtype)
        hblk = sharding.block_rows(n, world)

        def step():
            sharding.build_lower_sharded(ctx, code, es, L.NET_MLP, act, nl, 1.0, 1e-8, 1.0, x.ptr, n, d, d,
                                         rank, world, stage.ptr, None, 0)
            ctx.call("smn_lml_from_blocks", code, stage.ptr, n, world, hblk, y.ptr, eps, 0.0, 1.0, C.byref(lp),
                     C.byref(quad), C.byref(logdet), C.byref(info))

    def barrier():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    # Timed region: hipEvent pairs around the launches of the DOMINANT kernel only (category 5, the Cholesky trailing
    # update: 79 launches per step).  Pairs around all ~280 launches of a step cost ~2 ms of queue time per step
    # (profiles/r01e_event_overhead.txt), so the other categories are timed in a separate, untimed pass below.
    CATS = ["prep", "build", "recursion", "panel", "strip", "trail", "misc"]
    ctx.call("smn_profile_enable", 2 << 5)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
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
    barrier()
    prof = {}
    for cat, name in enumerate(CATS):
        ms, cnt = C.c_double(), C.c_int()
        ctx.call("smn_profile_read", cat, C.byref(ms), C.byref(cnt))
        prof[name] = (ms.value / DETAIL_STEPS, cnt.value // DETAIL_STEPS)
    ctx.call("smn_profile_enable", 0)
    prof["trail"] = trail_timed
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3

    if rank == 0:
        n_total = n + TILE if world == 1 else n + TILE
        flops_counted = 2.0 * n * n * d + n ** 3 / 3.0                 # Gram 2N^2 d + Cholesky N^3/3
        trail_fl, strip_fl = fl[5], fl[4]
        peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_F64_MFMA_TFLOPS
        per = prof                                 # per step: `trail` from the timed region, the rest from the detail pass
        kp = ((d + 31) // 32 * 32) if args.dtype == "f32" else ((d + 15) // 16 * 16)
        if not sharded:
            t = n_total // TILE
            build_fl = (t * (t + 1) // 2) * TILE * TILE * 2.0 * kp
        else:
            from smnngp import sharding as S_
            tiles = 0                                                     # lower tiles rank 0's two blocks execute
            for b in S_.paired_blocks(world, 0):
                rb_, re_ = S_.block_range(n, world, b)
                tiles += sum(t + 1 for t in range(rb_ // TILE, -(-re_ // TILE)))
            build_fl = tiles * TILE * TILE * 2.0 * kp
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
            # HBM-side bytes per launch cannot be read without rocprofv3: taken from the committed PMC pass of this
            # exact workload (profiles/r01f_pmc_traffic.json: FETCH_SIZE x2 + WRITE_SIZE, separate passes), else null
            "traffic": pmc_traffic(args, sharded),
            "launches_per_step": per["trail"][1], "avg_launch_ms": trail_ms / max(per["trail"][1], 1),
            "flops_per_step": trail_fl,
        }
        # Look-ahead (default from N=8192): the far updates run on a CU-masked stream (num_cu - SMN_CHAIN_CUS CUs)
        # BESIDE the next super-panel's panel chain, so the launch durations above overlap with other kernels and
        # `frac` (kept as the contract defines it) understates the kernel.  Two more readings of the same kernel:
        min_n_env = os.environ.get("SMN_CHAIN_MIN_N")
        lookahead = n_total >= int(min_n_env or "8192")
        chol_wall_ms = ms_per_step - per["build"][0] - per["prep"][0] - per["misc"][0]
        if not sharded:
            roof["cholesky_wall_ms"] = chol_wall_ms
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
        out = {
            "metric": "kernel-build + Cholesky wallclock (ms) and GFLOP/s at N=%d, %d-layer %s NNGP" % (n, nl, args.act),
            "value": flops_counted / (ms_per_step * 1e-3) / 1e9, "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "SPR.loss: NNGP kernel build + jittered Cholesky + Gaussian LML, N=%d d=%d L=%d %s"
                                   % (n, d, nl, args.act),
                       "N": n, "d": d, "layers": nl, "act": args.act, "w_std": 1.0, "b_std": 1e-8, "last_w_std": 1.0,
                       "eps_abs": eps, "flops_counted": flops_counted,
                       "parallelism": "single GPU" if not sharded else "paired lower-block row shards x%d + one RCCL all-gather + replicated Cholesky" % world},
            "phases_ms": {k: round(v[0], 4) for k, v in per.items()},
            "phases_ms_source": "trail: hipEvents in the timed region; others: separate untimed pass with events on every launch",
            # the north-star's "kernel-build speed-up at N GPUs" reads off these two (rank 0; every rank builds the same
            # number of tiles): the fused Gram + recursion launch of this rank's shard, and all-gather + scatter
            "kernel_build_ms": round(per["build"][0], 4), "exchange_ms": round(per["misc"][0], 4) if sharded else 0.0,
            "result": {"logpdf": lp.value, "logdet": logdet.value, "info": info.value},
            "roofline": roof,
        }
        # stand-alone recursion (a3): HBM roofline probe on a stored K0, outside the timed region
        if world == 1 and not args.no_recursion_probe:
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
                        with open(os.path.join(ROOT, "profiles", "r01e_pmc_recursion.json")) as f:
                            rec_traffic = json.load(f)["traffic_bytes_per_launch"]
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
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, np_dtype, eps, lp.value)
        elif world == 1:
            out["cpu_baseline"] = None
        out_fd.emit(json.dumps(out))
    if dist is not None:
        dist.barrier()
        ctx.call("smn_comm_destroy")
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
