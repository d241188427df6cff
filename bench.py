#!/usr/bin/env python3
"""Benchmark of the hot path: fp64 LU (getrf) GFLOP/s + solve latency on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one LU factorisation (blocked, partial pivoting) of one resident
N x N matrix.  Every step factors its own pre-generated copy, so the timed
region holds no fill and no device-to-device copy; inputs are synthetic
(counter-based generator, lsx_fill_*_dev) and already in HBM.

N=1 workload: BASELINE config #3's factorisation, 8192 x 8192 fp64 -- the size
north_star quotes its targets on.  N>1: the same per-GPU flop count (weak
scaling): n = 8192 * N^(1/3) rounded to the panel width, the trailing update
sharded 1-D block-cyclic by columns with a panel broadcast per step
(linalg_solver_amd/dist.py); N=8 gives config #4's 16384 x 16384.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {"f64": 78.6, "f32": 157.3}  # dense MFMA TFLOP/s, MI355X (SURVEY section 6 / MICROARCH guide)
HBM_PEAK_GBS = 8000.0


def lu_flops(n: int) -> float:
    return 2.0 / 3.0 * n ** 3


def cpu_baseline(sample_n: int):
    """The oracle's C restatement of the reference's row_reduce (1 core: the reference is
    sequential by construction) on a bounded sample of the same generator, and LAPACK's
    dgetrf on all host cores of the box beside it (warmed up: the first call pays thread start-up)."""
    import numpy as np

    from linalg_solver_amd import gen
    from oracle import capi

    A, b = gen.system(gen.U11, 1, sample_n)
    aug = np.hstack([A, b[:, None]])
    t0 = time.perf_counter()
    red, piv, _ = capi.row_reduce(aug, want_steps=False)
    dt = time.perf_counter() - t0
    assert len(piv) == sample_n
    flops = float(sample_n) ** 3  # forward 2/3 n^3 + back-elimination 1/3 n^3 (linalg.py:587-621)
    out = {"value": flops / dt / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port",
           "sample": f"oracle/rowreduce_ref.c row_reduce([A|b]) at n={sample_n} u11 seed 1 "
                     f"({dt:.1f} s, n^3 flops); reference Python itself: 0.013 GFLOP/s (BASELINE.md)"}
    # LAPACK beside it, in a CHILD process: a crash inside a BLAS thread pool (seen under torch.distributed.run, which
    # exports OMP_NUM_THREADS=1 to its ranks) must not take the benchmark with it
    try:
        import subprocess

        env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS")}
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--lapack-child"], capture_output=True, text=True,
                           timeout=240, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        out["lapack_dgetrf"] = json.loads(line[-1]) if (r.returncode == 0 and line) else \
            {"error": f"child exit code {r.returncode}", "stderr_tail": r.stderr[-300:]}
    except Exception as e:
        out["lapack_dgetrf"] = {"error": str(e)}
    return out


def lapack_child():
    """`bench.py --lapack-child`: LAPACK's dgetrf through scipy on the host cores, one JSON line.  No GPU, no torch."""
    from linalg_solver_amd import gen

    out = {}
    try:
        import scipy.linalg as sl

        cores = len(os.sched_getaffinity(0))
        pools = None
        try:
            from threadpoolctl import threadpool_info

            pools = [{k: p.get(k) for k in ("internal_api", "num_threads", "version")} for p in threadpool_info()]
        except Exception:
            pass
        n2 = 4096
        A2 = gen.fill(gen.U11, 1, n2, n2)
        sl.lu_factor(A2[:1024, :1024].copy(), check_finite=False)   # warm-up: thread pool, code paths
        # ONE thread pool, sized explicitly: the box reports more hardware threads than this job's CPU share, and an
        # unbounded OpenBLAS pool beside an OpenMP pool oversubscribes (round 2's 30 GFLOP/s line).  The best of a few
        # pool sizes is reported with the size that gave it.
        results = []
        try:
            from threadpoolctl import threadpool_limits
        except Exception:
            threadpool_limits = None
        for nt in ([8, 16, 32, 64] if threadpool_limits else [0]):
            if nt > cores:
                continue
            best = 1e30
            ctx = threadpool_limits(limits=nt) if threadpool_limits else None
            try:
                for _ in range(2):
                    t0 = time.perf_counter()
                    sl.lu_factor(A2, check_finite=False)
                    best = min(best, time.perf_counter() - t0)
            finally:
                if ctx is not None:
                    ctx.restore_original_limits()
            results.append((lu_flops(n2) / best / 1e9, nt))
        gf, nt = max(results)
        out = {"value": gf, "unit": "GFLOP/s", "n": n2, "threads": nt or "default", "cores_visible": cores,
                                "by_threads": {str(t): round(g, 1) for g, t in results}, "threadpools": pools,
                                "note": "scipy.linalg.lu_factor, best of 2 per pool size after a warm-up call, every BLAS/OpenMP pool limited to `threads`"}
    except Exception as e:  # scipy is optional plumbing here
        out = {"error": str(e)}
    print(json.dumps(out))


def sharded_residual(torch, dist, slu, LUloc, A0loc, piv, rehearsal):
    """max |L U x - P A x| / (||A||_inf max|x|) of a column-sharded factorisation, x a fixed vector: every rank
    multiplies its own column blocks, three all-reduces of n-vectors put the pieces together (a wrong trailing
    update or a chunk-ordering bug shows here; info / |l| <= 1 / interchange range alone would pass it)."""
    n, nb = slu.n, slu.nb
    dev = LUloc.device
    x = torch.cos(torch.arange(n, dtype=torch.float64, device=dev) * 0.37) + 0.25
    rows = torch.arange(n, device=dev).unsqueeze(1)

    def allsum(v):
        if rehearsal:
            h = v.cpu(); dist.all_reduce(h); return h.to(dev)
        dist.all_reduce(v)
        return v

    y = torch.zeros(n, dtype=torch.float64, device=dev)
    ax = torch.zeros(n, dtype=torch.float64, device=dev)
    rowsum = torch.zeros(n, dtype=torch.float64, device=dev)
    for b in slu.my_blocks:
        o, w = slu.offset[b], slu.widths[b]
        cols = torch.arange(b * nb, b * nb + w, device=dev).unsqueeze(0)
        xs = x[b * nb:b * nb + w]
        y += (LUloc[:, o:o + w].double() * (rows <= cols)) @ xs
        ax += A0loc[:, o:o + w].double() @ xs
        rowsum += A0loc[:, o:o + w].double().abs().sum(dim=1)
    y, ax, rowsum = allsum(y), allsum(ax), allsum(rowsum)
    z = torch.zeros(n, dtype=torch.float64, device=dev)
    for b in slu.my_blocks:
        o, w = slu.offset[b], slu.widths[b]
        cols = torch.arange(b * nb, b * nb + w, device=dev).unsqueeze(0)
        z += (LUloc[:, o:o + w].double() * (rows > cols)) @ y[b * nb:b * nb + w]
    z = allsum(z) + y
    pax = ax.cpu().numpy().copy()
    pv = piv.cpu().numpy()
    for k in range(n):   # the interchanges in order (LAPACK convention)
        p_ = int(pv[k])
        if p_ != k:
            pax[k], pax[p_] = pax[p_], pax[k]
    err = float((z.cpu() - torch.from_numpy(pax)).abs().max())
    return err / (float(rowsum.max()) * float(x.abs().max()))


def bench_mg(args, torch):
    """--mg: ONE process, N handles on N devices, one C call per factorisation (lsx_getrf_mg_f64: the panel goes to
    every peer with hipMemcpyPeerAsync in row chunks).  LSX_BENCH_REHEARSAL=1: all handles on GPU 0."""
    import numpy as np

    from linalg_solver_amd import gen
    from linalg_solver_amd.device import DeviceSolver
    from linalg_solver_amd.dist import MultiDeviceLU

    P = args.gpus
    rehearsal = os.environ.get("LSX_BENCH_REHEARSAL") == "1"
    ndev = torch.cuda.device_count()
    if not rehearsal and ndev < P:
        raise SystemExit(f"--mg --gpus {P}: only {ndev} devices visible (LSX_BENCH_REHEARSAL=1 puts every handle on GPU 0)")
    devices = [0] * P if rehearsal else list(range(P))
    nb = args.nb or 128
    n = args.n or int(round(8192 * (P ** (1.0 / 3.0)) / 256.0)) * 256
    mg = MultiDeviceLU(n, devices, nb)
    fillers = {d: DeviceSolver(d) for d in sorted(set(devices))}
    total = args.steps + args.warmup

    def make(seed):
        locs = []
        for d in range(P):
            with torch.cuda.device(devices[d]):
                loc = torch.empty(n, mg.local_cols[d], dtype=torch.float64, device=f"cuda:{devices[d]}")
                off = 0
                for b in mg.blocks[d]:
                    w = min(nb, n - b * nb)
                    fillers[devices[d]].fill_(loc[:, off:off + w], gen.U11, seed, 0, b * nb)
                    off += w
            locs.append(loc)
        return locs

    mats = [make(1 + s) for s in range(total)]

    def sync():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    sync()
    res = None
    for i in range(args.warmup):
        res = mg.factor_(mats[i])
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = mg.factor_(mats[args.warmup + i])
    sync()
    dt = time.perf_counter() - t0
    ipiv, info = res
    # in-run check of the last factorisation: info, |l| <= 1, the pivot vectors of all devices agree
    LU = mats[args.warmup + args.steps - 1] if args.steps else mats[args.warmup - 1]
    worst, infos = 0.0, []
    for d in range(P):
        off = 0
        rows = torch.arange(n, device=LU[d].device).unsqueeze(1)
        for b in mg.blocks[d]:
            w = min(nb, n - b * nb)
            cols = torch.arange(b * nb, b * nb + w, device=LU[d].device).unsqueeze(0)
            worst = max(worst, float((LU[d][:, off:off + w].abs() * (rows > cols)).max()))
            off += w
        infos.append(int(info[d].item()))
    same_piv = all(bool(torch.equal(ipiv[0].cpu(), ipiv[d].cpu())) for d in range(1, P))
    ms = dt / max(args.steps, 1) * 1e3
    out = {"metric": "fp64_lu_gflops", "value": lu_flops(n) * args.steps / dt / 1e9, "unit": "GFLOP/s", "n_gpus": P,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"{n}x{n} f64 LU with partial pivoting, single-call multi-device driver (lsx_getrf_mg_f64)",
                      "n": n, "nb": nb, "parallelism": f"1-D block-cyclic columns x{P}, panel written to every peer "
                                                       f"(hipMemcpyPeerAsync, 4 row chunks), one process",
                      "devices": devices, "devices_visible": ndev, "rehearsal_all_handles_on_gpu0": rehearsal},
           "check": {"info": infos, "max_abs_multiplier": worst, "pivot_vectors_agree": same_piv,
                     "ok": bool(all(v == 0 for v in infos) and worst <= 1.0 + 1e-12 and same_piv)},
           "roofline": None, "cpu_baseline": None}
    print(json.dumps(out))
    return 0


def main():
    if "--lapack-child" in sys.argv:
        return lapack_child()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=0, help="matrix order (default: 8192 per-GPU-flop-equivalent)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--nb", type=int, default=0)
    ap.add_argument("--panel", type=int, default=-1)
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip solve-latency / 4096 side measurements")
    ap.add_argument("--mg", action="store_true",
                    help="ONE process driving --gpus N devices through lsx_getrf_mg_f64 (peer copies, no torch.distributed)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and not args.mg and world != args.gpus:
        if "WORLD_SIZE" in os.environ:
            raise SystemExit(f"--gpus {args.gpus} under a launcher with WORLD_SIZE={world}")
        # Started bare with --gpus N: become the launcher.  The ranks are CHILD processes started before this
        # process has made any GPU call (a process that has initialised the GPU must never exec another program);
        # rank 0's JSON line is relayed, the exit code is the launcher's.
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        line = None
        for ln in proc.stdout.splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                line = ln
            else:
                print(ln, file=sys.stderr)
        if line:
            print(line)
        raise SystemExit(proc.returncode if proc.returncode else (0 if line else 1))

    import torch

    if args.mg:
        return bench_mg(args, torch)
    # Rehearsal switch for a one-GPU box (not used by the driver): LSX_BENCH_REHEARSAL=1 runs every
    # rank on GPU 0 over gloo with the panel broadcast staged through the host.
    rehearsal = os.environ.get("LSX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    comm = None
    if dist is not None:
        comm = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                "devices_visible": torch.cuda.device_count(), "rehearsal_all_ranks_on_gpu0": rehearsal}

    from linalg_solver_amd import gen
    from linalg_solver_amd.device import DeviceSolver

    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    dev = DeviceSolver(local_rank)
    if args.nb:
        dev.h.set_option("nb", args.nb)
    if args.panel >= 0:
        dev.h.set_option("panel", args.panel)
    nb = dev.h.get_option("nb")
    n = args.n or int(round(8192 * (world ** (1.0 / 3.0)) / 256.0)) * 256

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    total = args.steps + args.warmup
    if world == 1:
        mats = []
        for s in range(total):
            A = torch.empty(n, n, dtype=tdt, device="cuda")
            dev.fill_(A, gen.U11, 1 + s)
            mats.append(A)
        ipiv = torch.empty(n, dtype=torch.int32, device="cuda")
        info = torch.zeros(1, dtype=torch.int32, device="cuda")

        def step(i):
            dev.getrf_(mats[i], ipiv, info)
    else:
        from linalg_solver_amd.dist import ShardedLU

        bcast = None
        if rehearsal:
            def bcast(t, src):
                hbuf = t.cpu()
                dist.broadcast(hbuf, src=src)
                t.copy_(hbuf)
        # the panel travels in 4 row chunks: receivers update each row range as it lands (dist.py)
        # two consecutive column blocks per owner: inside a pair the next panel starts without waiting for a broadcast
        # (the owner goes on while its panel is still on the wire), so only every second step has one on the chain
        slu = ShardedLU(dev, n, nb, rank, world, dtype=tdt, bcast=bcast, chunks=4, dist_block=2)
        shards = [slu.fill(gen.U11, 1 + s) for s in range(total)]

        last = {}

        def step(i):
            last["ipiv"], last["info"] = slu.factor_(shards[i])
            last["i"] = i

    for i in range(args.warmup):
        step(i)
    barrier()
    # live HIP events over the timed region, on the launch stream, for the TIME-DOMINANT kernel only: the panel
    # factorisation (one launch per 128 columns, on the look-ahead driver's side stream).  Bracketing every
    # launch of every phase costs ~7 % of the step; the trailing update is measured by an extra, untimed pass of
    # the same driver below (`roofline_update`), the full per-phase breakdown by a sequential step.
    dev.h.prof_reset()
    PANEL_SAMPLE = 13 if world == 1 else 1  # every 13th panel launch (13 and the 64 panels of a step are coprime:
    dev.h.set_option("prof_sample", PANEL_SAMPLE)   # over the steps every panel height is sampled): the events sit
    dev.h.prof_enable(buckets=("panel",))           # on the panel-to-panel chain and would cost 4 % if all were bracketed
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    dev.h.prof_enable(False)
    dev.h.set_option("prof_sample", 1)
    prof = dev.h.prof_read()
    # the trailing update under the SAME driver (look-ahead: it runs beside the next panel), untimed extra pass
    prof_upd = None
    if world == 1:
        dev.h.prof_reset()
        dev.fill_(mats[0], gen.U11, 1)
        barrier()
        dev.h.prof_enable(buckets=("gemm",))
        step(0)
        barrier()
        dev.h.prof_enable(False)
        prof_upd = dev.h.prof_read()["gemm"]
    # Per-phase breakdown: one more, untimed step with every phase bracketed by events.  On one GPU it
    # runs the SEQUENTIAL driver (lookahead=0): under look-ahead the panel and the update overlap and
    # their brackets no longer add up to the step.  The same step gives the dominant kernel's duration
    # with the chip to itself (`roofline_sequential`).
    dev.h.prof_reset()
    look_default = dev.h.get_option("lookahead")
    if world == 1:
        dev.fill_(mats[0], gen.U11, 1)
        dev.h.set_option("lookahead", 0)
    else:
        shards[0] = slu.fill(gen.U11, 1)
    barrier()
    dev.h.prof_enable(True)
    step(0)
    barrier()
    dev.h.prof_enable(False)
    dev.h.set_option("lookahead", look_default)
    phases = dev.h.prof_read()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if world == 1:
        assert int(info.item()) == 0, "benchmark matrix was singular"
    shard_check = None
    if world > 1:
        # in-run check of the last sharded factorisation (untimed): info word, interchanges in range, and the
        # partial-pivoting invariant |l_ij| <= 1 on this rank's columns, reduced over the ranks
        Ash, piv = shards[last["i"]], last["ipiv"]
        rows = torch.arange(n, device="cuda").unsqueeze(1)
        worst = torch.zeros(1, dtype=torch.float64, device="cuda")
        for b in slu.my_blocks:
            o, w = slu.offset[b], slu.widths[b]
            cols = torch.arange(b * nb, b * nb + w, device="cuda").unsqueeze(0)
            worst = torch.maximum(worst, (Ash[:, o:o + w].abs() * (rows > cols)).max().double().reshape(1))
        k = torch.arange(n, device="cuda", dtype=torch.int32)
        bad = ((piv < k) | (piv >= n)).sum().double().reshape(1)
        vec = torch.cat([worst, bad, last["info"].double().abs().reshape(1)])
        if rehearsal:
            hv = vec.cpu(); dist.all_reduce(hv, op=dist.ReduceOp.MAX); vec = hv
        else:
            dist.all_reduce(vec, op=dist.ReduceOp.MAX)
        A0 = slu.fill(gen.U11, 1 + last["i"])   # the same matrix again: the factorisation overwrote its shards
        resid = sharded_residual(torch, dist, slu, Ash, A0, piv, rehearsal)
        del A0
        shard_check = {"info": int(vec[2].item()), "max_abs_multiplier": float(vec[0].item()),
                        "interchanges_out_of_range": int(vec[1].item()),
                        "scaled_residual_LUx_minus_PAx": resid, "bound": 1e-11,
                        "ok": bool(vec[2].item() == 0 and vec[1].item() == 0 and vec[0].item() <= 1.0 + 1e-12
                                   and resid < 1e-11)}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # HBM traffic of the dominant kernel: PMC counters cannot be read in-process, so the ratio
    # (FETCH_SIZE*2 + WRITE_SIZE) / algorithmic bytes measured by the committed rocprofv3 --pmc pass
    # on the largest trailing-update launch is applied to this run's per-launch algorithmic bytes.
    pmc_ratio, pmc_src = None, None
    try:
        import glob
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_gemm.json")))[-1]
        pj = json.load(open(f))
        want_k = nb * dev.h.get_option("kblock")
        cand = [l for l in pj["launches"] if l["k"] == want_k] or pj["launches"]
        pmc_ratio, pmc_src = cand[0]["ratio"], os.path.relpath(f, ROOT) + f" (m=n={cand[0]['m']}, k={cand[0]['k']})"
    except Exception:
        pass
    mfma_pmc = None   # MFMA utilisation of the update from the committed counter passes (tools/pmc_mfma_summary.py)
    try:
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_gemm_mfma.json")))[-1]
        l0 = [l for l in json.load(open(f))["launches"] if l["k"] == nb * dev.h.get_option("kblock")][0]
        mfma_pmc = {"mfma_util": l0["mfma_util"], "shader_clock_ghz": l0["shader_clock_ghz_under_profiler"],
                    "source": os.path.relpath(f, ROOT) + f" (static grid alone, m=n={l0['m']}, k={l0['k']}; SQ_VALU_MFMA_BUSY_CYCLES / "
                                                          "(GRBM_GUI_ACTIVE per XCD x 1024 SIMDs))"}
    except Exception:
        pass
    sustained = None
    if world == 1 and not args.no_extras:
        sustained = dev.h.mfma_peak(args.dtype == "f32", 100000, 1)[0]

    ms_per_step = dt / args.steps * 1e3
    value = lu_flops(n) * args.steps / dt / 1e9
    pl = prof["panel"]                          # live, timed region
    g = prof_upd if prof_upd else prof["gemm"]
    gemm_tflops = (g["flops"] / (g["ms"] * 1e-3) / 1e12) if g["ms"] > 0 else 0.0
    p = phases["panel"]
    panel_gbs = (pl["bytes"] / (pl["ms"] * 1e-3) / 1e9) if pl["ms"] > 0 else 0.0
    panels_per_step = (n + nb - 1) // nb
    panel_ms_per_step = pl["ms"] / max(pl["launches"], 1) * panels_per_step   # sampled launches -> all of a step's
    pmc_panel = None
    try:
        import glob
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_panel.json")))[-1]
        pmc_panel = json.load(open(f))
        pmc_panel["source"] = os.path.relpath(f, ROOT)
    except Exception:
        pass
    hops = None
    if world == 1 and not args.no_extras:
        try:   # the exchange under the panel's column chain, measured in this run
            hops = {"xcd_scope_one_way_us": dev.h.xchg_probe(2, 32, 8, False, 2000)[0] / 2,
                    "device_scope_one_way_us": dev.h.xchg_probe(2, 32, 1, True, 2000)[0] / 2,
                    "xcd_scope_allgather_plus_row_us": dev.h.xchg_probe(1, 32, 8, False, 2000)[0]}
        except Exception as e:  # noqa: BLE001
            hops = {"error": str(e)}
    out = {
        "metric": "fp64_lu_gflops" if args.dtype == "f64" else "fp32_lu_gflops",
        "value": value, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{n}x{n} {args.dtype} LU with partial pivoting (getrf), u11 generator, resident in HBM",
                   "n": n, "nb": nb, "panel_mode": dev.h.get_option("panel"),
                   "lookahead": look_default if world == 1 else "depth-1, sharded driver",
                   "parallelism": "single GPU" if world == 1 else f"1-D block-cyclic columns x{world} (2 column blocks per owner), panel broadcast (RCCL) in 4 row chunks"},
        # the time-dominant kernel of the step: the panel factorisation.  SURVEY 8d prices it against HBM
        # (algorithmic bytes = each panel read + written once = 2 * sizeof(T) * m * nb per launch); what actually bounds
        # it is one cross-CU pivot exchange per column -- latency, stated beside the fraction.
        "roofline": {"bound": "hbm", "limiter": "latency (one cross-CU pivot exchange per column), not bandwidth",
                     "kernel": "panel_x_kernel (panel factorisation, XCD-scope pivot exchange; panels taller than one "
                               "XCD holds: panel_pipe_kernel)" if dev.h.get_option("panel") == 4 else "panel_pipe_kernel",
                     "achieved": panel_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": panel_gbs / HBM_PEAK_GBS,
                     "traffic": (pmc_panel or {}).get("hbm_bytes_per_launch"),
                     "traffic_note": (f"PMC pass {pmc_panel['source']}: {pmc_panel.get('note', '')}" if pmc_panel
                                      else "no PMC pass available"),
                     "launches": pl["launches"], "avg_launch_ms": pl["ms"] / max(pl["launches"], 1),
                     "algorithmic_bytes_per_launch": pl["bytes"] / max(pl["launches"], 1),
                     "algorithmic_bytes": pl["bytes"],
                     "share_of_step": panel_ms_per_step / ms_per_step if ms_per_step > 0 else None,
                     "us_per_column": (pl["ms"] * 1e3 / max(pl["launches"], 1) / nb) if world == 1 else None,
                     "sampling": f"every {PANEL_SAMPLE}th launch bracketed by HIP events on its launch stream",
                     "latency_bound": {"note": "true partial pivoting needs one cross-CU exchange per column; the kernel's "
                                               "floor is columns x (one-way hop + the owner wave's instruction stream), "
                                               "not bytes / bandwidth", "hops": hops},
                     "whole_lu_frac_of_mfma_peak": value / 1e3 / PEAK[args.dtype]},
        "roofline_update": {"bound": "mfma", "kernel": "gemm_sub_queue_kernel / gemm_sub_kernel<T,.,true,128> (trailing update "
                                                       "C -= L21*U12, 128x128 tiles)",
                            "achieved": gemm_tflops, "peak": PEAK[args.dtype], "unit": "TFLOP/s",
                            "frac": gemm_tflops / PEAK[args.dtype],
                            "cus": "7 of 8 XCDs while a panel runs on the eighth (look-ahead driver)" if dev.h.get_option("panel") == 4 else "all",
                            "frac_of_the_cus_it_runs_on": gemm_tflops / (PEAK[args.dtype] * 7 / 8) if dev.h.get_option("panel") == 4 else None,
                            "traffic": (g["bytes"] / max(g["launches"], 1) * pmc_ratio) if pmc_ratio else None,
                            "traffic_note": (f"avg HBM bytes per launch = algorithmic x {pmc_ratio:.3f}, ratio from PMC pass "
                                             f"{pmc_src}") if pmc_ratio else "no PMC pass available",
                            "algorithmic_bytes_per_launch": g["bytes"] / max(g["launches"], 1),
                            "mfma_sustained_tflops_microbench": sustained,
                            "mfma_util_pmc": mfma_pmc,
                            "launches": g["launches"], "avg_launch_ms": g["ms"] / max(g["launches"], 1),
                            "note": "extra untimed pass of the same look-ahead driver with only these launches bracketed"},
        "phases_ms_per_step": {k: v["ms"] for k, v in phases.items()},
        "phases_note": "one extra untimed step of the sequential driver (lookahead=0) with every phase bracketed by events",
    }

    if g["launches"] == 0:   # sharded run: the update is not bracketed (its per-rank share is in phases_ms_per_step)
        out["roofline_update"] = None
    if world > 1:
        out["check"] = shard_check
        out["config"]["comm"] = comm
    gs = phases["gemm"]
    if world == 1 and gs["ms"] > 0:
        seq_tf = gs["flops"] / (gs["ms"] * 1e-3) / 1e12
        out["roofline_sequential"] = {"achieved": seq_tf, "peak": PEAK[args.dtype], "unit": "TFLOP/s",
                                      "frac": seq_tf / PEAK[args.dtype], "launches": gs["launches"],
                                      "avg_launch_ms": gs["ms"] / max(gs["launches"], 1),
                                      "note": "the update kernel with the chip to itself (lookahead=0 step)"}
    if world == 1 and not args.no_extras and n >= 3072:
        # the sequential driver end to end, for comparison (bit-identical factors)
        dev.h.set_option("lookahead", 0)
        ts = []
        for r in range(3):
            dev.fill_(mats[0], gen.U11, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.getrf_(mats[0], ipiv, info)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        dev.h.set_option("lookahead", look_default)
        out["sequential"] = {"ms_per_step": min(ts[1:]) * 1e3, "gflops": lu_flops(n) / min(ts[1:]) / 1e9,
                             "note": "lookahead=0, untimed extra"}
    if world == 1 and not args.no_extras:
        # config #2: 4096 x 4096 LU + single right-hand-side solve latency
        n2 = 4096
        A2 = torch.empty(n2, n2, dtype=tdt, device="cuda")
        ip2 = torch.empty(n2, dtype=torch.int32, device="cuda")
        b2 = torch.empty(n2, 1, dtype=tdt, device="cuda")
        reps = 3
        ts = []
        for r in range(reps + 1):
            dev.fill_(A2, gen.U11, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.getrf_(A2, ip2, info)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        lu4096 = min(ts[1:])
        ts = []
        for r in range(reps + 1):
            dev.fill_(b2, gen.U11, 1, col_off=gen.RHS_COL)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.getrs_(A2, ip2, b2)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out["config2_4096"] = {"lu_ms": lu4096 * 1e3, "lu_gflops": lu_flops(n2) / lu4096 / 1e9,
                               "solve_1rhs_latency_ms": min(ts[1:]) * 1e3}
        # parity check inside the run: factor a fresh matrix of the benchmark shape, solve one
        # right-hand side, report info and the scaled residual against the unfactored copy
        A_chk = torch.empty(n, n, dtype=tdt, device="cuda")
        dev.fill_(A_chk, gen.U11, 12345)
        LU_chk = A_chk.clone()
        ip_chk = torch.empty(n, dtype=torch.int32, device="cuda")
        info_chk = torch.zeros(1, dtype=torch.int32, device="cuda")
        dev.getrf_(LU_chk, ip_chk, info_chk)
        b_chk = torch.empty(n, 1, dtype=tdt, device="cuda")
        dev.fill_(b_chk, gen.U11, 12345, col_off=gen.RHS_COL)
        x_chk = b_chk.clone()
        dev.getrs_(LU_chk, ip_chk, x_chk)
        r = (A_chk.double() @ x_chk.double() - b_chk.double()).abs().max().item()
        scale = (A_chk.double().abs().sum(dim=1).max().item() * x_chk.double().abs().max().item()
                 + b_chk.double().abs().max().item())
        bound = 1e-9 if args.dtype == "f64" else 1e-4
        out["check"] = {"info": int(info_chk.item()), "solve_scaled_residual": r / scale, "bound": bound,
                        "ok": bool(int(info_chk.item()) == 0 and r / scale < bound)}
        if args.dtype == "f64":
            # config #3: inverse from the factors just checked
            LU = LU_chk
            inv = torch.empty(n, n, dtype=tdt, device="cuda")
            dev.getri(LU, ip_chk, inv)   # untimed: the first call allocates the n x n workspace
            t_inv = float("inf")
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dev.getri(LU, ip_chk, inv)
                torch.cuda.synchronize()
                t_inv = min(t_inv, time.perf_counter() - t0)
            ident_err = (A_chk @ inv - torch.eye(n, dtype=tdt, device="cuda")).abs().max().item()
            out["config3_inverse"] = {"getri_ms": t_inv * 1e3, "lu_plus_inverse_ms": t_inv * 1e3 + ms_per_step,
                                      "gflops_2n3": 2.0 * n ** 3 / (t_inv + ms_per_step * 1e-3) / 1e9,
                                      "max_abs_A_inv_minus_I": ident_err,
                                      "note": "inverse from the factors (lsx_getri_f64_dev), best of 2 after one untimed call"}
    if world == 1 and not args.no_extras and args.dtype == "f64":
        # the same factorisation through the HOST-buffer entry point (lsx_getrf_f64: H2D, LU, D2H of a
        # 512 MiB matrix in pageable memory) -- never the headline value, reported beside it
        import ctypes as _C

        import numpy as _np

        from linalg_solver_amd import dense as _dense
        dev.fill_(mats[0], gen.U11, 1)
        Ah = mats[0].cpu().numpy()
        hh = _dense._h(None)
        ts, inf = [], _C.c_int(0)
        for _ in range(2):
            LUh = Ah.copy()
            pvh = _np.zeros(n, dtype=_np.int32)
            t0 = time.perf_counter()
            rc_ = hh.lib.lsx_getrf_f64(hh.ptr, n, _dense._ptr(LUh, _C.c_double), n, _dense._ptr(pvh, _C.c_int32), _C.byref(inf))
            ts.append(time.perf_counter() - t0)
            assert rc_ == 0
        out["pcie_inclusive"] = {"n": n, "ms": min(ts) * 1e3, "gflops": lu_flops(n) / min(ts) / 1e9, "info": int(inf.value),
                                 "note": "lsx_getrf_f64 on host buffers: upload + factorisation + download of a 512 MiB matrix, "
                                         "pageable host memory (the C call only)"}
        del Ah, LUh
    if world == 1 and not args.no_extras and args.dtype == "f64":
        # config #5: 8192 x 8192 in fp32 -- factorisation time, and the SOLUTION against the fp64 result
        # ("tolerance 1e-4"): fp32 factors + fp64 residuals (lsx_gesv_f32_refined_dev), 4 right-hand sides
        n5 = 8192
        A32 = torch.empty(n5, n5, dtype=torch.float32, device="cuda")
        B32 = torch.empty(n5, 4, dtype=torch.float32, device="cuda")
        dev.fill_(A32, gen.U11, 5)
        dev.fill_(B32, gen.U11, 6)
        LU5 = A32.clone()
        ip5 = torch.empty(n5, dtype=torch.int32, device="cuda")
        ts = []
        for r in range(3):
            LU5.copy_(A32)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.getrf_(LU5, ip5, info)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        X64, X32, _, _, info5, st5 = dev.gesv_refined(A32, B32, sweeps=3)
        torch.cuda.synchronize()
        t_ref = time.perf_counter() - t0
        A64 = A32.double()
        Xd = B32.double().clone()
        ip64, info64 = dev.getrf_(A64)
        dev.getrs_(A64, ip64, Xd)
        X0 = B32.clone()
        dev.getrs_(LU5, ip5, X0)
        torch.cuda.synchronize()
        sc = float(Xd.abs().max())
        out["config5_fp32"] = {"lu_ms": min(ts[1:]) * 1e3, "lu_gflops": lu_flops(n5) / min(ts[1:]) / 1e9,
                               "refined_solve_4rhs_ms_incl_lu": t_ref * 1e3, "sweeps": 3,
                               "forward_error_vs_fp64_refined": float((X32.double() - Xd).abs().max()) / sc,
                               "forward_error_vs_fp64_unrefined": float((X0.double() - Xd).abs().max()) / sc,
                               "tolerance": 1e-4, "info": int(info5.item())}
        del A64, Xd, X0, LU5, A32
    if world == 1 and not args.no_extras and args.dtype == "f64":
        # SURVEY 8f item 1 at scale: rank-revealing row reduction of an 8192 x 8192 matrix of rank 4096 (blocked
        # column-skip elimination + blocked back substitution); pivots against the planted structure
        from linalg_solver_amd import _native as _N
        nr, rk = 8192, 4096
        Bm = torch.empty(nr, rk, dtype=torch.float64, device="cuda")
        Cm = torch.empty(rk, nr - rk, dtype=torch.float64, device="cuda")
        dev.fill_(Bm, gen.U11, 3)
        dev.fill_(Cm, gen.U11, 4)
        Ar = torch.cat([Bm, Bm @ Cm / 64.0], dim=1).contiguous()
        ts = []
        for _ in range(2):
            Rr = Ar.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pv, rkk = dev.rref_(Rr, bar_col=nr, pivot_rule=_N.PIVOT_MAX)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        pvh = pv[:2 * rk].view(rk, 2).cpu()
        out["rref_8192_rank4096_ms"] = min(ts) * 1e3
        out["rref_8192_rank4096"] = {"rank": int(rkk.item()),
                                      "pivots_match_planted": bool(int(rkk.item()) == rk and bool((pvh[:, 1] == torch.arange(rk)).all())),
                                      "max_abs_err_of_reduced_block": float((Rr[:rk, rk:] - Cm / 64.0).abs().max()),
                                      "note": "device tensors, max-|a| rule (rank and pivot columns only)"}
        # the PRODUCT surface on the same matrix with bar_col < n, i.e. with carried-along columns that follow the
        # reference's first-non-zero rule: Matrix.row_reduce_array on host arrays (upload, rank-revealing pass, blocked
        # LU of the pivot columns under the first-non-zero rule, solve + MFMA update, download)
        import linalg_solver_amd as _la
        Ah_r = Ar.cpu().numpy()
        bar_r = 6000
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            Rm, pm = _la.Matrix.from_numpy(Ah_r).row_reduce_array(bar_col=bar_r)
            ts.append(time.perf_counter() - t0)
        out["matrix_row_reduce_8192_rank4096"] = {
            "s": min(ts), "bar_col": bar_r, "pivots_match_planted": bool(pm == [(k, k) for k in range(rk)]),
            "blocked_first_rule_used": bool(_la.default_handle().get_option("rref_first_used")),
            "max_abs_err_of_reduced_block": float(abs(Rm[:rk, rk:] - (Cm / 64.0).cpu().numpy()).max()),
            "note": "Matrix.from_numpy(A).row_reduce_array(bar_col=6000), host arrays in and out (2 x 512 MiB over PCIe included)"}
        del Bm, Cm, Ar, Rr, Ah_r, Rm
    if world == 1 and not args.no_extras and args.dtype == "f64":
        # config #1 (the reference's own CPU-runnable case): 64 x 64 ints in [-5,5] as floats + rhs through
        # the Matrix surface -- fast path, traced path (reference-order arithmetic + step list) and traced
        # path with every intermediate LaTeX matrix, which is what the reference spends 12.3 s on
        import random as _random
        import linalg_solver_amd as la
        _random.seed(2026)
        A1 = [[float(_random.randint(-5, 5)) for _ in range(64)] for _ in range(64)]
        b1 = [float(_random.randint(-5, 5)) for _ in range(64)]
        aug1 = la.Matrix([r + [v] for r, v in zip(A1, b1)])
        res = {}
        for key, kw in (("fast_ms", {}), ("traced_steps_ms", {"trace": "steps"}), ("traced_full_latex_ms", {"trace": True})):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                r1 = aug1.row_reduce(**kw)
                best = min(best, time.perf_counter() - t0)
            res[key] = best * 1e3
            res[key.replace("_ms", "_steps")] = len(r1[3])
        res["reference_python_s"] = 12.33
        res["note"] = "reference figure measured in the build container (BASELINE.md); traced results are bit-identical to it"
        out["config1_64"] = res
    if not args.no_cpu and world == 1:   # (the contract: on rank 0 at N = 1 only)
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
