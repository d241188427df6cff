#!/usr/bin/env python3
"""Kernel-level micro-benchmarks on one MI355X (development tool, not the contract bench).

    python tools/kbench.py mfma | gemm | panel | lu [--n N] [--nb NB]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from linalg_solver_amd import gen  # noqa: E402
from linalg_solver_amd.device import DeviceSolver  # noqa: E402


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="+")
    ap.add_argument("--f32", action="store_true", help="lu: factor in fp32")
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--nb", type=int, default=128)
    args = ap.parse_args()
    dev = DeviceSolver()
    if "mfma" in args.what:
        for bpc in (1, 2):
            t64, mhz = dev.h.mfma_peak(False, 200000, bpc)
            t32, _ = dev.h.mfma_peak(True, 200000, bpc)
            cyc = 2048.0 * 4 * 256 * bpc * 0 + (mhz * 1e6) / (t64 * 1e12 / (2048.0 * 1024 * (bpc if bpc < 2 else 1) )) if mhz else 0
            print(f"mfma f64 ({bpc} WG/CU = {bpc} wave/SIMD): {t64:.1f} TFLOP/s at {mhz:.0f} MHz in-kernel clock "
                  f"-> {t64 * 1e12 / 1024 / (mhz * 1e6) if mhz else 0:.1f} flop/clk/SIMD "
                  f"(v_mfma_f64_16x16x4 = 2048 flop);   f32: {t32:.1f} TFLOP/s", flush=True)
    if "gemm" in args.what:
        for dt, gw in ((torch.float64, 4), (torch.float64, 8), (torch.float32, 4), (torch.float32, 8)):
            dev.h.set_option("gemm_waves", gw)
            for m, k in ((8064, 128), (4096, 128), (2048, 128), (8064, 256), (4096, 256)):
                A = torch.randn(m, k, dtype=dt, device="cuda")
                B = torch.randn(k, m, dtype=dt, device="cuda")
                Cm = torch.randn(m, m, dtype=dt, device="cuda")
                tmin, tmed = timeit(lambda: dev.gemm_sub_(Cm, A, B))
                fl = 2.0 * m * m * k
                by = 2.0 * Cm.element_size() * m * m
                print(f"gemm_sub {str(dt)[6:]} waves={gw} m=n={m} k={k}: {tmin:.3f} ms  {fl / tmin / 1e9:.1f} TFLOP/s  "
                      f"C traffic {by / tmin / 1e6:.0f} GB/s", flush=True)
    if "gemmq" in args.what:
        # the work-queue form of the update alone: all XCDs / XCD 0 left out, against the plain grid
        dt = torch.float64
        dev.h.set_option("gemm_waves", 0)
        for m, k in ((8064, 128), (6016, 128), (4096, 128), (2048, 128)):
            A = torch.randn(m, k, dtype=dt, device="cuda")
            B = torch.randn(k, m, dtype=dt, device="cuda")
            Cm = torch.randn(m, m, dtype=dt, device="cuda")
            for q in (0, 1, 2):
                dev.h.set_option("gemm_queue_test", q)
                tmin, tmed = timeit(lambda: dev.gemm_sub_(Cm, A, B), reps=7, warm=2)
                fl = 2.0 * m * m * k
                print(f"gemm_sub f64 m=n={m} k={k} queue={q}: min {tmin * 1e3:.1f} us  med {tmed * 1e3:.1f} us  {fl / tmin / 1e9:.1f} TFLOP/s", flush=True)
            dev.h.set_option("gemm_queue_test", 0)
    if "panel3tall" in args.what:
        # the device-scope panel on panels taller than one XCD holds (hybrid driver's first phase, multi-GPU): slice
        # heights (threads per workgroup x rows per thread tile)
        dev.h.set_option("panel", 3)
        for m in (16384, 12288, 9216):
            for nt, rt in ((0, 4), (256, 4), (512, 4), (512, 8), (256, 2), (512, 2)):
                try:
                    dev.h.set_option("panel_nt", nt)
                    dev.h.set_option("panel_rt", rt)
                except Exception as e:
                    print(f"panel3 m={m} nt={nt} rt={rt}: option refused ({e})", flush=True)
                    continue
                P0 = torch.empty(m, args.nb, dtype=torch.float64, device="cuda")
                dev.fill_(P0, gen.U11, 3)
                ipiv = torch.zeros(args.nb, dtype=torch.int32, device="cuda")
                info = torch.zeros(1, dtype=torch.int32, device="cuda")
                P = P0.clone()

                def run():
                    P.copy_(P0)
                    dev.panel_(P, 0, ipiv, info)
                try:
                    tmin, tmed = timeit(run, reps=5, warm=2)
                    tcopy, _ = timeit(lambda: P.copy_(P0), reps=5, warm=2)
                    t = tmin - tcopy
                    print(f"panel3 m={m} nt={nt} rt={rt}: {t * 1e3:.1f} us  ({t * 1e3 / args.nb:.2f} us/col) info={int(info.item())} fallbacks(diag)={dev.h.get_option('diag_panels')}", flush=True)
                except Exception as e:
                    print(f"panel3 m={m} nt={nt} rt={rt}: failed ({e})", flush=True)
        dev.h.set_option("panel_nt", 0); dev.h.set_option("panel_rt", 4); dev.h.set_option("panel", 4)
    if "panel" in args.what:
        for mode, nt, rt in ((1, 256, 4), (1, 512, 4)):
            dev.h.set_option("panel", mode)
            dev.h.set_option("panel_rt", rt)
            dev.h.set_option("panel_nt", nt)
            for m in (args.n, args.n // 2, args.n // 8, 256):
                P0 = torch.empty(m, args.nb, dtype=torch.float64, device="cuda")
                dev.fill_(P0, gen.U11, 3)
                ipiv = torch.zeros(args.nb, dtype=torch.int32, device="cuda")
                info = torch.zeros(1, dtype=torch.int32, device="cuda")
                P = P0.clone()

                def run():
                    P.copy_(P0)
                    dev.panel_(P, 0, ipiv, info)
                tmin, tmed = timeit(run, reps=5, warm=1)
                tcopy, _ = timeit(lambda: P.copy_(P0), reps=5, warm=1)
                t = tmin - tcopy
                print(f"panel mode={mode} nt={nt} rt={rt} m={m} nb={args.nb}: {t * 1e3:.1f} us  ({t * 1e3 / args.nb:.2f} us/col)  "
                      f"{2 * 8 * m * args.nb / t / 1e6:.1f} GB/s", flush=True)
    if "panel3" in args.what:
        import numpy as np
        # pipelined panel (mode 3): bitwise check against mode 1, timing, then the stamped build
        dev.h.set_option("panel_rt", 4)
        for m in (args.n, args.n // 2, 1024, 384, 128, 100):
            for nbw in (args.nb, 100):
                P0 = torch.empty(m, nbw, dtype=torch.float64, device="cuda")
                dev.fill_(P0, gen.U11, 3)
                outs = []
                for mode, nt in ((4, 0), (3, 0), (3, 512)):
                    dev.h.set_option("panel", mode)
                    dev.h.set_option("panel_nt", nt)
                    P = P0.clone()
                    ipiv = torch.zeros(nbw, dtype=torch.int32, device="cuda")
                    info = torch.zeros(1, dtype=torch.int32, device="cuda")
                    dev.panel_(P, 0, ipiv, info)
                    torch.cuda.synchronize()
                    outs.append((P, ipiv.clone(), int(info.item())))
                for k in (1, 2):
                    same = torch.equal(outs[0][0], outs[k][0]) and torch.equal(outs[0][1], outs[k][1]) \
                        and outs[0][2] == outs[k][2]
                    print(f"panel3 check m={m} jb={nbw} variant={k}: {'bit-identical' if same else 'MISMATCH'} "
                          f"info={outs[k][2]}", flush=True)
        for mode, nt in ((4, 0), (3, 512), (3, 256)):
            dev.h.set_option("panel", mode)
            dev.h.set_option("panel_nt", nt)
            for m in (args.n, args.n // 2, args.n // 8, 256):
                P0 = torch.empty(m, args.nb, dtype=torch.float64, device="cuda")
                dev.fill_(P0, gen.U11, 3)
                ipiv = torch.zeros(args.nb, dtype=torch.int32, device="cuda")
                info = torch.zeros(1, dtype=torch.int32, device="cuda")
                P = P0.clone()

                def run():
                    P.copy_(P0)
                    dev.panel_(P, 0, ipiv, info)
                tmin, tmed = timeit(run, reps=5, warm=1)
                tcopy, _ = timeit(lambda: P.copy_(P0), reps=5, warm=1)
                t = tmin - tcopy
                print(f"panel mode={mode} nt={nt} m={m} nb={args.nb}: {t * 1e3:.1f} us  ({t * 1e3 / args.nb:.2f} us/col)",
                      flush=True)
        names = ["O1:poll headers+winner", "O2:row granules", "O3:multipliers+block", "next cand+header",
                 "barrier", "bulk+publish", "block boundary", "-"]
        dev.h.set_option("panel", 3)
        dev.h.set_option("panel_debug", 1)
        for nt, m in ((512, args.n), (256, args.n), (512, 1024), (512, 128)):
            dev.h.set_option("panel_nt", nt)
            P = torch.empty(m, args.nb, dtype=torch.float64, device="cuda")
            ipiv = torch.zeros(args.nb, dtype=torch.int32, device="cuda")
            info = torch.zeros(1, dtype=torch.int32, device="cuda")
            for rep in range(2):
                dev.fill_(P, gen.U11, 3)
                dev.panel_(P, 0, ipiv, info)
            torch.cuda.synchronize()
            rows = nt // 16 * 4
            G = (m + rows - 1) // rows
            need = 256 + 2 * G * 512 + 2 * G * 128 * 16
            off = (need + 255) & ~255
            raw = np.frombuffer(dev.h.read_scratch(off, G * 64), dtype=np.uint64).reshape(G, 8).astype(np.float64)
            us = raw / 100.0 / args.nb
            print(f"stamps3 nt={nt} m={m} G={G}: owner-wave us per column (mean | max)   total {us.sum(1).mean():.2f}")
            for i, nm in enumerate(names[:7]):
                print(f"   {nm:24s} {us[:, i].mean():7.3f} | {us[:, i].max():7.3f}")
        dev.h.set_option("panel_debug", 0)
        dev.h.set_option("panel_nt", 0)
        dev.h.set_option("panel", 3)
    if "xchg" in args.what:
        # floor of the panel's per-column exchange: one all-gather round by store policy and placement
        for mode, nm in ((2, "ping-pong (one-way = half)"), (0, "header all-gather"), (1, "header all-gather + winner row")):
            for G, stride, wt in ((32, 8, False), (32, 8, True), (32, 1, True), (64, 1, True), (32, 1, False)):
                if mode == 2 and G == 64:
                    continue
                us, ids, nf = dev.h.xchg_probe(mode, G, stride, wt, 4000)
                print(f"xchg {nm}: G={G} stride={stride} stores={'sc1' if wt else 'plain'}: {us:.3f} us/round  "
                      f"failures={nf}  xcc ids={sorted(set(ids))} (participant 0 on {ids[0]}, 1 on {ids[1]})", flush=True)
    if "panelx" in args.what or "pstamps" in args.what:
        import numpy as np
        # device-scope pipelined panel (panel=3) / the same with XCD-scope stores (panel_xcd=1) / XCD kernel (panel=4)
        variants = (("p3", 3, 0), ("p3x", 3, 1), ("p4", 4, 0))
        dev.h.set_option("panel_nt", 0)
        dev.h.set_option("panel_rt", 4)
        dt = torch.float32 if args.f32 else torch.float64
        for m in (8192, 6000, 4096, 3000, 2048, 1024, 384, 257, 100) if "panelx" in args.what else ():
            for nbw in (args.nb, 100, 17):
                for kind in (gen.U11, gen.INT5):
                    P0 = torch.empty(m, nbw, dtype=dt, device="cuda")
                    dev.fill_(P0, kind, 3)
                    outs = []
                    for nm, mode, xcd in variants:
                        dev.h.set_option("panel", mode)
                        dev.h.set_option("panel_xcd", xcd)
                        P = P0.clone()
                        ipiv = torch.zeros(nbw, dtype=torch.int32, device="cuda")
                        info = torch.zeros(1, dtype=torch.int32, device="cuda")
                        dev.panel_(P, 0, ipiv, info)
                        torch.cuda.synchronize()
                        outs.append((P, ipiv.clone(), int(info.item())))
                    res = []
                    for k in (1, 2):
                        same = torch.equal(outs[0][0], outs[k][0]) and torch.equal(outs[0][1], outs[k][1]) and outs[0][2] == outs[k][2]
                        res.append(f"{variants[k][0]}={'same' if same else 'MISMATCH'}(info {outs[k][2]})")
                    print(f"panelx check m={m} jb={nbw} kind={kind}: " + " ".join(res), flush=True)
        for nm, mode, xcd in variants if "panelx" in args.what else ():
            dev.h.set_option("panel", mode)
            dev.h.set_option("panel_xcd", xcd)
            for m in (8192, 6144, 4096, 2048, 1024, 256):
                P0 = torch.empty(m, args.nb, dtype=dt, device="cuda")
                dev.fill_(P0, gen.U11, 3)
                ipiv = torch.zeros(args.nb, dtype=torch.int32, device="cuda")
                info = torch.zeros(1, dtype=torch.int32, device="cuda")
                P = P0.clone()

                def run():
                    P.copy_(P0)
                    dev.panel_(P, 0, ipiv, info)
                tmin, tmed = timeit(run, reps=7, warm=2)
                tcopy, _ = timeit(lambda: P.copy_(P0), reps=7, warm=2)
                t = tmin - tcopy
                print(f"panel {nm} m={m} nb={args.nb}: {t * 1e3:.1f} us  ({t * 1e3 / args.nb:.2f} us/col)", flush=True)
        names = ["own: headers+winner", "own: multipliers+granules", "own: update+choose+announce", "own: barrier",
                 "next wave: barrier wait", "next wave: far granules+update", "next wave: publish", "own: second block behind the barrier"]
        dev.h.set_option("panel", 4)
        dev.h.set_option("panel_xcd", 0)
        dev.h.set_option("panel_debug", 1)
        for m in (8192, 4096, 2048, 256):
            rows = 64 * (1 if m <= 2048 else 2 if m <= 4096 else 4 if (m <= 8192 or not args.f32) else 8)
            P = torch.empty(m, args.nb, dtype=dt, device="cuda")
            ipiv = torch.zeros(args.nb, dtype=torch.int32, device="cuda")
            info = torch.zeros(1, dtype=torch.int32, device="cuda")
            for rep in range(2):
                dev.fill_(P, gen.U11, 3)
                dev.panel_(P, 0, ipiv, info)
            torch.cuda.synchronize()
            G = (m + rows - 1) // rows
            need = 256 + 2 * G * 128 + 4 * G * 128 * 16
            off = (need + 255) & ~255
            raw = np.frombuffer(dev.h.read_scratch(off, G * 128), dtype=np.uint64).reshape(G, 16)
            us = raw.astype(np.float64) / 100.0 / args.nb
            tag = f" xcc ids {sorted(set((raw[:, 15] >> 8).tolist()))} same-flag {sorted(set((raw[:, 15] & 1).tolist()))}"
            print(f"stamps4 m={m} G={G}: us per column (mean | max over workgroups)   owner total {us[:, [0, 1, 2, 3, 7]].sum(1).mean():.2f}{tag}")
            for i, nm in enumerate(names[:8]):
                print(f"   {nm:30s} {us[:, i].mean():7.3f} | {us[:, i].max():7.3f}")
        dev.h.set_option("panel_debug", 0)
        dev.h.set_option("panel", 3)
    if "cumask" in args.what:
        from collections import Counter
        pats = {
            "no mask": None,
            "bits 0..223": range(0, 224),
            "bits 32..255": range(32, 256),
            "bits i%8!=0": [i for i in range(256) if i % 8 != 0],
            "bits i%8==0": [i for i in range(256) if i % 8 == 0],
            "bits 0..31": range(0, 32),
            "bits i%8==3": [i for i in range(256) if i % 8 == 3],
            "bits 0..127": range(0, 128),
            "bits i%2==0": range(0, 256, 2),
        }
        for nm, bits in pats.items():
            try:
                res = dev.h.cu_mask_probe(bits, 2048)
            except Exception as e:  # noqa: BLE001
                print(f"cumask {nm}: FAILED {e}")
                continue
            xc = Counter(x for x, _ in res)
            cus = Counter((x, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15) for x, hw in res)
            first8 = [x for x, _ in res[:16]]
            print(f"cumask {nm}: blocks per xcc {dict(sorted(xc.items()))}  distinct (xcc,se,sh,cu) {len(cus)}  first blocks' xcc {first8}", flush=True)
    if "lu4" in args.what:
        # whole factorisation: device-scope panel (3) against the XCD-scope panel (4), both drivers; same bits
        dt = torch.float32 if args.f32 else torch.float64
        for n in (args.n,) if args.n != 8192 else (8192, 4096, 12288):
            A0 = torch.empty(n, n, dtype=dt, device="cuda")
            dev.fill_(A0, gen.U11, 1)
            A = A0.clone()
            ipiv = torch.zeros(n, dtype=torch.int32, device="cuda")
            info = torch.zeros(1, dtype=torch.int32, device="cuda")
            ref = None
            for mode, look in ((3, 0), (3, 1), (4, 0), (4, 1)):
                dev.h.set_option("panel", mode)
                dev.h.set_option("lookahead", look)
                dev.h.set_option("lookahead_min", 1024 if look else 0)

                def run():
                    A.copy_(A0)
                    dev.getrf_(A, ipiv, info)
                tmin, _ = timeit(run, reps=5, warm=2)
                tcopy, _ = timeit(lambda: A.copy_(A0), reps=3, warm=1)
                t = tmin - tcopy
                torch.cuda.synchronize()
                if ref is None:
                    ref = (A.clone(), ipiv.clone())
                    same = "reference"
                else:
                    same = "bit-identical" if (torch.equal(ref[0], A) and torch.equal(ref[1], ipiv)) else "MISMATCH"
                dev.h.prof_reset(); dev.h.prof_enable(True); run(); torch.cuda.synchronize()
                dev.h.prof_enable(False)
                pr = dev.h.prof_read()
                print(f"getrf n={n} panel={mode} lookahead={look}: {t:.2f} ms  {2 / 3 * n ** 3 / t / 1e9:.2f} TFLOP/s  info={int(info.item())} {same}  phases "
                      + " ".join(f"{k}={v['ms']:.2f}" for k, v in pr.items() if v['ms'] > 0), flush=True)
        dev.h.set_option("panel", 3); dev.h.set_option("lookahead", 1); dev.h.set_option("lookahead_min", 0)
    if "hyb" in args.what:
        # matrices taller than one XCD holds: shared-CU schedule throughout (hybrid=0) against the hand-over to the
        # XCD-scope driver once the trailing matrix fits (hybrid=1, default); sequential driver as the reference
        dt = torch.float32 if args.f32 else torch.float64
        for n in (args.n,) if args.n != 8192 else (8320, 10240, 12288, 16384):
            A0 = torch.empty(n, n, dtype=dt, device="cuda")
            dev.fill_(A0, gen.U11, 1)
            A = A0.clone()
            ipiv = torch.zeros(n, dtype=torch.int32, device="cuda")
            info = torch.zeros(1, dtype=torch.int32, device="cuda")
            ref = None
            dev.h.set_option("panel", 4)
            for look, hyb in ((0, 1), (1, 0), (1, 1)):
                dev.h.set_option("lookahead", look)
                dev.h.set_option("hybrid", hyb)

                def run():
                    A.copy_(A0)
                    dev.getrf_(A, ipiv, info)
                tmin, _ = timeit(run, reps=4, warm=2)
                tcopy, _ = timeit(lambda: A.copy_(A0), reps=3, warm=1)
                t = tmin - tcopy
                torch.cuda.synchronize()
                if ref is None:
                    ref = (A.clone(), ipiv.clone())
                    same = "reference"
                else:
                    same = "bit-identical" if (torch.equal(ref[0], A) and torch.equal(ref[1], ipiv)) else "MISMATCH"
                print(f"getrf n={n} lookahead={look} hybrid={hyb}: {t:.2f} ms  {2 / 3 * n ** 3 / t / 1e9:.2f} TFLOP/s  info={int(info.item())} {same}", flush=True)
            del A0, A
        dev.h.set_option("lookahead", 1); dev.h.set_option("hybrid", 1)
    if "pmc" in args.what:
        # workload for `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE`: two kernels of known byte counts
        # (calibration) followed by the trailing-update kernel at LU-like shapes
        m = 8064
        Cm = torch.empty(m, m, dtype=torch.float64, device="cuda")
        dev.fill_(Cm, gen.U11, 1)                   # lsx fill_kernel: writes m*m*8 B (8 B per lane)
        Cc = Cm.clone()                             # torch copy: reads m*m*8 B, writes m*m*8 B
        dev.h.set_option("gemm_waves", 0)
        for k in (128, 256):
            A = torch.randn(m, k, dtype=torch.float64, device="cuda")
            B = torch.randn(k, m, dtype=torch.float64, device="cuda")
            dev.gemm_sub_(Cm, A, B)                 # algorithmic: read + write C = 2*m*m*8 B (+ A, B slabs)
        torch.cuda.synchronize()
        print(f"pmc workload done: m={m}, C bytes one way = {m * m * 8}")
    if "pmcpanel" in args.what:
        # workload for `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE`: a copy of known size (calibration), then the
        # XCD-scope panel kernel on a 8192 x 128 panel (algorithmic: read + write the panel once = 2 * 8192 * 128 * 8 B)
        m = 8192
        dev.h.set_option("panel", 4)
        big = torch.empty(m, m, dtype=torch.float64, device="cuda")
        dev.fill_(big, gen.U11, 1)
        bigc = big.clone()                          # calibration: reads and writes m*m*8 bytes
        P = torch.empty(m, args.nb, dtype=torch.float64, device="cuda")
        ipiv = torch.zeros(args.nb, dtype=torch.int32, device="cuda")
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        for rep in range(3):
            dev.fill_(P, gen.U11, 3 + rep)
            dev.panel_(P, 0, ipiv, info)
        torch.cuda.synchronize()
        print(f"pmcpanel workload done: panel {m} x {args.nb}, bytes one way = {m * args.nb * 8}; calibration copy {m * m * 8} one way")
    if "stamps" in args.what:
        import numpy as np
        names = ["1:col update+cand", "barrier A", "2:reduce+publish", "3:bulk update", "4a:poll headers",
                 "4b:row granules", "barrier C", "5:multipliers"]
        dev.h.set_option("panel", 1)
        dev.h.set_option("panel_debug", 1)
        for nt, rt, m in ((256, 4, args.n), (512, 4, args.n), (256, 4, 1024), (256, 4, 128)):
            dev.h.set_option("panel_rt", rt)
            dev.h.set_option("panel_nt", nt)
            P = torch.empty(m, args.nb, dtype=torch.float64, device="cuda")
            ipiv = torch.zeros(args.nb, dtype=torch.int32, device="cuda")
            info = torch.zeros(1, dtype=torch.int32, device="cuda")
            for rep in range(2):
                dev.fill_(P, gen.U11, 3)
                dev.panel_(P, 0, ipiv, info)
            torch.cuda.synchronize()
            G = (m + nt // 16 * rt - 1) // (nt // 16 * rt)
            need = 256 + 2 * G * 512 + 2 * G * 128 * 16
            off = (need + 255) & ~255
            raw = np.frombuffer(dev.h.read_scratch(off, G * 64), dtype=np.uint64).reshape(G, 8).astype(np.float64)
            us = raw / 100.0 / args.nb  # 100 MHz ticks -> us per column
            print(f"stamps nt={nt} rt={rt} m={m} G={G}: per-column us (mean over WGs | max)   total {us.sum(1).mean():.2f}")
            for i, nm in enumerate(names):
                print(f"   {nm:22s} {us[:, i].mean():7.3f} | {us[:, i].max():7.3f}")
        dev.h.set_option("panel_debug", 0)
    if "stamps2" in args.what:
        import numpy as np
        names = ["1:candidates", "barrier A", "2:reduce+publish", "3:poll records", "3b:winner reduce+bcast",
                 "barrier C", "4:multipliers+block update", "block end (per block/8)"]
        dev.h.set_option("panel", 2)
        dev.h.set_option("panel_debug", 1)
        for m in (args.n, 1024, 128):
            P = torch.empty(m, args.nb, dtype=torch.float64, device="cuda")
            ipiv = torch.zeros(args.nb, dtype=torch.int32, device="cuda")
            info = torch.zeros(1, dtype=torch.int32, device="cuda")
            for rep in range(2):
                dev.fill_(P, gen.U11, 3)
                dev.panel_(P, 0, ipiv, info)
            torch.cuda.synchronize()
            G = (m + 127) // 128
            need = 256 + 2 * G * 256 + 2 * 8 * 128 * 16
            off = (need + 255) & ~255
            raw = np.frombuffer(dev.h.read_scratch(off, G * 64), dtype=np.uint64).reshape(G, 8).astype(np.float64)
            us = raw / 100.0 / args.nb
            print(f"stamps2 (blocked) m={m} G={G}: per-column us (mean | max)   total {us.sum(1).mean():.2f}")
            for i, nm in enumerate(names):
                print(f"   {nm:28s} {us[:, i].mean():7.3f} | {us[:, i].max():7.3f}")
        dev.h.set_option("panel_debug", 0)
        dev.h.set_option("panel", 1)
    if "lu" in args.what:
        n = args.n
        A0 = torch.empty(n, n, dtype=torch.float32 if args.f32 else torch.float64, device="cuda")
        dev.fill_(A0, gen.U11, 1)
        A = A0.clone()
        ipiv = torch.zeros(n, dtype=torch.int32, device="cuda")
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        dev.h.set_option("panel_nt", 0)
        dev.h.set_option("gemm_waves", 0)
        for mode, rt, look, nb, kb in ((3, 4, 0, 128, 1), (3, 4, 2, 128, 1), (3, 4, 1, 128, 1), (3, 4, 0, 128, 2), (3, 4, 0, 64, 1), (3, 4, 0, 96, 1), (4, 4, 0, 128, 1), (4, 4, 1, 128, 1)):
            if True:
                dev.h.set_option("kblock", kb)
                dev.h.set_option("panel", mode)
                dev.h.set_option("panel_rt", rt)
                dev.h.set_option("lookahead", look)
                dev.h.set_option("nb", nb)

                def run():
                    A.copy_(A0)
                    dev.getrf_(A, ipiv, info)
                tmin, _ = timeit(run, reps=3, warm=1)
                tcopy, _ = timeit(lambda: A.copy_(A0), reps=3, warm=1)
                t = tmin - tcopy
                dev.h.prof_reset(); dev.h.prof_enable(True); run(); torch.cuda.synchronize()
                dev.h.prof_enable(False)
                pr = dev.h.prof_read()
                print(f"getrf n={n} panel={mode} rt={rt} lookahead={look} nb={nb} kblock={kb}: {t:.2f} ms  {2 / 3 * n ** 3 / t / 1e9:.2f} TFLOP/s  phases "
                      + " ".join(f"{k}={v['ms']:.2f}" for k, v in pr.items()), flush=True)
        dev.h.set_option("kblock", 1); dev.h.set_option("panel", 4); dev.h.set_option("lookahead", 1)
        dev.h.set_option("nb", 128)


if __name__ == "__main__":
    main()
