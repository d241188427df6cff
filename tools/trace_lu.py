#!/usr/bin/env python3
"""One look-ahead LU under `rocprofv3 --kernel-trace`: workload + analysis of the panel-to-panel chain.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/trace_lu.py run [--n N] [--panel P]
    python3 tools/trace_lu.py analyse DIR
"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(argv):
    import argparse
    import torch
    from linalg_solver_amd import gen
    from linalg_solver_amd.device import DeviceSolver
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--panel", type=int, default=4)
    ap.add_argument("--lookahead", type=int, default=1)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args(argv)
    dev = DeviceSolver()
    dev.h.set_option("panel", a.panel)
    dev.h.set_option("lookahead", a.lookahead)
    A0 = torch.empty(a.n, a.n, dtype=torch.float64, device="cuda")
    dev.fill_(A0, gen.U11, 1)
    A = A0.clone()
    ipiv = torch.zeros(a.n, dtype=torch.int32, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    for _ in range(a.reps):
        A.copy_(A0)
        dev.getrf_(A, ipiv, info)
    torch.cuda.synchronize()
    print("done", int(info.item()))


def analyse(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    short = lambda n: n.split("(")[0].replace("void lsx::", "")[:60]
    panels = [i for i, r in enumerate(rows) if "panel_x_kernel" in r[2] or "panel_pipe" in r[2]]
    # the last factorisation only: panels after the last big gap
    if not panels:
        print("no panel kernels found")
        return
    # the last factorisation: panel durations shrink along a factorisation, so it starts where one jumps back up
    last = [panels[-1]]
    for i in reversed(panels[:-1]):
        if (rows[i][1] - rows[i][0]) < 0.6 * (rows[last[0]][1] - rows[last[0]][0]) and len(last) > 8:
            break
        last.insert(0, i)
    print(f"{len(last)} panels in the last factorisation")
    tot_panel = sum(rows[i][1] - rows[i][0] for i in last) / 1e3
    span = (rows[last[-1]][1] - rows[last[0]][0]) / 1e3
    print(f"sum of panel kernels {tot_panel:.0f} us, first panel start -> last panel end {span:.0f} us, "
          f"between panels {span - tot_panel:.0f} us = {(span - tot_panel) / max(1, len(last) - 1):.1f} us per step")
    gq = [r for r in rows if "gemm_sub_queue" in r[2] and r[0] >= rows[last[0]][0] - 1000]
    print("step: panel us | queue gemm us (start rel. to panel start)")
    for k in range(0, len(last), max(1, len(last) // 16)):
        a = last[k]
        g = [x for x in gq if abs(x[0] - rows[a][0]) < 60_000]
        gs = f"{(g[0][1] - g[0][0]) / 1e3:7.1f} ({(g[0][0] - rows[a][0]) / 1e3:+.1f})" if g else "   -"
        gap = (rows[a][0] - rows[last[k - 1]][1]) / 1e3 if k else 0.0
        print(f"  {k:3d}: {(rows[a][1] - rows[a][0]) / 1e3:7.1f} | {gs}   gap before {gap:6.1f}")
    for k in (1, len(last) // 2, len(last) - 3):
        if k + 1 >= len(last) or k < 0:
            continue
        a, b = last[k], last[k + 1]
        t0 = rows[a][1]
        print(f"--- between panel {k} (len {(rows[a][1] - rows[a][0]) / 1e3:.1f} us) and panel {k + 1}: "
              f"{(rows[b][0] - t0) / 1e3:.1f} us; kernels active from panel {k}'s start on:")
        for r in rows:
            if r[1] >= rows[a][0] and r[0] <= rows[b][0] + 1000:
                print(f"    {(r[0] - t0) / 1e3:9.1f} .. {(r[1] - t0) / 1e3:9.1f}  ({(r[1] - r[0]) / 1e3:7.1f} us)  {short(r[2])}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2:])
    else:
        analyse(sys.argv[2])
