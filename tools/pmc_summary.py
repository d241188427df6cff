#!/usr/bin/env python3
"""Summarise the two PMC passes over `tools/kbench.py pmc` into profiles/rNN_pmc_gemm.json.

    python tools/pmc_summary.py FETCH.csv WRITE.csv OUT.json

FETCH.csv / WRITE.csv are the `*_counter_collection.csv` files of
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python3 tools/kbench.py pmc
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR -- python3 tools/kbench.py pmc
(separate passes, as MI355X_MICROARCH.md prescribes).  Units: KiB as reported; on gfx950 FETCH_SIZE counts half
of a streamed read, so it is doubled -- the `__amd_rocclr_copyBuffer` dispatch of the same run (reads and writes
exactly m*m*8 bytes) is the calibration and is reported next to the result.
"""
import csv
import json
import sys


def per_dispatch(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d = int(r["Dispatch_Id"])
        name, val = r["Kernel_Name"], float(r["Counter_Value"])
        if d in out:
            out[d] = (name, out[d][1] + val, out[d][2])
        else:
            out[d] = (name, val, int(r["VGPR_Count"]))
    return [out[d] for d in sorted(out)]


def panel_main():
    """python tools/pmc_summary.py panel FETCH.csv WRITE.csv OUT.json  (workload: tools/kbench.py pmcpanel)"""
    fetch_csv, write_csv, dst = sys.argv[2:5]
    m, nb = 8192, 128
    fetch = per_dispatch(fetch_csv, "FETCH_SIZE")
    write = per_dispatch(write_csv, "WRITE_SIZE")
    assert [f[0] for f in fetch] == [w[0] for w in write], "the two passes ran different dispatch sequences"
    cal, launches = {}, []
    for (name, f, vg), (_, w, _) in zip(fetch, write):
        if "copyBuffer" in name and not cal:
            one_way = m * m * 8
            cal = {"copyBuffer_fetch_kib_raw": f, "copyBuffer_write_kib": w, "bytes_one_way": one_way,
                   "fetch_x2_over_bytes": 2 * f * 1024 / one_way, "write_over_bytes": w * 1024 / one_way}
        elif "panel_x_kernel" in name:
            launches.append({"fetch_kib_raw": f, "write_kib": w, "hbm_bytes": (2 * f + w) * 1024})
    alg = 2 * m * nb * 8
    hbm = sum(l["hbm_bytes"] for l in launches) / max(len(launches), 1)
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/kbench.py pmcpanel; "
                     "summarised by tools/pmc_summary.py panel",
           "kernel": "panel_x_kernel<double, 4, false>", "m": m, "nb": nb, "calibration": cal, "launches": launches,
           "algorithmic_bytes_per_launch": alg, "hbm_bytes_per_launch": hbm, "ratio": hbm / alg,
           "note": f"{m} x {nb} fp64 panel: HBM-side bytes per launch = {hbm / 1e6:.2f} MB = {hbm / alg:.2f} x the algorithmic "
                   f"{alg / 1e6:.2f} MB (FETCH_SIZE doubled per the gfx950 correction; the pivot exchange stays in the XCD's L2)"}
    json.dump(res, open(dst, "w"), indent=1)
    print(res["note"])


def main():
    if sys.argv[1] == "panel":
        return panel_main()
    fetch_csv, write_csv, dst = sys.argv[1:4]
    m = 8064
    one_way = m * m * 8
    fetch = per_dispatch(fetch_csv, "FETCH_SIZE")
    write = per_dispatch(write_csv, "WRITE_SIZE")
    assert [f[0] for f in fetch] == [w[0] for w in write], "the two passes ran different dispatch sequences"
    res = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/kbench.py pmc; "
                  "summarised by tools/pmc_summary.py",
        "unit": "KiB as reported; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of streamed reads; "
                "see the copyBuffer calibration)",
        "calibration": {},
        "launches": [],
    }
    ks = iter((128, 256))
    for (name, f, vg), (_, w, _) in zip(fetch, write):
        if "copyBuffer" in name:
            res["calibration"] = {"copyBuffer_fetch_kib_raw": f, "copyBuffer_write_kib": w, "bytes_one_way": one_way,
                                  "fetch_x2_over_bytes": 2 * f * 1024 / one_way}
        elif "gemm_sub_kernel" in name:
            k = next(ks)
            hbm = (2 * f + w) * 1024
            alg = 2 * one_way
            res["launches"].append({
                "kernel": name.split("(")[0].replace("void lsx::", ""), "m": m, "n": m, "k": k,
                "fetch_kib_raw": f, "write_kib": w, "hbm_bytes": hbm, "algorithmic_bytes": alg,
                "algorithmic_bytes_with_slabs": alg + 2 * m * k * 8, "ratio": hbm / alg})
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res["launches"], indent=1))


if __name__ == "__main__":
    main()
