"""MFMA utilisation of the trailing-update kernel from two rocprofv3 PMC passes over `tools/kbench.py pmc`
(SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE, SQ_BUSY_CU_CYCLES + SQ_INSTS_VALU_MFMA_MOPS_F64).
usage: pmc_mfma_summary.py <dir of pass 1> <dir of pass 2> <out.json>"""
import csv
import glob
import json
import sys


def main(d1, d2, out_path):
    out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE / --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 "
                     "(separate passes) --kernel-trace -- python3 tools/kbench.py pmc",
           "note": "MfmaUtil as rocprofv3 defines it: sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs).  "
                   "GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (divided by 8 here); a v_mfma_f64_16x16x4 occupies its "
                   "SIMD for 64 cycles (busy cycles / instruction count).",
           "launches": []}
    a = {}
    for d in (d1, d2):
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
        for r in csv.DictReader(open(f)):
            if "gemm_sub" not in r["Kernel_Name"]:
                continue
            e = a.setdefault(r["Dispatch_Id"], {"kernel": r["Kernel_Name"][:60],
                                                "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
            e[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, (m, kk) in zip(sorted(a, key=int), ((8064, 128), (8064, 256))):   # the two launches of `kbench.py pmc`
        v = a[k]
        n_mfma = (m // 16) ** 2 * (kk // 4)
        gui = v["GRBM_GUI_ACTIVE"] / 8
        v.update({"m": m, "n": m, "k": kk, "mfma_instructions": n_mfma,
                  "busy_cycles_per_mfma": v["SQ_VALU_MFMA_BUSY_CYCLES"] / n_mfma,
                  "gui_active_cycles_per_xcd": gui, "mfma_util": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024),
                  "shader_clock_ghz_under_profiler": gui / v["ns"]})
        out["launches"].append(v)
    json.dump(out, open(out_path, "w"), indent=1)
    for v in out["launches"]:
        print(f"gemm_sub m=n={v['m']} k={v['k']}: MFMA utilisation {100 * v['mfma_util']:.1f} %  clock {v['shader_clock_ghz_under_profiler']:.2f} GHz")


if __name__ == "__main__":
    main(*sys.argv[1:4])
