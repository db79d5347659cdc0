#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE pass.

usage: summarize_mfma.py <dir-with-*_counter_collection.csv> <out_summary.csv> [<out_gemm.json>]

Per dispatch: SQ_VALU_MFMA_BUSY_CYCLES is summed over every SIMD of the chip (one v_mfma_f32_32x32x2_f32 = 64 cycles of its SIMD:
the round-1 calibration on a GEMM of known shape gave exactly 64 x #MFMA), GRBM_GUI_ACTIVE is summed over the 8 XCDs
(MI355X_MICROARCH.md, DVFS section), so
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
is the fraction of the dispatch's SIMD-cycles on which the matrix pipe was busy -- the counter-side twin of roofline.frac (which is
flops / time / peak and also pays for the clock the chip held).  SQ_BUSY_CU_CYCLES is kept in the table as read.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from summarize_pmc import shorten  # noqa: E402

N_SIMD, N_XCD = 1024, 8


def main():
    d, out = sys.argv[1], sys.argv[2]
    out_json = sys.argv[3] if len(sys.argv) > 3 else None
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit("no counter_collection.csv under " + d)
    per = defaultdict(dict)   # (file, dispatch id) -> {counter: value, "k": kernel}
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                key = (f, row["Dispatch_Id"])
                per[key]["k"] = row["Kernel_Name"]
                per[key][row["Counter_Name"]] = per[key].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for v in per.values():
        a = agg[shorten(v["k"])]
        a[0] += 1
        a[1] += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[2] += v.get("SQ_BUSY_CU_CYCLES", 0.0)
        a[3] += v.get("GRBM_GUI_ACTIVE", 0.0)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    lines = ["kernel,launches,mean_SQ_VALU_MFMA_BUSY_CYCLES,mean_SQ_BUSY_CU_CYCLES,mean_GRBM_GUI_ACTIVE,mfma_busy"]
    gn = gm = gg = 0.0
    for k, (n, m, b, g) in rows:
        util = m / (N_SIMD * g / N_XCD) if g > 0 else 0.0
        lines.append('"%s",%d,%.0f,%.0f,%.0f,%.4f' % (k, n, m / n, b / n, g / n, util))
        if k.startswith("gemm_f32_mfma"):
            gn += n
            gm += m
            gg += g
    text = "\n".join(lines) + "\n"
    open(out, "w").write(text)
    sys.stdout.write(text[:3000])
    if out_json and gg > 0:
        j = {
            "kernel": "gemm_f32_mfma* (all instantiations, cycle-weighted)",
            "launches_sampled": int(gn),
            "mfma_busy": round(gm / (N_SIMD * gg / N_XCD), 4),
            "method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE (own pass) over `python3 bench.py --steps 2 "
                      "--warmup 1 --no-cpu-baseline --no-secondary`; mfma_busy = sum MFMA_BUSY / (1024 SIMDs x sum GRBM_GUI_ACTIVE / 8 XCDs)",
            "source": os.path.join("profiles", os.path.basename(out)),
        }
        json.dump(j, open(out_json, "w"), indent=1)
        print(json.dumps(j))


if __name__ == "__main__":
    main()
