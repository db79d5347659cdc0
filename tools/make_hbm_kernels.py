#!/usr/bin/env python3
"""HBM-side bytes and achieved TB/s per kernel, for the kernels of the offline batch that are bound by memory traffic rather than by the
matrix pipe: `roofline.hbm_kernels` of the bench line (north_star: "rocprof HBM GB/s ... against gfx950 peak").

usage: make_hbm_kernels.py <pmc_fetch_summary.csv> <pmc_write_summary.csv> <offline_kernel_stats.csv> <out.json>
bytes per launch = FETCH_SIZE x 2 (gfx950 tallies 128-B requests as 64 B: MI355X_MICROARCH.md, HBM section; already applied by
tools/summarize_pmc.py in `mean_bytes_corrected`) + WRITE_SIZE, both from their own rocprofv3 --pmc pass; microseconds per launch from the
rocprofv3 --kernel-trace --stats pass of the same command (tools/summarize_stats.py); achieved = bytes / us; frac = achieved / 8 TB/s."""
import csv
import json
import os
import sys

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md


def read(path, col):
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            out[row["kernel"]] = float(row[col])
    return out


def main():
    fetch, write, stats, dst = sys.argv[1:5]
    fb, wb = read(fetch, "mean_bytes_corrected"), read(write, "mean_bytes_corrected")
    us, calls, tot = read(stats, "avg_us"), read(stats, "calls_per_step"), read(stats, "total_ms")
    step_ms = sum(v for k, v in tot.items() if not k.startswith("#"))
    rows = []
    for k in us:
        if k.startswith("#") or k.startswith("gemm_f32_mfma") or k.startswith("__amd") or k not in fb:
            continue
        b = fb[k] + wb.get(k, 0.0)
        gbs = b / (us[k] * 1e-6) / 1e9 if us[k] > 0 else 0.0
        rows.append({"kernel": k, "launches_per_batch": round(calls[k], 1), "avg_us": round(us[k], 2), "fetch_bytes": int(fb[k]), "write_bytes": int(wb.get(k, 0.0)),
                     "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                     "ms_per_batch": round(us[k] * calls[k] * 1e-3, 3)})
    rows.sort(key=lambda r: -r["ms_per_batch"])
    rows = [r for r in rows if r["ms_per_batch"] >= 0.02 and r["launches_per_batch"] >= 1][:12]
    d = {"what": "the non-GEMM kernels of one B = 32 x 10 s offline batch, by time: HBM-side bytes per launch (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, "
                 "separate passes) over the launch's mean duration (rocprofv3 --kernel-trace --stats of the same command) against the 8 TB/s HBM3E peak; "
                 "a fraction far below 1 on a kernel whose bytes are re-read from the memory-side cache (MALL) or that is latency-bound says so in DESIGN 4",
         "kernels": rows, "non_gemm_ms_per_batch": round(sum(us[k] * calls[k] * 1e-3 for k in us if not k.startswith(('#', 'gemm_f32_mfma', 'k_greedy', 'k_decoder_table', '__amd'))), 3),
         "all_kernels_ms_per_batch_traced": round(step_ms, 3),
         "source": ["profiles/" + os.path.basename(p) for p in (fetch, write, stats)]}
    with open(dst, "w") as f:
        json.dump(d, f, indent=1)
    print(json.dumps(d)[:1500])


if __name__ == "__main__":
    main()
