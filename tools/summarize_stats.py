#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats *_kernel_stats.csv: short kernel names, and the
launch-weighted mean duration over all gemm_f32_mfma* instantiations (the figure bench.py's
roofline.avg_launch_us must agree with).

usage: summarize_stats.py <kernel_stats.csv> <steps-in-run> [out.csv]
"""
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from summarize_pmc import shorten  # noqa: E402


def main():
    f, steps = sys.argv[1], float(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else None
    rows = list(csv.DictReader(open(f)))
    lines = ["kernel,calls,calls_per_step,avg_us,total_ms,pct"]
    gn = gt = 0.0
    for r in rows:
        name = shorten(r["Name"])
        calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
        if name.startswith("gemm_f32_mfma"):
            gn += calls
            gt += tot
        lines.append('"%s",%d,%.1f,%.2f,%.3f,%s' % (name, calls, calls / steps, float(r["AverageNs"]) / 1e3, tot / 1e6, r["Percentage"]))
    if gn:
        lines.append('"# all gemm_f32_mfma* launches",%d,%.1f,%.2f,%.3f,' % (gn, gn / steps, gt / gn / 1e3, gt / 1e6))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text[:4000])


if __name__ == "__main__":
    main()
