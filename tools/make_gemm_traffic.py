#!/usr/bin/env python3
"""Launch-weighted HBM bytes per GEMM launch from the two PMC summaries (tools/summarize_pmc.py output).
usage: make_gemm_traffic.py <fetch_summary.csv> <write_summary.csv> <out.json>"""
import csv
import json
import sys


def gemm_rows(path):
    n = tot = 0.0
    for r in csv.DictReader(open(path)):
        if r["kernel"].startswith("gemm_f32_mfma"):
            n += int(r["launches"])
            tot += int(r["launches"]) * float(r["mean_bytes_corrected"])
    return n, tot


def main():
    fetch, write, out = sys.argv[1:4]
    nf, tf = gemm_rows(fetch)
    nw, tw = gemm_rows(write)
    d = {
        "kernel": "gemm_f32_mfma* (all instantiations, launch-weighted)",
        "launches_sampled": int(nf),
        "fetch_bytes_per_launch": int(tf / nf),
        "write_bytes_per_launch": int(tw / nw),
        "hbm_bytes_per_launch": int(tf / nf + tw / nw),
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline --no-secondary`; KB x 1024; FETCH_SIZE doubled (gfx950 tallies 128-B requests as 64 B, MI355X_MICROARCH.md HBM "
                  "section); WRITE_SIZE as read",
        "source": ["profiles/" + fetch.rsplit("/", 1)[-1], "profiles/" + write.rsplit("/", 1)[-1]],
    }
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d))


if __name__ == "__main__":
    main()
