#!/bin/bash
# usage (GPU box, repo root): bash tools/probes/run_with_smi.sh <out.txt> <command ...>
# runs the command while sampling `rocm-smi --showpower --showclocks` every 0.4 s; prints the median power and shader clock
OUT=$1
shift
( for i in $(seq 1 200); do rocm-smi --showpower --showclocks --csv 2>/dev/null | grep card0; sleep 0.4; done ) > $OUT.smi &
SMI=$!
"$@" > $OUT 2> $OUT.err
kill $SMI 2>/dev/null
wait $SMI 2>/dev/null
python3 - $OUT.smi <<'PY'
import re, sys, statistics
rows = [l for l in open(sys.argv[1]) if l.startswith("card0")]
pw = [float(r.strip().split(",")[-1]) for r in rows]
clk = [int(re.findall(r"\((\d+)Mhz\)", r)[2]) for r in rows]
busy = [(p, c) for p, c in zip(pw, clk) if p > 400]
print("samples", len(rows), "busy", len(busy))
if busy:
    print("power W: median %.0f max %.0f | sclk MHz: median %d min %d" % (statistics.median(p for p, _ in busy), max(p for p, _ in busy), statistics.median(c for _, c in busy), min(c for _, c in busy)))
PY
