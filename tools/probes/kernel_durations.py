"""Dev probe: mean duration of one kernel per grid shape from a rocprofv3 --kernel-trace CSV.
usage: kernel_durations.py <dir-with-*_kernel_trace.csv> <kernel-name-substring>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        d[(int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    print(k, len(v), round(sum(v) / len(v), 2), "us")
