"""Idle time between the kernels of a step, from a rocprofv3 kernel trace of bench.py:
   tools/gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "ffq::k_" not in n or "synth" in n:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("ffq::")[1].split("(")[0].split("<")[0]))
rows.sort()
rows = rows[len(rows) // 2:]          # the later half: steady state
gap = collections.defaultdict(list)
dur = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gap[(n0, n1)].append(s1 - e0)
    dur[n0].append(e0 - s0)
for k, v in sorted(gap.items(), key=lambda kv: -len(kv[1])):
    if len(v) > 5:
        v.sort()
        print("%-14s -> %-14s  n=%4d  gap median %7.2f us  (min %.2f max %.2f)" % (k[0], k[1], len(v), v[len(v) // 2] / 1e3, v[0] / 1e3, v[-1] / 1e3))
for k, v in dur.items():
    v.sort(); print("%-14s duration median %.2f us" % (k, v[len(v) // 2] / 1e3))
