"""From a rocprofv3 --hip-trace --kernel-trace run of host_breakdown_advance.py: for every hipGraphLaunch, how long after
the call began did the first kernel of that graph start, and how long did the previous graph's last kernel end before it."""
import csv, glob, sys
d = sys.argv[1]
api = [r for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
ker = [r for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
launches = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in api if r["Function"] == "hipGraphLaunch")
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in ker)
import bisect
starts = [k[0] for k in ks]
rows = []
for b, e in launches[20:-5]:
    i = bisect.bisect_left(starts, b)
    if i >= len(ks) or i == 0:
        continue
    rows.append((ks[i][0] - b, e - b, b - ks[i - 1][1]))
rows.sort()
n = len(rows)
med = lambda j: sorted(r[j] for r in rows)[n // 2] / 1e3
print(f"{n} graph launches: first kernel starts {med(0):.1f} us after hipGraphLaunch begins (median); the call takes {med(1):.1f} us; "
      f"the previous graph's last kernel had ended {med(2):.1f} us before the call began")
