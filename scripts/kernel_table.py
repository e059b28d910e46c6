"""Per-kernel totals from a rocprofv3 --kernel-trace --stats run: python scripts/kernel_table.py <dir> [steps]."""
import csv, glob, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):7d}  avg {float(r["AverageNs"]) / 1e3:8.2f} us  '
          f'total/step {float(r["TotalDurationNs"]) / 1e3 / steps:9.1f} us  {100 * float(r["TotalDurationNs"]) / tot:5.1f}%')
print(f"all kernels: {tot / 1e3 / steps:.1f} us per step")
