"""Print (and optionally save) a compact per-kernel table from a rocprofv3 *_kernel_stats.csv."""
import csv
import sys

src = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dst = sys.argv[3] if len(sys.argv) > 3 else None
flt = sys.argv[4] if len(sys.argv) > 4 else ""
rows = [r for r in csv.DictReader(open(src)) if flt in r["Name"]]
rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
lines = ["%-72s %6s %10s %10s %6s" % ("kernel", "calls", "avg_us", "total_ms", "pct")]
for r in rows[:top]:
    lines.append("%-72s %6s %10.2f %10.3f %6s" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                  int(r["TotalDurationNs"]) / 1e6, r["Percentage"][:5]))
print("\n".join(lines))
if dst:
    open(dst, "w").write("\n".join(lines) + "\n")
