"""Average per-dispatch counter values per kernel from a rocprofv3 *_counter_collection.csv."""
import csv
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(src)):
    k = r["Kernel_Name"][:70]
    c = r["Counter_Name"]
    acc[k][c] += float(r["Counter_Value"])
    cnt[k][c] += 1
lines = []
for k in sorted(acc):
    parts = ["%s=%.1f" % (c, acc[k][c] / cnt[k][c]) for c in sorted(acc[k])]
    lines.append("%-72s n=%d  %s" % (k, max(cnt[k].values()), "  ".join(parts)))
print("\n".join(lines))
open(dst, "w").write("\n".join(lines) + "\n")
