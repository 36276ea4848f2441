import numpy as np, sys
a = np.loadtxt(sys.argv[1]).astype(np.int64)
t = a[:, :6].astype(np.float64); tot = a[:, 6]; sm = a[:, 7]
t0 = t[:, 0].min()
us = (t - t0) / 100.0   # wall_clock64 = 100 MHz
print("waves", len(a), "records/wave mean", tot.mean(), "max", tot.max())
print("start: min %.1f median %.1f max %.1f us" % (us[:,0].min(), np.median(us[:,0]), us[:,0].max()))
print("end(after drain): median %.1f max %.1f us" % (np.median(us[:,5]), us[:,5].max()))
names = ["dir+clear", "sweep1", "scan", "sweep2 issue", "drain"]
for k in range(5):
    d = us[:, k+1] - us[:, k]
    print("%-14s mean %.2f  p50 %.2f  p90 %.2f  max %.2f" % (names[k], d.mean(), np.median(d), np.percentile(d, 90), d.max()))
dur = us[:,5]-us[:,0]
print("wave duration mean %.2f p90 %.2f max %.2f ; corr with records %.2f" % (dur.mean(), np.percentile(dur,90), dur.max(), np.corrcoef(dur, tot)[0,1]))
heavy = np.argsort(-dur)[:5]
for h in heavy: print("  slow wave", h, "records", tot[h], "start %.1f dur %.1f" % (us[h,0], dur[h]), "phases", np.round(np.diff(us[h]),2))
# placement: which workgroups share a CU (smid = hardware id register: se / cu bits), and the load per CU
import collections
by = collections.defaultdict(list)
for i in range(len(a)):
    by[(i % 8, int(sm[i]))].append(i)
loads = {k: tot[v].sum() for k, v in by.items()}
durs = {k: dur[v].mean() for k, v in by.items()}
ks = sorted(loads, key=lambda k: -loads[k])
print("CUs seen", len(by), "waves per CU min/max", min(len(v) for v in by.values()), max(len(v) for v in by.values()))
print("records per CU: mean %.0f max %.0f min %.0f" % (np.mean(list(loads.values())), max(loads.values()), min(loads.values())))
print("corr(CU load, CU mean wave duration) %.2f" % np.corrcoef([loads[k] for k in ks], [durs[k] for k in ks])[0, 1])
for k in ks[:3] + ks[-2:]:
    print("  xcd %d smid %5d: waves %s... load %d mean dur %.1f" % (k[0], k[1], by[k][:6], loads[k], durs[k]))
