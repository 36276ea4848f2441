import numpy as np, sys
a = np.loadtxt(sys.argv[1]).astype(np.int64)
t = a[:, :6].astype(np.float64); new = a[:, 6]; tile = a[:, 7]
us = (t - t[:, 0].min()) / 100.0
print("wgs", len(a), "start max %.1f end max %.1f" % (us[:,0].max(), us[:,5].max()))
names = ["loads+scan", "popcount", "compact", "barrier", "outputs", "drain"]
for k in range(5):
    d = us[:, k+1] - us[:, k]
    print("%-12s mean %.2f p90 %.2f max %.2f" % (names[k], d.mean(), np.percentile(d, 90), d.max()))
dur = us[:,5]-us[:,0]
for h in np.argsort(-dur)[:6]: print("  slow wg", h, "tile", tile[h], "new", new[h], "start %.1f dur %.1f" % (us[h,0], dur[h]), np.round(np.diff(us[h]),2))
dense = new > 2000
print("dense tiles", dense.sum(), "mean dur %.1f ; sparse mean dur %.1f" % (dur[dense].mean(), dur[~dense].mean()))
