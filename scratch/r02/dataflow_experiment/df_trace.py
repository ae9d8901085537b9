"""Reads a SMN_DF_TRACE file: per task type durations, waits, the critical path, slot utilisation."""
import sys, numpy as np
raw = open(sys.argv[1], "rb").read()
n = int(np.frombuffer(raw[:8], np.int64)[0])
tk = np.frombuffer(raw[8:8 + 12 * n], np.uint16).reshape(n, 6)
tr = np.frombuffer(raw[8 + 12 * n:], np.int64).reshape(n, 4)
typ, ti, tc, k0, nk = tk[:, 0], tk[:, 1], tk[:, 2], tk[:, 3], tk[:, 4]
t0 = tr[:, 0].min()
claim, ready, done, wg = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0, tr[:, 3]   # microseconds
print("tasks %d  makespan %.3f ms  workgroups %d" % (n, done.max() / 1e3, len(set(wg))))
names = {0: "POTRF", 1: "TRSM", 2: "UPDATE"}
for t in (0, 1, 2):
    for kk in sorted(set(nk[typ == t])):
        m = (typ == t) & (nk == kk)
        print("  %-6s nk=%d  x%-6d run %7.1f us (p10 %.1f p90 %.1f)   wait %7.1f us (p90 %.1f)   sum run %.2f ms  sum wait %.2f ms" % (
            names[t], kk, m.sum(), np.median(done[m] - ready[m]), np.percentile(done[m] - ready[m], 10), np.percentile(done[m] - ready[m], 90),
            np.median(ready[m] - claim[m]), np.percentile(ready[m] - claim[m], 90), (done[m] - ready[m]).sum() / 1e3, (ready[m] - claim[m]).sum() / 1e3))
busy = (done - ready).sum(); wait = (ready - claim).sum()
nw = len(set(wg))
print("slot-time: run %.1f %%  wait %.1f %%  other/idle %.1f %%" % (100 * busy / (nw * done.max()), 100 * wait / (nw * done.max()), 100 * (1 - (busy + wait) / (nw * done.max()))))
# POTRF chain
p = np.where(typ == 0)[0]; p = p[np.argsort(tc[p])]
print("POTRF(j) done times (ms), every 8th:", np.round(done[p][::8] / 1e3, 2))
gaps = np.diff(done[p])
print("POTRF-to-POTRF interval us: median %.1f  first 16 %s  last 16 %s" % (np.median(gaps), np.round(gaps[:16]).astype(int), np.round(gaps[-16:]).astype(int)))
