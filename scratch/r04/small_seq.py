"""Kernel sequence (start, gap, duration) of the LAST loss / loss_and_grad call of scratch/small_trace.py from a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("void (anonymous namespace)::", "").split("(")[0][:60]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Grid_Size_X"] if "Grid_Size_X" in r else "?") for r in rows]
# split into calls at gaps > 300 us
calls, cur = [], [ks[0]]
for k in ks[1:]:
    if k[0] - cur[-1][1] > 300_000:
        calls.append(cur); cur = []
    cur.append(k)
calls.append(cur)
print(len(calls), "calls:", [len(c) for c in calls])
for idx in (2, 4, 6):
    if idx >= len(calls): continue
    c = calls[idx]
    t0 = c[0][0]
    print("--- call", idx, "launches", len(c), "span %.1f us" % ((c[-1][1] - t0) / 1e3), "busy %.1f us" % (sum(k[1] - k[0] for k in c) / 1e3))
    prev = t0
    for k in c:
        print("  %7.1f us  +%5.1f gap  %6.1f us  %s  grid %s" % ((k[0] - t0) / 1e3, (k[0] - prev) / 1e3, (k[1] - k[0]) / 1e3, k[2], k[3]))
        prev = k[1]
