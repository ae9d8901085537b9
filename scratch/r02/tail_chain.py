"""Per-kernel timeline of the LAST timed step's panel chain: for every 1024-column super-panel the main-stream launches
(name, duration, gap in front) and the F1 launch on the bulk stream.  Reads a rocprofv3 --kernel-trace directory."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    return r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
# last step = from the last build_kernel on
bi = max(i for i, r in enumerate(rows) if nm(r).startswith("build_kernel"))
step = rows[bi:]
t0 = int(step[0]["Start_Timestamp"])
queues = sorted({r["Queue_Id"] for r in step})
mainq = max(queues, key=lambda q: sum(1 for r in step if r["Queue_Id"] == q))
panels = [r for r in step if nm(r).startswith("panel")]
print("queues", {q: sum(1 for r in step if r["Queue_Id"] == q) for q in queues}, "panels", len(panels))
# group main-stream kernels by super-panel: 8 panels each
main = [r for r in step if r["Queue_Id"] == mainq]
sp, cnt, prev_end = 0, 0, None
acc = {}
lines = []
for r in main:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = nm(r)
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    key = n.split("<")[0] + ("<" + n.split("<")[1] if "<" in n else "")
    acc.setdefault(sp, []).append((n, (e - s) / 1e3, gap, (s - t0) / 1e3, r.get("Grid_Size", "")))
    prev_end = e
    if n.startswith("panel"):
        cnt += 1
        if cnt % 8 == 0:
            sp += 1
for k in sorted(acc):
    items = acc[k]
    dur = sum(d for _, d, _, _, _ in items); gaps = sum(g for _, _, g, _, _ in items)
    span = items[-1][3] + items[-1][1] - items[0][3]
    print("super-panel %2d: %2d launches, kernel time %7.1f us, gaps %6.1f us, span %7.1f us (from %8.1f us)" % (k, len(items), dur, gaps, span, items[0][3]))
for k in (10, 13):
    if k in acc:
        print("--- super-panel", k)
        for n, d, g, t, grid in acc[k]:
            print("   +%5.1f gap %7.1f us  %-40s grid %s" % (g, d, n[:40], grid))
