"""Timeline of the last bench step from a rocprofv3 kernel trace: bulk-stream far updates, main-queue busy time, gaps."""
import csv, collections, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n): return n.replace("void (anonymous namespace)::", "").split("(")[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"], int(r["Grid_Size_X"])) for r in rows)
b = [i for i, k in enumerate(ks) if k[2].startswith("build_kernel")]
step = ks[b[-1]:]
t0 = step[0][0]
qs = collections.Counter(k[3] for k in step)
bulkq = min(qs, key=qs.get) if len(qs) > 1 else None
print("queues", dict(qs), "bulk", bulkq, "span ms %.3f" % ((max(k[1] for k in step) - t0) / 1e6))
main = [k for k in step if k[3] != bulkq]
prev_end = None
for k in step:
    if k[3] == bulkq:
        # chain activity between consecutive F1 starts
        print("F1 %7.3f -> %7.3f (%.3f ms)" % ((k[0] - t0) / 1e6, (k[1] - t0) / 1e6, (k[1] - k[0]) / 1e6))
c = collections.defaultdict(lambda: [0, 0.0])
for k in main:
    c[k[2]][0] += 1; c[k[2]][1] += (k[1] - k[0]) / 1e6
for n, v in sorted(c.items(), key=lambda kv: -kv[1][1]): print("  main %-40s x%-4d %.3f ms" % (n[:40], v[0], v[1]))
gaps = [(main[i + 1][0] - main[i][1]) / 1e3 for i in range(len(main) - 1)]
print("main: busy %.3f ms, gaps: median %.1f us, sum %.3f ms" % (sum(k[1] - k[0] for k in main) / 1e6, statistics.median(gaps), sum(gaps) / 1e3))
# per super-panel (between far-update launches on main = update_kernel<float,1> with big grid?) print panel durations
for nm in ("potrf_kernel", "trsm_kernel"):
    pan = [(k[1] - k[0]) / 1e3 for k in main if k[2].startswith(nm)]
    print(nm, "us: first 16", [round(p) for p in pan[:16]], "last 16", [round(p) for p in pan[-16:]])
# chain spans: from the first potrf of a block to the last trsm before the next bulk launch

