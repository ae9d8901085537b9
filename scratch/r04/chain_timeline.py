"""Per super-panel timeline of the last factorisation in a rocprofv3 kernel trace (C4 step): when the bulk far update F1(s) runs,
when the chain of the next super-panel (panels, strips, near updates, F0) runs beside it, and who waits for whom."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n): return n.replace("void (anonymous namespace)::", "").split("(")[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"], int(r["Grid_Size_X"])) for r in rows)
b = [i for i, k in enumerate(ks) if k[2].startswith("build_kernel")]
step = ks[b[-1]:]
t0 = step[0][0]
qs = collections.Counter(k[3] for k in step)
bulkq = min(qs, key=qs.get)
ms = lambda t: (t - t0) / 1e6
print("step span %.3f ms; build %.3f ms" % (ms(max(k[1] for k in step)), ms(step[0][1])))
f1 = [k for k in step if k[3] == bulkq]
main = [k for k in step[1:] if k[3] != bulkq]
print("%3s %9s %9s %7s | %9s %9s %7s | %s" % ("s", "F1 start", "F1 end", "F1 ms", "chain beg", "chain end", "chain ms", "who waits"))
prev_f1_end = None
for i, k in enumerate(f1):
    nxt = f1[i + 1][0] if i + 1 < len(f1) else max(x[1] for x in step)
    # main-queue kernels that start after this F1 was launched and before the next F1 starts: chain of super-panel s+1 (+ F0)
    ch = [x for x in main if k[0] <= x[0] < nxt]
    cb = min(x[0] for x in ch) if ch else k[0]
    ce = max(x[1] for x in ch) if ch else k[0]
    busy = sum(x[1] - x[0] for x in ch) / 1e6
    who = "chain (F1 done %.3f ms earlier)" % ((ce - k[1]) / 1e6) if ce > k[1] else "F1 (chain done %.3f ms earlier)" % ((k[1] - ce) / 1e6)
    print("%3d %9.3f %9.3f %7.3f | %9.3f %9.3f %7.3f (busy %.3f, %d launches) | %s" % (i, ms(k[0]), ms(k[1]), (k[1] - k[0]) / 1e6, ms(cb), ms(ce), (ce - cb) / 1e6, busy, len(ch), who))
# before the first F1: chain of super-panel 0
ch0 = [x for x in main if x[0] < f1[0][0]]
print("chain(0) before the first F1: %.3f -> %.3f ms (%d launches, busy %.3f)" % (ms(ch0[0][0]), ms(ch0[-1][1]), len(ch0), sum(x[1] - x[0] for x in ch0) / 1e6))
tail = [x for x in main if x[0] >= f1[-1][1]]
if tail: print("after the last F1: %.3f -> %.3f ms (%d launches)" % (ms(tail[0][0]), ms(tail[-1][1]), len(tail)))
c = collections.defaultdict(lambda: [0, 0.0])
for k in main: c[k[2]][0] += 1; c[k[2]][1] += (k[1] - k[0]) / 1e6
for n, v in sorted(c.items(), key=lambda kv: -kv[1][1])[:8]: print("  main %-44s x%-4d %.3f ms" % (n[:44], v[0], v[1]))
