"""First 40 launches of the last smn_spr_loss call in a rocprofv3 kernel trace: start, duration, queue, name."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("void (anonymous namespace)::", "").split("(")[0][:44]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"], r["Grid_Size_X"]) for r in rows]
b = [i for i, k in enumerate(ks) if k[2].startswith("pad_rows")]
step = ks[b[-2]:]
t0 = step[0][0]
print("step span %.3f ms" % ((max(k[1] for k in step) - t0) / 1e6))
for k in step[:44]:
    print("%9.1f us  %8.1f us  q%s  %s  grid %s" % ((k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[3], k[2], k[4]))
