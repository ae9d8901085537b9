"""Vector-instruction budget of the layer loop of recursion_sym_kernel, from the compiler's ISA (hipcc -S).
usage: isa_budget.py kernel_build.s   -> table: per variant, instructions per element and layer by opcode group."""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
names = {"f": "f32", "d": "f64"}
out = []
i = 0
while i < len(src):
    m = re.match(r"^(_ZN12_GLOBAL__N_120recursion_sym_kernelI([fd])Li(\d)ELi(\d)ELb(\d)EEEvNS_7RecArgsIT_EE):", src[i])
    if not m:
        i += 1
        continue
    sym, ty, net, act, ntk = m.groups()
    j = i + 1
    body = []
    while not src[j].startswith(".Lfunc_end"):
        body.append(src[j]); j += 1
    # innermost loops: label ... backward branch to the label; take the loop with the most v_ instructions that contains
    # a transcendental or sqrt (the layer loop; the compiler unrolls the two passes x VEC elements inside it)
    labels = {l.split(":")[0].strip(): k for k, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for k, l in enumerate(body):
        mb = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if mb and mb.group(1) in labels and labels[mb.group(1)] < k:
            seg = body[labels[mb.group(1)]:k + 1]
            ins = [x.split()[0] for x in seg if re.match(r"\s+[vsd]s?_", x)]
            nv = sum(1 for x in ins if x.startswith("v_"))
            if any(x.startswith(("v_sqrt", "v_rsq", "v_rcp")) for x in ins) and (best is None or nv > best[0]):
                best = (nv, ins)
    if best is None:
        i = j; continue
    ins = best[1]
    elems = 2 * (4 if ty == "f" else 2)          # two passes x VEC elements per lane and loop trip
    groups = collections.Counter()
    for x in ins:
        if x.startswith(("v_sqrt", "v_rsq", "v_rcp", "v_log", "v_exp")): groups["transcendental"] += 1
        elif x.startswith(("v_fma", "v_fmac", "v_pk_fma")): groups["fma"] += 1
        elif x.startswith(("v_mul", "v_pk_mul")): groups["mul"] += 1
        elif x.startswith(("v_add", "v_sub")): groups["add"] += 1
        elif x.startswith(("v_cndmask", "v_cmp", "v_med3", "v_max", "v_min")): groups["select/clamp/compare"] += 1
        elif x.startswith("v_"): groups["other valu (mov, cvt, and, bfi, ...)"] += 1
        elif x.startswith("ds_"): groups["lds"] += 1
        elif x.startswith("s_"): groups["scalar"] += 1
    nv = sum(v for k, v in groups.items() if k not in ("lds", "scalar"))
    out.append((names[ty], ["mlp", "resnet", "none"][int(net)], ["relu", "erf"][int(act)], "nngp+ntk" if ntk == "1" else "nngp",
                nv / elems, {k: round(v / elems, 2) for k, v in sorted(groups.items())}))
    i = j
for r in sorted(out):
    print("%-4s %-6s %-4s %-8s  VALU per element-layer %5.1f   %s" % r)
