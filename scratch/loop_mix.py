"""Instruction mix of the main k-loop (the largest loop containing MFMAs) of every engine instantiation."""
import re, sys, collections
txt = open(sys.argv[1]).read()
for m in re.finditer(r"^(_Z15lvt_gemm_kernelILi(\d)ELi(\d)ELi(\d+)ELi(\d+)ELi\dELi\dELi(\d)EEv7KParams):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
    a, b, bm, bn, math, body = m.group(2), m.group(3), m.group(4), m.group(5), m.group(6), m.group(7)
    lines = body.split("\n")
    labels = {l.split(":")[0].strip(): i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(lines):
        mm = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            h = labels[mm.group(1)]
            if any("v_mfma" in x for x in lines[h:i]) and (best is None or i - h > best[1] - best[0]):
                best = (h, i)
    if best is None:
        continue
    loop = [l.strip().split()[0] for l in lines[best[0] + 1:best[1] + 1] if l.strip() and not l.strip().startswith((";", "."))]
    c = collections.Counter()
    for op in loop:
        k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "ds" if op.startswith("ds_") else
             "vmem" if op.startswith(("global_", "buffer_")) else "salu" if op.startswith("s_") else "other")
        c[k] += 1
    nreg = re.search(r"NumVgprs: (\d+)", txt[m.end():m.end() + 6000])
    print("A%s B%s %sx%s math%s: " % (a, b, bm, bn, math) + " ".join("%s=%d" % kv for kv in sorted(c.items())) +
          "  valu/mfma=%.1f  vgprs=%s" % (c["valu"] / max(c["mfma"], 1), nreg.group(1) if nreg else "?"))
