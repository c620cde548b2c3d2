"""Dev tool: one-line map of a kernel's ISA (m = MFMA, B = s_barrier, S/L = scratch store/load, w = global store,
D = buffer_load..lds, W = s_waitcnt vmcnt, [n] = basic-block label).  usage: isa_map.py file.s <mangled-substring>"""
import re, sys
txt = open(sys.argv[1]).read()
key = sys.argv[2]
lines = txt.split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN5gance19modconv_mfma_kernel") and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\ts_endpgm"))
ev = []
for l in lines[start:end]:
    if "scratch_" in l: ev.append("S" if "store" in l else "L")
    elif "v_mfma" in l: ev.append("m")
    elif "s_barrier" in l: ev.append("B")
    elif re.match(r"^\.LBB\d+_\d+:", l): ev.append("[" + l.split(":")[0].split("_")[-1] + "]")
    elif "global_store" in l or "buffer_store" in l: ev.append("w")
    elif "buffer_load" in l and "lds" in l: ev.append("D")
    elif "s_waitcnt vmcnt" in l: ev.append("W")
    elif "s_cbranch" in l or "s_branch" in l: ev.append("j")
s = "".join(ev)
print(re.sub(r"(.)\1{3,}", lambda m: f"{m.group(1)}x{len(m.group(0))} ", s))
