"""instruction mix between the s_barriers of a kernel in a hipcc -S listing:  python tools/isa_loop_mix.py file.s kernel_name_substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
m = re.search(r"^(_Z\S*%s\S*):" % re.escape(sys.argv[2]), s, re.M)
a = m.start(); b = s.index(".Lfunc_end", a)
body = s[a:b].split("\n")
def stats(lines):
    c = Counter()
    for l in lines:
        l = l.split(";")[0].strip()
        if not l or l.startswith(".") or l.endswith(":"): continue
        op = l.split()[0]
        if op.startswith("v_") and "f64" in op: c["fp64"] += 1
        elif op.startswith("v_readlane"): c["readlane"] += 1
        elif op.startswith("v_writelane"): c["writelane"] += 1
        elif op.startswith("v_accvgpr"): c["accvgpr"] += 1
        elif op.startswith("v_"): c["valu_other"] += 1
        elif op.startswith("ds_"): c[op] += 1
        elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif op.startswith("s_load"): c["s_load"] += 1
        elif op.startswith("buffer_") or op.startswith("global_"): c[op] += 1
        elif op.startswith("s_"): c["salu"] += 1
        else: c[op] += 1
    return c
idx = [i for i, l in enumerate(body) if "s_barrier" in l]
print("lines", len(body), "barriers at", idx)
tot = Counter()
for i in range(len(idx)):
    seg = body[idx[i]:idx[i + 1]] if i + 1 < len(idx) else body[idx[i]:]
    c = stats(seg); tot += c
    print(dict(c))
print("from the first barrier on:", dict(tot))
