"""Scans a kernel's ISA (hipcc -S output) for the split scalar-load pattern of device_utils.h (sload12_issue / sload12_wait):
between an s_load and the next `s_waitcnt lgkmcnt(0)` no instruction may read or write the load's destination SGPRs
(the compiler regards them as defined, so a spill or a copy there would move stale data and free the registers), and no label
or branch may lie in between either (this scan is linear: with control flow inside a window it would prove nothing) -- except
for loads off the kernarg pointer s[0:1], which the compiler itself issues and waits for.
usage: check_sload_pipeline.py file.s kernel_name_substring"""
import re
import sys

path, name = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(name), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
rng = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def sregs(text):
    out = set()
    for m in rng.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


inflight, own, bad, loads, windows = set(), set(), 0, 0, 0      # own: in-flight destinations of loads that are not kernarg loads
for i in range(start + 1, end):
    ins = lines[i].split(";")[0].strip()
    if not ins or (ins.startswith(".") and not ins.endswith(":")):
        continue
    if ins.endswith(":") or re.match(r"s_(c?branch|setpc|call|swappc)", ins):
        if own:
            print("line %d: control flow inside a request..wait window (%d registers in flight): %s" % (i + 1, len(own), ins)); bad += 1
        continue
    if ins.startswith("s_load_dword"):
        ops = ins.split(None, 1)[1].split(",")
        used = sregs(",".join(ops[1:]))
        if used & inflight:
            print("line %d reads in-flight registers: %s" % (i + 1, ins)); bad += 1
        inflight |= sregs(ops[0])
        if sregs(ops[1]) != {0, 1}:
            own |= sregs(ops[0])
        loads += 1
        continue
    if ins.startswith("s_waitcnt") and ("lgkmcnt(0)" in ins):
        if inflight:
            windows += 1
        inflight, own = set(), set()
        continue
    if inflight and (sregs(ins) & inflight):
        print("line %d touches in-flight registers %s: %s" % (i + 1, sorted(sregs(ins) & inflight), ins)); bad += 1
print("%s: %d scalar loads, %d request..wait windows, %d violations" % (name, loads, windows, bad))
sys.exit(1 if bad else 0)
