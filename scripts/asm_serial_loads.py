"""Static check: chains of global / scratch loads that are each waited for (s_waitcnt vmcnt(0)) before the next one is issued - every
link is a round trip to L2 / HBM.  Reads the assembly that scripts/asm_hot_regions.py left in the temp directory.
usage: python scripts/asm_serial_loads.py [BLOCK MODE] [min chain length]"""
import os
import re
import sys
import tempfile

block, mode = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("512", "0")
min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lines = open(os.path.join(tempfile.gettempdir(), "csdo_variant_%s_%s.s" % (block, mode))).read().split("\n")
mark, chain, start, pending, gap = "?", 0, 0, False, 0
out = []
for i, l in enumerate(lines):
    if "CSDO_MARK" in l:
        mark = l.split("CSDO_MARK")[1].strip()
    ins = l.strip().split(" ")[0] if l.strip() else ""
    if re.match(r"(global|scratch|flat)_load", ins):
        if not pending:
            if chain == 0:
                start, start_mark = i, mark
            pending, gap = True, 0
        else:
            gap = 0          # several loads in flight: still one link
    elif ins == "s_waitcnt" and "vmcnt(0)" in l and pending:
        chain += 1
        pending, gap = False, 0
    elif ins and not ins.startswith((";", ".")):
        gap += 1
        if gap > 40 and not pending:
            if chain >= min_len:
                out.append((start, i, chain, start_mark))
            chain = 0
            gap = 0
for s, e, c, m in out:
    print("lines %6d..%6d  %3d waits in a row   after mark %s" % (s, e, c, m))

# per chain: how many loads each wait covers (1 = a load, its wait, the next load ...: an element-wise copy or a dependent chase)
if os.environ.get("CSDO_CHAIN_DETAIL"):
    for s, e, c, m in out:
        loads, per = 0, []
        for l in lines[s:e]:
            ins = l.strip().split(" ")[0] if l.strip() else ""
            if re.match(r"(global|scratch|flat)_load", ins):
                loads += 1
            elif ins == "s_waitcnt" and "vmcnt(0)" in l and loads:
                per.append(loads)
                loads = 0
        print("lines %6d..%6d after %-14s loads per wait: %s" % (s, e, m, per))
