"""Where the scratch accesses of a variant sit: per CSDO_MARK region and loop depth (LLVM's `Depth=` annotations), from the assembly
scripts/asm_hot_regions.py left in the temp directory.  A scratch access at depth >= 4 runs once per ADMM iteration / level / pass.
usage: python scripts/asm_scratch_depth.py [BLOCK MODE] [min depth]"""
import os
import re
import sys
import tempfile
from collections import Counter

block, mode = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("512", "0")
min_depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lines = open(os.path.join(tempfile.gettempdir(), "csdo_variant_%s_%s%s.s" % (block, mode, os.environ.get("CSDO_ASM_TAG", "")))).read().split("\n")
marks = [i for i, l in enumerate(lines) if "CSDO_MARK" in l]
half = marks[(len(marks)) // 2] if marks else 0
mark, depth, fn = "?", 0, 0
out = Counter()
where = {}
for i, l in enumerate(lines):
    if "CSDO_MARK" in l:
        mark = l.split("CSDO_MARK")[1].strip()
    m = re.search(r"Depth=(\d+)", l)
    if m and ("in Loop" in l or "Loop Header" in l):
        depth = int(m.group(1))
    elif re.match(r"^\.LBB\d+_\d+:\s*$", l) or (l.startswith(".LBB") and "Loop" not in l and "Depth" not in l):
        depth = 0 if "Loop" not in "".join(lines[i:i + 3]) else depth
    if "scratch_load" in l or "scratch_store" in l:
        role = "solver" if i < half else "row"
        key = (role, mark, depth, "ld" if "scratch_load" in l else "st")
        out[key] += 1
        where.setdefault(key, i)
for (role, mark, depth, kind), n in sorted(out.items(), key=lambda kv: (-kv[0][2], kv[0][0], kv[0][1])):
    if depth >= min_depth:
        print("%-6s %-14s depth %d  %s x %3d   (first at line %d)" % (role, mark, depth, kind, n, where[(role, mark, depth, kind)] + 1))
