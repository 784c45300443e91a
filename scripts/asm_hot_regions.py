"""Static check of the register allocation of one kernel variant before spending GPU time on it: compiles the variant to
assembly and counts scratch / global accesses between the CSDO_MARK labels of the program (the first half of the labels is
the solver role's instantiation, the second half the row role's).  A scratch access inside `solve_begin` or `update` is a
reload in every ADMM iteration.

usage: python scripts/asm_hot_regions.py [BLOCK MODE]        (default 512 0)
"""
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "csdotrajectoryplanning_amd", "csrc")
block, mode = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("512", "0")
out = os.path.join(tempfile.gettempdir(), "csdo_variant_%s_%s%s.s" % (block, mode, os.environ.get("CSDO_ASM_TAG", "")))
extra = ["-DCSDO_ASM_MARKS"] if os.environ.get("CSDO_ASM_MARKS") else []
cmd = ["/opt/rocm/bin/hipcc", *os.environ.get("CSDO_XFLAGS", "").split(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
       "-DCSDO_V_BLOCK=" + block, "-DCSDO_V_MODE=" + mode, "-DCSDO_V_SPLIT=" + os.environ.get("CSDO_V_SPLIT", "1"), "-S", "--cuda-device-only", *extra,
       "-Rpass-analysis=kernel-resource-usage", "-I" + csrc, os.path.join(csrc, "dsqp_variant.hip"), "-o", out]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
for line in err.splitlines():
    if re.search(r"remark:\s+(VGPRs|ScratchSize|SGPRs Spill|VGPRs Spill)", line):
        print(line.split("remark:")[1].split("[")[0].strip())
lines = open(out).read().split("\n")
marks = [(i, l.split("CSDO_MARK")[1].strip()) for i, l in enumerate(lines) if "CSDO_MARK" in l] + [(len(lines), "end")]
half = (len(marks) - 1) // 2
print("%d lines of assembly" % len(lines))
for k, ((i, name), (j, _)) in enumerate(zip(marks[:-1], marks[1:])):
    seg = lines[i:j]
    if len(seg) < 200:
        continue
    print("%-6s %-14s %5d lines  scratch ld %3d st %3d   global ld %3d st %3d" % (
        "solver" if k < half else "row", name, len(seg), sum("scratch_load" in l for l in seg),
        sum("scratch_store" in l for l in seg), sum(bool(re.search(r"\bglobal_load", l)) for l in seg),
        sum(bool(re.search(r"\bglobal_store", l)) for l in seg)))
