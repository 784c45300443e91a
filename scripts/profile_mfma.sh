#!/bin/bash
# MFMA counters of the tail-inversion microbenchmark (scripts/microbench_mfma_tail.hip).  usage (through gpurun, repo root): profile_mfma.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-mf}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/scripts/microbench_mfma_tail.bin > $O/mfma_tail.txt 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES -d $O/pmc -o pmc -- $R/scripts/microbench_mfma_tail.bin > $O/pmc.log 2>&1
python3 - $O <<'PY'
import csv, glob, json, os, sys
o = sys.argv[1]
per = {}
for f in glob.glob(os.path.join(o, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = "invert_mfma" if "invert_mfma" in r["Kernel_Name"] else ("invert_lds" if "invert_lds" in r["Kernel_Name"] else None)
        if k:
            per.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in per.items()}
out["_note"] = ("rocprofv3 --pmc ... -- scripts/microbench_mfma_tail.bin; mean per dispatch (each dispatch = 200 inversions of one 36 x 36 "
                "SPD matrix); cycles per inversion in mfma_tail.txt")
json.dump(out, open(os.path.join(o, "mfma_tail_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
cat $O/mfma_tail.txt
