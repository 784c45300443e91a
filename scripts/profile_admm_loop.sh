#!/bin/bash
# Counters of the ADMM loop by itself: the fixed-work build (make ../libcsdo_hip_abl_FIXED.so: every QP runs osqp_max_iter
# iterations, ten SQP iterations; 4000 iterations per agent, 97 % of the time inside the iteration) under the same SQ counter pass as
# the shipped library, on the first instances of the map100 set.  Results of that build are meaningless; its counters are the loop's.
#   usage (gpurun, repo root): bash scripts/profile_admm_loop.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
ARGS="--workload map100 --instances 12 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --setup-procs 1 --skip-single-instance"
for which in shipped admm_loop; do
  O=$R/gpurun_out/${TAG}_loop_$which
  mkdir -p $O
  if [ $which = admm_loop ]; then export CSDO_DIAG_LIB=$R/csdotrajectoryplanning_amd/libcsdo_hip_abl_FIXED.so; else unset CSDO_DIAG_LIB; fi
  timeout -s KILL 900 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/sq -o sq -- python3 $R/bench.py $ARGS > $O/sq.log 2>&1
  timeout -s KILL 900 rocprofv3 --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS -d $O/sq2 -o sq2 -- python3 $R/bench.py $ARGS > $O/sq2.log 2>&1
  tail -n 1 $O/sq.log | cut -c1-200
done
python3 - "$R" "$TAG" <<'P'
import csv, glob, json, os, sys
R, tag = sys.argv[1], sys.argv[2]
out = {}
for which in ("shipped", "admm_loop"):
    tot = {}
    for f in glob.glob(os.path.join(R, "gpurun_out", "%s_loop_%s" % (tag, which), "sq*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "dsqp_agent_kernel" in row.get("Kernel_Name", ""):
                    tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    wc = tot.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    d = {"counters_summed_over_the_agent_kernel_dispatches": tot}
    d["waves_waiting"] = tot.get("SQ_WAIT_ANY", 0.0) / wc
    d["waves_waiting_for_an_instruction_slot"] = tot.get("SQ_WAIT_INST_ANY", 0.0) / wc
    d["waves_issuing"] = tot.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    if tot.get("SQ_INSTS_LDS"):
        d["lds_bank_conflict_cycles_per_lds_instruction"] = tot.get("SQ_LDS_BANK_CONFLICT", 0.0) / tot["SQ_INSTS_LDS"]
        d["valu_per_lds_instruction"] = tot.get("SQ_INSTS_VALU", 0.0) / tot["SQ_INSTS_LDS"]
    out[which] = d
out["_note"] = ("rocprofv3 --pmc, two passes per library; shipped = libcsdo_hip.so on 12 instances of the map100 set (600 agents); admm_loop = the "
                "fixed-work build (every QP 400 iterations, 10 SQP iterations): its counters are the ADMM iteration's, the difference of "
                "the two mixes is the set-up stages'")
json.dump(out, open(os.path.join(R, "gpurun_out", "%s_admm_loop_counters.json" % tag), "w"), indent=1, sort_keys=True)
for k in ("shipped", "admm_loop"):
    print(k, {kk: round(v, 3) for kk, v in out[k].items() if not isinstance(v, dict)})
P
