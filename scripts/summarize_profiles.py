"""Turns the rocprofv3 output directories of a profiling call (gpurun_out/r01_*) into the committed summaries under
profiles/: kernel-trace statistics (csv as written by --stats) and the per-dispatch PMC means as JSON.

usage: python scripts/summarize_profiles.py gpurun_out r01
"""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(out_dir, exist_ok=True)


def find(pattern):
    return sorted(glob.glob(os.path.join(src, pattern), recursive=True))


# kernel stats
for f in find("%s_trace/**/*kernel_stats.csv" % tag):
    shutil.copy(f, os.path.join(out_dir, "%s_kernel_stats.csv" % tag))
    print("kernel stats <-", f)
    with open(f) as fh:
        for row in list(csv.DictReader(fh))[:4]:
            print("   ", {k: row[k] for k in ("Name", "Calls", "AverageNs", "Percentage") if k in row})

summary = {}
STEPS = 4   # the profiled command runs 1 warm-up + 3 timed steps; a launch group may be two dispatches sharing one queue
dominant = "dsqp_agent_kernel<512, 0"
for sub in ("fetch", "write", "sq"):
    for f in find("%s_%s/**/*counter_collection.csv" % (tag, sub)):
        per = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if "dsqp_agent_kernel" not in name:
                    continue
                key = (row["Counter_Name"], "dominant" if name.replace("(int)", "").startswith(dominant) or
                       "<512, 0" in name else "other")
                per.setdefault(key, {}).setdefault(row["Dispatch_Id"], 0.0)
                per[key][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for (cname, which), d in per.items():
            vals = list(d.values())
            summary.setdefault(cname, {})[which] = {"dispatches": len(vals), "mean_per_dispatch": sum(vals) / len(vals),
                                                    "per_step": sum(vals) / STEPS}
        print("pmc <-", f)
if summary:
    f_kib = summary.get("FETCH_SIZE", {}).get("dominant", {}).get("per_step")
    w_kib = summary.get("WRITE_SIZE", {}).get("dominant", {}).get("per_step")
    if f_kib is not None and w_kib is not None:
        # MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B
        # per 128-B request, i.e. half the bytes of wide coalesced reads: doubled here (upper bound for this kernel's
        # 8-B-per-lane accesses, which the guide calls uncalibrated); WRITE_SIZE is taken as is.
        summary["hbm_bytes_per_launch_dominant_kernel"] = 2.0 * f_kib * 1024.0 + w_kib * 1024.0
    summary["_note"] = ("rocprofv3 --pmc, one pass per counter group (FETCH_SIZE and WRITE_SIZE in separate passes), command: "
                        "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --setup-procs 1 --skip-single-instance; per_step = sum over the "
                        "dsqp_agent_kernel dispatches / 4 steps (a launch group is a first launch on its share of the CUs plus a "
                        "second launch on the same queue); 'dominant' = the <512, 0, true> instantiation")
    with open(os.path.join(out_dir, "%s_pmc_summary.json" % tag), "w") as fh:
        json.dump(summary, fh, indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1, sort_keys=True)[:1500])
