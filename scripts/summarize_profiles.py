"""Turns the rocprofv3 output directories of a profiling call (scripts/profile_round.sh -> gpurun_out/<tag>_<workload>/) into
the committed summaries under profiles/: kernel-trace statistics (csv as written by --stats) and the PMC means as JSON
with the derived figures bench.py reports next to the nominal roofline.

usage: python scripts/summarize_profiles.py gpurun_out r02 map100
"""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
wl = sys.argv[3] if len(sys.argv) > 3 else "map100"
base = os.path.join(src, "%s_%s" % (tag, wl))
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(out_dir, exist_ok=True)
STEPS = 4        # the profiled command runs 1 warm-up + 3 timed steps, one dispatch of the agent kernel per step and class


def find(pattern):
    return sorted(glob.glob(os.path.join(base, pattern), recursive=True))


summary = {"workload": wl}
# which kernel the counters belong to: the bench line of the traced run carries the library's csdo_source_hash() (bench.py quotes a
# summary only when it equals the running library's)
try:
    with open(os.path.join(base, "trace.log")) as fh:
        line = [l for l in fh.read().splitlines() if l.startswith("{")][-1]
    summary["kernel_source_hash"] = json.loads(line).get("kernel_source_hash")
except Exception as e:
    print("no bench line in trace.log (%s): the summary carries no kernel_source_hash and bench.py will not quote it" % e)
for f in find("trace/**/*kernel_stats.csv"):
    shutil.copy(f, os.path.join(out_dir, "%s_%s_kernel_stats.csv" % (tag, wl)))
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for row in rows[:4]:
        print("   ", {k: row[k] for k in ("Name", "Calls", "AverageNs", "Percentage") if k in row})
    agent = [r for r in rows if "dsqp_agent_kernel" in r.get("Name", "")]
    if agent:
        dom = max(agent, key=lambda r: float(r["TotalDurationNs"]))
        summary["dominant_kernel"] = dom["Name"]
        summary["dominant_kernel_avg_ms"] = float(dom["AverageNs"]) * 1e-6
        summary["ms_per_step"] = sum(float(r["TotalDurationNs"]) for r in agent) * 1e-6 / STEPS if len(agent) == 1 else \
            float(dom["AverageNs"]) * 1e-6
# A launch group may be two dispatches of the kernel on one queue (its CU share first, the rest when CUs are released): the
# group's time per step is the span from its first start to its last end, which is what bench.py's HIP events measure.
for f in find("trace/**/*kernel_trace.csv"):
    with open(f) as fh:
        disp = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(fh)
                if "dsqp_agent_kernel" in r.get("Kernel_Name", "")]
    if disp and summary.get("dominant_kernel"):
        disp.sort(key=lambda d: d[1])
        t0 = disp[0][1]
        steps, cur_end = [], None      # a step = dispatches of any agent kernel that overlap in time
        for name, a, b in disp:
            # (a group's second launch that finds its queue empty is a dispatch of a few microseconds, possibly behind the end of
            #  every other kernel of its step: it belongs to that step, it is not a step)
            if cur_end is None or (a > cur_end and (b - a) > 200000):
                steps.append([])
                cur_end = b
            steps[-1].append((name, a, b))
            cur_end = max(cur_end, b)
        dom_spans = []
        for st in steps:
            d = [(a, b) for name, a, b in st if name == summary["dominant_kernel"]]
            if d:
                dom_spans.append((max(b for _, b in d) - min(a for a, _ in d)) * 1e-6)
        step_spans = [(max(b for _, _, b in st) - min(a for _, a, _ in st)) * 1e-6 for st in steps]
        if dom_spans:
            summary["dominant_kernel_span_ms_per_step"] = sum(dom_spans) / len(dom_spans)
            summary["ms_per_step"] = sum(step_spans) / len(step_spans)
            summary["steps_in_trace"] = len(steps)
pmc = {}
for sub in ("fetch", "write", "sq", "tcc", "tcc2"):
    for f in find("%s/**/*counter_collection.csv" % sub):
        per = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if "dsqp_agent_kernel" not in name:
                    continue
                per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for cname, d in per.items():
            vals = list(d.values())
            pmc[cname] = {"dispatches": len(vals), "per_step": sum(vals) / STEPS}
        print("pmc <-", f)
summary["pmc_per_step"] = pmc
g = lambda k: pmc.get(k, {}).get("per_step")
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    # MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per
    # 128-B request, i.e. half the bytes of wide coalesced reads: doubled here (an upper bound for this kernel's narrower
    # accesses, which the guide calls uncalibrated); WRITE_SIZE is taken as is.
    hbm = 2.0 * g("FETCH_SIZE") * 1024.0 + g("WRITE_SIZE") * 1024.0
    summary["hbm_bytes_per_launch_dominant_kernel"] = hbm
    if summary.get("ms_per_step"):
        summary["hbm_counter_GBps"] = hbm / (summary["ms_per_step"] * 1e-3) / 1e9
        summary["hbm_counter_frac_of_peak"] = summary["hbm_counter_GBps"] / 8000.0
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    summary["sq_wait_any_over_wave_cycles"] = g("SQ_WAIT_ANY") / wc if g("SQ_WAIT_ANY") else None
    summary["sq_active_inst_any_over_wave_cycles"] = g("SQ_ACTIVE_INST_ANY") / wc if g("SQ_ACTIVE_INST_ANY") else None
    if g("SQ_INSTS_LDS") and g("SQ_LDS_BANK_CONFLICT") is not None:
        summary["lds_bank_conflict_cycles_per_lds_inst"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_INSTS_LDS")
    if g("SQ_INSTS_VALU") and summary.get("ms_per_step"):
        # upper bound of the fp64 issue utilisation: every VALU wave-instruction priced as a 4-cycle fp64 issue slot on one
        # of the 1024 SIMDs at 2.4 GHz
        summary["valu_fp64_issue_frac"] = g("SQ_INSTS_VALU") * 4.0 / (1024.0 * 2.4e9 * summary["ms_per_step"] * 1e-3)
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
    # the guide's L2 hit rate; every request that misses goes to the Infinity Cache / HBM (a 2 - 3 k cycle trip for a dependent load)
    summary["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    summary["l2_misses_per_launch"] = g("TCC_MISS_sum")
if g("TCP_TCC_READ_REQ_sum") and g("TCC_EA0_RDREQ_sum") is not None:
    summary["l2_read_requests_per_launch"] = g("TCP_TCC_READ_REQ_sum")
    summary["fabric_read_requests_over_l2_read_requests"] = g("TCC_EA0_RDREQ_sum") / g("TCP_TCC_READ_REQ_sum")
summary["_note"] = ("rocprofv3 --pmc, one pass per counter group (FETCH_SIZE and WRITE_SIZE in separate passes), command: python3 "
                    "bench.py --workload %s --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --setup-procs 1 --skip-single-instance; "
                    "per_step = sum over the dsqp_agent_kernel dispatches / 4 steps" % wl)
with open(os.path.join(out_dir, "%s_%s_pmc_summary.json" % (tag, wl) if wl != "map100" else "%s_pmc_summary.json" % tag), "w") as fh:
    json.dump(summary, fh, indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in summary.items() if k != "pmc_per_step"}, indent=1, sort_keys=True))
