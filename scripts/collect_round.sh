#!/bin/bash
# After scripts/gpu_run.sh <tag> tests round profile:<workload>... (+ phase profiles, stand-in benches): copy what is judged from gpurun_out/ into profiles/.
#   usage: collect_round.sh <tag>
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r04}
for w in map100_refine1 map100_refine2; do
  [ -d gpurun_out/${TAG}_$w ] && python scripts/summarize_profiles.py gpurun_out $TAG $w > /dev/null 2>&1
done
for w in map100 map50 synth1024 room50 agents100; do
  python scripts/summarize_profiles.py gpurun_out $TAG $w > /dev/null 2>&1
  [ -f gpurun_out/$TAG/bench_$w.json ] && tail -1 gpurun_out/$TAG/bench_$w.json > profiles/${TAG}_bench_$w.json
done
[ -f gpurun_out/$TAG/bench_map100_force_dist.json ] && tail -1 gpurun_out/$TAG/bench_map100_force_dist.json > profiles/${TAG}_bench_map100_force_dist.json
[ -f gpurun_out/$TAG/parity_map100.json ] && cp gpurun_out/$TAG/parity_map100.json profiles/${TAG}_parity_map100.json
[ -f gpurun_out/$TAG/parity_map50.json ] && cp gpurun_out/$TAG/parity_map50.json profiles/${TAG}_parity_map50.json
[ -f gpurun_out/$TAG/stream_map100.txt ] && grep -v amdgpu gpurun_out/$TAG/stream_map100.txt > profiles/${TAG}_stream_map100.txt
[ -f gpurun_out/$TAG/single_instance_times.txt ] && grep -v amdgpu gpurun_out/$TAG/single_instance_times.txt > profiles/${TAG}_single_instance_times.txt
[ -f gpurun_out/$TAG/host_info.txt ] && cp gpurun_out/$TAG/host_info.txt profiles/${TAG}_host_info.txt
[ -f gpurun_out/$TAG/do_phase_times.txt ] && grep -v amdgpu gpurun_out/$TAG/do_phase_times.txt > profiles/${TAG}_do_phase_times.txt
[ -f gpurun_out/$TAG/host_stage_times.txt ] && grep -v amdgpu gpurun_out/$TAG/host_stage_times.txt > profiles/${TAG}_host_stage_times.txt
# (a phase profile is a table, not an error message: a stale libcsdo_hip_prof.so - `make -C csdotrajectoryplanning_amd/csrc prof` after every
#  change of the sources - leaves a load error in the file, which is not copied)
for pair in phases_map100:phase_profile_map100 phases_map50:phase_profile_map50 phases_room50_long:phase_profile_room50_long_horizons; do
  f=gpurun_out/$TAG/${pair%%:*}.txt
  [ -f $f ] || continue
  if grep -q "Traceback\|Error" $f; then echo "NOT collected (failed run): $f"; else cp $f profiles/${TAG}_${pair#*:}.txt; fi
done
[ -f gpurun_out/$TAG/group_times_room50.txt ] && grep -v amdgpu gpurun_out/$TAG/group_times_room50.txt > profiles/${TAG}_group_times_room50.txt
grep -E "passed|failed" gpurun_out/$TAG/pytest.log | tail -1 > profiles/${TAG}_pytest_gpu_summary.txt
grep -E "^PASSED|^FAILED" gpurun_out/$TAG/pytest.log >> profiles/${TAG}_pytest_gpu_summary.txt
for w in map100 map50; do [ -f gpurun_out/${TAG}s/bench_$w.json ] && tail -1 gpurun_out/${TAG}s/bench_$w.json > profiles/${TAG}_standin_bench_$w.json; done
python - <<PY
import json
for w in ("map100", "map50", "synth1024", "room50", "agents100"):
    d = json.loads(open("profiles/${TAG}_bench_%s.json" % w).read())
    r = d["roofline"]
    print(w, "%.2f M it/s" % (d["value"] / 1e6), "%.1f ms" % d["ms_per_step"], "bound", r["bound"], "frac %.3f" % r["frac"], "pmc", r["pmc_source"])
for w in ("map100", "map50"):
    try:
        d = json.loads(open("profiles/${TAG}_standin_bench_%s.json" % w).read())
        print("stand-in", w, "%.2f M it/s" % (d["value"] / 1e6), "%.1f ms" % d["ms_per_step"])
    except Exception as e:
        print("stand-in", w, "n/a")
PY
