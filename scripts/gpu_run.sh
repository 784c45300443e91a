#!/bin/bash
# ONE parametrised GPU runner (round 6; replaces the per-experiment gpu_*.sh scripts of rounds 3 - 5).
#   gpu_run.sh <tag> <step> [<step> ...]          steps, in the order given:
#     tests[:<pytest -k expression>]        pytest -m gpu
#     bench:<workload>[,<workload>...]      default bench.py lines into gpurun_out/<tag>/bench_<workload>.json
#     regress:<lib>[:<emu lib>][@workloads] HIP library against the lane-serial build of the same source, bit for bit
#     ab:<workloads>:<libA>,<libB>[,...]    interleaved A/B timing (scripts/ab_bench.py), one table per workload
#     single:<lib>[,<lib>...]               one 50-agent instance alone (scripts/single_instance_times.py)
#     phases:<workload>[:<lib>]             in-kernel phase profile (prof build or the library given)
#     refined:<workload>[,...]              bench lines with csdo_qp_parm::solve_refinement = 1
#     smoke                                 __graft_entry__.smoke()
#     round                                 the round's record: all bench lines, phase profiles, sweep, ... (then scripts/collect_round.sh)
#     boxes                                 safe-box ablation builds
#     profile:<workload>                    rocprofv3 kernel trace + PMC passes (scripts/profile_round.sh), summaries into profiles/ by collect
#     summarize                             ... and on the box (put it between the profile steps and `round`)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for step in "$@"; do
  kind=${step%%:*}; arg=${step#*:}; [ "$arg" == "$step" ] && arg=""
  echo "=== $step"
  case $kind in
    tests)
      timeout 2400 python -m pytest tests -m gpu -q --tb=short -x ${arg:+-k "$arg"} > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log ;;
    bench)
      for w in ${arg//,/ }; do
        timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
        python - $O/bench_$w.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("  %.2f M it/s  %.2f ms/step  single %s ms  e2e %s ms  roofline %s %.3f  pmc %s" % (
        d["value"] / 1e6, d["ms_per_step"], d["single_instance"] and round(d["single_instance"]["do_phase_ms"]["solve_kernel"], 2),
        d["do_phase_e2e"] and round(d["do_phase_e2e"]["total_ms"], 1), d["roofline"]["bound"], d["roofline"]["frac"], d["roofline"]["pmc_source"]))
except Exception as e:
    print("  ERR", e)
PY
      done ;;
    regress)
      lib=${arg%%@*}; wls=${arg#*@}; [ "$wls" == "$arg" ] && wls="map100,map50,room50"
      emu=""; case $lib in *:*) emu=${lib#*:}; lib=${lib%%:*};; esac
      CSDO_DIAG_LIB=$lib CSDO_EMU_LIB=${emu:+$PWD/$emu} timeout 1500 python scripts/gpu_regress.py --against-emu --workload $wls > $O/regress_$(basename $lib .so).txt 2>&1
      grep "HIP vs lane-serial\|Error\|error" $O/regress_$(basename $lib .so).txt | head -12 ;;
    ab)
      wls=${arg%%:*}; libs=${arg#*:}
      for w in ${wls//,/ }; do
        timeout 1500 python scripts/ab_bench.py --rounds 3 --workload $w ${libs//,/ } > $O/ab_$w.txt 2>&1; echo "-- $w"; cat $O/ab_$w.txt
      done ;;
    single)
      for lib in ${arg//,/ }; do
        echo "-- $lib"; CSDO_DIAG_LIB=$lib timeout 600 python scripts/single_instance_times.py 2>&1 | tee $O/single_$(basename $lib .so).txt | tail -6
      done ;;
    phases)
      w=${arg%%:*}; lib=${arg#*:}; [ "$lib" == "$arg" ] && lib=""
      tagl=$(basename ${lib:-prof} .so)
      CSDO_DIAG_LIB=${lib:-csdotrajectoryplanning_amd/libcsdo_hip_prof.so} timeout 900 python scripts/profile_phases_sum.py 0,1,2,4,5,6,7,9 $w > $O/phases_${w}_$tagl.txt 2>&1; head -12 $O/phases_${w}_$tagl.txt ;;
    profile)   # profile:<workload>[:<solve_refinement>]
      bash scripts/profile_round.sh $TAG ${arg%%:*} $([ "${arg#*:}" != "$arg" ] && echo ${arg#*:}) ;;
    summarize)  # the counter summaries of this call's profile:<workload> steps into profiles/ ON THE BOX, so that bench lines run later in
                # the same call (round) quote counters taken from the very library they time (bench.py: _newest_pmc compares the hashes)
      for w in map100 map50 synth1024 room50 agents100 map100_refine1 map100_refine2; do
        [ -d gpurun_out/${TAG}_$w ] && python scripts/summarize_profiles.py gpurun_out $TAG $w > /dev/null 2>&1
      done
      grep -h kernel_source_hash profiles/${TAG}_*pmc_summary.json | sort | uniq -c ;;
    refined)   # what csdo_qp_parm::solve_refinement costs: the same bench lines with the flag on
      for w in ${arg//,/ }; do
        for m in 1 2; do
          timeout 600 python bench.py --workload $w --solve-refinement $m --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_refined${m}_$w.json 2> $O/bench_refined${m}_$w.err; echo "bench solve_refinement=$m $w rc=$?"
          python -c "import json,sys; d=json.load(open(sys.argv[1])); print('  solve_refinement %s, %s: %.2f M it/s  %.2f ms/step  single %.2f ms' % (sys.argv[3], sys.argv[2], d['value']/1e6, d['ms_per_step'], d['single_instance']['do_phase_ms']['solve_kernel']))" $O/bench_refined${m}_$w.json $w $m
        done
      done ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log ;;
    round)     # everything a round's record needs beside the tests: host info, all five bench lines, the one-rank RCCL line, phase
               # profiles, streamed chunkings, the authors' sweep, single instances, launch-group times (collect: scripts/collect_round.sh <tag>)
      python -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))" > $O/host_info.txt 2>&1
      for w in map100 map50 synth1024 room50 agents100; do
        timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
      done
      timeout 600 python bench.py --force-dist --no-cpu-baseline > $O/bench_map100_force_dist.json 2> $O/bench_map100_force_dist.err; echo "bench force-dist rc=$?"
      timeout 600 python scripts/profile_phases_sum.py 0,1,2,4,5,6,7,9 map100 > $O/phases_map100.txt 2>&1
      timeout 600 python scripts/profile_phases_sum.py 0,2,3,4,5,6,7,9 map50 > $O/phases_map50.txt 2>&1
      timeout 600 python scripts/profile_phases_sum.py 1,2 room50 > $O/phases_room50_long.txt 2>&1
      timeout 600 python scripts/stream_fractions.py map100 "0.08,0.27,0.65" > $O/stream_map100.txt 2>&1
      timeout 900 python scripts/authors_sweep.py $O/authors_sweep.json > $O/authors_sweep.log 2>&1
      timeout 300 python scripts/single_instance_times.py > $O/single_instance_times.txt 2>&1
      timeout 300 python scripts/group_times.py room50 > $O/group_times_room50.txt 2>&1
      timeout 600 python scripts/do_phase_times.py > $O/do_phase_times.txt 2>&1          # csdo_do_phase against the Python-driven pipeline, all workloads
      timeout 300 python scripts/host_stage_times.py map100 60 > $O/host_stage_times.txt 2>&1
      CSDO_HOST_THREADS=1 timeout 300 python scripts/host_stage_times.py map100 60 >> $O/host_stage_times.txt 2>&1 ;;
    boxes)     # what the pieces of a safe box cost: phase-timer builds that do one piece twice (scripts/build_prof_ablation.sh)
      for t in "" _box2x_ALL _box2x_PASSES _box2x_REPLAY _box2x_CULL; do
        CSDO_PROF_LIB=libcsdo_hip_prof$t.so timeout 600 python scripts/profile_phases_sum.py 0,1,2,4,5,6,7,9 map100 > $O/phases$t.txt 2>&1
        echo "prof$t: $(sed -n 1p $O/phases$t.txt | cut -c1-80)"; grep -m1 "cycles per SQP iteration: corridor" $O/phases$t.txt
      done ;;
    *) echo "unknown step $step" ;;
  esac
done
