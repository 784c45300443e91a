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
#     profile:<workload>                    rocprofv3 kernel trace + PMC passes (scripts/profile_round.sh), summaries into profiles/ by collect
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
    profile)
      bash scripts/profile_round.sh $TAG $arg ;;
    refined)   # what csdo_qp_parm::solve_refinement costs: the same bench lines with the flag on
      for w in ${arg//,/ }; do
        timeout 600 python bench.py --workload $w --solve-refinement --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_refined_$w.json 2> $O/bench_refined_$w.err; echo "bench refined $w rc=$?"
        python -c "import json,sys; d=json.load(open(sys.argv[1])); print('  refined %s: %.2f M it/s  %.2f ms/step  single %.2f ms' % (sys.argv[2], d['value']/1e6, d['ms_per_step'], d['single_instance']['do_phase_ms']['solve_kernel']))" $O/bench_refined_$w.json $w
      done ;;
    *) echo "unknown step $step" ;;
  esac
done
