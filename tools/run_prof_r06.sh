# round 6, VERDICT r05 item 2: a SAME-LEASE pair for the roofline kernel -- the un-profiled bench line, the same command under
# rocprofv3 --kernel-trace --stats, and the clocks rocm-smi reports before / between / after -- so that the gap between the driver's
# un-profiled in-step figure and the profiled ones in profiles/ is a measurement, not a sentence (DVFS give-back item 2: profiled
# passes run at a lower clock).  Then the step profiles of config 3 / the RoBERTa step / c5 and the kernel timeline of one headline step.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
clk() { echo "== rocm-smi clocks $1"; (rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -6; rocm-smi --showpower 2>&1 | grep -i "power" | head -3) || true; }
{
clk "before (idle)"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config3 --no-extra-legs > gpurun_out/r06_bench_line_unprofiled.json 2> gpurun_out/r06_unprof.err
clk "after the un-profiled run"
rm -rf gpurun_out/prof_bench
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config3 --no-extra-legs > gpurun_out/r06_bench_line_under_rocprof.json 2> gpurun_out/prof_bench.err
clk "after the profiled run"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config3 --no-extra-legs > gpurun_out/r06_bench_line_unprofiled_2.json 2> gpurun_out/r06_unprof2.err
clk "after the second un-profiled run"
} > gpurun_out/r06_same_lease_pair.txt 2>&1
f=$(ls gpurun_out/prof_bench/*/*kernel_stats.csv | head -1) && cp $f gpurun_out/r06_bench_step_kernel_stats.csv && rm -rf gpurun_out/prof_bench
python3 - >> gpurun_out/r06_same_lease_pair.txt <<'PY'
import json
for tag in ("unprofiled", "under_rocprof", "unprofiled_2"):
    try:
        d = json.loads([l for l in open("gpurun_out/r06_bench_line_%s.json" % tag) if l.startswith('{"metric')][-1])
        r = d["roofline"]
        print("%-14s ms_per_step %.3f  in-step roofline kernel: %d launches avg %.2f us = %.4f of 2.5 PF" % (tag, d["ms_per_step"], r["launches_timed"], r["avg_launch_us"], r["frac"]))
    except Exception as e:
        print(tag, "unreadable:", e)
PY
python3 tools/kstats.py gpurun_out/r06_bench_step_kernel_stats.csv 14 >> gpurun_out/r06_same_lease_pair.txt
cat gpurun_out/r06_same_lease_pair.txt
run() {  # tag, bench flags...
  tag=$1; shift
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-extra-legs "$@" > gpurun_out/r06_${tag}_line_under_rocprof.json 2> gpurun_out/prof_$tag.err || return 1
  f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1) && cp $f gpurun_out/r06_${tag}_step_kernel_stats.csv && rm -rf gpurun_out/prof_$tag
  python3 tools/kstats.py gpurun_out/r06_${tag}_step_kernel_stats.csv 16
}
for t in ${TAGS:-c3 roberta c5}; do
  echo "== $t"
  case $t in
    c3) run c3 --train-encoders ;;
    roberta) run roberta --with-roberta ;;
    c5) run c5 --config c5 ;;
  esac || exit 1
done
rm -rf gpurun_out/prof_tl
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-config3 --no-extra-legs > gpurun_out/prof_tl.log 2>&1 &&
python3 tools/step_timeline.py gpurun_out/prof_tl gpurun_out/r06_step_timeline.txt && rm -rf gpurun_out/prof_tl && head -1 gpurun_out/r06_step_timeline.txt
# the many-rank rehearsal of the N > 1 plumbing on this one-GPU box OUTSIDE pytest -- six ranks (the pool's process guard
# allows six processes on the card; the suite's own case runs four beside the pytest process), toy dims, collectives over gloo
timeout -k 10 300 python3 bench.py --gpus 6 --rehearse-on-one-gpu --config toy --steps 5 --warmup 2 --no-cpu-baseline --no-config3 > gpurun_out/r06_rehearse_6_ranks.json 2> gpurun_out/r06_rehearse_6_ranks.err; tail -c 900 gpurun_out/r06_rehearse_6_ranks.json
