# round-5 step profiles under rocprofv3 (kernel trace + stats only): the headline step, config 3 (both encoders trained) and the
# step with the RoBERTa body -> gpurun_out/r05_{bench,c3,roberta}_step_kernel_stats.csv + the bench lines printed under the profiler
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # tag, bench flags...
  tag=$1; shift
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-extra-legs "$@" > gpurun_out/r05_${tag}_line_under_rocprof.json 2> gpurun_out/prof_$tag.err || return 1
  f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1) && cp $f gpurun_out/r05_${tag}_step_kernel_stats.csv && rm -rf gpurun_out/prof_$tag
  python3 tools/kstats.py gpurun_out/r05_${tag}_step_kernel_stats.csv 22
}
for t in ${TAGS:-bench c3 roberta}; do
  echo "== $t"
  case $t in
    bench) run bench ;;
    c3) run c3 --train-encoders ;;
    roberta) run roberta --with-roberta ;;
    c5) run c5 --config c5 ;;
  esac || exit 1
done
