# step profile of one bench.py mode under rocprofv3: bash tools/run_prof_mode.sh --train-encoders  ->  gpurun_out/mode_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_mode
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mode -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-config3 "$@" > gpurun_out/prof_mode.log 2>&1 &&
f=$(ls gpurun_out/prof_mode/*/*kernel_stats.csv | head -1) && cp $f gpurun_out/mode_kernel_stats.csv && rm -rf gpurun_out/prof_mode && tail -c 300 gpurun_out/prof_mode.log
