# kernel timeline of one bench.py step -> gpurun_out/step_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_tl
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-config3 "$@" > gpurun_out/prof_tl.log 2>&1 &&
python tools/step_timeline.py gpurun_out/prof_tl gpurun_out/step_timeline.txt && rm -rf gpurun_out/prof_tl && tail -c 300 gpurun_out/prof_tl.log
