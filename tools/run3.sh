cd multimodal-context-reasoning_amd && timeout 600 python run_PMR_ModCR.py --do_train --do_eval --max_steps 6 --logging_steps 2 --valid_steps 4 --synthetic_train_examples 256 --synthetic_val_examples 16 --per_gpu_train_batch_size 16 --output_dir /tmp/out 2>&1 | grep -v amdgpu.ids | tail -8
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py 2>&1 | tail -1 > gpurun_out/bench_r1.json; cat gpurun_out/bench_r1.json | cut -c1-1500
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r1c.log 2>&1
