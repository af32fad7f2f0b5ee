# round 6, VERDICT r05 item 1: the persistent GEMMs' K-loop skeleton -- static wave priority A/B (tuning library), the in-kernel clock of the
# FFN-up kernel under load, and a fresh counter pass of FFN-up FORWARD (product library).  Outputs under gpurun_out/r06_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/ab_gemm_order.py > gpurun_out/r06_ab_gemm_setprio.log 2>&1; tail -6 gpurun_out/r06_ab_gemm_setprio.log
N=512 S=180 ATTN_DROPOUT=0.1 ROUNDS=9 timeout -k 10 200 python3 tools/ab_attn.py "default=" "prio_waves4to7=MODCR_ATTN_DEBUG=256" "prio_waves0to3=MODCR_ATTN_DEBUG=512" > gpurun_out/r06_ab_attn_setprio.log 2>&1; cat gpurun_out/r06_ab_attn_setprio.log
N=512 S=180 ROUNDS=9 timeout -k 10 200 python3 tools/ab_attn.py "default=" "prio_waves4to7=MODCR_ATTN_DEBUG=256" "prio_waves0to3=MODCR_ATTN_DEBUG=512" > gpurun_out/r06_ab_attn_setprio_eval.log 2>&1; cat gpurun_out/r06_ab_attn_setprio_eval.log
HEAT=4000 timeout -k 10 200 python3 tools/trace_gemm.py > gpurun_out/r06_gemm_tile_trace_clock.txt 2>&1; grep -n "clock\|spec=" gpurun_out/r06_gemm_tile_trace_clock.txt
HEAT=4000 SHAPE=92160x768x3072 ACT=0 timeout -k 10 200 python3 tools/trace_gemm.py > gpurun_out/r06_gemm_tile_trace_clock_ffn_down.txt 2>&1; grep -n "clock\|spec=" gpurun_out/r06_gemm_tile_trace_clock_ffn_down.txt
rm -rf gpurun_out/pmc_*
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  t=$(echo $set | cut -d' ' -f1)
  M=92160 NN=3072 K=768 ACT=1 F16=0 timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$t -- python3 tools/prof_gemm.py > gpurun_out/pmc_$t.log 2>&1 || echo "pass $t failed"
done
python3 tools/pmc_summary.py gpurun_out linear_bf16 > gpurun_out/r06_pmc_ffn_up_fwd.txt 2>&1
rm -rf gpurun_out/pmc_*
M=92160 NN=3072 K=768 ACT=1 F16=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_g -- python3 tools/prof_gemm.py > gpurun_out/kt_g.log 2>&1
echo "# kernel-trace of the same command:" >> gpurun_out/r06_pmc_ffn_up_fwd.txt
python3 tools/kstats.py $(ls -t gpurun_out/kt_g/*/*kernel_stats.csv | head -1) 3 >> gpurun_out/r06_pmc_ffn_up_fwd.txt
rm -rf gpurun_out/kt_g
cat gpurun_out/r06_pmc_ffn_up_fwd.txt
# the two global_enc passes as ONE batch of rows through the token-wise blocks (Abstract_Specific.batch_global_passes; round 1 measured it slower,
# 34.2 vs 33.5 ms at 64 examples, with that round's kernels): re-measured with the round-6 kernels, three interleaved rounds
FLAGS_A="" FLAGS_B="modeling_ensemble.BATCH_GLOBAL_PASSES=1" LEG="--optimizer hf" STEPS=20 bash tools/run_ab_flags.sh > gpurun_out/r06_ab_batch_global_passes.log 2>&1; cat gpurun_out/r06_ab_batch_global_passes.log
# the phase-3 call <3,192,1,2,0> with the align map summed before / after the dropout, one process (553 -> 604 us between
# the r04 and r05 timelines was never priced), and the same call in eval mode
AB_CALL=phase3 N=512 S=180 ATTN_DROPOUT=0.1 ROUNDS=9 timeout -k 10 200 python3 tools/ab_attn.py "map_undropped=" "map_post_dropout=SIDE_POST=1" > gpurun_out/r06_ab_phase3_side_post.log 2>&1; cat gpurun_out/r06_ab_phase3_side_post.log
AB_CALL=phase3 N=512 S=180 ROUNDS=5 timeout -k 10 200 python3 tools/ab_attn.py "eval_mode=" >> gpurun_out/r06_ab_phase3_side_post.log 2>&1; tail -1 gpurun_out/r06_ab_phase3_side_post.log
