# round 6: the 128-token tile without its all-padding row block (S = 101 image-only pass, P + S = 111 RoBERTa body) and the alternating wave
# priority of phase B -- same-process A/Bs against the tuning library's knobs (MODCR_ATTN_DEBUG bit 12 = no skip; bits 10 / 11 = alternate)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== S=101 H=768 N=512, training mode (the image-only pass of the step)"
N=512 S=101 ATTN_DROPOUT=0.1 ROUNDS=9 timeout -k 10 200 python3 tools/ab_attn.py "skip_padded_block=" "all_blocks=MODCR_ATTN_DEBUG=4096"
echo "== S=101 H=768 N=512, eval"
N=512 S=101 ROUNDS=7 timeout -k 10 200 python3 tools/ab_attn.py "skip_padded_block=" "all_blocks=MODCR_ATTN_DEBUG=4096"
echo "== S=111 H=1024 N=512, training mode (the RoBERTa body's shape, without the dump)"
N=512 S=111 H=1024 ATTN_DROPOUT=0.1 ROUNDS=7 timeout -k 10 200 python3 tools/ab_attn.py "skip_padded_block=" "all_blocks=MODCR_ATTN_DEBUG=4096"
echo "== S=120 H=768 N=512 (no block to skip: the two must agree)"
N=512 S=120 ATTN_DROPOUT=0.1 ROUNDS=5 timeout -k 10 200 python3 tools/ab_attn.py "skip_padded_block=" "all_blocks=MODCR_ATTN_DEBUG=4096"
echo "== phase B, alternating priority: S=180 H=768 N=512 training mode"
N=512 S=180 ATTN_DROPOUT=0.1 ROUNDS=9 timeout -k 10 200 python3 tools/ab_attn.py "default=" "alternate_per_tile=MODCR_ATTN_DEBUG=1024" "alternate_per_2_tiles=MODCR_ATTN_DEBUG=2048"
echo "== phase B, alternating priority: eval"
N=512 S=180 ROUNDS=7 timeout -k 10 200 python3 tools/ab_attn.py "default=" "alternate_per_tile=MODCR_ATTN_DEBUG=1024" "alternate_per_2_tiles=MODCR_ATTN_DEBUG=2048"
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_ab_attn_skip_and_alternate.log
cat gpurun_out/r06_ab_attn_skip_and_alternate.log
