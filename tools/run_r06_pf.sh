# round 6: the L2 warm-up of the seamless-ring GEMM's epilogue (MODCR_GEMM_PF, tuning library) -- tile traces of several workgroups with and
# without it, then the interleaved A/B on the FFN-up shapes and a check that the other shapes do not move
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for wg in 0 40 104 200; do
  for pf in 0 1; do
    echo "== workgroup $wg, MODCR_GEMM_PF=$pf"
    MODCR_GEMM_PF=$pf MODCR_GEMM_TRACE_WG=$wg SPECS=1 HEAT=500 timeout -k 10 120 python3 tools/trace_gemm.py 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/r06_gemm_tile_trace_pf.txt 2>&1
grep -c kloop gpurun_out/r06_gemm_tile_trace_pf.txt
python3 - <<'PY'
import re
cur=None; acc={}
for l in open("gpurun_out/r06_gemm_tile_trace_pf.txt"):
    if l.startswith("=="): cur=l.strip(); acc[cur]=[]
    m=re.search(r"kloop\s+(\d+).*tile\s+(\d+)", l)
    if m and cur: acc[cur].append((int(m.group(1)), int(m.group(2))))
for k,v in acc.items():
    if v: print("%-40s tiles %2d  mean K loop %6.0f  mean tile %6.0f (both waves)" % (k, len(v), sum(a for a,_ in v)/len(v), sum(b for _,b in v)/len(v)))
PY
VARIANTS="pf1:MODCR_GEMM_PF=1,pf0:MODCR_GEMM_PF=0" CASES="92160x3072x768:1:bf16,51712x3072x768:1:bf16,92160x3072x768:0:bf16,25600x768x2112:0:bf16,46080x4096x1024:1:bf16" ROUNDS=9 timeout -k 10 400 python3 tools/ab_gemm_order.py > gpurun_out/r06_ab_gemm_pf.log 2>&1; cat gpurun_out/r06_ab_gemm_pf.log
