cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wg in 40 200; do for pf in 0 1; do echo "== chain, workgroup $wg, MODCR_GEMM_PF=$pf"; CHAIN=1 MODCR_GEMM_PF=$pf MODCR_GEMM_TRACE_WG=$wg SPECS=1 HEAT=200 timeout -k 10 120 python3 tools/trace_gemm.py 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/r06_gemm_tile_trace_chain.txt 2>&1
python3 - <<'PY'
import re
cur=None; acc={}
for l in open("gpurun_out/r06_gemm_tile_trace_chain.txt"):
    if l.startswith("=="): cur=l.strip(); acc[cur]=[]
    m=re.search(r"kloop\s+(\d+).*rest\s+(\d+).*tile\s+(\d+)", l)
    if m and cur: acc[cur].append(tuple(int(x) for x in m.groups()))
for k,v in acc.items():
    if v: print("%-44s tiles %2d  mean K loop %6.0f  mean rest %6.0f  mean tile %6.0f" % (k, len(v), sum(a for a,_,_ in v)/len(v), sum(b for _,b,_ in v)/len(v), sum(c for _,_,c in v)/len(v)))
PY
