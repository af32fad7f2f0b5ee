# same-box A/B of hip_layers flags on a bench leg: FLAGS_A / FLAGS_B (see tools/ab_flags.py), LEG = bench flags; three interleaved rounds
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for f in "${FLAGS_A:-}" "${FLAGS_B:-PACK_FUSED=0}"; do
    FLAGS="$f" timeout -k 10 300 python3 tools/ab_flags.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-config3 --no-extra-legs ${LEG:---train-encoders} 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags [%s]: %.3f ms/step' % ('$f', d['ms_per_step']))" || exit 1
  done
done
