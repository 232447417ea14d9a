cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for lib in tools/bin/libfa_bwd_old.so ""; do
    for w in c2_bwd c3_bwd; do
      if [ -n "$lib" ]; then export FA_FWD_LIB=$lib; else unset FA_FWD_LIB; fi
      timeout -k 10 120 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'old' if '$lib' else 'new', d['value'], d['ms_per_step'])"
    done
  done
done
