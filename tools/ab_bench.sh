# Developer aid: A/B on one box, FA_FWD_LIB=<other build> vs the in-tree library, interleaved.  Usage: bash tools/ab_bench.sh <old.so>
run() { if [ -n "$1" ]; then FA_FWD_LIB=$1 "${@:2}"; else "${@:2}"; fi; }
for i in 1 2; do
  for lib in "$1" ""; do
    for w in c2 c3; do
      run "$lib" python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'old' if '$lib' else 'new', d['value'], d['roofline']['kernel_ms_min'])"
    done
  done
done
for lib in "$1" ""; do echo "grid lib=${lib:-new}"; run "$lib" python tools/fwd_grid.py 0 2>&1 | grep -E "d128 causal=. s *(512|1024|2048|8192) |d 64 causal=. s *(2048|8192) "; done
