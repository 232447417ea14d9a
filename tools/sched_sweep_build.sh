#!/bin/bash
# Developer aid (round 3): build schedule / ablation variants of the generated forward loop into tools/bin/.
# Usage: bash tools/sched_sweep_build.sh VARIANTS_FILE     lines: NAME | generator args (--slice with _ for spaces) | c (cycles build) and/or p (plain build)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin/gen
build_one() {
  IFS='|' read -r name args kinds <<< "$1"
  name=$(echo $name); kinds=$(echo $kinds)
  hdr=$PWD/tools/bin/gen/$name.h
  python tools/gen_fwd_loop.py $args --out $hdr > /dev/null
  if [[ $kinds == *c* ]]; then bash tools/build_variant.sh cyc_$name -DFA_CYCLES "-DFA_LOOP_GEN_HEADER=\"$hdr\"" > /dev/null; fi
  if [[ $kinds == *p* ]]; then bash tools/build_variant.sh $name "-DFA_LOOP_GEN_HEADER=\"$hdr\"" > /dev/null; fi
  echo done $name
}
export -f build_one
grep -v '^#' "$1" | grep . | tr '\n' '\0' | xargs -0 -P 6 -I{} bash -c 'build_one "$1"' _ {}
