#!/bin/bash
# Developer aid (round 3): build the schedule / ablation variants of the generated forward loop.
# Each line: NAME | generator args | c (cycles build) and/or p (plain build)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin/gen
VARIANTS=(
 "base||cp"
 "s1|--sched 1|cp"
 "s2|--sched 2|cp"
 "s1d1|--sched 1 --dspos 1|cp"
 "s1l1|--sched 1 --lapos 1|cp"
 "s1d1l1|--sched 1 --dspos 1 --lapos 1|cp"
 "s0d1|--dspos 1|cp"
 "g2|--ablate 2|c"
 "g2_32|--ablate 34|c"
 "g2_64|--ablate 66|c"
 "g2_128|--ablate 130|c"
 "g2_256|--ablate 258|c"
 "g2_512|--ablate 514|c"
 "g2_1024|--ablate 1026|c"
 "g2_s1|--ablate 2 --sched 1|c"
)
build_one() {
  IFS='|' read -r name args kinds <<< "$1"
  hdr=$PWD/tools/bin/gen/$name.h
  python tools/gen_fwd_loop.py $args --out $hdr > /dev/null
  if [[ $kinds == *c* ]]; then bash tools/build_variant.sh cyc_$name -DFA_CYCLES "-DFA_LOOP_GEN_HEADER=\"$hdr\"" > /dev/null; fi
  if [[ $kinds == *p* ]]; then bash tools/build_variant.sh $name "-DFA_LOOP_GEN_HEADER=\"$hdr\"" > /dev/null; fi
  echo done $name
}
export -f build_one
printf '%s\n' "${VARIANTS[@]}" | xargs -P 6 -I{} bash -c 'build_one "{}"'
