#!/bin/bash
# GPU side of tools/sched_sweep_build.sh: cycles per tile of every instrumented variant, then an interleaved A/B of the plain ones.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sched_sweep.txt
: > $OUT
for f in tools/bin/libfa_cyc_*.so; do
  n=$(basename $f .so); echo "== $n" >> $OUT
  FA_FWD_LIB=$f timeout -k 10 120 python tools/loop_cycles.py 2>&1 | grep -v amdgpu.ids | head -2 >> $OUT || exit 1
done
echo "== interleaved A/B" >> $OUT
PLAIN=$(ls tools/bin/libfa_*.so | grep -v cyc_)
timeout -k 10 400 python tools/ab_interleaved.py --rounds 5 --shapes c2,s2048c,d64,d64c,d96 $PLAIN 2>&1 | grep -v amdgpu.ids >> $OUT
cat $OUT
