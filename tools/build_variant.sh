#!/bin/bash
# Developer aid: build a variant of the library into tools/bin/ (git-ignored, travels to the GPU box).
# Usage: bash tools/build_variant.sh NAME [hipcc flags ...]   e.g.  build_variant.sh cyc_s1 -DFA_CYCLES -DFA_LOOP_GEN_HEADER='"/abs/gen.h"'
# The backward object is built once (tools/bin/fa_bwd_api.o) and reused: variants only touch the forward translation unit.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/bin
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I flash_attention_annotated_amd/csrc"
if [ ! -f tools/bin/fa_bwd_api.o ] || [ flash_attention_annotated_amd/csrc/fa_bwd_kernel.h -nt tools/bin/fa_bwd_api.o ]; then
  $HIPCC $FLAGS -c flash_attention_annotated_amd/csrc/fa_bwd_api.hip -o tools/bin/fa_bwd_api.o
fi
$HIPCC $FLAGS "$@" -c flash_attention_annotated_amd/csrc/fa_fwd_api.hip -o tools/bin/fwd_$name.o
$HIPCC --offload-arch=gfx950 -shared -fPIC tools/bin/fwd_$name.o tools/bin/fa_bwd_api.o -o tools/bin/libfa_$name.so
rm -f tools/bin/fwd_$name.o
echo built tools/bin/libfa_$name.so
