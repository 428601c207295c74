#!/bin/bash
# Dev tool: build the library of a git revision (default HEAD) as yet-another-bpe_amd/csrc/libyabpe_base.so, for A/B runs
# inside ONE gpurun call (boxes differ by up to 30 %):   YABPE_LIB=$PWD/yet-another-bpe_amd/csrc/libyabpe_base.so python tools/quick_job.py
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/pkg/csrc $T/include
for f in yabpe.hip yabpe_kernels.h yabpe_aux_kernels.h yabpe_pretok_kernels.h pretok_logic.h unicode_classes.inc tile_logic.h; do
  git -C "$ROOT" show "$REV:yet-another-bpe_amd/csrc/$f" > $T/pkg/csrc/$f
done
git -C "$ROOT" show "$REV:include/yabpe.h" > $T/include/yabpe.h
# (the flags of THAT revision's Makefile: an A/B run must not credit a build flag to the change under test)
git -C "$ROOT" show "$REV:yet-another-bpe_amd/csrc/Makefile" > $T/pkg/csrc/Makefile
FLAGS=$(sed -n 's/^CXXFLAGS ?= //p' $T/pkg/csrc/Makefile)
(cd $T/pkg/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -shared -o "$ROOT/yet-another-bpe_amd/csrc/libyabpe_base.so" yabpe.hip)
rm -rf $T
echo "built libyabpe_base.so from $REV"
