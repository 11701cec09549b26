#!/bin/bash
# Build libgradus_mi355x.so of a given git revision (or the working tree with "WORK") into ab/<name>.so
# usage: scripts/build_variant.sh <rev|WORK> <name> [extra hipcc flags...]
# FAST=1 reuses the in-tree fp32 object (gradus.jl_amd/csrc/gradus_mi355x_f32.o) instead of recompiling that unit:
# fine for A/B timing of the fp64 kernels, NOT for anything that runs the fp32 kernels.
set -e
REV=$1; NAME=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/ab; mkdir -p $OUT
TMP=$(mktemp -d)
if [ "$REV" = "WORK" ]; then
  cp -r "$ROOT/gradus.jl_amd/csrc" $TMP/csrc; mkdir -p $TMP/inc; cp "$ROOT/include/gradus_mi355x.h" $TMP/inc/
else
  mkdir -p $TMP/csrc $TMP/inc
  for f in gr_device.hpp gr_kernels.hpp gradus_mi355x.hip gradus_mi355x_f32.hip; do git -C $ROOT show "$REV:gradus.jl_amd/csrc/$f" > $TMP/csrc/$f; done
  git -C $ROOT show "$REV:include/gradus_mi355x.h" > $TMP/inc/gradus_mi355x.h
fi
mkdir -p $TMP/a/b; mv $TMP/csrc $TMP/a/b/csrc; mkdir -p $TMP/a/include; cp $TMP/inc/gradus_mi355x.h $TMP/a/include/
# the sources include "../../include/gradus_mi355x.h" relative to csrc
cd $TMP/a/b/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm $@"
hipcc $F -c gradus_mi355x.hip -o main.o &
if [ "${FAST:-0}" = "1" ] && [ -f "$ROOT/gradus.jl_amd/csrc/gradus_mi355x_f32.o" ]; then
  cp "$ROOT/gradus.jl_amd/csrc/gradus_mi355x_f32.o" f32.o
else
  hipcc $F -Xclang -cl-single-precision-constant -c gradus_mi355x_f32.hip -o f32.o &
fi
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so main.o f32.o
rm -rf $TMP
echo built $OUT/$NAME.so
