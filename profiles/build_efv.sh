#!/bin/bash
# experiment build of the library with a variant of ONE source:  bash profiles/build_efv.sh <name> <source.hip> "<-D flags>"
# -> bsms-gnn_amd/lib_<name>.so.keep (the other objects come from the experiment build in _build_exp/)
set -e
cd "$(dirname "$0")/../bsms-gnn_amd"
name=$1; src=$2; flags=$3
mkdir -p _build_exp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DBSMS_EXPERIMENTS"
srcof() { if [ -f csrc/$1.hip ]; then echo csrc/$1.hip; else echo csrc/experiments/$1.hip; fi; }   # efuse32.hip lives in csrc/experiments/ (not a product source)
for s in plan rowsum chain efuse efuse32 efwd wgrad gmp bsgmp optim hierarchy sim; do
  f=$(srcof $s)
  if [ ! -f _build_exp/$s.o ] || [ $f -nt _build_exp/$s.o ] || [ csrc/chain.h -nt _build_exp/$s.o ] || [ csrc/chain_dev.h -nt _build_exp/$s.o ]; then
    extra=""; { [ $s = rowsum ] || [ $s = sim ]; } && extra="-ffp-contract=off"
    /opt/rocm/bin/hipcc $F $extra -c $f -o _build_exp/$s.o &
  fi
done
wait
b=${src%.hip}
/opt/rocm/bin/hipcc $F $flags -c $(srcof $b) -o _build_exp/${b}_$name.o
objs=""
for s in plan rowsum chain efuse efuse32 efwd wgrad gmp bsgmp optim hierarchy sim; do
  if [ $s = $b ]; then objs="$objs _build_exp/${b}_$name.o"; else objs="$objs _build_exp/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib_$name.so.keep $objs
echo "built lib_$name.so.keep ($src $flags)"
