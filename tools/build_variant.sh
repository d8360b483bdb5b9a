#!/bin/bash
# Tuning aid: build libmlpk with ONE source recompiled under extra flags -> jittor-mlp_amd/lib/variants/libmlpk_<tag>.so
# usage: tools/build_variant.sh <tag> <source.hip> <flags...>;  run with MLPK_LIB_PATH=<that file>
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
P=jittor-mlp_amd
mkdir -p $P/lib/variants /tmp/mlpk_var_$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result "$@" -c $P/csrc/$src -o /tmp/mlpk_var_$tag/v.o
objs=""
for f in $P/build/mlpk_*.o; do
  case $f in *-hip-*|*-host-*) continue;; esac
  [ "$(basename $f)" = "${src%.hip}.o" ] && continue
  objs="$objs $f"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/variants/libmlpk_$tag.so $objs /tmp/mlpk_var_$tag/v.o
echo $P/lib/variants/libmlpk_$tag.so
