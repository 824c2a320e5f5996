#!/bin/bash
# Build an EXPERIMENT variant of the library: tools/build_variant.sh NAME SOURCE.hip "-DSWITCH ..." -> sslrec_amd/csrc/exp/libsslrec_hip_NAME.so
# (the one source recompiled with the switches, the other objects of the default build linked in; `*.so` is git-ignored but travels to the
# GPU box).  Run a command on it with SSLREC_HIP_LIBRARY=sslrec_amd/csrc/exp/libsslrec_hip_NAME.so (sslrec_amd/_lib.py).
set -e
cd "$(dirname "$0")/../sslrec_amd/csrc"
name=$1; src=$2; flags=$3
make -s
mkdir -p exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c "$src" -o "exp/${src%.*}_$name.o"
objs=""
for o in spmm.o spmm_swept.o losses.o infonce.o eval.o mt19937.o plan.o; do
    if [ "$o" = "${src%.*}.o" ]; then objs="$objs exp/${src%.*}_$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs -o "exp/libsslrec_hip_$name.so"
echo "exp/libsslrec_hip_$name.so"
