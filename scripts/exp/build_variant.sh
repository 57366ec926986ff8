#!/bin/bash
# build_variant.sh NAME -DFOO=1 ... : product library with experiment macros -> scripts/exp/libs/lib_NAME.so
R=$(cd "$(dirname "$0")/../.." && pwd)
n=$1; shift
mkdir -p $R/scripts/exp/libs
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -shared -o $R/scripts/exp/libs/lib_$n.so $R/linreg-mpc_amd/csrc/liblinreg_gc.hip 2>&1 | grep -E " error" 
exit 0
