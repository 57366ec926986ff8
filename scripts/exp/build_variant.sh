#!/bin/bash
# build_variant.sh NAME -DFOO=1 ... : the whole product library (all translation units) with experiment macros
# -> scripts/exp/libs/lib_NAME.so  (run with LGC_LIB=... ; scripts/exp/mac_ab.sh)
R=$(cd "$(dirname "$0")/../.." && pwd)
n=$1; shift
D=/tmp/lgc_var/a/$n
mkdir -p $R/scripts/exp/libs $D /tmp/lgc_var/include
cp $R/include/*.h /tmp/lgc_var/include/
cp $R/linreg-mpc_amd/csrc/*.h $R/linreg-mpc_amd/csrc/*.hip $R/linreg-mpc_amd/csrc/Makefile $D/
(cd $D && make -s JOBS=${JOBS:-8} EXTRA="$*" 2>&1 | grep -E "error|Error" )
cp $D/liblinreg_gc.so $R/scripts/exp/libs/lib_$n.so && echo "built lib_$n.so"
