#!/bin/bash
# Variant build of ONE source file of the library (the other objects are the product's): tools/build_variant.sh <name> <file.hip> [flags / sed script]
#   tools/build_variant.sh lb4 dcn6_kernels.hip -DRVSR_X=1            ->  realvsr_amd/csrc/librealvsr_lb4.so
#   SED='s/a/b/' tools/build_variant.sh v2 dcn6_kernels.hip           (the sed script is applied to a copy of the file first)
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../realvsr_amd/csrc"
F=$SRC
if [ -n "$SED" ]; then F=_variant_$NAME.hip; sed "$SED" $SRC > $F; fi
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function "$@" -c $F -o /tmp/variant_$NAME.o
[ "$F" != "$SRC" ] && rm -f $F
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/variant_$NAME.o -o librealvsr_$NAME.so
echo built librealvsr_$NAME.so
