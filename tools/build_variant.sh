#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>  ->  build/variants/librmd_<name>.so
# A/B builds of the library for kernel tuning; select with RMD_LIB_PATH=build/variants/librmd_<name>.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build/variants/$NAME
for f in raymarchdenoisercuda_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -w "$@" -c $f -o build/variants/$NAME/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/librmd_$NAME.so build/variants/$NAME/*.o
echo built build/variants/librmd_$NAME.so
