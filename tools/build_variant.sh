#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>  ->  build/variants/librmd_<name>.so
# A/B builds of the library for kernel tuning; select with RMD_LIB_PATH=build/variants/librmd_<name>.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
rm -rf build/variants/$NAME
mkdir -p build/variants/$NAME
pids=()
for f in raymarchdenoisercuda_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -w "$@" -c $f -o build/variants/$NAME/$(basename $f .hip).o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done      # a failed compile fails the build (no stale object is linked)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/librmd_$NAME.so build/variants/$NAME/*.o
echo built build/variants/librmd_$NAME.so
