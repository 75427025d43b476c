#!/bin/bash
# Developer tool: build a VARIANT of libastroburst_hip.so for same-box A/B runs (AB_LIB_PATH=<variant> python tools/time_stack_bench_data.py).
#   tools/build_variant.sh <name> <file.hip> [-DFLAG ...]   ->  astroburst_amd/csrc/build/variants/libab_<name>.so
# Only <file.hip> is recompiled with the extra flags; every other object comes from the regular build.
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../astroburst_amd/csrc"
make -s -j8 dev
mkdir -p build/variants
obj=build/variants/${name}_$(basename "$src" .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -DAB_DEV_ABLATION "$@" -c "$src" -o "$obj"
others=$(ls build_dev/*.o | grep -v "build_dev/$(basename "$src" .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libab_${name}.so $obj $others
echo "built astroburst_amd/csrc/build/variants/libab_${name}.so"
