#!/bin/bash
# Kernel experiments: tools/build_variant.sh <tag> <source.hip> <object name> <extra hipcc flags...>
# builds ONE object of libpstat.so with extra flags and links polymer_stats_amd/csrc/build/var_<tag>/libpstat.so from it and
# the regular build's other objects (run `make -C polymer_stats_amd/csrc` first).  Select with PSTAT_LIB=<that path>.
set -euo pipefail
tag=$1; src=$2; obj=$3; shift 3
here=$(cd "$(dirname "$0")/../polymer_stats_amd/csrc" && pwd)
out=$here/build/var_$tag
mkdir -p "$out"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function "$@" -c -o "$out/$obj" "$here/$src"
objs=$(ls "$here"/build/*.o | grep -v "/$obj\$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libpstat.so" $objs "$out/$obj"
echo "$out/libpstat.so"
