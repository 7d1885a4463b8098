#!/usr/bin/env bash
# tuning build of the library: tools/build_variant.sh <name> <extra hipcc flags...>  ->  take_amd/variants/lib_<name>.so
# (selected at run time with TAKE_HIP_LIB=...; tools/variants.sh runs bench.py over several of them)
set -e
name="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/take_amd/variants"
cd "$root/take_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wl,-soname,libtake_hip.so -I../../include \
  -Wall -Wno-unused -pthread "$@" -o "../variants/lib_$name.so" tk_api.hip 2>&1 | grep -E "error|spill|Scratch" || true
ls -la "../variants/lib_$name.so"
