#!/bin/bash
# Builds libhavac_dev.so (HIP kernels + C ABI) for gfx950.  Run from anywhere.
# (written under a temporary name and renamed: a snapshot of the tree never sees a half-written library)
set -e
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/.."
mkdir -p "$here/../../build"
cd "$here/../../build"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -o "$out/libhavac_dev.so.tmp" "$here/havac_dev.hip" "$here/havac_pipe.hip" "$here/havac_gather.hip" -ldl "$@"
mv -f "$out/libhavac_dev.so.tmp" "$out/libhavac_dev.so"
