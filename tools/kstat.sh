#!/bin/bash
# Compiles havac_dev.hip with -save-temps into build/asm2 (library -> $1, default build/ab/libB.so) and prints what the
# SSV kernel costs: VGPRs, SGPRs, scratch, occupancy, code size, scratch instructions.   bash tools/kstat.sh [out.so]
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
out="${1:-$root/build/ab/libB.so}"
mkdir -p "$root/build/asm2" "$(dirname "$out")"
cd "$root/build/asm2"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -save-temps -o "$out" "$root/havac_amd/csrc/havac_dev.hip" > compile.log 2>&1 || { grep -B2 -A6 "error" compile.log | head -40; exit 1; }
s=havac_dev-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_ZN5havac15ssv_diag_kernel/ {f=1} f && /^; (NumVgprs|NumSgprs|ScratchSize|Occupancy|codeLenInByte)/ {print} f && /^; Occupancy/ {exit}' $s
echo "scratch instructions in the whole object: $(grep -c 'scratch_' $s || true)"
