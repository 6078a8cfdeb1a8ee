#!/bin/bash
# Builds libhavac.so (the C++ `Havac` API + readers + projection + C wrappers) and the
# havac_benchmark executable.  Needs ../../libhavac_dev.so (run ../build.sh first).
# -ffp-contract=off: the int8 projection must not be contracted into FMAs.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../.."
CXXFLAGS="-O2 -std=c++17 -fPIC -pthread -ffp-contract=off -Wall -Wextra -Wno-unused-parameter"
# (written under a temporary name and renamed: a snapshot of the tree never sees a half-written library)
g++ $CXXFLAGS -shared -o "$out/libhavac.so.tmp" \
    "$here/FastaVector.cpp" "$here/p7HmmReader.cpp" "$here/PhmmReprojection.cpp" \
    "$here/SequencePreprocessor.cpp" "$here/PhmmPreprocessor.cpp" "$here/Havac.cpp" "$here/havac_host_c.cpp" \
    -L"$out" -lhavac_dev -Wl,-rpath,'$ORIGIN'
mv -f "$out/libhavac.so.tmp" "$out/libhavac.so"
g++ $CXXFLAGS -o "$out/havac_benchmark" "$here/havac_benchmark.cpp" -L"$out" -lhavac -lhavac_dev -Wl,-rpath,'$ORIGIN'
