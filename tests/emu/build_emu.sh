#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compiles the unchanged csrc/*.hip with g++ against the SIMT interpreter in
# tests/emu/include (see the header there).  Output: tests/emu/libmcq_emu.so, loaded only by tests.
set -e
cd "$(dirname "$0")"
SRC=../../global_racetrajectory_optimization_amd/csrc
F="-O2 -std=c++17 -fPIC -pthread -x c++ -I include -Wno-unused-result -Wno-attributes"
TMP=$(mktemp -d -p .)          # (here, not in /tmp: the final rename must stay inside one file system)
trap 'rm -rf "$TMP"' EXIT
g++ $F -c -o $TMP/kernels.o $SRC/mcq_kernels.hip &
P1=$!
g++ $F -c -o $TMP/api.o $SRC/mcq_api.hip &
P2=$!
wait $P1 || exit 1
wait $P2 || exit 1
g++ -shared -fPIC -pthread -o $TMP/libmcq_emu.so $TMP/kernels.o $TMP/api.o -ldl
mv -f $TMP/libmcq_emu.so libmcq_emu.so      # (atomic: a process that has the old library mapped keeps it)
