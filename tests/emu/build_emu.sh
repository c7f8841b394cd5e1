#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compiles the unchanged csrc/*.hip with g++ against the SIMT interpreter in
# tests/emu/include (see the header there).  Output: tests/emu/libmcq_emu.so, loaded only by tests.
set -e
cd "$(dirname "$0")"
SRC=../../global_racetrajectory_optimization_amd/csrc
# mcq_kernels.hip twice, like csrc/build.sh: the library's kernels (saddle-point core) and namespace mcq_band (bordered-band core)
F="-O2 -std=c++17 -fPIC -x c++ -I include -Wno-unused-result -Wno-attributes"
g++ $F -c -o /tmp/mcq_emu_kkt.o $SRC/mcq_kernels.hip &
g++ $F -DMCQ_CORE_BAND -c -o /tmp/mcq_emu_band.o $SRC/mcq_kernels.hip &
g++ $F -c -o /tmp/mcq_emu_api.o $SRC/mcq_api.hip &
wait
g++ -shared -fPIC -o libmcq_emu.so /tmp/mcq_emu_kkt.o /tmp/mcq_emu_band.o /tmp/mcq_emu_api.o
