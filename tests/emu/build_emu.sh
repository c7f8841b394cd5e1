#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compiles the unchanged csrc/*.hip with g++ against the SIMT interpreter in
# tests/emu/include (see the header there).  Output: tests/emu/libmcq_emu.so, loaded only by tests.
set -e
cd "$(dirname "$0")"
SRC=../../global_racetrajectory_optimization_amd/csrc
g++ -O2 -std=c++17 -fPIC -shared -x c++ -I include -o libmcq_emu.so $SRC/mcq_kernels.hip $SRC/mcq_api.hip -Wno-unused-result -Wno-attributes
