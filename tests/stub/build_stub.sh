#!/bin/bash
# TEST INFRASTRUCTURE ONLY: builds tests/stub/librccl_stub.so (hipcc, asynchronous, for the GPU box) and librccl_stub_sync.so (g++, for the
# CPU tests on the SIMT-interpreted library).  See rccl_stub.cpp.
set -e
cd "$(dirname "$0")"
g++ -O2 -std=c++17 -fPIC -shared -DSTUB_SYNC -o librccl_stub_sync.so.tmp rccl_stub.cpp -lrt && mv -f librccl_stub_sync.so.tmp librccl_stub_sync.so
if [ "$1" != "sync-only" ]; then
  ${HIPCC:-/opt/rocm/bin/hipcc} -O2 -std=c++17 -fPIC -shared -x hip --offload-arch=gfx950 -o librccl_stub.so.tmp rccl_stub.cpp -lrt 2>/dev/null && mv -f librccl_stub.so.tmp librccl_stub.so
fi
