// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl.so.1 with the five entry points the engine binds (csrc/mcq_api.hip: rccl_load), so that
// the engine's collective path -- mcq_comm_init / mcq_comm_allgather / mcq_comm_wait, the comm stream behind the compute stream, the event
// ring, callers alternating two send buffers -- runs with MORE THAN ONE RANK where no multi-GPU node is at hand ("RCCL has
// never seen > 1 rank").  The ranks are processes on one machine; the "fabric" is a POSIX shared-memory segment named in the 128-byte id.
//
//   default build (hipcc, on the GPU box; tests/test_gpu_gi.py::test_two_ranks_on_one_gpu_through_the_engines_collective): ASYNCHRONOUS like the
//       real library -- ncclAllGather only ENQUEUES on the caller's stream: a device-to-host copy of `send` into this rank's slot of the
//       segment (the mapping is registered with the HIP runtime), a host function that waits for every rank at a barrier, host-to-device
//       copies of every rank's slot into `recv`, a second barrier (nobody overwrites a slot another rank has not read).  Two processes
//       sharing ONE GPU are enough to exercise every ordering the engine relies on.
//   -DSTUB_SYNC (g++, no HIP; the CPU tests on the SIMT-interpreted library, whose "device" memory is host memory and whose streams run
//       inline): the same exchange with memcpy, executed inside the call.
// Loaded through $MCQ_RCCL_LIB.  Nothing in the package refers to this file.
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

#ifndef STUB_SYNC
#include <hip/hip_runtime.h>
typedef hipStream_t stub_stream_t;
#else
typedef void* stub_stream_t;
#endif

struct StubId { char internal[128]; };
struct Seg {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
    int nranks;
    size_t slot_bytes;
};
struct Comm {
    Seg* seg;
    char* data;
    size_t map_bytes;
    int rank, nranks;
    char name[128];
};
static const size_t SLOT_BYTES = (size_t)32 << 20;

static void barrier(Comm* c)
{
    Seg* s = c->seg;
    const int gen = s->generation.load();
    if (s->arrived.fetch_add(1) + 1 == c->nranks) {
        s->arrived.store(0);
        s->generation.fetch_add(1);
        return;
    }
    const time_t t0 = time(nullptr);
    while (s->generation.load() == gen) {
        usleep(50);
        if (time(nullptr) - t0 > 120) { fprintf(stderr, "rccl_stub: rank %d waited 120 s at a barrier -- aborting\n", c->rank); abort(); }
    }
}

extern "C" int ncclGetUniqueId(StubId* id)
{
    static int counter = 0;
    memset(id->internal, 0, sizeof(id->internal));
    snprintf(id->internal, sizeof(id->internal), "/mcq_rccl_stub_%d_%d_%ld", (int)getpid(), counter++, (long)time(nullptr));
    return 0;
}

extern "C" int ncclCommInitRank(void** comm_out, int nranks, StubId id, int rank)
{
    if (!comm_out || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return 4;      // ncclInvalidArgument
    Comm* c = new Comm();
    c->rank = rank;
    c->nranks = nranks;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->map_bytes = 4096 + (size_t)nranks * SLOT_BYTES;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return 2; }                                                                  // ncclSystemError
    if (ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); delete c; return 2; }                       // (zero-filled: counters start at 0)
    void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return 2; }
    c->seg = (Seg*)p;
    c->data = (char*)p + 4096;
#ifndef STUB_SYNC
    if (hipHostRegister(p, c->map_bytes, hipHostRegisterDefault) != hipSuccess) { munmap(p, c->map_bytes); delete c; return 1; }
#endif
    c->seg->attached.fetch_add(1);
    const time_t t0 = time(nullptr);
    while (c->seg->attached.load() < nranks) {           // the real call is collective too
        usleep(100);
        if (time(nullptr) - t0 > 120) { fprintf(stderr, "rccl_stub: rank %d: not every rank attached within 120 s\n", rank); return 2; }
    }
    *comm_out = c;
    return 0;
}

static size_t dtype_size(int t) { return t == 8 ? 8 : (t == 7 ? 4 : (t == 2 ? 4 : 0)); }      // ncclFloat64 / ncclFloat32 / ncclInt32

#ifndef STUB_SYNC
static void barrier_cb(void* p) { barrier((Comm*)p); }
#endif

extern "C" int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, stub_stream_t stream)
{
    Comm* c = (Comm*)comm;
    const size_t es = dtype_size(dtype);
    if (!c || !send || !recv || es == 0) return 4;
    const size_t bytes = count * es;
    for (size_t off = 0; off < bytes; off += SLOT_BYTES) {
        const size_t len = bytes - off < SLOT_BYTES ? bytes - off : SLOT_BYTES;
#ifdef STUB_SYNC
        (void)stream;
        memcpy(c->data + (size_t)c->rank * SLOT_BYTES, (const char*)send + off, len);
        barrier(c);
        for (int r = 0; r < c->nranks; ++r) memcpy((char*)recv + (size_t)r * bytes + off, c->data + (size_t)r * SLOT_BYTES, len);
        barrier(c);
#else
        if (hipMemcpyAsync(c->data + (size_t)c->rank * SLOT_BYTES, (const char*)send + off, len, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
        if (hipLaunchHostFunc(stream, barrier_cb, c) != hipSuccess) return 1;
        for (int r = 0; r < c->nranks; ++r)
            if (hipMemcpyAsync((char*)recv + (size_t)r * bytes + off, c->data + (size_t)r * SLOT_BYTES, len, hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
        if (hipLaunchHostFunc(stream, barrier_cb, c) != hipSuccess) return 1;
#endif
    }
    return 0;
}

extern "C" int ncclCommDestroy(void* comm)
{
    Comm* c = (Comm*)comm;
    if (!c) return 0;
#ifndef STUB_SYNC
    (void)hipHostUnregister(c->seg);
#endif
    munmap(c->seg, c->map_bytes);
    shm_unlink(c->name);          // (the first rank to get here removes the name; the others' mappings stay valid until unmapped)
    delete c;
    return 0;
}

extern "C" const char* ncclGetErrorString(int rc)
{
    return rc == 0 ? "no error" : (rc == 1 ? "unhandled HIP error (stub)" : (rc == 2 ? "unhandled system error (stub)" : "invalid argument (stub)"));
}
