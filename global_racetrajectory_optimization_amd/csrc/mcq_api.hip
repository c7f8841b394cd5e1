// mcq_api.hip -- host side of libmcq.so: the C ABI declared in include/mcq.h.
//
// The handle owns one HIP stream and the per-batch device workspace ("slabs": band matrices, factor, vectors) on one
// device; it is grown on demand and reused across calls.  One process per GPU creates one handle (bench.py,
// engine.py); a multi-GPU job is N such processes, each solving a contiguous shard of the batch.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <cmath>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "mcq_kernels.h"

static thread_local std::string g_err;

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            char buf_[512];                                                                               \
            snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                     __LINE__);                                                                           \
            g_err = buf_;                                                                                 \
            return MCQ_E_DEVICE;                                                                          \
        }                                                                                                 \
    } while (0)

struct mcq_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool timing_valid = false;
    hipEvent_t ev_st[2] = {nullptr, nullptr};       // mcq_solve_device_stream: joins the second compute stream to the first and back
    hipEvent_t ev_span[2] = {nullptr, nullptr};     // mcq_timing_begin / mcq_timing_end: a span of launches on the compute stream
    int span_launches = 0;
    bool span_open = false;
    // workspace slabs
    size_t cap_elems = 0;   // batch * nmax the slabs are sized for
    size_t cap_batch = 0;
    double *L = nullptr, *vec = nullptr, *Z = nullptr;
    signed char* state = nullptr;
    signed char* state2 = nullptr;     // working sets carried through the IQP glue (warm start of the next pass)
    bool state2_valid = false;
    int state2_batch = 0, state2_nmax = 0;   // the layout mcq_relinearise_device wrote state2 with: a warm start is honoured only for it
    // staging for the host-buffer entry point
    double *d_ref = nullptr, *d_nv = nullptr, *d_sc = nullptr, *d_alpha = nullptr, *d_curv = nullptr, *d_kb = nullptr,
           *d_wv = nullptr;
    int *d_n = nullptr, *d_status = nullptr;
    mcq_info* d_info = nullptr;
    size_t stage_elems = 0, stage_batch = 0;
    double *d_ref2 = nullptr, *d_nv2 = nullptr;      // second set of the IQP double buffer (mcq_iqp_batch)
    size_t stage2_elems = 0;
    int* d_iqp = nullptr;                             // bookkeeping arrays of mcq_iqp_device, 8 ints per track + 1
    double* d_iqp_curv = nullptr;
    size_t iqp_batch = 0;
    void* pin = nullptr;                              // pinned host staging (packing of the host-buffer entries)
    size_t pin_bytes = 0;
    bool poison = false;    // MCQ_POISON=1 (debugging aid): every workspace / staging allocation is filled with 0xFF bytes (NaN as a
                            // double), so that a read of memory nothing has written shows up as a wrong result on every box instead of
                            // on the rare one whose recycled memory happens to hold garbage
    long long ws_bytes = 0;
    bool smem_attr_set = false;
    double* vel_scratch = nullptr;      // lap-doubled profiles of mcq_vel_profile_device, [2 nmax][batch]
    mcq_iqp_round_cb iqp_cb = nullptr;  // mcq_iqp_set_round_callback
    void* iqp_cb_user = nullptr;
    void* comm = nullptr;               // ncclComm_t of mcq_comm_init (RCCL, loaded with dlopen)
    int comm_rank = 0, comm_world = 0;
    hipStream_t comm_stream = nullptr;  // the gathers run here: ordered behind the solves by an event, overlapping the NEXT solve
    hipEvent_t comm_ready = nullptr, comm_t0[4] = {nullptr, nullptr, nullptr, nullptr}, comm_done[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned comm_seq = 0;              // gathers enqueued so far (event ring index)
    size_t vel_scratch_bytes = 0;
    double* kbig = nullptr;             // overflow slots of the curvature-row working set (MCQ_KBIG_SLOTS x MCQ_KBIG_SLOT doubles)
    int* slot_flags = nullptr;          // [MCQ_SLOT_FLAGS] = overflow slots | full Goldfarb-Idnani slots | small ones: 0 free / 1 taken, claimed and released by the workgroups (never reset by the host)
    double* gi = nullptr;               // FULL slots of the Goldfarb-Idnani path (mcq_gi.inc): gi_slots x MCQ_GI_SLOT_DOUBLES(gi_nmax, gi_nmax)
    int gi_slots = 0, gi_nmax = 0;
    long long gi_bytes = 0;
    size_t gi_none_nmax = 0;            // > 0: no full slot could be had for rings of this many waypoints (beyond the byte cap, or hipMalloc said no) -- not tried again
    double* gis = nullptr;              // SMALL slots (MCQ_ALG_GI: one per resident workgroup): gis_slots x MCQ_GI_SLOT_DOUBLES(gis_nmax, gi_small_qcap(gis_nmax))
    int gis_slots = 0, gis_nmax = 0;
    long long gis_bytes = 0;
    // mcq_solve_host_pipelined: a SECOND compute stream with a workspace of its own -- the kernels of consecutive steps run on alternating
    // streams, so the tail of one launch (its slowest problems, on CUs the others have left) overlaps the start of the next (round 5)
    hipStream_t stream2 = nullptr;
    double *L2 = nullptr, *vec2 = nullptr, *Z2 = nullptr, *kbig2 = nullptr, *gi2 = nullptr;
    signed char* state_alt = nullptr;
    int* slot_flags2 = nullptr;
    size_t alt_elems = 0, alt_batch = 0;
    int gi2_slots = 0, gi2_nmax = 0;
    long long alt_bytes = 0;
    double* d_org = nullptr;            // per-track origins of the fp32 row entries, [batch][2]
    double* d_trace = nullptr;          // curvature-error trace of mcq_iqp_batch, [batch][MCQ_IQP_TRACE] (grown with d_iqp)
    // mcq_solve_host_pipelined: two copy streams, the second set of staging buffers, one event triple per slot
    hipStream_t cs_in = nullptr, cs_out = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_down[2] = {nullptr, nullptr};
    hipEvent_t ev_slice[16] = {};      // mcq_solve_host / mcq_solve_batch in slices: upload / kernel done, per slice
    int last_upload_direct = 0;         // the last host-buffer batch went up without the packing pass (mcq_last_upload_was_direct)
    double *p_ref = nullptr, *p_nv = nullptr, *p_sc = nullptr, *p_alpha = nullptr, *p_curv = nullptr;
    int* p_status = nullptr;
    size_t pipe_elems = 0, pipe_batch = 0;
};

extern "C" const char* mcq_last_error(void) { return g_err.c_str(); }

extern "C" void mcq_default_opts(mcq_opts* o)
{
    if (!o) return;
    o->algorithm = MCQ_ALG_DEFAULT;
    o->max_ipm_iter = 60;
    o->max_as_iter = 60;
    o->refine_steps = 2;
    o->check_kappa = 1;
    o->objective = MCQ_OBJ_MIN_CURV;
    o->warm_start = 0;
}

static mcq_opts resolve_opts(const mcq_opts* in)
{
    mcq_opts o;
    mcq_default_opts(&o);
    if (in) {
        o.algorithm = in->algorithm == MCQ_ALG_GI ? MCQ_ALG_GI : MCQ_ALG_DEFAULT;    // (the field was `band_e` until round 4: 0 / 32 mean the default)
        if (in->max_ipm_iter > 0) o.max_ipm_iter = in->max_ipm_iter;
        if (in->max_as_iter > 0) o.max_as_iter = in->max_as_iter;
        if (in->refine_steps >= 0) o.refine_steps = in->refine_steps;
        o.check_kappa = in->check_kappa >= 0 ? 1 : 0;   // 0 (a zero-initialised struct) and 1: carried; < 0: skipped
        o.objective = in->objective == MCQ_OBJ_SHORTEST_PATH ? MCQ_OBJ_SHORTEST_PATH : MCQ_OBJ_MIN_CURV;
        o.warm_start = in->warm_start;
    }
    return o;
}

extern "C" void mcq_destroy(mcq_handle* h);
extern "C" int mcq_comm_destroy(mcq_handle* h);

extern "C" int mcq_create(int device_id, mcq_handle** out)
{
    if (!out) { g_err = "mcq_create: out is NULL"; return MCQ_E_ARG; }
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) { g_err = "mcq_create: no such device"; return MCQ_E_DEVICE; }
    HIP_TRY(hipSetDevice(device_id));
    mcq_handle* h = new mcq_handle();
    h->device = device_id;
    {
        const char* e = getenv("MCQ_POISON");
        h->poison = e && e[0] == '1';
    }
    hipError_t e = hipStreamCreate(&h->stream);
    for (int k = 0; k < 5 && e == hipSuccess; ++k) e = hipEventCreate(&h->ev[k]);
    if (e != hipSuccess) {
        g_err = std::string("mcq_create: ") + hipGetErrorString(e);
        mcq_destroy(h);         // releases whatever was created
        return MCQ_E_DEVICE;
    }
    *out = h;
    return 0;
}

static void free_ws(mcq_handle* h)
{
    (void)hipFree(h->L); (void)hipFree(h->vec); (void)hipFree(h->Z); (void)hipFree(h->state);
    (void)hipFree(h->state2);
    (void)hipFree(h->kbig); (void)hipFree(h->slot_flags); (void)hipFree(h->gi); (void)hipFree(h->gis);
    h->kbig = nullptr; h->slot_flags = nullptr; h->gi = nullptr; h->gis = nullptr;
    h->gi_slots = h->gi_nmax = 0;
    h->gis_slots = h->gis_nmax = 0;
    h->gi_bytes = h->gis_bytes = 0;
    h->L = h->vec = h->Z = nullptr;
    h->state = h->state2 = nullptr;
    h->state2_valid = false;
    h->cap_elems = h->cap_batch = 0;
}

static void free_alt(mcq_handle* h)
{
    (void)hipFree(h->L2); (void)hipFree(h->vec2); (void)hipFree(h->Z2); (void)hipFree(h->state_alt); (void)hipFree(h->kbig2);
    (void)hipFree(h->slot_flags2); (void)hipFree(h->gi2);
    h->L2 = h->vec2 = h->Z2 = h->kbig2 = h->gi2 = nullptr;
    h->state_alt = nullptr;
    h->slot_flags2 = nullptr;
    h->alt_elems = h->alt_batch = 0;
    h->gi2_slots = h->gi2_nmax = 0;
    h->alt_bytes = 0;
}

static void free_stage(mcq_handle* h)
{
    (void)hipFree(h->d_ref); (void)hipFree(h->d_nv); (void)hipFree(h->d_sc); (void)hipFree(h->d_alpha); (void)hipFree(h->d_curv); (void)hipFree(h->d_kb);
    (void)hipFree(h->d_wv); (void)hipFree(h->d_n); (void)hipFree(h->d_status); (void)hipFree(h->d_info);
    h->d_ref = h->d_nv = h->d_sc = h->d_alpha = h->d_curv = h->d_kb = h->d_wv = nullptr;
    h->d_n = h->d_status = nullptr;
    h->d_info = nullptr;
    h->stage_elems = h->stage_batch = 0;
    (void)hipFree(h->d_ref2); (void)hipFree(h->d_nv2); (void)hipFree(h->d_iqp); (void)hipFree(h->d_iqp_curv);
    h->d_ref2 = h->d_nv2 = h->d_iqp_curv = nullptr;
    h->d_iqp = nullptr;
    h->stage2_elems = h->iqp_batch = 0;
    (void)hipFree(h->d_org); (void)hipFree(h->d_trace);
    h->d_org = h->d_trace = nullptr;
}

static void free_pipe(mcq_handle* h)
{
    (void)hipFree(h->p_ref); (void)hipFree(h->p_nv); (void)hipFree(h->p_sc); (void)hipFree(h->p_alpha); (void)hipFree(h->p_curv); (void)hipFree(h->p_status);
    h->p_ref = h->p_nv = h->p_sc = h->p_alpha = h->p_curv = nullptr;
    h->p_status = nullptr;
    h->pipe_elems = h->pipe_batch = 0;
}

extern "C" void mcq_destroy(mcq_handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->cs_in) (void)hipStreamSynchronize(h->cs_in);
    if (h->cs_out) (void)hipStreamSynchronize(h->cs_out);
    (void)mcq_comm_destroy(h);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    free_ws(h);
    free_alt(h);
    free_stage(h);
    free_pipe(h);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    for (int k = 0; k < 2; ++k) {
        if (h->ev_up[k]) (void)hipEventDestroy(h->ev_up[k]);
        if (h->ev_done[k]) (void)hipEventDestroy(h->ev_done[k]);
        if (h->ev_down[k]) (void)hipEventDestroy(h->ev_down[k]);
    }
    for (int k = 0; k < 16; ++k) if (h->ev_slice[k]) (void)hipEventDestroy(h->ev_slice[k]);
    if (h->cs_in) (void)hipStreamDestroy(h->cs_in);
    if (h->cs_out) (void)hipStreamDestroy(h->cs_out);
    (void)hipFree(h->vel_scratch);
    if (h->pin) (void)hipHostFree(h->pin);
    for (int k = 0; k < 5; ++k) if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
    for (int k = 0; k < 2; ++k) if (h->ev_span[k]) (void)hipEventDestroy(h->ev_span[k]);
    for (int k = 0; k < 2; ++k) if (h->ev_st[k]) (void)hipEventDestroy(h->ev_st[k]);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// Goldfarb-Idnani slots (mcq_gi.inc).  Two pools:
//   FULL slots  -- a working set holds at most nmax independent constraints: nmax x nmax (Q) + nmax x nmax (R) doubles (64 MB at nmax = 2000).  The
//                  fallback's pool: as many as $MCQ_GI_BYTES (16 GB) hold, at most MCQ_GI_FULL_MAX = 512 (256 at nmax = 2000: a sweep over tight curvature
//                  bounds can send hundreds of problems of one launch down this path), never more than the batch.  A rare path must not cost the
//                  common one its launch: a slot beyond the byte cap (rings above ~32 000 waypoints) or a refused hipMalloc leaves the handle
//                  WITHOUT the pool -- the kernel then returns what its own phases left (MCQ_ITER_CAP, ...), as in round 4; not tried again for
//                  rings that long.
//   SMALL slots -- round 6, mcq_opts.algorithm = MCQ_ALG_GI (every problem takes the path: one slot per resident workgroup): working sets of up to
//                  gi_small_qcap(nmax) = nmax / 8 constraints (rounded up to 64, at least 128), what the path's working sets measure on every workload
//                  here (62 .. 100 active rows at nmax = 2000); 4.6 MB at nmax = 2000, 2.4 GB for 512 of them (round 5: full slots,
//                  33 GB).  A problem that outgrows its small slot moves into a full one in place (gi_grow, mcq_gi.inc).
static size_t gi_small_qcap(size_t nmax)
{
    const size_t q = std::max<size_t>(128, ((nmax / 8 + 63) / 64) * 64);
    return std::min(q, nmax);
}

static size_t gi_byte_cap()
{
    if (const char* e = getenv("MCQ_GI_BYTES")) { const long long v = atoll(e); if (v >= 0) return (size_t)v; }
    return (size_t)16 << 30;       // (288 GB of HBM: round 6 measured the 600-large-ring stress at 75 s with 4 GB of full slots and 44 s with 48 GB)
}

// hipMalloc of `slots` items of `per` bytes, halving the count until the allocation succeeds; returns the count had (0: none)
static int alloc_slots(double** out, int slots, size_t per)
{
    *out = nullptr;
    while (slots >= 1) {
        if (hipMalloc((void**)out, (size_t)slots * per) == hipSuccess) return slots;
        (void)hipGetLastError();        // (an out-of-memory is not sticky, but it stays in the last-error slot)
        *out = nullptr;
        slots /= 2;
    }
    return 0;
}

static void ensure_gi(mcq_handle* h, size_t batch, size_t nmax)
{
    if (h->gi_none_nmax && nmax >= h->gi_none_nmax) return;
    const size_t cap_bytes = gi_byte_cap();
    int want = MCQ_GI_FULL_MAX;
    if (const char* e = getenv("MCQ_GI_SLOTS")) want = std::min(std::max(atoi(e), 1), MCQ_GI_FULL_MAX);
    const size_t per = MCQ_GI_SLOT_DOUBLES(nmax, nmax) * sizeof(double);
    int slots = (int)std::min<size_t>((size_t)want, cap_bytes / per);
    slots = (int)std::min<size_t>((size_t)slots, std::max<size_t>(batch, 1));
    if (h->gi && (size_t)h->gi_nmax >= nmax && h->gi_slots >= slots) return;
    nmax = std::max(nmax, (size_t)h->gi_nmax);
    const size_t per2 = MCQ_GI_SLOT_DOUBLES(nmax, nmax) * sizeof(double);
    slots = std::max(slots, h->gi_slots);
    slots = (int)std::min<size_t>((size_t)slots, cap_bytes / per2);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->gi);
    h->gi = nullptr;
    h->gi_slots = h->gi_nmax = 0;
    h->gi_bytes = 0;
    slots = alloc_slots(&h->gi, slots, per2);
    if (slots < 1) { h->gi_none_nmax = nmax; return; }
    if (h->poison) (void)hipMemsetAsync(h->gi, 0xff, (size_t)slots * per2, h->stream);
    h->gi_slots = slots;
    h->gi_nmax = (int)nmax;
    h->gi_bytes = (long long)((size_t)slots * per2);
}

// the small pool of MCQ_ALG_GI; an error only if the handle ends up with no slot of either kind (the caller asked for this path by name)
static int ensure_gi_small(mcq_handle* h, size_t batch, size_t nmax)
{
    int slots = (int)std::min<size_t>(std::max<size_t>(batch, 1), MCQ_GI_SLOTS_MAX);
    if (!(h->gis && (size_t)h->gis_nmax >= nmax && h->gis_slots >= slots)) {
        nmax = std::max(nmax, (size_t)h->gis_nmax);
        slots = std::max(slots, h->gis_slots);
        const size_t per = MCQ_GI_SLOT_DOUBLES(nmax, gi_small_qcap(nmax)) * sizeof(double);
        const size_t cap_bytes = std::max(gi_byte_cap(), (size_t)8 << 30);
        slots = (int)std::min<size_t>((size_t)slots, cap_bytes / per);
        (void)hipStreamSynchronize(h->stream);
        (void)hipFree(h->gis);
        h->gis = nullptr;
        h->gis_slots = h->gis_nmax = 0;
        h->gis_bytes = 0;
        slots = alloc_slots(&h->gis, slots, per);
        if (slots >= 1) {
            if (h->poison) (void)hipMemsetAsync(h->gis, 0xff, (size_t)slots * per, h->stream);
            h->gis_slots = slots;
            h->gis_nmax = (int)nmax;
            h->gis_bytes = (long long)((size_t)slots * per);
        }
    }
    if (!h->gis && !h->gi) { g_err = "MCQ_ALG_GI: no device memory for a Goldfarb-Idnani slot at this ring size"; return MCQ_E_DEVICE; }
    return 0;
}

static int ensure_ws(mcq_handle* h, size_t batch, size_t nmax)
{
    const size_t elems = batch * nmax;
    if (elems <= h->cap_elems && batch <= h->cap_batch) { ensure_gi(h, batch, nmax); return 0; }
    HIP_TRY(hipStreamSynchronize(h->stream));
    free_ws(h);
    HIP_TRY(hipMalloc((void**)&h->L, elems * MCQ_LLD * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->vec, elems * MCQ_NVEC * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->Z, (elems + batch * (size_t)MCQ_KMAX * MCQ_KMAX) * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->state, elems));
    HIP_TRY(hipMalloc((void**)&h->state2, elems));
    HIP_TRY(hipMalloc((void**)&h->kbig, (size_t)MCQ_KBIG_SLOTS * MCQ_KBIG_SLOT * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->slot_flags, MCQ_SLOT_FLAGS * sizeof(int)));
    HIP_TRY(hipMemsetAsync(h->slot_flags, 0, MCQ_SLOT_FLAGS * sizeof(int), h->stream));
    ensure_gi(h, batch, nmax);
    if (h->poison) HIP_TRY(hipMemsetAsync(h->kbig, 0xff, (size_t)MCQ_KBIG_SLOTS * MCQ_KBIG_SLOT * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(h->state, 0, elems, h->stream));
    HIP_TRY(hipMemsetAsync(h->state2, 0, elems, h->stream));
    if (h->poison) {
        HIP_TRY(hipMemsetAsync(h->L, 0xff, elems * MCQ_LLD * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(h->vec, 0xff, elems * MCQ_NVEC * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(h->Z, 0xff, (elems + batch * (size_t)MCQ_KMAX * MCQ_KMAX) * sizeof(double), h->stream));
    }
    h->cap_elems = elems;
    h->cap_batch = batch;
    h->ws_bytes = (long long)(elems * ((MCQ_LLD + MCQ_NVEC + 1) * sizeof(double) + 2) +
                              batch * (size_t)MCQ_KMAX * MCQ_KMAX * sizeof(double) +
                              (size_t)MCQ_KBIG_SLOTS * MCQ_KBIG_SLOT * sizeof(double));
    return 0;
}

// the second workspace of the pipelined host entry (same slabs as ensure_ws, half the Goldfarb-Idnani slots)
static int ensure_alt(mcq_handle* h, size_t batch, size_t nmax)
{
    if (!h->stream2) HIP_TRY(hipStreamCreate(&h->stream2));
    const size_t elems = batch * nmax;
    if (elems <= h->alt_elems && batch <= h->alt_batch && (size_t)h->gi2_nmax >= nmax) return 0;
    HIP_TRY(hipStreamSynchronize(h->stream2));
    free_alt(h);
    HIP_TRY(hipMalloc((void**)&h->L2, elems * MCQ_LLD * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->vec2, elems * MCQ_NVEC * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->Z2, (elems + batch * (size_t)MCQ_KMAX * MCQ_KMAX) * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->state_alt, elems));
    HIP_TRY(hipMalloc((void**)&h->kbig2, (size_t)MCQ_KBIG_SLOTS * MCQ_KBIG_SLOT * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->slot_flags2, MCQ_SLOT_FLAGS * sizeof(int)));
    HIP_TRY(hipMemsetAsync(h->slot_flags2, 0, MCQ_SLOT_FLAGS * sizeof(int), h->stream2));
    const size_t per = MCQ_GI_SLOT_DOUBLES(nmax, nmax) * sizeof(double);
    // half the first workspace's full slots (none where it has none: the kernel then returns what its own phases left)
    int slots = h->gi && (size_t)h->gi_nmax >= nmax ? alloc_slots(&h->gi2, std::max(1, h->gi_slots / 2), per) : 0;
    HIP_TRY(hipMemsetAsync(h->state_alt, 0, elems, h->stream2));
    if (h->poison) {
        HIP_TRY(hipMemsetAsync(h->L2, 0xff, elems * MCQ_LLD * sizeof(double), h->stream2));
        HIP_TRY(hipMemsetAsync(h->vec2, 0xff, elems * MCQ_NVEC * sizeof(double), h->stream2));
        HIP_TRY(hipMemsetAsync(h->Z2, 0xff, (elems + batch * (size_t)MCQ_KMAX * MCQ_KMAX) * sizeof(double), h->stream2));
        HIP_TRY(hipMemsetAsync(h->kbig2, 0xff, (size_t)MCQ_KBIG_SLOTS * MCQ_KBIG_SLOT * sizeof(double), h->stream2));
        if (h->gi2) HIP_TRY(hipMemsetAsync(h->gi2, 0xff, (size_t)slots * per, h->stream2));
    }
    h->alt_elems = elems;
    h->alt_batch = batch;
    h->gi2_slots = slots;
    h->gi2_nmax = (int)nmax;      // (the ring size this workspace was sized for, with or without slots)
    h->alt_bytes = (long long)(elems * ((MCQ_LLD + MCQ_NVEC + 1) * sizeof(double) + 1) + batch * (size_t)MCQ_KMAX * MCQ_KMAX * sizeof(double) +
                               (size_t)MCQ_KBIG_SLOTS * MCQ_KBIG_SLOT * sizeof(double) + (size_t)slots * per);
    return 0;
}

static int ensure_stage(mcq_handle* h, size_t batch, size_t nmax)
{
    const size_t elems = batch * nmax;
    if (elems <= h->stage_elems && batch <= h->stage_batch) return 0;
    HIP_TRY(hipStreamSynchronize(h->stream));
    free_stage(h);
    HIP_TRY(hipMalloc((void**)&h->d_ref, elems * 4 * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->d_nv, elems * 2 * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->d_sc, elems * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->d_alpha, elems * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->d_curv, batch * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->d_kb, batch * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->d_wv, batch * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->d_n, batch * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&h->d_status, batch * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&h->d_info, batch * sizeof(mcq_info)));
    HIP_TRY(hipMalloc((void**)&h->d_org, batch * 2 * sizeof(double)));
    if (h->poison) {
        HIP_TRY(hipMemsetAsync(h->d_ref, 0xff, elems * 4 * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(h->d_nv, 0xff, elems * 2 * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(h->d_sc, 0xff, elems * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(h->d_alpha, 0xff, elems * sizeof(double), h->stream));
    }
    h->stage_elems = elems;
    h->stage_batch = batch;
    return 0;
}

// alt: on the handle's second compute stream and workspace (ensure_alt; mcq_solve_host_pipelined's odd steps) -- no event timing there
// workspace, options and slots of a solver launch (everything of McqBatch the caller does not set)
static int fill_batch(mcq_handle* h, McqBatch& B, const mcq_opts& o, bool alt)
{
    B.L = alt ? h->L2 : h->L; B.vec = alt ? h->vec2 : h->vec; B.Z = alt ? h->Z2 : h->Z; B.state = alt ? h->state_alt : h->state;
    B.max_ipm_iter = o.max_ipm_iter;
    B.max_as_iter = o.max_as_iter;
    B.refine_steps = o.refine_steps;
    B.check_kappa = o.check_kappa;
    B.objective = o.objective;
    B.poison_lds = h->poison ? 1 : 0;
    B.kbig = alt ? h->kbig2 : h->kbig;
    B.slot_flags = alt ? h->slot_flags2 : h->slot_flags;      // (zero between launches: the workgroups release what they claim)
    B.kbig_slots = MCQ_KBIG_SLOTS;
    B.algorithm = o.algorithm;
    // the Goldfarb-Idnani slots hold working sets of up to gi_nmax constraints on rings of up to gi_nmax waypoints (slot stride: B.nmax)
    const bool gi_path = o.objective == MCQ_OBJ_MIN_CURV && !B.prep_only;
    B.gis = nullptr;
    B.gis_slots = B.gis_qcap = 0;
    if (!alt && o.algorithm == MCQ_ALG_GI && gi_path) {
        // every problem takes the Goldfarb-Idnani path: a (small) slot per resident workgroup, so that they run side by side
        if (int rc = ensure_gi_small(h, (size_t)B.batch, (size_t)B.nmax)) return rc;
        if (h->gis && B.nmax <= h->gis_nmax) {
            B.gis = h->gis;
            B.gis_slots = h->gis_slots;
            B.gis_qcap = (int)gi_small_qcap((size_t)B.nmax);
        }
    }
    double* gi_mem = alt ? h->gi2 : h->gi;
    const bool gi_on = gi_mem && (alt ? h->gi2_slots : h->gi_slots) > 0 && B.nmax <= (alt ? h->gi2_nmax : h->gi_nmax) && gi_path;
    B.gi = gi_on ? gi_mem : nullptr;
    B.gi_slots = gi_on ? (alt ? h->gi2_slots : h->gi_slots) : 0;
    B.gi_qcap = B.nmax;
    // The overflow slots of the curvature-row working set (rounds 3-4) serve only where no Goldfarb-Idnani slot exists: which of more than eight
    // such problems of a launch got one depended on the order the GPU scheduled workgroups in, and so did the last bits of their results
    // (ADVICE r5).  With a Goldfarb-Idnani pool every working set beyond MCQ_KMAX rows takes THAT path -- one route per problem, whatever else
    // is in the launch.
    if (B.gi || B.gis) B.kbig_slots = 0;
    return 0;
}

static int launch(mcq_handle* h, McqBatch& B, const mcq_opts& o, bool alt = false)
{
    hipStream_t st = alt ? h->stream2 : h->stream;
    if (int rc = fill_batch(h, B, o, alt)) return rc;
    // warm start: only the working sets mcq_relinearise_device carried over for exactly this batch layout
    B.warm = (!alt && o.warm_start > 0 && h->state2_valid && !B.prep_only && h->state2_batch == B.batch && h->state2_nmax == B.nmax)
                 ? h->state2 : nullptr;
    if (!B.prep_only && !alt) h->state2_valid = false;
    if (B.objective == MCQ_OBJ_SHORTEST_PATH && !B.prep_only) {
        if (!B.nv) { g_err = "shortest-path objective: normvec is required"; return MCQ_E_ARG; }
        B.check_kappa = 0;
        if (!alt) HIP_TRY(hipEventRecord(h->ev[0], st));
        hipLaunchKernelGGL(mcq_assemble_sp_kernel, dim3(B.batch), dim3(256), 0, st, B);
        HIP_TRY(hipGetLastError());
        if (!alt) HIP_TRY(hipEventRecord(h->ev[1], st));
        if (!alt) HIP_TRY(hipEventRecord(h->ev[2], st));
        hipLaunchKernelGGL(mcq_solve_kernel, dim3(B.batch), dim3(256), 0, st, B);
        HIP_TRY(hipGetLastError());
        if (!alt) HIP_TRY(hipEventRecord(h->ev[3], st));
        if (!alt) HIP_TRY(hipEventRecord(h->ev[4], st));
        if (!alt) h->timing_valid = true;
        return 0;
    }
    if (!alt) HIP_TRY(hipEventRecord(h->ev[0], st));
    if (B.prep_only) {
        hipLaunchKernelGGL(mcq_assemble_kernel, dim3(B.batch), dim3(256), 0, st, B);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // (no assembly launch: the solver kernel assembles its problem itself -- assemble_problem() -- since round 4)
    if (!alt) HIP_TRY(hipEventRecord(h->ev[1], st));
    if (!alt) HIP_TRY(hipEventRecord(h->ev[2], st));
    hipLaunchKernelGGL(mcq_solve_kernel, dim3(B.batch), dim3(256), 0, st, B);
    HIP_TRY(hipGetLastError());
    if (!alt) HIP_TRY(hipEventRecord(h->ev[3], st));
    if (!alt) HIP_TRY(hipEventRecord(h->ev[4], st));
    if (!alt) h->timing_valid = true;
    if (!alt && h->span_open) ++h->span_launches;
    return 0;
}

extern "C" int mcq_solve_device(mcq_handle* h, int batch, int n, const double* reftrack, const double* normvec,
                                const double* scaling, double kappa_bound, double w_veh, const mcq_opts* opts,
                                double* alpha_out, double* curv_err_out, int* status_out, mcq_info* info_out)
{
    if (!h || batch <= 0 || n <= 0 || !reftrack || !alpha_out || !curv_err_out || !status_out) {
        g_err = "mcq_solve_device: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    const mcq_opts o = resolve_opts(opts);
    int rc = ensure_ws(h, (size_t)batch, (size_t)n);
    if (rc) return rc;
    McqBatch B;
    memset(&B, 0, sizeof(B));
    B.batch = batch;
    B.n = n;
    B.nmax = n;
    B.n_list = nullptr;
    B.ref = reftrack;
    B.nv = normvec;
    B.sc = scaling;
    B.alpha = alpha_out;
    B.curv_err = curv_err_out;
    B.status = status_out;
    B.info = info_out;
    B.kappa_bound = kappa_bound;
    B.w_veh = w_veh;
    return launch(h, B, o);
}

static int convert_launch(mcq_handle* h, const float* src, double* dst, size_t count)
{
    const unsigned blocks = (unsigned)((count / 4 + 255) / 256 < 4096 ? (count / 4 + 255) / 256 + 1 : 4096);
    hipLaunchKernelGGL(mcq_widen_kernel, dim3(blocks), dim3(256), 0, h->stream, src, dst, count);
    HIP_TRY(hipGetLastError());
    return 0;
}

// fp32 at the boundary, fp64 inside (BASELINE config 5): the float rows are widened into the handle's staging buffers, the
// fp64 engine runs unchanged, alpha is narrowed on the way out.  The conversions stream 28 + 12 bytes per waypoint next
// to the ~93 KB per waypoint the solver moves: not measurable.
extern "C" int mcq_solve_device_f32(mcq_handle* h, int batch, int n, const float* reftrack, const float* normvec,
                                    const float* scaling, double kappa_bound, double w_veh, const mcq_opts* opts,
                                    float* alpha_out, double* curv_err_out, int* status_out, mcq_info* info_out)
{
    if (!h || batch <= 0 || n <= 0 || !reftrack || !alpha_out || !curv_err_out || !status_out) {
        g_err = "mcq_solve_device_f32: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    const mcq_opts o = resolve_opts(opts);
    int rc = ensure_ws(h, (size_t)batch, (size_t)n);
    if (rc) return rc;
    rc = ensure_stage(h, (size_t)batch, (size_t)n);
    if (rc) return rc;
    const size_t elems = (size_t)batch * n;
    rc = convert_launch(h, reftrack, h->d_ref, elems * 4);
    if (rc) return rc;
    if (normvec) { rc = convert_launch(h, normvec, h->d_nv, elems * 2); if (rc) return rc; }
    if (scaling) { rc = convert_launch(h, scaling, h->d_sc, elems); if (rc) return rc; }
    McqBatch B;
    memset(&B, 0, sizeof(B));
    B.batch = batch;
    B.n = n;
    B.nmax = n;
    B.ref = h->d_ref;
    B.nv = normvec ? h->d_nv : nullptr;
    B.sc = scaling ? h->d_sc : nullptr;
    B.alpha = h->d_alpha;
    B.curv_err = curv_err_out;
    B.status = status_out;
    B.info = info_out;
    B.kappa_bound = kappa_bound;
    B.w_veh = w_veh;
    rc = launch(h, B, o);
    if (rc) return rc;
    const unsigned blocks = (unsigned)((elems / 4 + 255) / 256 < 4096 ? (elems / 4 + 255) / 256 + 1 : 4096);
    hipLaunchKernelGGL(mcq_narrow_kernel, dim3(blocks), dim3(256), 0, h->stream, (const double*)h->d_alpha, alpha_out, elems);
    HIP_TRY(hipGetLastError());
    return 0;
}

// fp32 rows in either layout (include/mcq.h: MCQ_F32_ABSOLUTE / MCQ_F32_INCREMENTS), device-resident: the rows are rebuilt as fp64
// [x, y, w_r, w_l] in the handle's staging buffer (mcq_widen_rows_kernel: running sums in fp64, closure defect spread over the
// ring), normals and scalings are derived on the device, alpha is narrowed on the way out.
static int solve_f32_rows(mcq_handle* h, int batch, int n, int layout, const float* d_rows, const double* d_origin, double kappa_bound,
                          double w_veh, const mcq_opts& o, float* d_alpha_out, double* curv_err_out, int* status_out, mcq_info* info_out)
{
    const size_t elems = (size_t)batch * n;
    hipLaunchKernelGGL(mcq_widen_rows_kernel, dim3((unsigned)batch), dim3(64), 0, h->stream, d_rows, d_origin, h->d_ref, n, layout);
    HIP_TRY(hipGetLastError());
    McqBatch B;
    memset(&B, 0, sizeof(B));
    B.batch = batch;
    B.n = n;
    B.nmax = n;
    B.ref = h->d_ref;
    B.alpha = h->d_alpha;
    B.curv_err = curv_err_out;
    B.status = status_out;
    B.info = info_out;
    B.kappa_bound = kappa_bound;
    B.w_veh = w_veh;
    int rc = launch(h, B, o);
    if (rc) return rc;
    const unsigned blocks = (unsigned)((elems / 4 + 255) / 256 < 4096 ? (elems / 4 + 255) / 256 + 1 : 4096);
    hipLaunchKernelGGL(mcq_narrow_kernel, dim3(blocks), dim3(256), 0, h->stream, (const double*)h->d_alpha, d_alpha_out, elems);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mcq_solve_device_f32_rows(mcq_handle* h, int batch, int n, int layout, const float* reftrack, const double* origin,
                                         double kappa_bound, double w_veh, const mcq_opts* opts, float* alpha_out,
                                         double* curv_err_out, int* status_out, mcq_info* info_out)
{
    if (!h || batch <= 0 || n <= 0 || !reftrack || !alpha_out || !curv_err_out || !status_out ||
        (layout != MCQ_F32_ABSOLUTE && layout != MCQ_F32_INCREMENTS)) {
        g_err = "mcq_solve_device_f32_rows: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    const mcq_opts o = resolve_opts(opts);
    int rc = ensure_ws(h, (size_t)batch, (size_t)n);
    if (rc) return rc;
    rc = ensure_stage(h, (size_t)batch, (size_t)n);
    if (rc) return rc;
    return solve_f32_rows(h, batch, n, layout, reftrack, origin, kappa_bound, w_veh, o, alpha_out, curv_err_out, status_out, info_out);
}

extern "C" int mcq_solve_device_ragged(mcq_handle* h, int batch, int nmax, const int* n_list, const double* reftrack,
                                       const double* normvec, const double* scaling, double kappa_bound, double w_veh,
                                       const mcq_opts* opts, double* alpha_out, double* curv_err_out, int* status_out,
                                       mcq_info* info_out)
{
    if (!h || batch <= 0 || nmax <= 0 || !n_list || !reftrack || !alpha_out || !curv_err_out || !status_out) {
        g_err = "mcq_solve_device_ragged: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    const mcq_opts o = resolve_opts(opts);
    int rc = ensure_ws(h, (size_t)batch, (size_t)nmax);
    if (rc) return rc;
    McqBatch B;
    memset(&B, 0, sizeof(B));
    B.batch = batch;
    B.n = nmax;
    B.nmax = nmax;
    B.n_list = n_list;
    B.ref = reftrack;
    B.nv = normvec;
    B.sc = scaling;
    B.alpha = alpha_out;
    B.curv_err = curv_err_out;
    B.status = status_out;
    B.info = info_out;
    B.kappa_bound = kappa_bound;
    B.w_veh = w_veh;
    return launch(h, B, o);
}

// The ragged device entry with per-problem vehicle parameters (device arrays): the shape of a vehicle-width sweep whose
// tracks are already resident (BASELINE config 4).  Either list may be NULL (the scalar applies to every problem).
extern "C" int mcq_solve_device_ragged_params(mcq_handle* h, int batch, int nmax, const int* n_list, const double* reftrack,
                                              const double* normvec, const double* scaling, double kappa_bound, double w_veh,
                                              const double* kappa_bound_list, const double* w_veh_list, const mcq_opts* opts,
                                              double* alpha_out, double* curv_err_out, int* status_out, mcq_info* info_out)
{
    if (!h || batch <= 0 || nmax <= 0 || !n_list || !reftrack || !alpha_out || !curv_err_out || !status_out) {
        g_err = "mcq_solve_device_ragged_params: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    const mcq_opts o = resolve_opts(opts);
    int rc = ensure_ws(h, (size_t)batch, (size_t)nmax);
    if (rc) return rc;
    McqBatch B;
    memset(&B, 0, sizeof(B));
    B.batch = batch;
    B.n = nmax;
    B.nmax = nmax;
    B.n_list = n_list;
    B.ref = reftrack;
    B.nv = normvec;
    B.sc = scaling;
    B.alpha = alpha_out;
    B.curv_err = curv_err_out;
    B.status = status_out;
    B.info = info_out;
    B.kappa_bound = kappa_bound;
    B.w_veh = w_veh;
    B.kappa_bound_list = kappa_bound_list;
    B.w_veh_list = w_veh_list;
    return launch(h, B, o);
}

extern "C" int mcq_prep_device(mcq_handle* h, int batch, int nmax, const int* n_list, const double* reftrack,
                               double* normvec_out, double* scaling_out, int* status_out)
{
    if (!h || batch <= 0 || nmax <= 0 || !reftrack || !status_out || (!normvec_out && !scaling_out)) {
        g_err = "mcq_prep_device: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    const mcq_opts o = resolve_opts(nullptr);
    int rc = ensure_ws(h, (size_t)batch, (size_t)nmax);
    if (rc) return rc;
    McqBatch B;
    memset(&B, 0, sizeof(B));
    B.batch = batch;
    B.n = nmax;
    B.nmax = nmax;
    B.n_list = n_list;
    B.ref = reftrack;
    B.status = status_out;
    B.nv_out = normvec_out;
    B.sc_out = scaling_out;
    B.prep_only = 1;
    B.w_veh = 0.0;              // the bounds computed along the way are not used: any widths are "feasible"
    B.kappa_bound = 1.0;
    return launch(h, B, o);
}

extern "C" int mcq_normals_crossing_device(mcq_handle* h, int batch, int nmax, const int* n_list, const double* reftrack,
                                           const double* normvec, int horizon, int* crossing_out)
{
    if (!h || batch <= 0 || nmax <= 0 || !reftrack || !normvec || !crossing_out) {
        g_err = "mcq_normals_crossing_device: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(mcq_normals_crossing_kernel, dim3(batch), dim3(256), 0, h->stream, nmax, n_list, reftrack, normvec,
                       horizon, crossing_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mcq_relinearise_device(mcq_handle* h, int batch, int nmax, const int* n_in, const double* reftrack_in,
                                      const double* normvec_in, const double* alpha, const int* live, double alpha_scale,
                                      double stepsize, double* reftrack_out, double* normvec_out, int* n_out,
                                      int* status_out)
{
    if (!h || batch <= 0 || nmax <= 0 || !n_in || !reftrack_in || !normvec_in || !alpha || !reftrack_out || !normvec_out ||
        !n_out || !status_out || !(stepsize > 0.0) || reftrack_in == reftrack_out || normvec_in == normvec_out) {
        g_err = "mcq_relinearise_device: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    int rc = ensure_ws(h, (size_t)batch, (size_t)nmax);
    if (rc) return rc;
    McqRelin R;
    memset(&R, 0, sizeof(R));
    R.batch = batch;
    R.nmax = nmax;
    R.n_in = n_in;
    R.ref_in = reftrack_in;
    R.nv_in = normvec_in;
    R.alpha = alpha;
    R.live = live;
    R.alpha_scale = alpha_scale;
    R.stepsize = stepsize;
    R.ref_out = reftrack_out;
    R.nv_out = normvec_out;
    R.n_out = n_out;
    R.status = status_out;
    R.vec = h->vec;
    R.state_in = h->state;          // what the last solve on this handle left (the caller solves, then re-linearises, the same batch)
    R.state_out = h->state2;
    h->state2_valid = true;
    h->state2_batch = batch;
    h->state2_nmax = nmax;
    hipLaunchKernelGGL(mcq_relinearise_kernel, dim3(batch), dim3(256), 0, h->stream, R);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int vel_profile_launch(mcq_handle* h, int batch, int n, int nmax, const int* n_of_track, const int* track_of,
                              const double* kappa, const double* el_lengths, const double* ggv, int n_ggv,
                              const double* ax_max_machines, int n_machines, const double* drag_coeff, const double* m_veh,
                              const double* v_max, double dyn_model_exp, double* vx_out, double* lap_time_out,
                              const double* mu = nullptr, int filt_window = 0)
{
    if (!h || batch <= 0 || (!n_of_track && n < 2) || nmax < n || nmax < 2 || !kappa || !el_lengths || !ggv || n_ggv < 1 ||
        !ax_max_machines || n_machines < 1 || !drag_coeff || !m_veh || !v_max || !vx_out || !lap_time_out ||
        !(dyn_model_exp > 0.0)) {
        g_err = "mcq_vel_profile_device: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    const size_t need = (size_t)2 * nmax * batch * sizeof(double);
    if (need > h->vel_scratch_bytes) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->vel_scratch);
        h->vel_scratch = nullptr;
        h->vel_scratch_bytes = 0;
        HIP_TRY(hipMalloc((void**)&h->vel_scratch, need));
        h->vel_scratch_bytes = need;
    }
    McqVel V;
    memset(&V, 0, sizeof(V));
    V.batch = batch; V.n = n; V.nmax = nmax;
    V.n_of_track = n_of_track;
    V.track_of = track_of; V.kappa = kappa; V.el = el_lengths;
    V.ggv = ggv; V.ng = n_ggv; V.axm = ax_max_machines; V.nam = n_machines;
    V.drag = drag_coeff; V.mass = m_veh; V.vmax = v_max; V.dyn_exp = dyn_model_exp;
    V.mu = mu; V.filt_window = filt_window;
    V.scratch = h->vel_scratch; V.vx_out = vx_out; V.lap_time = lap_time_out;
    hipLaunchKernelGGL(mcq_vel_profile_kernel, dim3((batch + 63) / 64), dim3(64), 0, h->stream, V);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mcq_vel_profile_device(mcq_handle* h, int batch, int n, int nmax, const int* track_of, const double* kappa,
                                      const double* el_lengths, const double* ggv, int n_ggv, const double* ax_max_machines,
                                      int n_machines, const double* drag_coeff, const double* m_veh, const double* v_max,
                                      double dyn_model_exp, double* vx_out, double* lap_time_out)
{
    return vel_profile_launch(h, batch, n, nmax, nullptr, track_of, kappa, el_lengths, ggv, n_ggv, ax_max_machines,
                              n_machines, drag_coeff, m_veh, v_max, dyn_model_exp, vx_out, lap_time_out);
}

extern "C" int mcq_vel_profile_device_ragged(mcq_handle* h, int batch, int nmax, const int* n_of_track, const int* track_of,
                                             const double* kappa, const double* el_lengths, const double* ggv, int n_ggv,
                                             const double* ax_max_machines, int n_machines, const double* drag_coeff,
                                             const double* m_veh, const double* v_max, double dyn_model_exp, double* vx_out,
                                             double* lap_time_out)
{
    if (!n_of_track) { g_err = "mcq_vel_profile_device_ragged: n_of_track is NULL"; return MCQ_E_ARG; }
    return vel_profile_launch(h, batch, 0, nmax, n_of_track, track_of, kappa, el_lengths, ggv, n_ggv, ax_max_machines,
                              n_machines, drag_coeff, m_veh, v_max, dyn_model_exp, vx_out, lap_time_out);
}

extern "C" int mcq_vel_profile_device_opts(mcq_handle* h, int batch, int n, int nmax, const int* n_of_track, const int* track_of,
                                           const double* kappa, const double* el_lengths, const double* ggv, int n_ggv,
                                           const double* ax_max_machines, int n_machines, const double* drag_coeff,
                                           const double* m_veh, const double* v_max, const mcq_vel_opts* opts, double* vx_out,
                                           double* lap_time_out)
{
    if (!opts) { g_err = "mcq_vel_profile_device_opts: opts is NULL"; return MCQ_E_ARG; }
    if (opts->filt_window < 0) { g_err = "mcq_vel_profile_device_opts: negative filt_window"; return MCQ_E_ARG; }
    return vel_profile_launch(h, batch, n_of_track ? 0 : n, nmax, n_of_track, track_of, kappa, el_lengths, ggv, n_ggv, ax_max_machines,
                              n_machines, drag_coeff, m_veh, v_max, opts->dyn_model_exp, vx_out, lap_time_out, opts->mu,
                              opts->filt_window);
}

extern "C" int mcq_raceline_device(mcq_handle* h, int batch, int nmax, const int* n_in, const double* reftrack,
                                   const double* normvec, const double* alpha, double stepsize, int mmax,
                                   double* raceline_out, double* psi_out, double* kappa_out, double* el_lengths_out,
                                   int* m_out, int* status_out)
{
    if (!h || batch <= 0 || nmax < 3 || mmax < 2 || !reftrack || !normvec || !alpha || !(stepsize > 0.0) || !kappa_out ||
        !el_lengths_out || !m_out || !status_out) {
        g_err = "mcq_raceline_device: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    int rc = ensure_ws(h, (size_t)batch, (size_t)nmax);
    if (rc) return rc;
    McqRace Q;
    memset(&Q, 0, sizeof(Q));
    Q.batch = batch; Q.nmax = nmax; Q.mmax = mmax;
    Q.n_in = n_in; Q.ref = reftrack; Q.nv = normvec; Q.alpha = alpha; Q.stepsize = stepsize;
    Q.xy_out = raceline_out; Q.psi_out = psi_out; Q.kappa_out = kappa_out; Q.el_out = el_lengths_out;
    Q.m_out = m_out; Q.status = status_out;
    Q.vec = h->vec;
    hipLaunchKernelGGL(mcq_raceline_kernel, dim3(batch), dim3(256), 0, h->stream, Q);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- device memory plumbing for callers that keep data resident between calls without a second HIP runtime in the
//      process (the Python IQP driver): plain allocate / free / copy on the handle's device and stream ------------------
extern "C" int mcq_device_alloc(mcq_handle* h, size_t bytes, void** out)
{
    if (!h || !out || bytes == 0) { g_err = "mcq_device_alloc: bad argument"; return MCQ_E_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMalloc(out, bytes));
    HIP_TRY(hipMemsetAsync(*out, 0, bytes, h->stream));
    return 0;
}

extern "C" int mcq_device_free(mcq_handle* h, void* ptr)
{
    if (!h) { g_err = "mcq_device_free: NULL handle"; return MCQ_E_ARG; }
    if (!ptr) return 0;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIP_TRY(hipStreamSynchronize(h->comm_stream));      // (a gather in flight may read or write the buffer)
    HIP_TRY(hipFree(ptr));
    return 0;
}

extern "C" int mcq_copy_to_device(mcq_handle* h, void* dst, const void* src, size_t bytes)
{
    if (!h || !dst || !src) { g_err = "mcq_copy_to_device: bad argument"; return MCQ_E_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));          // the host buffer may be reused on return
    return 0;
}

extern "C" int mcq_copy_to_host(mcq_handle* h, void* dst, const void* src, size_t bytes)
{
    if (!h || !dst || !src) { g_err = "mcq_copy_to_host: bad argument"; return MCQ_E_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    // a gather enqueued before this copy may be writing `src` on the comm stream: the copy is ordered behind the latest one (ADVICE r4:
    // parallel.solve_sharded read its receive buffer before the gather had finished)
    if (h->comm_stream && h->comm_seq > 0) HIP_TRY(hipStreamWaitEvent(h->stream, h->comm_done[(h->comm_seq - 1) & 3u], 0));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int mcq_sync(mcq_handle* h)
{
    if (!h) { g_err = "mcq_sync: NULL handle"; return MCQ_E_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIP_TRY(hipStreamSynchronize(h->comm_stream));      // gathers in flight (mcq_comm_allgather) are work of the handle too
    if (h->stream2) HIP_TRY(hipStreamSynchronize(h->stream2));
    return 0;
}

extern "C" void* mcq_stream(mcq_handle* h) { return h ? (void*)h->stream : nullptr; }

extern "C" int mcq_last_timing(mcq_handle* h, float ms[5])
{
    if (!h || !ms || !h->timing_valid) { g_err = "mcq_last_timing: no timed launch"; return MCQ_E_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventSynchronize(h->ev[4]));
    HIP_TRY(hipEventElapsedTime(&ms[0], h->ev[0], h->ev[1]));      // shortest path: mcq_assemble_sp_kernel; otherwise the gap between two event records
    ms[1] = 0.0f;                                                   // (no such kernel since round 4)
    HIP_TRY(hipEventElapsedTime(&ms[2], h->ev[2], h->ev[3]));
    ms[3] = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms[4], h->ev[0], h->ev[4]));
    return 0;
}

// A span of launches on the handle's compute stream, timed on the device (include/mcq.h)
extern "C" int mcq_timing_begin(mcq_handle* h)
{
    if (!h) { g_err = "mcq_timing_begin: NULL handle"; return MCQ_E_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    for (int k = 0; k < 2; ++k) if (!h->ev_span[k]) HIP_TRY(hipEventCreate(&h->ev_span[k]));
    HIP_TRY(hipEventRecord(h->ev_span[0], h->stream));
    h->span_launches = 0;
    h->span_open = true;
    return 0;
}

extern "C" int mcq_timing_end(mcq_handle* h, float* ms_out, int* launches_out)
{
    if (!h || !ms_out || !launches_out) { g_err = "mcq_timing_end: bad argument"; return MCQ_E_ARG; }
    if (!h->span_open) { g_err = "mcq_timing_end: no span open (mcq_timing_begin)"; return MCQ_E_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord(h->ev_span[1], h->stream));
    HIP_TRY(hipEventSynchronize(h->ev_span[1]));
    HIP_TRY(hipEventElapsedTime(ms_out, h->ev_span[0], h->ev_span[1]));
    *launches_out = h->span_launches;
    h->span_open = false;
    return 0;
}

extern "C" int mcq_last_upload_was_direct(mcq_handle* h) { return h ? h->last_upload_direct : 0; }

extern "C" long long mcq_workspace_bytes(mcq_handle* h) { return h ? h->ws_bytes + h->gi_bytes + h->gis_bytes + h->alt_bytes : 0; }

static int ensure_pin(mcq_handle* h, size_t bytes)
{
    if (bytes <= h->pin_bytes) return 0;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->pin) (void)hipHostFree(h->pin);
    h->pin = nullptr;
    h->pin_bytes = 0;
    HIP_TRY(hipHostMalloc(&h->pin, bytes, hipHostMallocDefault));
    if (h->poison) memset(h->pin, 0xff, bytes);
    h->pin_bytes = bytes;
    return 0;
}

// error paths of the host-buffer entries: copies from / to the pinned staging may still be queued
#define HIP_TRY_SYNC(expr)                                                                                \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            char buf_[512];                                                                               \
            snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                     __LINE__);                                                                           \
            g_err = buf_;                                                                                 \
            (void)hipStreamSynchronize(h->stream);                                                        \
            return MCQ_E_DEVICE;                                                                          \
        }                                                                                                 \
    } while (0)

// ---- the n spline scalings out of the dense matrix the reference passes (include/mcq.h); host code, no GPU ----------------------------------
extern "C" int mcq_les_scalings(const double* A, int n, double* s_out, int check)
{
    if (!A || !s_out || n < 3) { g_err = "mcq_les_scalings: bad argument"; return MCQ_E_ARG; }
    const size_t m = (size_t)4 * n;
    for (int i = 0; i + 1 < n; ++i) s_out[i] = -A[((size_t)4 * i + 2) * m + 4 * i + 5];
    s_out[n - 1] = A[(m - 2) * m + 1];
    if (!check) return 0;
    // Row by row: the entries the closed-spline system has in that row, at their values (the wrap-around rows carry the
    // opposite sign), and NOTHING else -- a row's non-zeros are counted while it streams through.
    // one thread per 16 MB of matrix, 32 at most (512 MB at n = 2000: 3.4 ms on 8 threads of the bench host, memory-bound); $MCQ_PACK_THREADS overrides
    int nthreads = (int)std::min<size_t>(32, std::max<size_t>(1, (m * m * sizeof(double)) >> 24));
    if (const char* e = getenv("MCQ_PACK_THREADS")) nthreads = atoi(e);
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && nthreads > hw) nthreads = hw;
    if (nthreads > n) nthreads = n;
    if (nthreads < 1) nthreads = 1;
    std::vector<long long> bad((size_t)nthreads, -1);          // first offending (row * m + column) of a thread's blocks
    auto rel = [](double a, double b) { return fabs(a - b) <= 1e-12 * fabs(b); };
    auto scan = [&](int t) {
        const int i0 = (int)((long long)n * t / nthreads), i1 = (int)((long long)n * (t + 1) / nthreads);
        for (int i = i0; i < i1 && bad[t] < 0; ++i) {
            const size_t j = (size_t)4 * i;
            const bool last = i == n - 1;
            const double s = s_out[i];
            if (!(s > 0.0) || !std::isfinite(s)) { bad[t] = (long long)((j + 2) * m + (last ? 1 : j + 5)); break; }
            for (int r = 0; r < 4; ++r) {
                const double* row = A + (j + r) * m;
                size_t nz = 0;
                for (size_t c = 0; c < m; ++c) nz += row[c] != 0.0;
                bool ok;
                if (r == 0) ok = nz == 1 && row[j] == 1.0;
                else if (r == 1) ok = nz == 4 && row[j] == 1.0 && row[j + 1] == 1.0 && row[j + 2] == 1.0 && row[j + 3] == 1.0;
                else if (r == 2) ok = nz == 4 && (last ? (row[j + 1] == -1.0 && row[j + 2] == -2.0 && row[j + 3] == -3.0 && row[1] == s)
                                                       : (row[j + 1] == 1.0 && row[j + 2] == 2.0 && row[j + 3] == 3.0 && row[j + 5] == -s));
                else ok = nz == 3 && (last ? (row[j + 2] == -2.0 && row[j + 3] == -6.0 && rel(row[2], 2.0 * s * s))
                                           : (row[j + 2] == 2.0 && row[j + 3] == 6.0 && rel(-row[j + 6], 2.0 * s * s)));
                if (!ok) { bad[t] = (long long)((j + r) * m); break; }
            }
        }
    };
    if (nthreads > 1) {
        std::vector<std::thread> th;
        int taken = 1;
        try {
            for (; taken < nthreads; ++taken) th.emplace_back(scan, taken);
        } catch (...) {}
        scan(0);
        for (int t = taken; t < nthreads; ++t) scan(t);
        for (auto& t : th) t.join();
    } else scan(0);
    for (int t = 0; t < nthreads; ++t) {
        if (bad[t] >= 0) {
            char buf[160];
            snprintf(buf, sizeof buf, "mcq_les_scalings: row %lld of A does not have the structure of calc_splines' closed-spline system", bad[t] / (long long)m);
            g_err = buf;
            return MCQ_E_ARG;
        }
    }
    return 0;
}

extern "C" int mcq_host_alloc(mcq_handle* h, size_t bytes, void** out)
{
    if (!h || !out || bytes == 0) { g_err = "mcq_host_alloc: bad argument"; return MCQ_E_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return 0;
}

extern "C" int mcq_host_free(mcq_handle* h, void* ptr)
{
    if (!h) { g_err = "mcq_host_free: NULL handle"; return MCQ_E_ARG; }
    if (!ptr) return 0;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipHostFree(ptr));
    return 0;
}

// the knobs of the sliced host entries (mcq_solve_host, mcq_solve_batch): slices for this batch / these options, or 1
static int host_slices(int batch, const mcq_opts& o)
{
    int slice_min = 512;           // ($MCQ_HOST_SLICE_MIN: the tests slice a batch of eleven)
    if (const char* e = getenv("MCQ_HOST_SLICE_MIN")) slice_min = std::max(atoi(e), 4);
    // TWO slices, one per compute stream: both are in flight as soon as their uploads are, so a batch with a heavy tail -- a sweep over tight
    // curvature bounds, where single problems take seconds in the Goldfarb-Idnani path -- loses nothing against the one launch.  Four slices
    // gain 0.3 ms more on uniform batches (11.5 against 11.8 ms for 1024 x N = 2000; one launch: 13.0) but make slice k + 2 wait for the slowest
    // problem of slice k on its stream: 600 curvature-tight long rings then take 76 s instead of 45 (docs/NOTEBOOK.md R6.5).  $MCQ_HOST_SLICES: 2 .. 8
    int nsl = 2;
    if (const char* e = getenv("MCQ_HOST_SLICES")) nsl = std::min(std::max(atoi(e), 2), 8);
    return (batch >= slice_min && o.objective == MCQ_OBJ_MIN_CURV && o.algorithm != MCQ_ALG_GI && !getenv("MCQ_HOST_ONE_LAUNCH")) ? nsl : 1;
}
// the compute stream of slice k: the handle's two, in turn.  (Tried, round 6: a stream per slice -- 14.0 ms where two streams give 11.5 for four
// slices of 256 problems, and the 600 curvature-tight long rings 105 s where the one launch takes 45: more than two queues of large workgroups
// cost more in the hardware scheduler than they overlap.)
static hipStream_t slice_stream(mcq_handle* h, int k) { return (k & 1) ? h->stream2 : h->stream; }
static int ensure_slice_streams(mcq_handle* h)
{
    if (!h->cs_in) { HIP_TRY(hipStreamCreate(&h->cs_in)); HIP_TRY(hipStreamCreate(&h->cs_out)); }
    if (!h->stream2) HIP_TRY(hipStreamCreate(&h->stream2));
    for (int k = 0; k < 16; ++k) if (!h->ev_slice[k]) HIP_TRY(hipEventCreate(&h->ev_slice[k]));
    HIP_TRY(hipStreamSynchronize(h->stream));          // whatever ran before on the handle is done with the staging buffers and the workspace
    HIP_TRY(hipStreamSynchronize(h->stream2));
    return 0;
}

// (the slices of mcq_solve_host: every stream that may hold copies from / into the caller's buffers is drained before an error is reported)
#define HIP_TRY_SLICE(expr)                                                                               \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            char buf_[512];                                                                               \
            snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                     __LINE__);                                                                           \
            g_err = buf_;                                                                                 \
            (void)hipStreamSynchronize(h->cs_in);                                                         \
            (void)hipStreamSynchronize(h->stream);                                                        \
            (void)hipStreamSynchronize(h->stream2);                                                       \
            (void)hipStreamSynchronize(h->cs_out);                                                        \
            return MCQ_E_DEVICE;                                                                          \
        }                                                                                                 \
    } while (0)

extern "C" int mcq_solve_host(mcq_handle* h, int batch, int n, const double* reftrack, const double* normvec,
                              const double* scaling, double kappa_bound, double w_veh, const mcq_opts* opts, double* alpha_out,
                              double* curv_err_out, int* status_out, mcq_info* info_out)
{
    if (!h || batch <= 0 || n <= 0 || !reftrack || !alpha_out || !curv_err_out || !status_out) {
        g_err = "mcq_solve_host: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    const mcq_opts o = resolve_opts(opts);
    int rc = ensure_ws(h, (size_t)batch, (size_t)n);
    if (rc) return rc;
    rc = ensure_stage(h, (size_t)batch, (size_t)n);
    if (rc) return rc;
    const size_t elems = (size_t)batch * n;
    McqBatch B;
    memset(&B, 0, sizeof(B));
    B.batch = batch;
    B.n = n;
    B.nmax = n;
    B.ref = h->d_ref;
    B.nv = normvec ? h->d_nv : nullptr;
    B.sc = scaling ? h->d_sc : nullptr;
    B.alpha = h->d_alpha;
    B.curv_err = h->d_curv;
    B.status = h->d_status;
    B.info = info_out ? h->d_info : nullptr;
    B.kappa_bound = kappa_bound;
    B.w_veh = w_veh;
    // A large batch in (two) SLICES (round 6): the upload of slice k + 1 and the download of slice k - 1 run while slice k's kernel does -- on a
    // copy stream each way --, and the kernels of consecutive slices go to the handle's two compute streams; all of them work on disjoint rows of
    // the ONE workspace (McqBatch.pb0).  The blocking single-batch entry had paid its 115 MB of H2D and 16 MB of D2H in full next to the kernel: 13.3 ms
    // for a 10.4 ms kernel at 1024 x N = 2000.  Results bitwise those of the one launch: the problems are independent.  $MCQ_HOST_ONE_LAUNCH=1:
    // the one launch (A/B knob).
    const int nsl = host_slices(batch, o);
    if (nsl > 1) {
        if (int src = ensure_slice_streams(h)) return src;
        if (int frc = fill_batch(h, B, o, false)) return frc;
        B.warm = nullptr;
        h->state2_valid = false;
        h->timing_valid = false;
        for (int k = 0; k < nsl; ++k) {
            const int b0 = (int)((long long)batch * k / nsl), b1 = (int)((long long)batch * (k + 1) / nsl);
            const size_t off = (size_t)b0 * n, cnt = (size_t)(b1 - b0) * n;
            HIP_TRY_SLICE(hipMemcpyAsync(h->d_ref + off * 4, reftrack + off * 4, cnt * 4 * sizeof(double), hipMemcpyHostToDevice, h->cs_in));
            if (normvec) HIP_TRY_SLICE(hipMemcpyAsync(h->d_nv + off * 2, normvec + off * 2, cnt * 2 * sizeof(double), hipMemcpyHostToDevice, h->cs_in));
            if (scaling) HIP_TRY_SLICE(hipMemcpyAsync(h->d_sc + off, scaling + off, cnt * sizeof(double), hipMemcpyHostToDevice, h->cs_in));
            HIP_TRY_SLICE(hipEventRecord(h->ev_slice[k], h->cs_in));
            hipStream_t st = slice_stream(h, k);
            HIP_TRY_SLICE(hipStreamWaitEvent(st, h->ev_slice[k], 0));
            McqBatch S = B;
            S.pb0 = b0;
            hipLaunchKernelGGL(mcq_solve_kernel, dim3(b1 - b0), dim3(256), 0, st, S);
            HIP_TRY_SLICE(hipGetLastError());
            HIP_TRY_SLICE(hipEventRecord(h->ev_slice[8 + k], st));
            HIP_TRY_SLICE(hipStreamWaitEvent(h->cs_out, h->ev_slice[8 + k], 0));
            HIP_TRY_SLICE(hipMemcpyAsync(alpha_out + off, h->d_alpha + off, cnt * sizeof(double), hipMemcpyDeviceToHost, h->cs_out));
        }
        // (cs_out is behind every slice's kernel now: the small arrays follow the last slice's alpha)
        HIP_TRY_SLICE(hipMemcpyAsync(curv_err_out, h->d_curv, batch * sizeof(double), hipMemcpyDeviceToHost, h->cs_out));
        HIP_TRY_SLICE(hipMemcpyAsync(status_out, h->d_status, batch * sizeof(int), hipMemcpyDeviceToHost, h->cs_out));
        if (info_out) HIP_TRY_SLICE(hipMemcpyAsync(info_out, h->d_info, batch * sizeof(mcq_info), hipMemcpyDeviceToHost, h->cs_out));
        HIP_TRY(hipStreamSynchronize(h->cs_out));
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream2));
        return 0;
    }
    HIP_TRY_SYNC(hipMemcpyAsync(h->d_ref, reftrack, elems * 4 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (normvec) HIP_TRY_SYNC(hipMemcpyAsync(h->d_nv, normvec, elems * 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (scaling) HIP_TRY_SYNC(hipMemcpyAsync(h->d_sc, scaling, elems * sizeof(double), hipMemcpyHostToDevice, h->stream));
    rc = launch(h, B, o);
    if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }
    HIP_TRY_SYNC(hipMemcpyAsync(alpha_out, h->d_alpha, elems * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY_SYNC(hipMemcpyAsync(curv_err_out, h->d_curv, batch * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY_SYNC(hipMemcpyAsync(status_out, h->d_status, batch * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (info_out) HIP_TRY_SYNC(hipMemcpyAsync(info_out, h->d_info, batch * sizeof(mcq_info), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    // (MCQ_KAPPA_NO_SLOT never arrives here since round 5: the Goldfarb-Idnani path inside the solver kernel takes such problems)
    return 0;
}

// Host-buffer fp32 entry (mcq_solve_batch_f32): float rows in, float alpha out -- half the PCIe bytes of the
// fp64 entry.  The float rows land in the staging buffer of the (unused) normals -- batch * n * 2 doubles = batch * n * 4 floats --,
// the float alpha in the one of the scalings; the fp64 rows are rebuilt next to them (solve_f32_rows).
extern "C" int mcq_solve_batch_f32(mcq_handle* h, int batch, int n, int layout, const float* reftrack, const double* origin,
                                   double kappa_bound, double w_veh, const mcq_opts* opts, float* alpha_out, double* curv_err_out,
                                   int* status_out, mcq_info* info_out)
{
    if (!h || batch <= 0 || n <= 0 || !reftrack || !alpha_out || !curv_err_out || !status_out ||
        (layout != MCQ_F32_ABSOLUTE && layout != MCQ_F32_INCREMENTS)) {
        g_err = "mcq_solve_batch_f32: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    const mcq_opts o = resolve_opts(opts);
    int rc = ensure_ws(h, (size_t)batch, (size_t)n);
    if (rc) return rc;
    rc = ensure_stage(h, (size_t)batch, (size_t)n);
    if (rc) return rc;
    const size_t elems = (size_t)batch * n;
    float* d_rows = (float*)h->d_nv;
    float* d_al32 = (float*)h->d_sc;
    HIP_TRY_SYNC(hipMemcpyAsync(d_rows, reftrack, elems * 4 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (origin) HIP_TRY_SYNC(hipMemcpyAsync(h->d_org, origin, (size_t)batch * 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    rc = solve_f32_rows(h, batch, n, layout, d_rows, origin ? h->d_org : nullptr, kappa_bound, w_veh, o, d_al32, h->d_curv, h->d_status,
                        info_out ? h->d_info : nullptr);
    if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }
    HIP_TRY_SYNC(hipMemcpyAsync(alpha_out, d_al32, elems * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY_SYNC(hipMemcpyAsync(curv_err_out, h->d_curv, batch * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY_SYNC(hipMemcpyAsync(status_out, h->d_status, batch * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (info_out) HIP_TRY_SYNC(hipMemcpyAsync(info_out, h->d_info, batch * sizeof(mcq_info), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// ---- a stream of uniform host batches with the PCIe behind the kernels (include/mcq.h) ------------------------------------------
static int ensure_pipe(mcq_handle* h, size_t batch, size_t nmax)
{
    if (!h->cs_in) {                    // (mcq_solve_host's slices may have created the two copy streams already)
        HIP_TRY(hipStreamCreate(&h->cs_in));
        HIP_TRY(hipStreamCreate(&h->cs_out));
    }
    if (!h->ev_up[0]) {
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(hipEventCreate(&h->ev_up[k]));
            HIP_TRY(hipEventCreate(&h->ev_done[k]));
            HIP_TRY(hipEventCreate(&h->ev_down[k]));
        }
    }
    const size_t elems = batch * nmax;
    if (elems <= h->pipe_elems && batch <= h->pipe_batch) return 0;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipStreamSynchronize(h->cs_in));
    HIP_TRY(hipStreamSynchronize(h->cs_out));
    free_pipe(h);
    HIP_TRY(hipMalloc((void**)&h->p_ref, elems * 4 * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->p_nv, elems * 2 * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->p_sc, elems * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->p_alpha, elems * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->p_curv, batch * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->p_status, batch * sizeof(int)));
    h->pipe_elems = elems;
    h->pipe_batch = batch;
    return 0;
}

// every stream of the pipeline is drained before an error is reported: copies from / into the caller's buffers may be queued
#define HIP_TRY_PIPE(expr)                                                                                \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            char buf_[512];                                                                               \
            snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                     __LINE__);                                                                           \
            g_err = buf_;                                                                                 \
            (void)hipStreamSynchronize(h->cs_in);                                                         \
            (void)hipStreamSynchronize(h->stream);                                                        \
            if (h->stream2) (void)hipStreamSynchronize(h->stream2);                                       \
            (void)hipStreamSynchronize(h->cs_out);                                                        \
            return MCQ_E_DEVICE;                                                                          \
        }                                                                                                 \
    } while (0)

extern "C" int mcq_solve_host_pipelined(mcq_handle* h, int steps, int batch, int n, const double* const* reftrack,
                                        const double* const* normvec, const double* const* scaling, double kappa_bound, double w_veh,
                                        const mcq_opts* opts, double* const* alpha_out, double* const* curv_err_out,
                                        int* const* status_out)
{
    if (!h || steps <= 0 || batch <= 0 || n <= 0 || !reftrack || !alpha_out || !curv_err_out || !status_out) {
        g_err = "mcq_solve_host_pipelined: bad argument";
        return MCQ_E_ARG;
    }
    for (int k = 0; k < steps; ++k) {
        if (!reftrack[k] || !alpha_out[k] || !curv_err_out[k] || !status_out[k]) { g_err = "mcq_solve_host_pipelined: NULL buffer in step list"; return MCQ_E_ARG; }
    }
    HIP_TRY(hipSetDevice(h->device));
    const mcq_opts o = resolve_opts(opts);
    int rc = ensure_ws(h, (size_t)batch, (size_t)n);
    if (rc) return rc;
    rc = ensure_stage(h, (size_t)batch, (size_t)n);
    if (rc) return rc;
    rc = ensure_pipe(h, (size_t)batch, (size_t)n);
    if (rc) return rc;
    // curv_error / status of every step land in PINNED memory of the handle and reach the caller's arrays at the end: a device-to-host copy into
    // pageable memory -- what these two small arrays usually are -- blocks the host until the step's kernels have finished, i.e. the host could
    // not enqueue step k + 1 before step k was over (round 5: this was the 0.5 ms per step that rounds 3-4 reported as "not hidden")
    rc = ensure_pin(h, (size_t)steps * batch * (sizeof(double) + sizeof(int)));
    if (rc) return rc;
    double* pin_cu = (double*)h->pin;
    int* pin_st = (int*)(pin_cu + (size_t)steps * batch);
    // kernels of odd steps on the second compute stream / workspace (the variable: A/B knob); MCQ_ALG_GI stays on one stream: its slots
    // (one per resident workgroup) belong to the first workspace (the second one's few slots made odd steps run ~32 problems at a time)
    const bool two = steps > 1 && !getenv("MCQ_PIPE_ONE_STREAM") && o.algorithm != MCQ_ALG_GI;
    if (two) { rc = ensure_alt(h, (size_t)batch, (size_t)n); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(h->stream));          // whatever ran before on the handle's stream is done with the staging buffers
    const size_t elems = (size_t)batch * n;
    double* s_ref[2] = {h->d_ref, h->p_ref};
    double* s_nv[2] = {h->d_nv, h->p_nv};
    double* s_sc[2] = {h->d_sc, h->p_sc};
    double* s_al[2] = {h->d_alpha, h->p_alpha};
    double* s_cu[2] = {h->d_curv, h->p_curv};
    int* s_st[2] = {h->d_status, h->p_status};
    // Order of the host's enqueues: upload(0); then per step  kernels(k), upload(k + 1), download(k).  The copies of both directions
    // drain through one in-order copy queue of the runtime (rocprofv3 --memory-copy-trace, round 3: with download(k) enqueued ahead
    // of upload(k + 1) the upload sat behind it, i.e. behind the kernels of step k, and nothing overlapped): the upload of the NEXT
    // step therefore goes in first -- it only depends on kernels two steps back -- and runs while step k computes.
    auto upload = [&](int k) -> int {
        const int s = k & 1;
        const double* nv_k = normvec ? normvec[k] : nullptr;
        const double* sc_k = scaling ? scaling[k] : nullptr;
        // the kernels of step k - 2 were the last readers of this slot's rows
        if (k >= 2) HIP_TRY_PIPE(hipStreamWaitEvent(h->cs_in, h->ev_done[s], 0));
        HIP_TRY_PIPE(hipMemcpyAsync(s_ref[s], reftrack[k], elems * 4 * sizeof(double), hipMemcpyHostToDevice, h->cs_in));
        if (nv_k) HIP_TRY_PIPE(hipMemcpyAsync(s_nv[s], nv_k, elems * 2 * sizeof(double), hipMemcpyHostToDevice, h->cs_in));
        if (sc_k) HIP_TRY_PIPE(hipMemcpyAsync(s_sc[s], sc_k, elems * sizeof(double), hipMemcpyHostToDevice, h->cs_in));
        HIP_TRY_PIPE(hipEventRecord(h->ev_up[s], h->cs_in));
        return 0;
    };
    rc = upload(0);
    if (rc) return rc;
    const bool trace = getenv("MCQ_PIPE_TRACE") != nullptr;
    struct timespec ts0;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    for (int k = 0; k < steps; ++k) {
        const int s = k & 1;
        if (trace) { struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1); fprintf(stderr, "pipe: step %d enqueued from %.3f ms\n", k, (t1.tv_sec - ts0.tv_sec) * 1e3 + (t1.tv_nsec - ts0.tv_nsec) * 1e-6); }
        const double* nv_k = normvec ? normvec[k] : nullptr;
        const double* sc_k = scaling ? scaling[k] : nullptr;
        // kernels of step k: after its upload, and after the download of step k - 2 has left the slot's result buffers
        // (slot s's kernels always run on compute stream s: step k - 2, the last user of this slot's workspace and buffers, precedes in stream order)
        hipStream_t cst = (two && s) ? h->stream2 : h->stream;
        HIP_TRY_PIPE(hipStreamWaitEvent(cst, h->ev_up[s], 0));
        if (k >= 2) HIP_TRY_PIPE(hipStreamWaitEvent(cst, h->ev_down[s], 0));
        McqBatch B;
        memset(&B, 0, sizeof(B));
        B.batch = batch;
        B.n = n;
        B.nmax = n;
        B.ref = s_ref[s];
        B.nv = nv_k ? s_nv[s] : nullptr;
        B.sc = sc_k ? s_sc[s] : nullptr;
        B.alpha = s_al[s];
        B.curv_err = s_cu[s];
        B.status = s_st[s];
        B.kappa_bound = kappa_bound;
        B.w_veh = w_veh;
        rc = launch(h, B, o, two && s);
        if (rc) { (void)hipStreamSynchronize(h->cs_in); (void)hipStreamSynchronize(h->stream); if (h->stream2) (void)hipStreamSynchronize(h->stream2); (void)hipStreamSynchronize(h->cs_out); return rc; }
        HIP_TRY_PIPE(hipEventRecord(h->ev_done[s], cst));
        if (k + 1 < steps) { rc = upload(k + 1); if (rc) return rc; }
        // download of step k
        HIP_TRY_PIPE(hipStreamWaitEvent(h->cs_out, h->ev_done[s], 0));
        HIP_TRY_PIPE(hipMemcpyAsync(alpha_out[k], s_al[s], elems * sizeof(double), hipMemcpyDeviceToHost, h->cs_out));
        HIP_TRY_PIPE(hipMemcpyAsync(pin_cu + (size_t)k * batch, s_cu[s], batch * sizeof(double), hipMemcpyDeviceToHost, h->cs_out));
        HIP_TRY_PIPE(hipMemcpyAsync(pin_st + (size_t)k * batch, s_st[s], batch * sizeof(int), hipMemcpyDeviceToHost, h->cs_out));
        HIP_TRY_PIPE(hipEventRecord(h->ev_down[s], h->cs_out));
    }
    HIP_TRY_PIPE(hipStreamSynchronize(h->cs_in));
    HIP_TRY_PIPE(hipStreamSynchronize(h->stream));
    if (two) HIP_TRY_PIPE(hipStreamSynchronize(h->stream2));
    HIP_TRY_PIPE(hipStreamSynchronize(h->cs_out));
    for (int k = 0; k < steps; ++k) {
        memcpy(curv_err_out[k], pin_cu + (size_t)k * batch, batch * sizeof(double));
        memcpy(status_out[k], pin_st + (size_t)k * batch, batch * sizeof(int));
    }
    return 0;
}

// ---- tph.iqp_handler as one call (include/mcq.h) ---------------------------------------------------------------------------------
static int ensure_iqp(mcq_handle* h, size_t batch)
{
    if (batch <= h->iqp_batch) return 0;
    HIP_TRY(hipStreamSynchronize(h->stream));
    (void)hipFree(h->d_iqp); (void)hipFree(h->d_iqp_curv); (void)hipFree(h->d_trace);
    h->d_iqp = nullptr; h->d_iqp_curv = nullptr; h->d_trace = nullptr; h->iqp_batch = 0;
    HIP_TRY(hipMalloc((void**)&h->d_iqp, (batch * 8 + 16) * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&h->d_iqp_curv, batch * 2 * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->d_trace, batch * MCQ_IQP_TRACE * sizeof(double)));     // mcq_iqp_batch's trace staging
    h->iqp_batch = batch;
    return 0;
}

extern "C" int mcq_iqp_device(mcq_handle* h, int batch, int nmax, int* n_io, double* reftrack_a, double* normvec_a,
                              double* reftrack_b, double* normvec_b, const double* scaling, double kappa_bound, double w_veh,
                              double stepsize_interp, int iters_min, double curv_error_allowed, int max_rounds,
                              const mcq_opts* opts, double* alpha_out, int* buf_out, double* curv_err_out, int* status_out,
                              int* rounds_out, double* curv_trace_out, mcq_iqp_stats* stats)
{
    if (!h || batch <= 0 || nmax <= 0 || !n_io || !reftrack_a || !normvec_a || !reftrack_b || !normvec_b || !alpha_out ||
        !buf_out || !curv_err_out || !status_out || !rounds_out || !(stepsize_interp > 0.0) || iters_min < 1 || max_rounds < 1 ||
        reftrack_a == reftrack_b || normvec_a == normvec_b) {
        g_err = "mcq_iqp_device: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    mcq_opts o = resolve_opts(opts);
    const bool warm = !(opts && opts->warm_start < 0);        // passes 2+ start from the carried working sets unless switched off
    int rc = ensure_ws(h, (size_t)batch, (size_t)nmax);
    if (rc) return rc;
    rc = ensure_iqp(h, (size_t)batch);
    if (rc) return rc;
    const bool timed = stats && stats->timed;
    if (timed) { rc = ensure_stage(h, (size_t)batch, (size_t)nmax); if (rc) return rc; }
    // bookkeeping arrays: live | n of the other set | glue status | pass status | final_n (= n_io on return) ... live count
    int* live = h->d_iqp;
    int* n_b = live + batch;
    int* rst = n_b + batch;
    int* pass_status = rst + batch;
    int* d_final_n = pass_status + batch;
    int* live_count = d_final_n + batch;
    double* pass_curv = h->d_iqp_curv;
    int* n_set[2] = {n_io, n_b};
    double* ref_set[2] = {reftrack_a, reftrack_b};
    double* nv_set[2] = {normvec_a, normvec_b};
    // every track starts live (a track with n == 0 ends in its first, empty pass: status MCQ_BAD_INPUT, rounds_out 1)
    {
        std::vector<int> ones((size_t)batch, 1);
        HIP_TRY(hipMemcpyAsync(live, ones.data(), batch * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemsetAsync(rounds_out, 0, batch * sizeof(int), h->stream));
        HIP_TRY(hipMemsetAsync(buf_out, 0, batch * sizeof(int), h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));      // `ones` goes out of scope
    }
    McqIqpStep S;
    memset(&S, 0, sizeof(S));
    S.batch = batch;
    S.iters_min = iters_min;
    S.curv_allowed = curv_error_allowed;
    S.curv = pass_curv;
    S.status = pass_status;
    S.relin_status = rst;
    S.live = live;
    S.final_buf = buf_out;
    S.final_curv = curv_err_out;
    S.final_status = status_out;
    S.final_rounds = rounds_out;
    S.live_count = live_count;
    S.curv_trace = curv_trace_out;
    if (curv_trace_out) HIP_TRY(hipMemsetAsync(curv_trace_out, 0, (size_t)batch * MCQ_IQP_TRACE * sizeof(double), h->stream));
    const dim3 sgrid((unsigned)((batch + 255) / 256)), sblock(256);
    int cur = 0, rounds = 0, n_live = batch;
    long long solves = 0;
    if (stats) { stats->rounds = 0; stats->qp_solves = 0; for (int k = 0; k < 16; ++k) { stats->solver_ms[k] = 0.f; stats->fallbacks[k] = 0; } }
    // the final ring sizes are collected in a scratch array and moved to n_io at the end (n_io doubles as set 0's sizes)
    HIP_TRY(hipMemsetAsync(d_final_n, 0, batch * sizeof(int), h->stream));
    S.final_n = d_final_n;
    int err = 0;
    int it0 = 1;
    // ---- The first iters_min rounds as ONE launch.  A track's rounds depend on nothing but its own previous round, and no track can end before
    //      round iters_min, so nothing has to come back to the host in between; but launched round by round, every round ends with its slowest
    //      track -- a warm start that falls back to the cold path takes 8.7 ms where the median track takes 3 -- and the next round's launch
    //      waits for it with most compute units idle (third pass of the 1024 ovals: 6.5 ms of mean load, 11.5 ms of launch; neither another cap
    //      of the exchange nor the pass before says which tracks those will be: docs/NOTEBOOK.md R5.7).  In mcq_iqp_rounds_kernel every
    //      workgroup takes its track through those rounds on its own -- the bodies of the solver, bookkeeping and glue kernels between workgroup
    //      barriers, the same arrays --: nobody waits for anybody.  Results bitwise those of the round-by-round loop below, which takes over
    //      after round iters_min (and runs everything when per-round data must reach the host: timed statistics, the print_debug callback;
    //      $MCQ_IQP_FUSED=0: always). ----
    const char* fe = getenv("MCQ_IQP_FUSED");
    const bool fused = !(fe && fe[0] == '0') && !timed && !h->iqp_cb;
    if (fused) {
        const int ra = iters_min < max_rounds ? iters_min : max_rounds;
        McqIqpRounds F;
        memset(&F, 0, sizeof(F));
        F.B.batch = batch;
        F.B.n = nmax;
        F.B.nmax = nmax;
        F.B.alpha = alpha_out;
        F.B.curv_err = pass_curv;
        F.B.status = pass_status;
        F.B.kappa_bound = kappa_bound;
        F.B.w_veh = w_veh;
        if ((err = fill_batch(h, F.B, o, false)) != 0) return err;
        F.R.batch = batch;
        F.R.nmax = nmax;
        F.R.alpha = alpha_out;
        F.R.live = live;
        F.R.stepsize = stepsize_interp;
        F.R.status = rst;
        F.R.vec = h->vec;
        F.R.state_in = h->state;
        F.R.state_out = h->state2;
        F.S = S;
        F.S.live_count = live_count;
        for (int q = 0; q < 2; ++q) { F.n_set[q] = n_set[q]; F.ref_set[q] = ref_set[q]; F.nv_set[q] = nv_set[q]; }
        F.sc = scaling;
        F.warm = warm ? h->state2 : nullptr;
        F.rounds = ra;
        if (hipMemsetAsync(live_count, 0, sizeof(int), h->stream) != hipSuccess) err = MCQ_E_DEVICE;
        if (!err) {
            hipLaunchKernelGGL(mcq_iqp_rounds_kernel, dim3((unsigned)batch), dim3(256), 0, h->stream, F);
            if (hipGetLastError() != hipSuccess) err = MCQ_E_DEVICE;
        }
        // (the loop below may go on from the working sets the last round's glue carried over)
        h->state2_valid = true;
        h->state2_batch = batch;
        h->state2_nmax = nmax;
        if (!err && (hipMemcpyAsync(&n_live, live_count, sizeof(int), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                     hipStreamSynchronize(h->stream) != hipSuccess)) err = MCQ_E_DEVICE;
        cur = ra & 1;
        rounds = ra;
        it0 = ra + 1;
    }
    for (int it = it0; it <= max_rounds && n_live > 0 && !err; ++it) {
        rounds = it;
        McqBatch B;
        memset(&B, 0, sizeof(B));
        B.batch = batch;
        B.n = nmax;
        B.nmax = nmax;
        B.n_list = n_set[cur];
        B.ref = ref_set[cur];
        B.nv = nv_set[cur];
        B.sc = it == 1 ? scaling : nullptr;                 // the re-spline of passes 2+ uses unit scalings (upstream)
        B.alpha = alpha_out;
        B.curv_err = pass_curv;
        B.status = pass_status;
        B.info = timed ? h->d_info : nullptr;
        B.kappa_bound = kappa_bound;
        B.w_veh = w_veh;
        o.warm_start = (warm && it > 1) ? 1 : 0;
        if ((err = launch(h, B, o)) != 0) break;
        if (timed && it <= 16) {
            float ms[5];
            if (mcq_last_timing(h, ms) == 0) stats->solver_ms[it - 1] = ms[4];
            std::vector<mcq_info> info((size_t)batch);
            std::vector<int> lv((size_t)batch);
            if (hipMemcpyAsync(info.data(), h->d_info, batch * sizeof(mcq_info), hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
                hipMemcpyAsync(lv.data(), live, batch * sizeof(int), hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
                hipStreamSynchronize(h->stream) == hipSuccess) {
                int fb = 0;
                for (int k = 0; k < batch; ++k) if (lv[k] && (info[k].second_attempt & 2)) ++fb;
                stats->fallbacks[it - 1] = fb;
            }
        }
        if (h->iqp_cb) {        // print_debug: the pass's curvature errors and who ran it, before the termination test decides
            std::vector<double> cv((size_t)batch);
            std::vector<int> lv((size_t)batch);
            if (hipMemcpyAsync(cv.data(), pass_curv, batch * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipMemcpyAsync(lv.data(), live, batch * sizeof(int), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess) { err = MCQ_E_DEVICE; break; }
            h->iqp_cb(h->iqp_cb_user, it, batch, cv.data(), lv.data());
        }
        if (hipMemsetAsync(live_count, 0, sizeof(int), h->stream) != hipSuccess) { err = MCQ_E_DEVICE; break; }
        S.phase = 0;
        S.round = it;
        S.cur = cur;
        S.n_ring = n_set[cur];
        S.n_next = n_set[1 - cur];
        hipLaunchKernelGGL(mcq_iqp_step_kernel, sgrid, sblock, 0, h->stream, S);
        const double scale = it < iters_min ? (double)it / (double)iters_min : 1.0;
        if ((err = mcq_relinearise_device(h, batch, nmax, n_set[cur], ref_set[cur], nv_set[cur], alpha_out, live, scale,
                                          stepsize_interp, ref_set[1 - cur], nv_set[1 - cur], n_set[1 - cur], rst)) != 0) break;
        S.phase = 1;
        hipLaunchKernelGGL(mcq_iqp_step_kernel, sgrid, sblock, 0, h->stream, S);
        if (hipGetLastError() != hipSuccess) { err = MCQ_E_DEVICE; break; }
        cur = 1 - cur;
        // the first iters_min - 1 rounds cannot end a healthy track: no need to look
        if (it >= iters_min || it == max_rounds) {
            if (hipMemcpyAsync(&n_live, live_count, sizeof(int), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess) { err = MCQ_E_DEVICE; break; }
        }
    }
    if (!err && hipMemcpyAsync(n_io, d_final_n, batch * sizeof(int), hipMemcpyDeviceToDevice, h->stream) != hipSuccess) err = MCQ_E_DEVICE;
    if (hipStreamSynchronize(h->stream) != hipSuccess && !err) err = MCQ_E_DEVICE;
    h->state2_valid = false;        // the working sets belong to this run's last glue call, not to a later solve
    if (err) { if (err == MCQ_E_DEVICE && g_err.empty()) g_err = "mcq_iqp_device: HIP runtime error"; return err; }
    if (stats) {
        // QP passes summed over the tracks = sum of the rounds every track ran (the live count is only read back from round
        // iters_min on: counting launches x live tracks over-counted tracks that failed in an early round)
        // (a statistic: a failed read-back must not discard the run -- qp_solves = -1 then; tracks that came in empty ran no QP)
        std::vector<int> rv((size_t)batch), nf((size_t)batch);
        if (hipMemcpy(rv.data(), rounds_out, batch * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(nf.data(), n_io, batch * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int k = 0; k < batch; ++k) if (nf[k] > 0) solves += rv[k];
        } else solves = -1;
        stats->rounds = rounds;
        stats->qp_solves = (int)solves;
    }
    if (n_live > 0) {
        // tracks still iterating at max_rounds: report them (status MCQ_ITER_CAP), keep their last state
        std::vector<int> lv((size_t)batch), stv((size_t)batch);
        HIP_TRY(hipMemcpy(lv.data(), live, batch * sizeof(int), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(stv.data(), status_out, batch * sizeof(int), hipMemcpyDeviceToHost));
        for (int k = 0; k < batch; ++k) if (lv[k] && stv[k] == MCQ_OK) stv[k] = MCQ_ITER_CAP;
        HIP_TRY(hipMemcpy(status_out, stv.data(), batch * sizeof(int), hipMemcpyHostToDevice));
    }
    return 0;
}

// Packs the per-problem host buffers into the handle's pinned staging ([batch][nmax][*], zero padded) and queues the uploads.
// Layout of the pinned block: ref | nv | sc | kb | wv | n   (then, for the results: alpha | info).
struct PinLayout {
    double *ref, *nv, *sc, *kb, *wv, *alpha;
    int* n;
    mcq_info* info;
};

// after_chunk (round 6, mcq_solve_batch in slices): the uploads go to the handle's copy stream h->cs_in instead of its compute stream, in exactly
// `force_chunks` chunks of tracks, the per-problem scalars first, and after_chunk(k, b0, b1) is called as soon as chunk k's uploads are queued
// (it queues that slice's kernel behind them while the next chunk is packed).
static int pack_and_upload(mcq_handle* h, const mcq_problem* probs, int batch, size_t nmax, bool any_sc, bool with_nv,
                           size_t extra_bytes, PinLayout& P, const char* who, int force_chunks = 0,
                           const std::function<int(int, int, int)>& after_chunk = nullptr)
{
    const size_t elems = (size_t)batch * nmax;
    const size_t bytes = elems * (4 + 2 + 1 + 1) * sizeof(double) + (size_t)batch * (2 * sizeof(double) + sizeof(int) + sizeof(mcq_info)) + 64 + extra_bytes;
    int rc = ensure_pin(h, bytes);
    if (rc) return rc;
    char* base = (char*)h->pin;
    P.ref = (double*)base;
    P.nv = P.ref + elems * 4;
    P.sc = P.nv + elems * 2;
    P.alpha = P.sc + elems;
    P.kb = P.alpha + elems;
    P.wv = P.kb + batch;
    P.info = (mcq_info*)(P.wv + batch);
    P.n = (int*)(P.info + batch);
    // (the padding behind a track's n waypoints is never read by a kernel: no need to clear it)
    auto pack_range = [&](int b0, int b1) {
        for (int b = b0; b < b1; ++b) {
            const size_t n = (size_t)probs[b].n;
            memcpy(P.ref + (size_t)b * nmax * 4, probs[b].reftrack, n * 4 * sizeof(double));
            if (with_nv) memcpy(P.nv + (size_t)b * nmax * 2, probs[b].normvec, n * 2 * sizeof(double));
            if (any_sc) {
                if (probs[b].scaling) memcpy(P.sc + (size_t)b * nmax, probs[b].scaling, n * sizeof(double));
                else for (size_t q = 0; q < n; ++q) P.sc[(size_t)b * nmax + q] = 1.0;
            }
            P.kb[b] = probs[b].kappa_bound;
            P.wv[b] = probs[b].w_veh;
            P.n[b] = probs[b].n;
        }
    };
    // ---- no packing at all (round 6): a UNIFORM batch that lies in the caller's memory as one contiguous, page-locked block per array (what
    //      engine.py hands over for stacked arrays from mcq_host_alloc) goes to the device straight from there -- one strided copy per array
    //      (rows of n waypoints into rows of nmax) instead of a 115 MB memcpy into the staging first: that memcpy was 4-6 ms of a 36-42 ms
    //      mcq_iqp_batch call and the part of it that differs from host to host.  $MCQ_PACK_ALWAYS=1: the packing pass (A/B knob). ----
    {
        const int n0 = probs[0].n;
        bool direct = batch >= 2 && n0 > 0 && !getenv("MCQ_PACK_ALWAYS");
        for (int b = 1; b < batch && direct; ++b) {
            direct = probs[b].n == n0 && probs[b].reftrack == probs[0].reftrack + (size_t)b * n0 * 4
                     && (!with_nv || probs[b].normvec == probs[0].normvec + (size_t)b * n0 * 2)
                     && (probs[0].scaling ? probs[b].scaling == probs[0].scaling + (size_t)b * n0 : probs[b].scaling == nullptr);
        }
        auto pinned = [](const void* p) {
            hipPointerAttribute_t a;
            if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
            return a.type == hipMemoryTypeHost;
        };
        direct = direct && pinned(probs[0].reftrack) && (!with_nv || pinned(probs[0].normvec)) && (!probs[0].scaling || pinned(probs[0].scaling));
        h->last_upload_direct = direct ? 1 : 0;
        if (direct) {
            hipStream_t up = force_chunks > 0 ? h->cs_in : h->stream;
            for (int b = 0; b < batch; ++b) { P.kb[b] = probs[b].kappa_bound; P.wv[b] = probs[b].w_veh; P.n[b] = probs[b].n; }
            HIP_TRY_SYNC(hipMemcpyAsync(h->d_kb, P.kb, batch * sizeof(double), hipMemcpyHostToDevice, up));
            HIP_TRY_SYNC(hipMemcpyAsync(h->d_wv, P.wv, batch * sizeof(double), hipMemcpyHostToDevice, up));
            HIP_TRY_SYNC(hipMemcpyAsync(h->d_n, P.n, batch * sizeof(int), hipMemcpyHostToDevice, up));
            const size_t w = (size_t)n0 * sizeof(double), pitch = nmax * sizeof(double);
            const int nck = force_chunks > 0 ? std::min(force_chunks, batch) : 1;
            for (int ck = 0; ck < nck; ++ck) {
                const int b0 = (int)((long long)batch * ck / nck), b1 = (int)((long long)batch * (ck + 1) / nck);
                const size_t rows = (size_t)(b1 - b0);
                HIP_TRY_SYNC(hipMemcpy2DAsync(h->d_ref + (size_t)b0 * nmax * 4, pitch * 4, probs[b0].reftrack, w * 4, w * 4, rows, hipMemcpyHostToDevice, up));
                if (with_nv) HIP_TRY_SYNC(hipMemcpy2DAsync(h->d_nv + (size_t)b0 * nmax * 2, pitch * 2, probs[b0].normvec, w * 2, w * 2, rows, hipMemcpyHostToDevice, up));
                if (any_sc) HIP_TRY_SYNC(hipMemcpy2DAsync(h->d_sc + (size_t)b0 * nmax, pitch, probs[b0].scaling, w, w, rows, hipMemcpyHostToDevice, up));
                if (after_chunk) { if (int rc2 = after_chunk(ck, b0, b1)) return rc2; }
            }
            return 0;
        }
    }
    // A large batch (the 1024 N = 2000 tracks of the bench: 115 MB) is packed in four chunks of tracks by several host threads, and a
    // chunk's uploads are queued as soon as it is packed: one thread copies at ~10 GB/s -- 11 ms, a quarter of the whole mcq_iqp_batch
    // call -- and the DMA of chunk k runs while chunk k + 1 is packed.  $MCQ_PACK_THREADS: threads (default 8 for batches above 8 MB; 1 = the single loop).
    const size_t row_doubles = 4 + (with_nv ? 2 : 0) + (any_sc ? 1 : 0);
    const size_t payload = elems * row_doubles * sizeof(double);
    const char* pt = getenv("MCQ_PACK_THREADS");
    const bool forced = pt && *pt;          // (asked for explicitly: whatever the size of the batch -- the tests pack six small tracks on three threads)
    int nthreads = forced ? atoi(pt) : (payload > ((size_t)8 << 20) && batch >= 16 ? 8 : 1);
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && nthreads > hw) nthreads = hw;
    if (nthreads > batch) nthreads = batch;
    if (nthreads < 1) nthreads = 1;
    int nchunks = (forced && nthreads > 1) || (payload > ((size_t)32 << 20) && batch >= 16) ? 4 : 1;
    if (force_chunks > 0) nchunks = force_chunks;
    if (nchunks > batch) nchunks = batch;
    (void)who;
    hipStream_t up = force_chunks > 0 ? h->cs_in : h->stream;
    if (force_chunks > 0) {        // the slices' kernels read the per-problem scalars: first
        for (int b = 0; b < batch; ++b) { P.kb[b] = probs[b].kappa_bound; P.wv[b] = probs[b].w_veh; P.n[b] = probs[b].n; }
        HIP_TRY_SYNC(hipMemcpyAsync(h->d_kb, P.kb, batch * sizeof(double), hipMemcpyHostToDevice, up));
        HIP_TRY_SYNC(hipMemcpyAsync(h->d_wv, P.wv, batch * sizeof(double), hipMemcpyHostToDevice, up));
        HIP_TRY_SYNC(hipMemcpyAsync(h->d_n, P.n, batch * sizeof(int), hipMemcpyHostToDevice, up));
    }
    for (int ck = 0; ck < nchunks; ++ck) {
        const int b0 = (int)((long long)batch * ck / nchunks), b1 = (int)((long long)batch * (ck + 1) / nchunks);
        if (nthreads > 1) {
            auto sub = [&](int t) { return b0 + (int)((long long)(b1 - b0) * t / nthreads); };
            std::vector<std::thread> th;
            int taken = 1;          // sub-ranges 0 .. taken - 1 have a thread (0: this one)
            try {
                for (; taken < nthreads; ++taken) th.emplace_back(pack_range, sub(taken), sub(taken + 1));
            } catch (...) {}        // (no more threads to be had: their sub-ranges are packed here)
            pack_range(sub(0), sub(1));
            for (int t = taken; t < nthreads; ++t) pack_range(sub(t), sub(t + 1));
            for (auto& t : th) t.join();
        } else pack_range(b0, b1);
        const size_t off = (size_t)b0 * nmax, cnt = (size_t)(b1 - b0) * nmax;
        HIP_TRY_SYNC(hipMemcpyAsync(h->d_ref + off * 4, P.ref + off * 4, cnt * 4 * sizeof(double), hipMemcpyHostToDevice, up));
        if (with_nv) HIP_TRY_SYNC(hipMemcpyAsync(h->d_nv + off * 2, P.nv + off * 2, cnt * 2 * sizeof(double), hipMemcpyHostToDevice, up));
        if (any_sc) HIP_TRY_SYNC(hipMemcpyAsync(h->d_sc + off, P.sc + off, cnt * sizeof(double), hipMemcpyHostToDevice, up));
        if (after_chunk) { if (int rc2 = after_chunk(ck, b0, b1)) return rc2; }
    }
    if (force_chunks > 0) return 0;
    HIP_TRY_SYNC(hipMemcpyAsync(h->d_kb, P.kb, batch * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY_SYNC(hipMemcpyAsync(h->d_wv, P.wv, batch * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY_SYNC(hipMemcpyAsync(h->d_n, P.n, batch * sizeof(int), hipMemcpyHostToDevice, h->stream));
    return 0;
}

extern "C" int mcq_solve_batch(mcq_handle* h, const mcq_problem* probs, int batch, const mcq_opts* opts,
                               double* alpha_out, double* curv_err_out, int* status_out, mcq_info* info_out)
{
    if (!h || !probs || batch <= 0 || !alpha_out || !curv_err_out || !status_out) {
        g_err = "mcq_solve_batch: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    const mcq_opts o = resolve_opts(opts);
    size_t nmax = 3;
    bool any_sc = false;
    for (int b = 0; b < batch; ++b) {
        if (!probs[b].reftrack || probs[b].n < 0 || (probs[b].normvec == nullptr) != (probs[0].normvec == nullptr)) {
            g_err = "mcq_solve_batch: problem with NULL buffers (normvec must be given for all problems or for none)";
            return MCQ_E_ARG;
        }
        if ((size_t)probs[b].n > nmax) nmax = (size_t)probs[b].n;
        if (probs[b].scaling) any_sc = true;
    }
    int rc = ensure_ws(h, (size_t)batch, nmax);
    if (rc) return rc;
    rc = ensure_stage(h, (size_t)batch, nmax);
    if (rc) return rc;
    PinLayout P;
    McqBatch B;
    memset(&B, 0, sizeof(B));
    B.batch = batch;
    B.n = (int)nmax;
    B.nmax = (int)nmax;
    B.n_list = h->d_n;
    B.ref = h->d_ref;
    B.nv = probs[0].normvec ? h->d_nv : nullptr;      // none given: derived on the device (scalings too)
    B.sc = any_sc ? h->d_sc : nullptr;
    B.alpha = h->d_alpha;
    B.curv_err = h->d_curv;
    B.status = h->d_status;
    B.info = h->d_info;
    B.kappa_bound_list = h->d_kb;
    B.w_veh_list = h->d_wv;
    const int nsl = host_slices(batch, o);
    if (nsl > 1) {
        // in slices, as mcq_solve_host (see there): chunk k's kernel is queued behind its uploads while chunk k + 1 is packed; the padded alpha
        // rows come back slice by slice into the pinned staging
        rc = ensure_slice_streams(h);
        if (rc) return rc;
        rc = fill_batch(h, B, o, false);
        if (rc) return rc;
        B.warm = nullptr;
        h->state2_valid = false;
        h->timing_valid = false;
        PinLayout* Pp = &P;
        rc = pack_and_upload(h, probs, batch, nmax, any_sc, probs[0].normvec != nullptr, 0, P, "mcq_solve_batch", nsl,
                             [&](int k, int b0, int b1) -> int {
                                 HIP_TRY_SLICE(hipEventRecord(h->ev_slice[k], h->cs_in));
                                 hipStream_t st = slice_stream(h, k);
                                 HIP_TRY_SLICE(hipStreamWaitEvent(st, h->ev_slice[k], 0));
                                 McqBatch S = B;
                                 S.pb0 = b0;
                                 hipLaunchKernelGGL(mcq_solve_kernel, dim3(b1 - b0), dim3(256), 0, st, S);
                                 HIP_TRY_SLICE(hipGetLastError());
                                 HIP_TRY_SLICE(hipEventRecord(h->ev_slice[8 + k], st));
                                 HIP_TRY_SLICE(hipStreamWaitEvent(h->cs_out, h->ev_slice[8 + k], 0));
                                 const size_t off = (size_t)b0 * nmax, cnt = (size_t)(b1 - b0) * nmax;
                                 HIP_TRY_SLICE(hipMemcpyAsync(Pp->alpha + off, h->d_alpha + off, cnt * sizeof(double), hipMemcpyDeviceToHost, h->cs_out));
                                 return 0;
                             });
        if (rc) { (void)hipStreamSynchronize(h->cs_in); (void)hipStreamSynchronize(h->stream); (void)hipStreamSynchronize(h->stream2); (void)hipStreamSynchronize(h->cs_out); return rc; }
        HIP_TRY_SLICE(hipMemcpyAsync(curv_err_out, h->d_curv, batch * sizeof(double), hipMemcpyDeviceToHost, h->cs_out));
        HIP_TRY_SLICE(hipMemcpyAsync(status_out, h->d_status, batch * sizeof(int), hipMemcpyDeviceToHost, h->cs_out));
        HIP_TRY_SLICE(hipMemcpyAsync(P.info, h->d_info, batch * sizeof(mcq_info), hipMemcpyDeviceToHost, h->cs_out));
        HIP_TRY_SLICE(hipStreamSynchronize(h->cs_out));
        HIP_TRY_SLICE(hipStreamSynchronize(h->stream));
        HIP_TRY_SLICE(hipStreamSynchronize(h->stream2));
        size_t off2 = 0;
        for (int b = 0; b < batch; ++b) {
            const size_t nb = (size_t)probs[b].n;
            memcpy(alpha_out + off2, P.alpha + (size_t)b * nmax, nb * sizeof(double));
            off2 += nb;
            if (info_out) info_out[b] = P.info[b];
        }
        return 0;
    }
    rc = pack_and_upload(h, probs, batch, nmax, any_sc, probs[0].normvec != nullptr, 0, P, "mcq_solve_batch");
    if (rc) return rc;
    rc = launch(h, B, o);
    if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }

    HIP_TRY_SYNC(hipMemcpyAsync(P.alpha, h->d_alpha, (size_t)batch * nmax * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY_SYNC(hipMemcpyAsync(curv_err_out, h->d_curv, batch * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY_SYNC(hipMemcpyAsync(status_out, h->d_status, batch * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY_SYNC(hipMemcpyAsync(P.info, h->d_info, batch * sizeof(mcq_info), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    size_t off = 0;
    for (int b = 0; b < batch; ++b) {
        const size_t n = (size_t)probs[b].n;
        memcpy(alpha_out + off, P.alpha + (size_t)b * nmax, n * sizeof(double));
        off += n;
        if (info_out) info_out[b] = P.info[b];
    }
    return 0;
}

// ---- a stream of RESIDENT uniform batches on the handle's two compute streams (include/mcq.h) ---------------------------------------------
extern "C" int mcq_solve_device_stream(mcq_handle* h, int steps, int batch, int n, const double* const* reftrack, const double* const* normvec,
                                       const double* const* scaling, double kappa_bound, double w_veh, const mcq_opts* opts,
                                       double* const* alpha_out, double* const* curv_err_out, int* const* status_out)
{
    if (!h || steps <= 0 || batch <= 0 || n <= 0 || !reftrack || !alpha_out || !curv_err_out || !status_out) {
        g_err = "mcq_solve_device_stream: bad argument";
        return MCQ_E_ARG;
    }
    for (int k = 0; k < steps; ++k)
        if (!reftrack[k] || !alpha_out[k] || !curv_err_out[k] || !status_out[k]) { g_err = "mcq_solve_device_stream: NULL buffer in step list"; return MCQ_E_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    const mcq_opts o = resolve_opts(opts);
    int rc = ensure_ws(h, (size_t)batch, (size_t)n);
    if (rc) return rc;
    const bool two = steps > 1 && !getenv("MCQ_PIPE_ONE_STREAM") && o.algorithm != MCQ_ALG_GI;
    if (two) { rc = ensure_alt(h, (size_t)batch, (size_t)n); if (rc) return rc; }
    // the second stream starts behind whatever the caller has enqueued on the first (uploads of the inputs, an earlier call)
    if (two) {
        for (int k = 0; k < 2; ++k) if (!h->ev_st[k]) HIP_TRY(hipEventCreate(&h->ev_st[k]));
        HIP_TRY(hipEventRecord(h->ev_st[0], h->stream));
        HIP_TRY(hipStreamWaitEvent(h->stream2, h->ev_st[0], 0));
    }
    for (int k = 0; k < steps; ++k) {
        McqBatch B;
        memset(&B, 0, sizeof(B));
        B.batch = batch;
        B.n = n;
        B.nmax = n;
        B.ref = reftrack[k];
        B.nv = normvec ? normvec[k] : nullptr;
        B.sc = (scaling && B.nv) ? scaling[k] : nullptr;
        B.alpha = alpha_out[k];
        B.curv_err = curv_err_out[k];
        B.status = status_out[k];
        B.kappa_bound = kappa_bound;
        B.w_veh = w_veh;
        rc = launch(h, B, o, two && (k & 1));
        if (rc) { (void)hipStreamSynchronize(h->stream); if (h->stream2) (void)hipStreamSynchronize(h->stream2); return rc; }
    }
    // the handle's first stream ends behind the second one: mcq_sync / a later call on the handle sees every step done
    if (two) {
        HIP_TRY(hipEventRecord(h->ev_st[1], h->stream2));
        HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_st[1], 0));
    }
    return 0;
}

extern "C" int mcq_iqp_set_round_callback(mcq_handle* h, mcq_iqp_round_cb cb, void* user)
{
    if (!h) { g_err = "mcq_iqp_set_round_callback: NULL handle"; return MCQ_E_ARG; }
    h->iqp_cb = cb;
    h->iqp_cb_user = cb ? user : nullptr;
    return 0;
}

extern "C" int mcq_iqp_batch(mcq_handle* h, const mcq_problem* probs, int batch, double stepsize_interp, int iters_min,
                             double curv_error_allowed, int max_rounds, const mcq_opts* opts, int nmax_out, double* alpha_out,
                             double* reftrack_out, double* normvec_out, int* n_out, double* curv_err_out, int* status_out,
                             int* rounds_out, double* curv_trace_out, mcq_iqp_stats* stats)
{
    if (!h || !probs || batch <= 0 || nmax_out < 3 || !alpha_out || !reftrack_out || !normvec_out || !n_out || !curv_err_out ||
        !status_out || !rounds_out) {
        g_err = "mcq_iqp_batch: bad argument";
        return MCQ_E_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    const bool trace = getenv("MCQ_IQP_TRACE") != nullptr;          // host-side time stamps of this call on stderr
    struct timespec ts0;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    auto stamp = [&](const char* what) {
        if (!trace) return;
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        fprintf(stderr, "iqp_batch: %-28s at %.3f ms\n", what, (t1.tv_sec - ts0.tv_sec) * 1e3 + (t1.tv_nsec - ts0.tv_nsec) * 1e-6);
    };
    const size_t nmax = (size_t)nmax_out;
    bool any_sc = false;
    for (int b = 0; b < batch; ++b) {
        if (!probs[b].reftrack || !probs[b].normvec || probs[b].n < 0) { g_err = "mcq_iqp_batch: reftrack and normvec are required"; return MCQ_E_ARG; }
        if ((size_t)probs[b].n > nmax) { g_err = "mcq_iqp_batch: a track has more waypoints than nmax_out"; return MCQ_E_TOO_LARGE; }
        if (probs[b].kappa_bound != probs[0].kappa_bound || probs[b].w_veh != probs[0].w_veh) {
            g_err = "mcq_iqp_batch: one kappa_bound / w_veh per call";
            return MCQ_E_ARG;
        }
        if (probs[b].scaling) any_sc = true;
    }
    int rc = ensure_ws(h, (size_t)batch, nmax);
    if (rc) return rc;
    rc = ensure_stage(h, (size_t)batch, nmax);
    if (rc) return rc;
    const size_t elems = (size_t)batch * nmax;
    if (elems > h->stage2_elems) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_ref2); (void)hipFree(h->d_nv2);
        h->d_ref2 = h->d_nv2 = nullptr; h->stage2_elems = 0;
        HIP_TRY(hipMalloc((void**)&h->d_ref2, elems * 4 * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&h->d_nv2, elems * 2 * sizeof(double)));
        h->stage2_elems = elems;
    }
    PinLayout P;
    // results need room for both buffer sets' ref / nv in the worst case: one extra set
    stamp("checked, buffers ready");
    rc = pack_and_upload(h, probs, batch, nmax, any_sc, true, elems * 6 * sizeof(double) + (size_t)batch * 3 * sizeof(int), P, "mcq_iqp_batch");
    if (rc) return rc;
    stamp("packed, uploads queued");
    rc = ensure_iqp(h, (size_t)batch);                 // the trace staging lives in the handle (no allocation / hipFree -- an implicit
    if (rc) return rc;                                 // device-wide synchronisation -- per call)
    double* d_trace = curv_trace_out ? h->d_trace : nullptr;
    // d_kb / d_wv staging doubles as the per-track outputs of the loop: curvature error (double), buffer index / rounds (ints)
    int* d_buf = (int*)h->d_kb;                 // batch doubles >= 2 * batch ints
    int* d_rounds = d_buf + batch;
    rc = mcq_iqp_device(h, batch, (int)nmax, h->d_n, h->d_ref, h->d_nv, h->d_ref2, h->d_nv2, any_sc ? h->d_sc : nullptr,
                        probs[0].kappa_bound, probs[0].w_veh, stepsize_interp, iters_min, curv_error_allowed, max_rounds, opts,
                        h->d_alpha, d_buf, h->d_curv, h->d_status, d_rounds, d_trace, stats);
    stamp("rounds done");
    if (!rc && d_trace && hipMemcpy(curv_trace_out, d_trace, (size_t)batch * MCQ_IQP_TRACE * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
        g_err = "mcq_iqp_batch: read-back of the curvature-error trace failed";
        rc = MCQ_E_DEVICE;
    }
    if (rc) return rc;
    int* buf_h = P.n;                            // the pinned int block: n | (extra) buf, rounds
    HIP_TRY_SYNC(hipMemcpyAsync(n_out, h->d_n, batch * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY_SYNC(hipMemcpyAsync(buf_h, d_buf, batch * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY_SYNC(hipMemcpyAsync(rounds_out, d_rounds, batch * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY_SYNC(hipMemcpyAsync(curv_err_out, h->d_curv, batch * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY_SYNC(hipMemcpyAsync(status_out, h->d_status, batch * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    // alpha straight into the caller's [batch][nmax_out] array (its padding holds whatever the device buffer held)
    HIP_TRY_SYNC(hipMemcpyAsync(alpha_out, h->d_alpha, elems * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    stamp("alpha and per-track results down");
    bool use[2] = {false, false};
    for (int b = 0; b < batch; ++b) use[buf_h[b] ? 1 : 0] = true;
    const double* d_ref_set[2] = {h->d_ref, h->d_ref2};
    const double* d_nv_set[2] = {h->d_nv, h->d_nv2};
    if (use[0] != use[1]) {
        // the usual case (one iters_min for all: every track ends in the same round): the final set goes out in two copies
        const int q = use[1] ? 1 : 0;
        HIP_TRY_SYNC(hipMemcpyAsync(reftrack_out, d_ref_set[q], elems * 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY_SYNC(hipMemcpyAsync(normvec_out, d_nv_set[q], elems * 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        stamp("rings and normals down");
        return 0;
    }
    // tracks ended in different rounds: both sets through the pinned staging (set 0 into the now free input block, set 1 into
    // the extra block behind it), then track by track
    double* ref_h[2] = {P.ref, nullptr};
    double* nv_h[2] = {P.nv, nullptr};
    {
        size_t addr = (size_t)(P.n + batch + 16);
        addr = (addr + 7) & ~(size_t)7;                 // keep the extra block 8-byte aligned
        ref_h[1] = (double*)addr;
        nv_h[1] = ref_h[1] + elems * 4;
    }
    for (int q = 0; q < 2; ++q) {
        HIP_TRY_SYNC(hipMemcpyAsync(ref_h[q], d_ref_set[q], elems * 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY_SYNC(hipMemcpyAsync(nv_h[q], d_nv_set[q], elems * 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int b = 0; b < batch; ++b) {
        const int q = buf_h[b] ? 1 : 0;
        const size_t n = n_out[b] > 0 ? (size_t)n_out[b] : 0;
        memcpy(reftrack_out + (size_t)b * nmax * 4, ref_h[q] + (size_t)b * nmax * 4, n * 4 * sizeof(double));
        memcpy(normvec_out + (size_t)b * nmax * 2, nv_h[q] + (size_t)b * nmax * 2, n * 2 * sizeof(double));
    }
    return 0;
}


// =====================================================================================================================
// The one collective of a multi-GPU job (include/mcq.h): ncclAllGather of RCCL on the handle's COMM stream, ordered behind its compute stream by an event.  RCCL is loaded with dlopen on
// first use -- libmcq.so has no link-time dependency on it, and a single-GPU process never maps its 570 MB.  Only the five entry
// points below are bound; their types are restated here (rccl.h: ncclUniqueId = 128 opaque bytes, ncclComm_t an opaque pointer,
// ncclDataType_t: ncclInt32 = 2, ncclFloat32 = 7, ncclFloat64 = 8; ncclSuccess = 0).
// =====================================================================================================================
namespace {
struct RcclId { char internal[MCQ_COMM_ID_BYTES]; };
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string path;
};
Rccl g_rccl;

int rccl_load()
{
    if (g_rccl.lib) return 0;
    std::vector<std::string> cand;
    if (const char* e = getenv("MCQ_RCCL_LIB")) cand.push_back(e);
    if (const char* e = getenv("ROCM_PATH")) cand.push_back(std::string(e) + "/lib/librccl.so.1");
    cand.push_back("/opt/rocm/lib/librccl.so.1");
    cand.push_back("librccl.so.1");
    std::string tried;
    for (const std::string& p : cand) {
        void* lib = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!lib) { tried += p + " (" + (dlerror() ? dlerror() : "?") + "); "; continue; }
        g_rccl.GetUniqueId = (int (*)(RcclId*))dlsym(lib, "ncclGetUniqueId");
        g_rccl.CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(lib, "ncclCommInitRank");
        g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(lib, "ncclAllGather");
        g_rccl.CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
        g_rccl.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
        if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy) {
            tried += p + " (symbols missing); ";
            dlclose(lib);
            continue;
        }
        g_rccl.lib = lib;
        g_rccl.path = p;
        return 0;
    }
    g_err = "mcq_comm: RCCL could not be loaded: " + tried;
    return MCQ_E_DEVICE;
}

int rccl_fail(const char* what, int rc)
{
    g_err = std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?") + " (" + g_rccl.path + ")";
    return MCQ_E_DEVICE;
}
}   // namespace

extern "C" int mcq_comm_unique_id(unsigned char id_out[MCQ_COMM_ID_BYTES])
{
    if (!id_out) { g_err = "mcq_comm_unique_id: id_out is NULL"; return MCQ_E_ARG; }
    int rc = rccl_load();
    if (rc) return rc;
    RcclId id;
    const int r = g_rccl.GetUniqueId(&id);
    if (r != 0) return rccl_fail("ncclGetUniqueId", r);
    memcpy(id_out, id.internal, MCQ_COMM_ID_BYTES);
    return 0;
}

extern "C" int mcq_comm_init(mcq_handle* h, int rank, int world, const unsigned char id[MCQ_COMM_ID_BYTES])
{
    if (!h || !id || world < 1 || rank < 0 || rank >= world) { g_err = "mcq_comm_init: bad argument"; return MCQ_E_ARG; }
    if (h->comm) { g_err = "mcq_comm_init: the handle already has a communicator"; return MCQ_E_ARG; }
    int rc = rccl_load();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if (!h->comm_stream) {
        HIP_TRY(hipStreamCreate(&h->comm_stream));
        HIP_TRY(hipEventCreate(&h->comm_ready));
        for (int k = 0; k < 4; ++k) { HIP_TRY(hipEventCreate(&h->comm_t0[k])); HIP_TRY(hipEventCreate(&h->comm_done[k])); }
    }
    RcclId uid;
    memcpy(uid.internal, id, MCQ_COMM_ID_BYTES);
    void* comm = nullptr;
    const int r = g_rccl.CommInitRank(&comm, world, uid, rank);
    if (r != 0) return rccl_fail("ncclCommInitRank", r);
    h->comm = comm;
    h->comm_rank = rank;
    h->comm_world = world;
    h->comm_seq = 0;
    return 0;
}

extern "C" int mcq_comm_allgather(mcq_handle* h, const void* send, void* recv, size_t count, int dtype)
{
    if (!h || !send || !recv || dtype < MCQ_DT_F64 || dtype > MCQ_DT_I32) { g_err = "mcq_comm_allgather: bad argument"; return MCQ_E_ARG; }
    if (!h->comm) { g_err = "mcq_comm_allgather: mcq_comm_init has not been called on this handle"; return MCQ_E_ARG; }
    if (count == 0) return 0;
    HIP_TRY(hipSetDevice(h->device));
    // behind everything enqueued on the handle's stream so far (the solve that filled `send`), on a stream of its own: the handle's
    // stream goes on with the next solve while the gather runs
    HIP_TRY(hipEventRecord(h->comm_ready, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->comm_stream, h->comm_ready, 0));
    const unsigned slot = h->comm_seq & 3u;
    HIP_TRY(hipEventRecord(h->comm_t0[slot], h->comm_stream));
    const int nccl_t = dtype == MCQ_DT_F64 ? 8 : dtype == MCQ_DT_F32 ? 7 : 2;
    const int r = g_rccl.AllGather(send, recv, count, nccl_t, h->comm, h->comm_stream);
    if (r != 0) return rccl_fail("ncclAllGather", r);
    HIP_TRY(hipEventRecord(h->comm_done[slot], h->comm_stream));
    ++h->comm_seq;
    return 0;
}

extern "C" int mcq_comm_wait(mcq_handle* h, int lag, float* ms_out)
{
    if (!h || lag < 0 || lag > 2) { g_err = "mcq_comm_wait: bad argument"; return MCQ_E_ARG; }
    if (ms_out) *ms_out = 0.0f;
    if (!h->comm || h->comm_seq <= (unsigned)lag) return 0;          // nothing (that old) in flight
    HIP_TRY(hipSetDevice(h->device));
    const unsigned slot = (h->comm_seq - 1u - (unsigned)lag) & 3u;
    HIP_TRY(hipEventSynchronize(h->comm_done[slot]));
    if (ms_out) HIP_TRY(hipEventElapsedTime(ms_out, h->comm_t0[slot], h->comm_done[slot]));
    return 0;
}

extern "C" int mcq_comm_world(mcq_handle* h, int* rank_out, int* world_out)
{
    if (!h || !h->comm) { g_err = "mcq_comm_world: no communicator"; return MCQ_E_ARG; }
    if (rank_out) *rank_out = h->comm_rank;
    if (world_out) *world_out = h->comm_world;
    return 0;
}

extern "C" int mcq_comm_destroy(mcq_handle* h)
{
    if (!h) return MCQ_E_ARG;
    int ret = 0;
    (void)hipSetDevice(h->device);
    if (h->comm) {
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
        const int r = g_rccl.CommDestroy(h->comm);
        h->comm = nullptr;
        h->comm_rank = h->comm_world = 0;
        if (r != 0) ret = rccl_fail("ncclCommDestroy", r);
    }
    if (h->comm_stream) {
        (void)hipEventDestroy(h->comm_ready);
        for (int k = 0; k < 4; ++k) { (void)hipEventDestroy(h->comm_t0[k]); (void)hipEventDestroy(h->comm_done[k]); }
        (void)hipStreamDestroy(h->comm_stream);
        h->comm_stream = nullptr;
        h->comm_ready = nullptr;
        for (int k = 0; k < 4; ++k) h->comm_t0[k] = h->comm_done[k] = nullptr;
    }
    h->comm_seq = 0;      // no gather in flight any more: mcq_copy_to_host / mcq_device_free must not wait on the destroyed events
    return ret;
}
