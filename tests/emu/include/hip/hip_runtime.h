// TEST INFRASTRUCTURE ONLY -- a single-threaded SIMT interpreter for the HIP subset used by csrc/*.hip.
//
// The development container has no GPU; this header lets the *unchanged* kernel sources of
// global_racetrajectory_optimization_amd/csrc/ be compiled with g++ (tests/emu/build_emu.sh puts this directory in
// front of the include path, so `#include <hip/hip_runtime.h>` resolves here) and executed workgroup by workgroup with
// one ucontext fiber per work-item, so that index arithmetic, barrier placement and wave-level data flow of the
// kernels are exercised by `pytest -m "not gpu"`.  It is NOT a backend: the product library libmcq.so is built by
// hipcc for gfx950 only and engine.py loads nothing else; nothing under tests/emu is importable from the package.
//
// Semantics kept: 64-wide waves; __syncthreads() as a workgroup barrier; __shfl/__shfl_xor exchange through a
// per-wave slot array with wave-level rendezvous (a lane that skips a shuffle its wave executes deadlocks the
// interpreter -> reported as an error, which is exactly the bug it would be on hardware); `__shared__` = one static
// instance (workgroups run one after another); dynamic LDS via HIP_DYNAMIC_SHARED.
#pragma once
#include <ucontext.h>
#include <chrono>
inline double hipemu_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

using std::isfinite;
inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

namespace hipemu {
struct Idx { unsigned x, y, z; };
inline Idx& tidx() { static Idx v; return v; }
inline Idx& bidx() { static Idx v; return v; }
inline Idx& bdim() { static Idx v; return v; }
inline Idx& gdim() { static Idx v; return v; }

// Fiber switch.  glibc's swapcontext saves and restores the signal mask with one system call per switch -- hundreds of millions of them per test
// run, half the interpreter's time.  On x86-64 the switch is done by hand instead (callee-saved registers, mxcsr and the x87 control word on the
// fiber's own stack, then the stack pointer is exchanged); other hosts keep ucontext.
#if defined(__x86_64__)
#define HIPEMU_ASM_SWITCH 1
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
__asm__(".text\n"
        ".weak hipemu_switch\n"
        ".hidden hipemu_switch\n"
        ".type hipemu_switch,@function\n"
        "hipemu_switch:\n"
        "    pushq %rbp\n"
        "    pushq %rbx\n"
        "    pushq %r12\n"
        "    pushq %r13\n"
        "    pushq %r14\n"
        "    pushq %r15\n"
        "    subq $8, %rsp\n"
        "    stmxcsr (%rsp)\n"
        "    fnstcw 4(%rsp)\n"
        "    movq %rsp, (%rdi)\n"
        "    movq %rsi, %rsp\n"
        "    ldmxcsr (%rsp)\n"
        "    fldcw 4(%rsp)\n"
        "    addq $8, %rsp\n"
        "    popq %r15\n"
        "    popq %r14\n"
        "    popq %r13\n"
        "    popq %r12\n"
        "    popq %rbx\n"
        "    popq %rbp\n"
        "    ret\n"
        ".size hipemu_switch, .-hipemu_switch\n");
#endif

struct State {
#ifdef HIPEMU_ASM_SWITCH
    void* main_sp = nullptr;
    std::vector<void*> sp;
#else
    ucontext_t main_ctx;
    std::vector<ucontext_t> ctx;
#endif
    std::vector<char*> stacks;
    std::vector<char> done;
    int nt = 0, cur = 0, alive = 0;
    int bar_count = 0;
    unsigned bar_gen = 0;
    std::vector<int> wbar_count;
    std::vector<unsigned> wbar_gen;
    std::vector<double> slot_d, slot_e;
    std::vector<long long> slot_i;
    std::vector<char> dyn_smem;
    std::function<void()> body;
    long long switches = 0;
    bool deadlock = false;
};
inline State& st() { static State s; return s; }

inline void to_main(State& s)
{
#ifdef HIPEMU_ASM_SWITCH
    hipemu_switch(&s.sp[s.cur], s.main_sp);
#else
    swapcontext(&s.ctx[s.cur], &s.main_ctx);
#endif
}

inline void yield()
{
    State& s = st();
    ++s.switches;
    to_main(s);
}

inline void block_barrier()
{
    State& s = st();
    const unsigned gen = s.bar_gen;
    if (++s.bar_count == s.alive) { s.bar_count = 0; ++s.bar_gen; return; }
    while (s.bar_gen == gen) yield();
}

inline void wave_barrier()
{
    State& s = st();
    const int w = s.cur >> 6;
    int wave_size = s.nt - (w << 6);
    if (wave_size > 64) wave_size = 64;
    const unsigned gen = s.wbar_gen[w];
    if (++s.wbar_count[w] == wave_size) { s.wbar_count[w] = 0; ++s.wbar_gen[w]; return; }
    while (s.wbar_gen[w] == gen) yield();
}

inline void trampoline()
{
    State& s = st();
    s.body();
    s.done[s.cur] = 1;
    --s.alive;
    to_main(s);          // (never resumed)
    abort();
}

inline void* dyn_smem_ptr() { return st().dyn_smem.data(); }

template <typename F>
inline void run_block(F&& f, unsigned nt, size_t smem)
{
    State& s = st();
    const size_t STACK = 256 * 1024;
    if ((int)s.stacks.size() < (int)nt) {
        for (size_t k = s.stacks.size(); k < nt; ++k) s.stacks.push_back((char*)malloc(STACK));
    }
#ifdef HIPEMU_ASM_SWITCH
    s.sp.resize(nt);
#else
    s.ctx.resize(nt);
#endif
    s.done.assign(nt, 0);
    s.nt = (int)nt;
    s.alive = (int)nt;
    s.bar_count = 0;
    s.wbar_count.assign((nt + 63) / 64, 0);
    s.wbar_gen.assign((nt + 63) / 64, 0);
    s.slot_d.assign(nt, 0.0);
    s.slot_e.assign(nt, 0.0);
    s.slot_i.assign(nt, 0);
    if (s.dyn_smem.size() < smem + 64) s.dyn_smem.resize(smem + 64);
    s.body = f;
#ifdef HIPEMU_ASM_SWITCH
    unsigned int mxcsr;
    unsigned short fpcw;
    __asm__ volatile("stmxcsr %0" : "=m"(mxcsr));
    __asm__ volatile("fnstcw %0" : "=m"(fpcw));
    for (unsigned t = 0; t < nt; ++t) {
        // the frame hipemu_switch pops on the first switch into this fiber: control words, r15 r14 r13 r12 rbx rbp, then `ret` into
        // trampoline with the stack pointer where a call would have left it (16-byte boundary + 8)
        void** p = (void**)((size_t)(s.stacks[t] + STACK) & ~(size_t)15);
        *--p = nullptr;                            // the return address trampoline never uses
        *--p = (void*)(void (*)())trampoline;
        for (int k = 0; k < 6; ++k) *--p = nullptr;
        --p;
        ((unsigned int*)p)[0] = mxcsr;
        ((unsigned int*)p)[1] = fpcw;
        s.sp[t] = (void*)p;
    }
#else
    for (unsigned t = 0; t < nt; ++t) {
        getcontext(&s.ctx[t]);
        s.ctx[t].uc_stack.ss_sp = s.stacks[t];
        s.ctx[t].uc_stack.ss_size = STACK;
        s.ctx[t].uc_link = &s.main_ctx;
        makecontext(&s.ctx[t], (void (*)())trampoline, 0);
    }
#endif
    long long idle_rounds = 0;
    while (s.alive > 0) {
        const long long sw0 = s.switches;
        const int alive0 = s.alive;
        const unsigned g0 = s.bar_gen;
        // HIPEMU_REVERSE=1: the work-items of a block take their turns in reverse order (the LAST wave runs a phase first): an
        // ordering bug between waves that the natural order hides shows up deterministically
        static const bool reverse = getenv("HIPEMU_REVERSE") && getenv("HIPEMU_REVERSE")[0] == '1';
        for (unsigned tt = 0; tt < nt; ++tt) {
            const unsigned t = reverse ? nt - 1 - tt : tt;
            if (s.done[t]) continue;
            s.cur = (int)t;
            tidx().x = t;
#ifdef HIPEMU_ASM_SWITCH
            hipemu_switch(&s.main_sp, s.sp[t]);
#else
            swapcontext(&s.main_ctx, &s.ctx[t]);
#endif
        }
        (void)sw0;
        // progress detection: a full round in which nobody finished and no barrier generation advanced, repeated
        bool progressed = (s.alive != alive0) || (s.bar_gen != g0);
        if (!progressed) {
            unsigned wsum = 0;
            for (unsigned g : s.wbar_gen) wsum += g;
            static unsigned last_wsum = 0;
            if (wsum != last_wsum) { progressed = true; last_wsum = wsum; }
        }
        idle_rounds = progressed ? 0 : idle_rounds + 1;
        if (idle_rounds > 4) {
            fprintf(stderr, "hipemu: deadlock (divergent barrier / shuffle) in block (%u,%u)\n", bidx().x, bidx().y);
            s.deadlock = true;
            abort();
        }
    }
}
}  // namespace hipemu

#define threadIdx (hipemu::tidx())
#define blockIdx (hipemu::bidx())
#define blockDim (hipemu::bdim())
#define gridDim (hipemu::gdim())
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)hipemu::dyn_smem_ptr();

inline void __syncthreads() { hipemu::block_barrier(); }

inline double __shfl(double v, int src)
{
    hipemu::State& s = hipemu::st();
    const int t = s.cur;
    s.slot_d[t] = v;
    hipemu::wave_barrier();
    const double r = s.slot_d[(t & ~63) | (src & 63)];
    hipemu::wave_barrier();
    return r;
}
inline double __shfl_xor(double v, int m)
{
    hipemu::State& s = hipemu::st();
    return __shfl(v, (s.cur ^ m) & 63);
}
inline int __shfl(int v, int src) { return (int)__shfl((double)v, src); }
inline unsigned long long __ballot(int pred)
{
    hipemu::State& s = hipemu::st();
    const int t = s.cur, base = t & ~63;
    s.slot_i[t] = pred ? 1 : 0;
    hipemu::wave_barrier();
    unsigned long long m = 0;
    const int wave_size = (s.nt - base) > 64 ? 64 : (s.nt - base);
    for (int l = 0; l < wave_size; ++l) if (s.slot_i[base + l]) m |= 1ull << l;
    hipemu::wave_barrier();
    return m;
}


// v_permlane32_swap / v_permlane16_swap (gfx950): [0] = new vdst, [1] = new src0.
//   32: vdst lanes 32..63 <-> src0 lanes 0..31;   16: vdst rows 1,3 (lanes 16..31, 48..63) <-> src0 rows 0,2
struct hipemu_u2 {
    unsigned v[2];
    unsigned operator[](int i) const { return v[i]; }
};
inline hipemu_u2 hipemu_permlane_swap(unsigned d, unsigned s, int width)
{
    const int lane = hipemu::st().cur & 63;
    // the partner lane of the *other* operand
    const unsigned s_from = (unsigned)(long long)__shfl((double)s, lane - width);   // src0[lane - width] (meaningful for the upper half)
    const unsigned d_from = (unsigned)(long long)__shfl((double)d, lane + width);   // vdst[lane + width] (meaningful for the lower half)
    const bool upper = (lane & width) != 0;
    hipemu_u2 r;
    r.v[0] = upper ? s_from : d;
    r.v[1] = upper ? s : d_from;
    return r;
}
#define __builtin_amdgcn_permlane32_swap(d, s, fi, bc) hipemu_permlane_swap((unsigned)(d), (unsigned)(s), 32)
#define __builtin_amdgcn_permlane16_swap(d, s, fi, bc) hipemu_permlane_swap((unsigned)(d), (unsigned)(s), 16)

inline long long wall_clock64() { return (long long)(hipemu_now() * 1e5); }
inline long long clock64() { return (long long)(hipemu_now() * 1e6); }

// v_mfma_f64_16x16x4_f64: D(16x16) = A(16x4) B(4x16) + C.  Lane l holds A[l&15][l>>4], B[l>>4][l&15]; C/D register r of lane l
// is element (row (l>>4) + 4r, col l&15)   (/opt/skills/guides/cdna_hip_programming.md, section 3, f64 layout).
typedef double hipemu_v4d __attribute__((vector_size(32)));
inline hipemu_v4d hipemu_mfma_f64_16x16x4(double a, double b, hipemu_v4d c)
{
    hipemu::State& s = hipemu::st();
    const int t = s.cur, base = t & ~63, l = t & 63;
    s.slot_d[t] = a;
    s.slot_e[t] = b;
    hipemu::wave_barrier();
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) + 4 * r, col = l & 15;
        double acc = 0.0;
        for (int k = 0; k < 4; ++k) acc += s.slot_d[base + row + 16 * k] * s.slot_e[base + col + 16 * k];
        c[r] += acc;
    }
    hipemu::wave_barrier();
    return c;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) hipemu_mfma_f64_16x16x4((a), (b), (c))

// AMDGCN builtins used by the kernels
// DPP moves as the kernels use them: row_newbcast:n (dpp_ctrl 0x150 + n): lane n of every 16-lane row to all lanes of that row;
// row_shl:n (0x100 + n): lane i of a row takes lane i + n of the same row.  Lanes whose row / bank (4 lanes) is not enabled by row_mask /
// bank_mask keep `old`; a source beyond the row gives 0 with bound_ctrl, `old` without.
inline double hipemu_update_dpp(double old, double src, int ctrl, int rm, int bm, bool bc)
{
    const int lane = hipemu::st().cur & 63;
    const bool enabled = ((rm >> (lane >> 4)) & 1) && ((bm >> ((lane & 15) >> 2)) & 1);
    if (ctrl >= 0x101 && ctrl <= 0x10f) {
        const int from = (lane & 15) + (ctrl - 0x100);
        const double got = __shfl(src, (lane & ~15) | (from & 15));
        if (!enabled) return old;
        return from < 16 ? got : (bc ? 0.0 : old);
    }
    if (ctrl < 0x150 || ctrl > 0x15f) { fprintf(stderr, "hipemu: unsupported dpp_ctrl 0x%x\n", ctrl); abort(); }
    const double got = __shfl(src, (lane & ~15) | (ctrl - 0x150));
    return enabled ? got : old;
}
#define __builtin_amdgcn_rcp(x) (1.0 / (double)(x))              /* v_rcp_f64: the kernels refine it */
#define __builtin_amdgcn_rsq(x) (1.0 / sqrt((double)(x)))      /* v_rsq_f64: the kernels refine it */
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipemu_update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
// v_fmac_f64_dpp ... -b row_newbcast:N, which the kernels emit by name (mcq_kernels.hip: fnma_bcast_row16)
#define MCQ_HAVE_FNMA_BCAST_ROW16 1
template <int N> inline double fnma_bcast_row16(double a, double b, double c) { return fma(-hipemu_update_dpp(0.0, a, 0x150 + N, 0xf, 0xf, true), b, c); }
inline int hipemu_readlane(int v, int src) { return (int)__shfl((double)v, src); }
#define __builtin_amdgcn_readlane(v, l) hipemu_readlane((v), (l))
#define __builtin_amdgcn_s_barrier() hipemu::block_barrier()
#define __builtin_amdgcn_readfirstlane(v) (v)     /* only used on wave-uniform values */
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, size, id) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_s_waitcnt(n) ((void)0)
#define MCQ_PIN_SVV(sreg, vreg0, vreg1) ((void)0)
#define MCQ_PIN_SV(sreg, vreg) ((void)0)
#define __builtin_amdgcn_wave_barrier() hipemu::wave_barrier()
#define __builtin_amdgcn_fence(...) ((void)0)
inline int __double2loint(double d) { long long b; memcpy(&b, &d, 8); return (int)(b & 0xffffffffLL); }
inline int __double2hiint(double d) { long long b; memcpy(&b, &d, 8); return (int)((b >> 32) & 0xffffffffLL); }
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
inline double __hiloint2double(int hi, int lo)
{
    long long b = ((long long)(unsigned)hi << 32) | (unsigned)lo;
    double d; memcpy(&d, &b, 8); return d;
}
inline int __shfl_xor(int v, int m) { return (int)__shfl_xor((double)v, m); }
inline int atomicAdd(int* p, int v) { const int o = *p; *p = o + v; return o; }    /* one work-item runs at a time */
inline int atomicCAS(int* p, int cmp, int v) { const int o = *p; if (o == cmp) *p = v; return o; }
inline int atomicExch(int* p, int v) { const int o = *p; *p = v; return o; }
inline void __threadfence() {}

// ---- host runtime subset -------------------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
typedef void* hipStream_t;
struct hipemu_event { double t; };
typedef hipemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };

inline const char* hipGetErrorString(hipError_t) { return "hipemu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (void*)1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
enum { hipHostMallocDefault = 0 };
/* page-locked host memory: the interpreter keeps the ranges, so that hipPointerGetAttributes can tell them from pageable memory (the library
 * uploads a uniform batch straight from pinned arrays, without its packing pass) */
inline std::vector<std::pair<char*, size_t>>& hipemu_pinned_() { static std::vector<std::pair<char*, size_t>> v; return v; }
inline std::mutex& hipemu_pinned_mutex_() { static std::mutex m; return m; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned)
{
    *p = calloc(n ? n : 1, 1);
    if (!*p) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> g(hipemu_pinned_mutex_());
    hipemu_pinned_().push_back({(char*)*p, n ? n : 1});
    return hipSuccess;
}
inline hipError_t hipHostFree(void* p)
{
    {
        std::lock_guard<std::mutex> g(hipemu_pinned_mutex_());
        auto& v = hipemu_pinned_();
        for (size_t i = 0; i < v.size(); ++i) if (v[i].first == (char*)p) { v.erase(v.begin() + i); break; }
    }
    free(p);
    return hipSuccess;
}
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p)
{
    std::lock_guard<std::mutex> g(hipemu_pinned_mutex_());
    for (auto& r : hipemu_pinned_())
        if ((const char*)p >= r.first && (const char*)p < r.first + r.second) { a->type = hipMemoryTypeHost; a->device = 0; a->devicePointer = a->hostPointer = (void*)p; return hipSuccess; }
    return hipErrorInvalidValue;       /* (as the runtime answers for pageable memory) */
}
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t)
{
    for (size_t r = 0; r < height; ++r) memcpy((char*)d + r * dpitch, (const char*)s + r * spitch, width);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
inline double hipemu_now_() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event{0.0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = hipemu_now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   /* every emulated stream is synchronous */
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

template <typename K, typename... Args>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t smem, hipStream_t, Args... args)
{
    hipemu::bdim() = {block.x, block.y, block.z};
    hipemu::gdim() = {grid.x, grid.y, grid.z};
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
            hipemu::bidx() = {bx, by, 0};
            hipemu::tidx() = {0, 0, 0};
            hipemu::run_block([=]() { kernel(args...); }, block.x, smem);
        }
}
