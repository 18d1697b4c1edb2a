// Host shim used ONLY by tools/hostemu: compiles the kernel sources of aircompressor_amd/csrc with a host C++ compiler and runs them on
// the CPU -- test infrastructure for developing and fuzzing kernels without a GPU.  Not part of the product.
//
// Execution model: every thread of a workgroup is a FIBER (its own stack, cooperative switching).  A fiber runs until it reaches a
// cross-lane operation (__ballot, __shfl*, __syncthreads, wave_sync, readfirstlane ...); the scheduler then runs the other fibers to the same
// point, performs the operation over the wave (64 consecutive threads) or the workgroup, and releases them.  That is exact for kernels
// that (1) call cross-lane operations only in wave-uniform control flow (lanes that have left the kernel do not take part -- as on the
// hardware) and (2) separate memory traffic BETWEEN lanes with wave_sync() / __syncthreads() (on the device the former is only a compiler
// barrier: a wavefront's memory operations are performed in program order; here it is a rendezvous).  A cross-lane operation reached
// by only some lanes of a wave (at different source lines) is reported and aborts: such code would depend on EXEC-mask semantics that
// this shim does not model.  Lane-private kernels (no cross-lane operation at all) simply run one lane after the other.
// Kernels that give an item to an aligned GROUP of lanes (a quad: the Zstd sequence stage; 4 / 16 / 64 lanes: the ring decoders) have
// rendezvous of the group: quad_bcast / quad_sync, and group_sync, which achip_rings.h places -- for this shim only, the device code is
// untouched -- where the hardware's lockstep makes the lanes of a group meet: before anybody writes what somebody may still read, and
// before anybody reads what somebody may still write (within one copy step too: all loads, then all stores).
//
// HOSTEMU_ACCESS_LOCKSTEP (tools/hostemu/emu_enc.cpp: the encoders): the unit is built with clang's -fsanitize-coverage=trace-loads,trace-stores,
// whose callbacks make EVERY load and store of the kernel source a soft order point (below): a lane pauses before each memory access until
// no lane of its wave can run further, so all lanes perform access k before any performs access k+1 -- the lockstep of a wavefront at
// the granularity of memory operations.  That is what kernels need that run the same serial code on every lane and change shared state in
// place (table[h] read and then overwritten by all 64 lanes: every lane must see the old entry).  Accesses to a fiber's own stack do not
// pause; the shim's own functions are not instrumented (and not inlined into code that is).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#if defined(HOSTEMU_ACCESS_LOCKSTEP)
// (noinline: inlined into an instrumented kernel function the shim's own accesses would be instrumented with it -- and a pause between
// "me.wait = WAIT_BLOCK" and the switch would turn the barrier into nothing)
#pragma clang attribute push(__attribute__((no_sanitize("coverage"), noinline)), apply_to = function)
#endif

#define __global__
#define __device__
#define __host__
#define __constant__ const
#define __forceinline__ inline
#define __launch_bounds__(...)
#define ACHIP_WAVES_PER_EU(lo, hi)  // (a register cap of the device compiler: achip_device.h)
#define __shared__ static
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
#define hipErrorUnknown 999
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
// streams and events do not exist here: everything runs in launch order
typedef void* hipEvent_t;
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }

namespace hostemu {

constexpr int MAX_THREADS = 1024;
constexpr size_t STACK_BYTES = 512 << 10;
enum Wait { RUNNABLE = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, WAIT_QUAD = 3, WAIT_SOFT = 4 };  // (WAIT_QUAD: an aligned group of Fiber::gsize lanes, 4 by default)

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = true;
    int wait = RUNNABLE;
    const char* file = nullptr;
    int line = 0;
    uint64_t post = 0;
    int gsize = 4;  // group size of a WAIT_QUAD rendezvous
    long waits[5] = {0, 0, 0, 0, 0};  // (diagnosis: rendezvous of each kind this lane has been to)
    uintptr_t key[24];                 // (HOSTEMU_ACCESS_LOCKSTEP: where the lane pauses -- return addresses, outermost frame first)
    int keyLen = 0;
    int recent[64] = {0};              // (diagnosis: source lines of its latest rendezvous)
};

struct State {
    Fiber f[MAX_THREADS];
    int n = 0;
    int cur = -1;
    bool inFiber = false;  // a kernel lane is running (not the scheduler, not host code)
    void* schedSp = nullptr;
    std::function<void()> body;
};
inline State& S()
{
    static thread_local State s;
    return s;
}

extern "C" void hostemu_switch(void** saveSp, void* loadSp);
// x86-64 System V: callee-saved registers + stack pointer
__asm__(
    ".text\n"
    ".globl hostemu_switch\n"
    ".type hostemu_switch,@function\n"
    "hostemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size hostemu_switch,.-hostemu_switch\n");

inline bool key_less(const Fiber& a, const Fiber& b)
{
    const int n = a.keyLen < b.keyLen ? a.keyLen : b.keyLen;
    for (int i = 0; i < n; i++) {
        if (a.key[i] != b.key[i]) return a.key[i] < b.key[i];
    }
    return false;  // (equal as far as both go: together)
}

inline void fiber_entry()
{
    State& s = S();
    s.body();
    Fiber& me = s.f[s.cur];
    me.done = true;
    hostemu_switch(&me.sp, s.schedSp);
    abort();  // a finished fiber is never resumed
}

inline void wait_here(int kind, const char* file, int line)
{
    State& s = S();
    Fiber& me = s.f[s.cur];
    me.wait = kind;
    me.file = file;
    me.line = line;
    me.keyLen = 0;
    me.waits[kind]++;
    if (kind == WAIT_WAVE) me.recent[me.waits[kind] & 63] = line;
    hostemu_switch(&me.sp, s.schedSp);
}

inline void run_workgroup(int nThreads, const std::function<void()>& body)
{
    State& s = S();
    if (nThreads > MAX_THREADS) {
        fprintf(stderr, "hostemu: workgroup too large\n");
        abort();
    }
    s.n = nThreads;
    s.body = body;
    for (int t = 0; t < nThreads; t++) {
        Fiber& f = s.f[t];
        if (!f.stack) {
            f.stack = (char*)aligned_alloc(64, STACK_BYTES);
        }
        uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
        void** p = (void**)top;
        *--p = nullptr;                 // fake return address of fiber_entry
        *--p = (void*)&fiber_entry;     // `ret` of the first switch lands here
        for (int k = 0; k < 6; k++) *--p = nullptr;
        f.sp = (void*)p;
        f.done = false;
        f.wait = RUNNABLE;
        f.waits[WAIT_WAVE] = 0;
    }
    for (;;) {
        bool any = false, ran = false;
        static thread_local unsigned pass = 0;
        pass++;
        for (int t0 = 0; t0 < nThreads; t0++) {
            // lanes run in a different order from pass to pass (ascending, descending, interleaved): code that works only because a lower
            // lane happens to run first between two rendezvous is a race on the hardware and should fail here too
            static const int fixedOrder = getenv("HOSTEMU_ORDER") ? atoi(getenv("HOSTEMU_ORDER")) : -1;  // (diagnosis: one lane order for every pass)
            const unsigned ord = fixedOrder >= 0 ? (unsigned)fixedOrder : (pass & 3);
            const int t = ord == 0 ? t0 : (ord == 1 ? nThreads - 1 - t0 : (ord == 2 ? (t0 ^ 1) < nThreads ? (t0 ^ 1) : t0 : (t0 * 37 + 11) % nThreads));
            Fiber& f = s.f[t];
            if (f.done) continue;
            any = true;
            if (f.wait != RUNNABLE) continue;
            s.cur = t;
            threadIdx = dim3((unsigned)t);
            s.inFiber = true;
            hostemu_switch(&s.schedSp, f.sp);
            s.inFiber = false;
            ran = true;
        }
        if (!any) break;
        // release complete rendezvous
        bool released = false;
        bool blockReady = true;
        for (int t = 0; t < nThreads; t++) {
            if (!s.f[t].done && s.f[t].wait != WAIT_BLOCK) blockReady = false;
        }
        if (blockReady) {
            for (int t = 0; t < nThreads; t++) {
                if (!s.f[t].done) s.f[t].wait = RUNNABLE;
            }
            released = true;
        }
        else {
            for (int w = 0; w < nThreads; w += 64) {
                const int e = w + 64 < nThreads ? w + 64 : nThreads;
                bool ready = true, some = false;
                const char* file = nullptr;
                int line = 0;
                for (int t = w; t < e; t++) {
                    Fiber& f = s.f[t];
                    if (f.done) continue;
                    if (f.wait != WAIT_WAVE) {
                        ready = false;
                        continue;
                    }
                    if (!some) {
                        file = f.file;
                        line = f.line;
                        some = true;
                    }
                    else if (f.line != line || f.file != file) {
                        fprintf(stderr, "hostemu: lanes of one wave wait at different cross-lane operations: %s:%d and %s:%d (thread %d)\n", file, line, f.file, f.line, t);
                        if (getenv("HOSTEMU_VERBOSE")) {
                            for (int u = w; u < e; u++) {
                                fprintf(stderr, "  thread %d: %s at %s:%d  (wave rendezvous so far: %ld)\n", u, s.f[u].done ? "done" : (s.f[u].wait == RUNNABLE ? "runnable" : "waits"), s.f[u].file ? s.f[u].file : "?", s.f[u].line, s.f[u].waits[WAIT_WAVE]);
                                fprintf(stderr, "      latest:");
                                for (int k = 63; k >= 0; k--) fprintf(stderr, " %d", s.f[u].recent[(s.f[u].waits[WAIT_WAVE] - k) & 63]);
                                fprintf(stderr, "\n");
                            }
                        }
                        abort();
                    }
                }
                if (ready && some) {
                    for (int t = w; t < e; t++) {
                        if (!s.f[t].done) s.f[t].wait = RUNNABLE;
                    }
                    released = true;
                }
            }
        }
        if (!blockReady) {
            // quad-level rendezvous (kernels that give an item to four lanes: the Zstd pipeline's sequence stage): the lanes of a quad that
            // are still running all wait at the same operation
            for (int q = 0; q < nThreads;) {
                // the group of the first lane at or behind q that waits at a group rendezvous decides the stride (groups are aligned)
                int gs = 4;
                for (int t = q; t < nThreads && t < q + 64; t++) {
                    if (!s.f[t].done && s.f[t].wait == WAIT_QUAD) {
                        gs = s.f[t].gsize;
                        break;
                    }
                }
                q &= ~(gs - 1);
                const int qEnd = q + gs;
                bool ready = true, some = false;
                const char* file = nullptr;
                int line = 0;
                for (int t = q; t < qEnd && t < nThreads; t++) {
                    Fiber& f = s.f[t];
                    if (f.done) continue;
                    if (f.wait != WAIT_QUAD) {
                        ready = false;
                        continue;
                    }
                    if (!some) {
                        file = f.file;
                        line = f.line;
                        some = true;
                    }
                    else if (f.line != line || f.file != file) {
                        fprintf(stderr, "hostemu: lanes of one quad wait at different cross-lane operations: %s:%d and %s:%d (thread %d)\n", file, line, f.file, f.line, t);
                        abort();
                    }
                }
                if (ready && some) {
                    for (int t = q; t < qEnd && t < nThreads; t++) {
                        if (!s.f[t].done) s.f[t].wait = RUNNABLE;
                    }
                    released = true;
                }
                q = qEnd;
            }
        }
        // soft order points (wave_mem_order under HOSTEMU_ORDER_IS_SOFT): a lane that reaches one pauses until no lane of its wave can run any
        // further -- every other lane is at an order point of its own, at a rendezvous, or done.  For code that runs a wavefront in uniform
        // control flow that is a barrier (what the device's lockstep gives it); for lanes that go their own ways it is only a pause.
        for (int w = 0; w < nThreads; w += 64) {
            const int e = w + 64 < nThreads ? w + 64 : nThreads;
            bool anyRunnable = false, anySoft = false;
            for (int t = w; t < e; t++) {
                if (s.f[t].done) continue;
                anyRunnable = anyRunnable || s.f[t].wait == RUNNABLE;
                anySoft = anySoft || s.f[t].wait == WAIT_SOFT;
            }
            if (anySoft && !anyRunnable) {
                // Of the lanes that pause before a memory access, those EARLIEST IN THE PROGRAM go first -- the call stacks' return addresses
                // compared from the outermost frame inwards, smallest first: a lane inside an `if (lane == 0) { ... }` or still in a loop runs
                // until it has caught up with the lanes that wait behind the join.  That is the reconvergence a wavefront has by
                // construction (the other lanes are masked off, not ahead); code laid out in source order is what makes addresses a fair
                // stand-in for it.  (Lanes with equal stacks -- uniform code -- go together; pauses without a stack, keyLen 0, likewise.)
                // Not modelled: loop trips.  A lane back at a loop's head compares EARLIER than lanes in the previous trip's tail; kernels
                // whose lanes fall out of step inside a trip and have no cross-lane operation per trip (the ring decoders) need a rendezvous
                // of their own now and then -- where their device code has its wave_mem_order() is enough: tools/hostemu/emu_lockstep.cpp.
                int best = -1;
                for (int t = w; t < e; t++) {
                    if (s.f[t].done || s.f[t].wait != WAIT_SOFT) continue;
                    if (best < 0 || key_less(s.f[t], s.f[best])) best = t;
                }
                for (int t = w; t < e; t++) {
                    if (!s.f[t].done && s.f[t].wait == WAIT_SOFT && !key_less(s.f[best], s.f[t])) s.f[t].wait = RUNNABLE;
                }
                released = true;
            }
        }
        if (!ran && !released) {
            fprintf(stderr, "hostemu: deadlock -- some lanes wait at a cross-lane operation the others never reach:\n");
            for (int t = 0; t < nThreads; t++) {
                if (!s.f[t].done) fprintf(stderr, "  thread %d: wait %d at %s:%d\n", t, s.f[t].wait, s.f[t].file ? s.f[t].file : "?", s.f[t].line);
            }
            abort();
        }
    }
}

inline int lane_id() { return S().cur & 63; }
inline int wave_base() { return S().cur & ~63; }

// all lanes post a value; after the first rendezvous every lane may read every post; the second rendezvous keeps a fast lane from posting
// its NEXT value before a slow lane has read this one
template <typename F>
inline auto collective(uint64_t mine, const char* file, int line, F&& read) -> decltype(read())
{
    State& s = S();
    s.f[s.cur].post = mine;
    wait_here(WAIT_WAVE, file, line);
    auto r = read();
    wait_here(WAIT_WAVE, file, line);
    return r;
}
inline bool lane_active(int t)
{
    State& s = S();
    return t < s.n && !s.f[t].done;
}
inline unsigned long long ballot(int p, const char* file, int line)
{
    return collective(p ? 1 : 0, file, line, [&]() {
        unsigned long long m = 0;
        const int w = wave_base();
        for (int i = 0; i < 64; i++) {
            if (lane_active(w + i) && S().f[w + i].post) m |= 1ull << i;
        }
        return m;
    });
}
template <typename T>
inline uint64_t to_bits(T v)
{
    static_assert(sizeof(T) <= 8, "shuffle of a wide type");
    uint64_t b = 0;
    memcpy(&b, &v, sizeof(T));
    return b;
}
template <typename T>
inline T from_bits(uint64_t b)
{
    T v;
    memcpy(&v, &b, sizeof(T));
    return v;
}
template <typename T>
inline T shfl_from(T v, int srcLane, const char* file, int line)
{
    return collective(to_bits(v), file, line, [&]() {
        const int t = wave_base() + (srcLane & 63);
        // an inactive source lane: the hardware returns what its register holds; here: the reader's own value
        return lane_active(t) ? from_bits<T>(S().f[t].post) : v;
    });
}
inline void wave_sync(const char* file, int line) { wait_here(WAIT_WAVE, file, line); }
// value of lane k of the caller's aligned group of gs lanes (a shuffle whose source lies in the caller's own lane group: group-uniform control flow is enough)
template <typename T>
inline T group_from(T v, int gs, int k, const char* file, int line)
{
    State& s = S();
    s.f[s.cur].post = to_bits(v);
    s.f[s.cur].gsize = gs;
    wait_here(WAIT_QUAD, file, line);
    const int t = (s.cur & ~(gs - 1)) + (k & (gs - 1));
    const T r = lane_active(t) ? from_bits<T>(s.f[t].post) : v;
    s.f[s.cur].gsize = gs;
    wait_here(WAIT_QUAD, file, line);
    return r;
}
inline void order_point(const char* file, int line) { wait_here(WAIT_WAVE, file, line); }  // (wave_mem_order under HOSTEMU_ORDER_IS_RENDEZVOUS)
inline void soft_order_point(const char* file, int line) { wait_here(WAIT_SOFT, file, line); }  // (wave_mem_order under HOSTEMU_ORDER_IS_SOFT)
inline void quad_sync(const char* file, int line) { S().f[S().cur].gsize = 4; wait_here(WAIT_QUAD, file, line); }
// rendezvous of the caller's aligned group of gs lanes (gs a power of two, at most 64): kernels that give a block to a lane group
inline void group_sync(int gs, const char* file, int line) { S().f[S().cur].gsize = gs; wait_here(WAIT_QUAD, file, line); }
// value of lane k of the caller's quad (DPP quad_perm broadcast); quad-uniform control flow is enough
template <typename T>
inline T quad_from(T v, int k, const char* file, int line)
{
    State& s = S();
    s.f[s.cur].post = to_bits(v);
    s.f[s.cur].gsize = 4;
    wait_here(WAIT_QUAD, file, line);
    const int t = (s.cur & ~3) + (k & 3);
    const T r = lane_active(t) ? from_bits<T>(s.f[t].post) : v;
    wait_here(WAIT_QUAD, file, line);
    return r;
}
inline void block_sync(const char* file, int line) { wait_here(WAIT_BLOCK, file, line); }

}  // namespace hostemu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                             \
    do {                                                                                        \
        blockDim = block;                                                                       \
        gridDim = grid;                                                                         \
        for (unsigned bx_ = 0; bx_ < dim3(grid).x; bx_++) {                                     \
            blockIdx = dim3(bx_);                                                               \
            hostemu::run_workgroup((int)dim3(block).x, [&]() { kernel(__VA_ARGS__); });         \
        }                                                                                       \
    } while (0)

#define __syncthreads() hostemu::block_sync(__FILE__, __LINE__)
#define __ballot(p) hostemu::ballot((p), __FILE__, __LINE__)
#define __shfl(v, src) hostemu::shfl_from((v), (src), __FILE__, __LINE__)
#define __shfl_up(v, d) hostemu::shfl_from((v), hostemu::lane_id() >= (int)(d) ? hostemu::lane_id() - (int)(d) : hostemu::lane_id(), __FILE__, __LINE__)
#define __shfl_down(v, d) hostemu::shfl_from((v), hostemu::lane_id() + (int)(d) < 64 ? hostemu::lane_id() + (int)(d) : hostemu::lane_id(), __FILE__, __LINE__)
#define __shfl_xor(v, m) hostemu::shfl_from((v), hostemu::lane_id() ^ (int)(m), __FILE__, __LINE__)
#define __builtin_amdgcn_readfirstlane(v) hostemu::shfl_from((v), __builtin_ctzll(hostemu::ballot(1, __FILE__, __LINE__)), __FILE__, __LINE__)
#define __builtin_amdgcn_readlane(v, l) hostemu::shfl_from((v), (l), __FILE__, __LINE__)
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
inline int __ffsll(unsigned long long v) { return v == 0 ? 0 : __builtin_ctzll(v) + 1; }
// cross-lane builtins that only NOT emulated kernels of a shared header use (they must compile, they never run here)
inline int __builtin_amdgcn_update_dpp(int old, int src, int, int, int, bool) { (void)old; return src; }
template <typename T> inline T atomicAdd(T* p, T v) { T old = *p; *p += v; return old; }
template <typename T> inline T atomicOr(T* p, T v) { T old = *p; *p |= v; return old; }
template <typename T> inline T atomicExch(T* p, T v) { T old = *p; *p = v; return old; }
template <typename T> inline T atomicMax(T* p, T v) { T old = *p; if (v > old) *p = v; return old; }

// dynamic shared memory of the emulated launch: one arena, re-used by every "workgroup"
static uint8_t hostemu_dynamic_lds[1 << 20] __attribute__((aligned(64)));
inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

#if defined(HOSTEMU_ACCESS_LOCKSTEP)
namespace hostemu {
inline void access_point(const void* p, void* pc, void** callerFrame)  // pc: the access in the kernel source; callerFrame: that function's frame
{
    State& s = S();
    if (!s.inFiber) return;  // host code of the same unit
    Fiber& me = s.f[s.cur];
    if ((uintptr_t)p - (uintptr_t)me.stack < STACK_BYTES) return;  // the lane's own stack: private
    me.wait = WAIT_SOFT;
    me.file = "memory access";
    me.line = 0;
    {   // the call stack (units built with -fno-omit-frame-pointer): [saved frame pointer][return address] per frame, up to fiber_entry's null
        uintptr_t ra[24];
        int n = 0;
        ra[n++] = (uintptr_t)pc;
        void** fp = callerFrame;
        while (fp != nullptr && n < 24) {
            const uintptr_t r = (uintptr_t)fp[1];
            if (r == 0) break;
            ra[n++] = r;
            void** up = (void**)fp[0];
            if (up <= fp || (uintptr_t)up - (uintptr_t)me.stack >= STACK_BYTES) break;
            fp = up;
        }
        me.keyLen = 0;
        for (int i = n - 1; i >= 0; i--) me.key[me.keyLen++] = ra[i];
    }
    hostemu_switch(&me.sp, s.schedSp);
}
}  // namespace hostemu
#define HOSTEMU_TRACE_CALLBACK __attribute__((disable_tail_calls))  // (a frame of its own: its return address is the access, its saved frame pointer the kernel function's)
extern "C" {
HOSTEMU_TRACE_CALLBACK void __sanitizer_cov_load1(uint8_t* p) { hostemu::access_point(p, __builtin_return_address(0), (void**)*(void**)__builtin_frame_address(0)); }
HOSTEMU_TRACE_CALLBACK void __sanitizer_cov_load2(uint16_t* p) { hostemu::access_point(p, __builtin_return_address(0), (void**)*(void**)__builtin_frame_address(0)); }
HOSTEMU_TRACE_CALLBACK void __sanitizer_cov_load4(uint32_t* p) { hostemu::access_point(p, __builtin_return_address(0), (void**)*(void**)__builtin_frame_address(0)); }
HOSTEMU_TRACE_CALLBACK void __sanitizer_cov_load8(uint64_t* p) { hostemu::access_point(p, __builtin_return_address(0), (void**)*(void**)__builtin_frame_address(0)); }
HOSTEMU_TRACE_CALLBACK void __sanitizer_cov_load16(__int128* p) { hostemu::access_point(p, __builtin_return_address(0), (void**)*(void**)__builtin_frame_address(0)); }
HOSTEMU_TRACE_CALLBACK void __sanitizer_cov_store1(uint8_t* p) { hostemu::access_point(p, __builtin_return_address(0), (void**)*(void**)__builtin_frame_address(0)); }
HOSTEMU_TRACE_CALLBACK void __sanitizer_cov_store2(uint16_t* p) { hostemu::access_point(p, __builtin_return_address(0), (void**)*(void**)__builtin_frame_address(0)); }
HOSTEMU_TRACE_CALLBACK void __sanitizer_cov_store4(uint32_t* p) { hostemu::access_point(p, __builtin_return_address(0), (void**)*(void**)__builtin_frame_address(0)); }
HOSTEMU_TRACE_CALLBACK void __sanitizer_cov_store8(uint64_t* p) { hostemu::access_point(p, __builtin_return_address(0), (void**)*(void**)__builtin_frame_address(0)); }
HOSTEMU_TRACE_CALLBACK void __sanitizer_cov_store16(__int128* p) { hostemu::access_point(p, __builtin_return_address(0), (void**)*(void**)__builtin_frame_address(0)); }
void __sanitizer_cov_8bit_counters_init(char*, char*) {}
void __sanitizer_cov_pcs_init(const uintptr_t*, const uintptr_t*) {}
}
#pragma clang attribute pop
#endif
