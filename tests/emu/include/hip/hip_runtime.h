// tests/emu/include/hip/hip_runtime.h — a SIMT interpreter for KERNEL UNIT TESTS ON A GPU-LESS BOX.
//
// TEST INFRASTRUCTURE ONLY.  The product library (repaq_amd/lib/librfq_hip.so) is built by hipcc for gfx950 and has
// no CPU path.  This header lets the very same .hip sources be compiled by g++ (-x c++ -I tests/emu/include) into
// tests/emu/librfq_emu.so, where every HIP thread is a ucontext fiber and wave64 collectives (__ballot/__shfl/...)
// and __syncthreads() are rendezvous points that ABORT on divergent use.  It exists because the build container has
// no GPU and the GPU box is a scarce, minutes-per-call resource: logic bugs are found here, hardware behaviour and
// all timing on the MI355X.  It is not a backend, is never shipped and never loaded by repaq_amd.
#pragma once
#include <ucontext.h>
#include <sys/mman.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define RFQ_SIMT_EMULATION 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static thread_local
// dynamic shared memory (the launch's shmem argument): one 64 KB buffer per interpreter thread - a worker runs one block at a time
namespace emu { inline void* dyn_shared() { alignas(16) static thread_local unsigned char buf[65536]; return buf; } }
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct ulonglong2 { unsigned long long x, y; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }

namespace emu {
constexpr int WAVE = 64;
enum State { READY, WAIT_WAVE, WAIT_BLOCK, DONE };
enum Op { OP_NONE, OP_BALLOT, OP_SHFL, OP_SHFL_UP, OP_SHFL_DOWN, OP_SHFL_XOR, OP_SYNC };
struct Fiber { ucontext_t ctx; State st; Op op; uint64_t val, res; int arg, width; dim3 tid; void* stack; };
struct Block {
    std::vector<Fiber> f; ucontext_t sched; dim3 bid, bdim, gdim; int cur; const std::function<void()>* body;
};
inline thread_local Block* blk = nullptr;
constexpr size_t STACK = 128 * 1024;

inline void die(const char* msg) { fprintf(stderr, "[hip-emu] %s (block %u thread %d)\n", msg, blk ? blk->bid.x : 0, blk ? blk->cur : -1); abort(); }
inline void trampoline() { (*blk->body)(); blk->f[blk->cur].st = DONE; swapcontext(&blk->f[blk->cur].ctx, &blk->sched); }
inline uint64_t collective(Op op, uint64_t v, int arg, int width) {
    Fiber& me = blk->f[blk->cur];
    me.op = op; me.val = v; me.arg = arg; me.width = width; me.st = (op == OP_SYNC) ? WAIT_BLOCK : WAIT_WAVE;
    swapcontext(&me.ctx, &blk->sched);
    return me.res;
}
inline void resolve_wave(Block& b, int w0, int w1) {
    Op op = OP_NONE; uint64_t ballot = 0;
    for (int i = w0; i < w1; i++) if (b.f[i].st == WAIT_WAVE) { if (op == OP_NONE) op = b.f[i].op; else if (op != b.f[i].op) { b.cur = i; die("divergent wave collective: lanes of one wave are at different collective ops"); } }
    for (int i = w0; i < w1; i++) if (b.f[i].st == WAIT_WAVE && b.f[i].val && op == OP_BALLOT) ballot |= 1ull << (i - w0);
    for (int i = w0; i < w1; i++) {
        Fiber& f = b.f[i]; if (f.st != WAIT_WAVE) continue;
        int lane = i - w0, src = lane, wd = f.width > 0 ? f.width : WAVE, seg = lane / wd * wd;
        switch (op) {
            case OP_BALLOT: f.res = ballot; break;
            case OP_SHFL: src = seg + (((f.arg % wd) + wd) % wd); break;
            case OP_SHFL_UP: src = lane - f.arg < seg ? lane : lane - f.arg; break;
            case OP_SHFL_DOWN: src = lane + f.arg >= seg + wd ? lane : lane + f.arg; break;
            case OP_SHFL_XOR: src = (lane ^ f.arg); if (src >= seg + wd || src < seg) src = lane; break;
            default: break;
        }
        if (op != OP_BALLOT) { int si = w0 + src; f.res = (si < w1 && b.f[si].st == WAIT_WAVE) ? b.f[si].val : f.val; }
    }
    for (int i = w0; i < w1; i++) if (b.f[i].st == WAIT_WAVE) b.f[i].st = READY;
}
inline void run_block(Block& b, int nthreads) {
    blk = &b;
    for (int i = 0; i < nthreads; i++) {
        Fiber& f = b.f[i]; getcontext(&f.ctx); f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())trampoline, 0); f.st = READY; f.op = OP_NONE;
        f.tid = dim3(i % b.bdim.x, (i / b.bdim.x) % b.bdim.y, i / (b.bdim.x * b.bdim.y));
    }
    for (;;) {
        bool progress = false; int live = 0;
        for (int i = 0; i < nthreads; i++) {
            if (b.f[i].st == READY) { b.cur = i; swapcontext(&b.sched, &b.f[i].ctx); progress = true; }
            if (b.f[i].st != DONE) live++;
        }
        if (!live) break;
        for (int w0 = 0; w0 < nthreads; w0 += WAVE) {
            int w1 = w0 + WAVE < nthreads ? w0 + WAVE : nthreads; int waiting = 0, other = 0;
            for (int i = w0; i < w1; i++) { if (b.f[i].st == WAIT_WAVE) waiting++; else if (b.f[i].st != DONE) other++; }
            if (waiting && !other) { resolve_wave(b, w0, w1); progress = true; }
        }
        int wb = 0, oth = 0;
        for (int i = 0; i < nthreads; i++) { if (b.f[i].st == WAIT_BLOCK) wb++; else if (b.f[i].st != DONE) oth++; }
        if (wb && !oth) { for (int i = 0; i < nthreads; i++) if (b.f[i].st == WAIT_BLOCK) b.f[i].st = READY; progress = true; }
        if (!progress) { b.cur = -1; die("deadlock: divergent __syncthreads()/wave collective (some threads wait at a barrier, others at a wave op)"); }
    }
    blk = nullptr;
}
inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z); const long nblocks = (long)grid.x * grid.y * grid.z;
    if (nthreads <= 0 || nthreads > 1024) { fprintf(stderr, "[hip-emu] bad block size %d\n", nthreads); abort(); }
    if (nblocks <= 0) return;
    std::atomic<long> next{0};
    auto worker = [&]() {
        Block b; b.f.resize(nthreads); b.bdim = block; b.gdim = grid; b.body = &body;
        char* stacks = (char*)mmap(nullptr, STACK * nthreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stacks == MAP_FAILED) { perror("[hip-emu] mmap"); abort(); }
        for (int i = 0; i < nthreads; i++) b.f[i].stack = stacks + STACK * i;
        for (;;) { long k = next.fetch_add(1); if (k >= nblocks) break; b.bid = dim3((unsigned)(k % grid.x), (unsigned)((k / grid.x) % grid.y), (unsigned)(k / ((long)grid.x * grid.y))); run_block(b, nthreads); }
        munmap(stacks, STACK * nthreads);
    };
    static int ncpu = [] { const char* e = getenv("HIP_EMU_THREADS"); int n = e ? atoi(e) : (int)std::thread::hardware_concurrency(); return n < 1 ? 1 : (n > 16 ? 16 : n); }();
    int nw = (int)(nblocks < ncpu ? nblocks : ncpu);
    if (nw <= 1) { worker(); return; }
    std::vector<std::thread> th; for (int i = 0; i < nw; i++) th.emplace_back(worker);
    for (auto& t : th) t.join();
}
template <class T> inline uint64_t to_bits(T v) { uint64_t b = 0; static_assert(sizeof(T) <= 8, "shfl type"); memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
}  // namespace emu

#define threadIdx (emu::blk->f[emu::blk->cur].tid)
// LDS-DMA (global_load_lds): every active lane copies `size` bytes from ITS global address to (wave-uniform LDS base) + lane * size.
// The interpreter runs a lane at a time, so the copy is immediate; the address-space qualifiers of the product code are ignored by g++.
static inline void __builtin_amdgcn_global_load_lds(const void* g, void* l, unsigned size, int off, int aux) { (void)aux; memcpy((char*)l + off + size * (threadIdx.x & 63u), g, size); }
#define blockIdx (emu::blk->bid)
#define blockDim (emu::blk->bdim)
#define gridDim (emu::blk->gdim)
#define warpSize 64

static inline unsigned long long __ballot(int p) { return emu::collective(emu::OP_BALLOT, p ? 1 : 0, 0, 0); }
static inline int __any(int p) { return __ballot(p) != 0; }
static inline int __all(int p) { return __ballot(!p) == 0; }
static inline void __syncthreads() { emu::collective(emu::OP_SYNC, 0, 0, 0); }
template <class T> static inline T __shfl(T v, int src, int width = 64) { return emu::from_bits<T>(emu::collective(emu::OP_SHFL, emu::to_bits(v), src, width)); }
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) { return emu::from_bits<T>(emu::collective(emu::OP_SHFL_UP, emu::to_bits(v), (int)d, width)); }
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) { return emu::from_bits<T>(emu::collective(emu::OP_SHFL_DOWN, emu::to_bits(v), (int)d, width)); }
template <class T> static inline T __shfl_xor(T v, int m, int width = 64) { return emu::from_bits<T>(emu::collective(emu::OP_SHFL_XOR, emu::to_bits(v), m, width)); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline long long clock64() { return 0; }
static inline long long wall_clock64() { return 0; }
// wave-level barrier / fence builtins: a rendezvous of the wave's lanes in the interpreter
// v_perm_b32: result byte i = byte (sel byte i) of the 8-byte value {hi, lo} (selectors 0-3 lo, 4-7 hi; >= 0x0C constants, unused here)
static inline uint32_t __builtin_amdgcn_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const unsigned long long v = ((unsigned long long)hi << 32) | lo; uint32_t r = 0;
    for (int i = 0; i < 4; i++) { const uint32_t k = (sel >> (8 * i)) & 0xFFu; r |= (uint32_t)((k < 8 ? (v >> (8 * k)) & 0xFFu : (k == 0x0C ? 0u : 0xFFu))) << (8 * i); }
    return r;
}
// v_readfirstlane_b32 is only applied to wave-uniform values in the product code: identity here
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline void __builtin_amdgcn_wave_barrier() { (void)emu::collective(emu::OP_BALLOT, 0, 0, 0); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
static inline void __threadfence_block() {}

template <class T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicSub(T* p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicCAS(T* p, T c, T v) { __atomic_compare_exchange_n(p, &c, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return c; }
template <class T> static inline T atomicMin(T* p, T v) { T o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }

// ---- host runtime subset (synchronous; "device memory" is host memory) ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNoDevice = 100 };
typedef struct emu_stream_s* hipStream_t;
typedef struct emu_event_s { std::chrono::steady_clock::time_point t; }* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hip-emu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = nullptr; if (posix_memalign(p, 256, n ? n : 256)) return hipErrorOutOfMemory; memset(*p, 0xA5, n ? n : 256); /* poison: catch reads of uninitialised device memory */ return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
#define hipHostRegisterDefault 0u
static inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
#define hipStreamNonBlocking 1u
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event_s(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
#define hipEventDisableTiming 2u
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event_s(); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
enum { hipErrorNotReady = 600 };
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }   // launches run to completion in order
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch(dim3(grid), dim3(block), std::function<void()>([=]() { kernel(__VA_ARGS__); }))
