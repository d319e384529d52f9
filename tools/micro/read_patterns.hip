// Micro-benchmark: what a streaming read of N bytes costs with (a) 64 contiguous bytes per lane (four 16-byte loads at a lane stride of 64 bytes: the line
// index's pattern) and (b) fully coalesced 16-byte loads (lane stride 16 bytes), with and without the newline-mask arithmetic.  build: hipcc --offload-arch=gfx950 -O3 -o /tmp/rp tools/micro/read_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__device__ __forceinline__ uint32_t flags4(uint32_t w, uint32_t pat) { const uint32_t v = w ^ pat; return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu); }
__device__ __forceinline__ uint32_t mask8(uint32_t a, uint32_t b, uint32_t pat) { uint32_t x = (flags4(a, pat) >> 7) | (flags4(b, pat) >> 3); x |= x >> 7; x |= x >> 14; return x & 0xFFu; }
__device__ __forceinline__ uint32_t mask16(const uint4& q, uint32_t pat) { return mask8(q.x, q.y, pat) | (mask8(q.z, q.w, pat) << 8); }
template <int MODE, int TILES, bool MASK> __global__ void __launch_bounds__(256) k_read(const uint8_t* __restrict__ p, uint64_t n, uint32_t* out, uint32_t magic) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t wbase = (uint64_t)blockIdx.x * (TILES * 16384u) + (uint64_t)wv * (TILES * 4096u);
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < TILES; k++) {
        const uint64_t tb = wbase + (uint32_t)k * 4096u;
        if (tb + 4096 > n) break;
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            const uint4 q = MODE == 0 ? *(const uint4*)(p + tb + lane * 64 + q4 * 16) : *(const uint4*)(p + tb + q4 * 1024 + lane * 16);
            if (MASK) acc += __builtin_popcount(mask16(q, 0x0A0A0A0Au)); else acc |= q.x ^ q.y ^ q.z ^ q.w;
        }
    }
    if (acc == magic) out[0] = acc;
}
template <int MODE, int TILES, bool MASK> static void run(const char* name, const uint8_t* d, uint64_t n, uint32_t* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint32_t grid = (uint32_t)(n / (TILES * 16384ull));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_read<MODE, TILES, MASK>), dim3(grid), dim3(256), 0, 0, d, n, out, 777u);
    hipEventRecord(a, 0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_read<MODE, TILES, MASK>), dim3(grid), dim3(256), 0, 0, d, n, out, 777u);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-34s %7.3f ms  %7.1f GB/s\n", name, ms, n / ms / 1e6);
}
int main() {
    const uint64_t n = 8ull << 30; uint8_t* d; uint32_t* out;
    if (hipMalloc(&d, n) != hipSuccess) return 1; hipMalloc(&out, 64);
    hipMemset(d, 0x41, n); hipMemset(d, 0x0A, n / 64);
    run<0, 1, false>("lane64 tiles1  read", d, n, out);  run<1, 1, false>("coalesced tiles1 read", d, n, out);
    run<0, 8, false>("lane64 tiles8  read", d, n, out);  run<1, 8, false>("coalesced tiles8 read", d, n, out);
    run<0, 16, false>("lane64 tiles16 read", d, n, out); run<1, 16, false>("coalesced tiles16 read", d, n, out);
    run<0, 1, true>("lane64 tiles1  read+mask", d, n, out);  run<1, 1, true>("coalesced tiles1 read+mask", d, n, out);
    run<0, 8, true>("lane64 tiles8  read+mask", d, n, out);  run<1, 8, true>("coalesced tiles8 read+mask", d, n, out);
    run<0, 16, true>("lane64 tiles16 read+mask", d, n, out); run<1, 16, true>("coalesced tiles16 read+mask", d, n, out);
    return 0;
}
