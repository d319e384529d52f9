// How fast are 16-byte (and 8-byte) LDS reads and writes at byte granularity?  The tile kernels read their staged text with unaligned ds_read_b128
// (lds_get16) everywhere, and an output tile in LDS (k_dec_emit3<.., OUT>) would write records with unaligned ds_write_b128.
// Each lane touches 16 bytes per step; 256 threads, ITER steps over a 24 KB tile; patterns:
//   contiguous: lane i at 16 i + off      (off = 0 aligned, 1, 4, 8)
//   records:    4 lanes on 64 consecutive bytes of a record, records 357 bytes apart (emit3's / gather2's shape), base offset off
// Reported: wave instructions per second per CU in G/s and bytes per clock per CU (2.4 GHz assumed), best of 3.
// usage: lds_align   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_align tools/micro/lds_align.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct __attribute__((packed, aligned(1))) U16 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(1))) U8 { uint32_t a, b; };
#define TILE 24576
#define ITER 4096
// KIND 0: read 16, 1: write 16, 2: read 8, 3: write 8, 4: read 16 as two aligned reads + alignbyte, 5: read 16 as five aligned dwords + alignbit, 6: read 4, 7: read 8 as two aligned dwords + alignbit
template <int KIND, bool REC, int STRIDE = 357> __global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t off) {
    __shared__ uint4 t4[TILE / 16 + 8];
    uint8_t* const t = (uint8_t*)t4; const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < TILE / 16 + 8; i += 256) t4[i] = make_uint4(i, i * 3u, i * 5u, i * 7u);
    __syncthreads();
    uint32_t acc = 0;
    // address of my group in round r
    const uint32_t rec = tid >> 2, part = tid & 3u;
    for (uint32_t it = 0; it < ITER; it++) {
        const uint32_t r = it % 5u;                                    // 5 rounds of 4 groups cover 320 bytes of a 357-byte record
        uint32_t a = REC ? rec * (uint32_t)STRIDE + 16u * (part + 4u * r) + off : 16u * tid + off + 4096u * (it & 3u);
        if (a + 16u > TILE) a = off;
        if (KIND == 0) { const U16 v = *(const U16*)(t + a); acc += v.a ^ v.b ^ v.c ^ v.d; }
        else if (KIND == 1) { U16 v; v.a = acc + it; v.b = it; v.c = tid; v.d = a; *(U16*)(t + a) = v; acc += it; }
        else if (KIND == 2) { const U8 v = *(const U8*)(t + a); acc += v.a ^ v.b; }
        else if (KIND == 3) { U8 v; v.a = acc + it; v.b = a; *(U8*)(t + a) = v; acc += it; }
        else if (KIND == 5) { const uint32_t* p = (const uint32_t*)(t + (a & ~3u)); const uint32_t sh = 8u * (a & 3u); const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
               acc += (uint32_t)((((unsigned long long)d1 << 32) | d0) >> sh) ^ (uint32_t)((((unsigned long long)d2 << 32) | d1) >> sh) ^ (uint32_t)((((unsigned long long)d3 << 32) | d2) >> sh) ^ (uint32_t)((((unsigned long long)d4 << 32) | d3) >> sh); }
        else if (KIND == 6) { struct __attribute__((packed, aligned(1))) U4_ { uint32_t a; }; acc += ((const U4_*)(t + a))->a; }
        else if (KIND == 7) { const uint32_t* p = (const uint32_t*)(t + (a & ~3u)); const uint32_t sh = 8u * (a & 3u); const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
               acc += (uint32_t)((((unsigned long long)d1 << 32) | d0) >> sh) ^ (uint32_t)((((unsigned long long)d2 << 32) | d1) >> sh); }
        else { const uint32_t al = a & ~15u, sh = a & 15u; const uint4 x = *(const uint4*)(t + al), y = *(const uint4*)(t + al + 16u);
               // bytes sh .. sh + 15 of the 32: funnel shifts
               const uint32_t w[8] = { x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w }; const uint32_t d = sh >> 2, b = sh & 3u; uint32_t o[4];
#pragma unroll
               for (int q = 0; q < 4; q++) { uint32_t lo = 0, hi = 0;
#pragma unroll
                   for (int e = 0; e < 4; e++) if (d == (uint32_t)e) { lo = w[e + q]; hi = w[e + q + 1]; }
                   o[q] = (uint32_t)((((unsigned long long)hi << 32) | lo) >> (8u * b)); }
               acc += o[0] ^ o[1] ^ o[2] ^ o[3]; }
        if (KIND & 1) __builtin_amdgcn_s_waitcnt(0xc07f);              // (keep the writes from piling up unboundedly: lgkmcnt(0))
    }
    __syncthreads();
    if (acc == 0x12345u || KIND & 1) out[blockIdx.x * 256 + tid] = acc ^ t4[tid].x;
}
template <int KIND, bool REC, int STRIDE = 357> static void run(const char* name, uint32_t off, uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float best = 1e9f;
    const int grid = 256 * 6;                                         // 6 workgroups per CU, all resident
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL((k<KIND, REC, STRIDE>), dim3(grid), dim3(256), 0, 0, d, off); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double winst = (double)grid * 4 * ITER, per_cu_per_s = winst / 256.0 / (best * 1e-3), bytes_clk = per_cu_per_s * 64 * ((KIND == 2 || KIND == 3) ? 8 : 16) / 2.4e9;
    printf("%-34s off %2u  %8.3f ms  %6.2f G wave-inst/s/CU  %6.1f B/clk/CU\n", name, off, best, per_cu_per_s / 1e9, bytes_clk);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 6 * 256 * 4);
    const uint32_t offs[] = { 0, 1, 4, 8 };
    for (uint32_t o : offs) run<0, false>("read16 contiguous", o, d);
    for (uint32_t o : offs) run<0, true>("read16 records", o, d);
    for (uint32_t o : offs) run<4, true>("read16 records, 2 aligned + shift", o, d);
    for (uint32_t o : offs) run<5, true>("read16 records, 5 dwords + alignbit", o, d);
    for (uint32_t o : offs) run<6, true>("read4 records", o, d);
    for (uint32_t o : offs) run<7, true>("read8 records, 3 dwords + alignbit", o, d);
    for (uint32_t o : offs) run<0, true, 352>("read16 records of 352 bytes", o, d);
    for (uint32_t o : offs) run<0, true, 160>("read16 rows of 160 bytes", o, d);
    for (uint32_t o : offs) run<2, true, 360>("read8 records of 360 bytes", o, d);
    for (uint32_t o : offs) run<1, true, 352>("write16 records of 352 bytes", o, d);
    for (uint32_t o : offs) run<1, false>("write16 contiguous", o, d);
    for (uint32_t o : offs) run<1, true>("write16 records", o, d);
    for (uint32_t o : offs) run<2, true>("read8 records", o, d);
    for (uint32_t o : offs) run<3, true>("write8 records", o, d);
    for (uint32_t o : offs) run<3, false>("write8 contiguous", o, d);
    return 0;
}
