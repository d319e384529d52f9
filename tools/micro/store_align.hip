// How fast are 16-byte global stores at byte granularity?  (k_dec_emit3 writes every line of the text with them.)
// Each lane stores 16 bytes; patterns:
//   0  contiguous, 16-aligned (the reference: a plain copy's store side)
//   1  contiguous, every address + 1        (whole wave shifted by a byte)
//   2  contiguous, every address + 4
//   3  records: 4 lanes write 64 consecutive bytes of a record, records 357 bytes apart, 6 rounds cover the record (emit3's shape)
//   4  as 3, but every 16-byte group moved DOWN to a 4-aligned address (what a dword-aligned variant would issue)
//   5  as 3 with 16-aligned groups
//   6 / 7 / 8  as 3 with 8 / 16 / 32 lanes per record (runs of 128 / 256 / 357 contiguous bytes per instruction)
//   9   records of four lines (55 + 151 + 2 + 149 bytes) by 4 lanes: the name line as in 3, the two long lines as ONE unaligned head group, 16-ALIGNED body groups and one
//       unaligned tail group that ends with the line (the groups overlap; what an emitter that picks its group boundaries by the output address would issue)
//   10  as 9, the name line too
//   11  as 9, and the aligned groups dealt to the four lanes BY ADDRESS (group at 16 J to lane J & 3): one store instruction writes whole 64-byte sectors
// usage: store_align [MB]   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/store_align tools/micro/store_align.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct __attribute__((packed, aligned(1))) U16 { uint32_t a, b, c, d; };
template <int MODE> __global__ void __launch_bounds__(256) k(uint8_t* out, size_t n_groups, uint32_t seed) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthr = (size_t)gridDim.x * 256;
    U16 v; v.a = seed + (uint32_t)tid; v.b = v.a * 3u; v.c = v.a * 5u; v.d = v.a * 7u;
    if (MODE <= 2) {
        const size_t off = MODE == 1 ? 1 : (MODE == 2 ? 4 : 0);
        for (size_t g = tid; g < n_groups; g += nthr) *(U16*)(out + off + 16 * g) = v;
    } else if (MODE >= 9) {   // (11: the name line as in 3)
        const size_t n_rec = (n_groups * 16) / 357 - 2;
        for (size_t r = tid >> 2; r < n_rec; r += nthr >> 2) {
            const size_t base = r * 357; const uint32_t part = (uint32_t)(tid & 3);
            const uint32_t lo[3] = { 0u, 55u, 208u }, ln[3] = { 55u, 153u, 149u };      // name; bases + "\n+\n" riding on its tail; qualities
            for (int li = 0; li < 3; li++) {
                const size_t a0 = base + lo[li]; const uint32_t n = ln[li];
                if (li == 0 && MODE != 10) { for (uint32_t gi = part; gi < (n + 15u) / 16u; gi += 4u) { uint32_t p0 = 16u * gi; if (p0 + 16u > n) p0 = n - 16u; *(U16*)(out + a0 + p0) = v; } continue; }
                const uint32_t h = (uint32_t)((16u - ((uintptr_t)(out + a0) & 15u)) & 15u), nb = (n - h) / 16u, ntask = nb + 2u;       // head, nb aligned groups, tail
                if (MODE == 11) {
                    const uint32_t J0 = (uint32_t)(((uintptr_t)(out + a0) + h) >> 4) & 3u;
                    if (h && part == ((J0 + 3u) & 3u)) *(U16*)(out + a0) = v;
                    for (uint32_t b = (part - J0) & 3u; b < nb; b += 4u) *(U16*)(out + a0 + h + 16u * b) = v;
                    if (part == ((J0 + nb) & 3u)) *(U16*)(out + a0 + n - 16u) = v;
                    continue;
                }
                for (uint32_t t = part; t < ntask; t += 4u) {
                    const uint32_t p0 = t == 0 ? 0u : (t <= nb ? h + 16u * (t - 1u) : n - 16u);
                    if (t == 0 && h == 0) continue;
                    *(U16*)(out + a0 + p0) = v;
                }
            }
        }
    } else {
        // record r = thread / 4 (+ rounds), part = thread & 3; a record is 357 bytes = 22 groups + 5 bytes: groups part, part + 4, ...
        const size_t n_rec = (n_groups * 16) / 357 - 2;
        const int LS = MODE == 6 ? 3 : (MODE == 7 ? 4 : (MODE == 8 ? 5 : 2)); const uint32_t L = 1u << LS;
        for (size_t r = tid >> LS; r < n_rec; r += nthr >> LS) {
            const size_t base = r * 357;
            for (uint32_t gi = (uint32_t)(tid & (L - 1)); gi < 22; gi += L) {
                size_t a = base + 16 * gi;
                if (MODE == 4) a &= ~(size_t)3;
                if (MODE == 5) a &= ~(size_t)15;
                *(U16*)(out + a) = v;
            }
        }
    }
}
int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 4096, bytes = mb << 20, n_groups = bytes / 16 - 64;
    uint8_t* d; hipMalloc(&d, bytes + 4096); hipMemset(d, 0, bytes + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = { "contiguous aligned", "contiguous + 1 byte", "contiguous + 4 bytes", "records, byte-granular", "records, 4-aligned groups", "records, 16-aligned groups", "records, 8 lanes", "records, 16 lanes", "records, 32 lanes", "records, long lines aligned", "records, all lines aligned", "records, sectors by lane" };
    for (int mode = 0; mode < 12; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            const int grid = 256 * 24;
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, n_groups, rep); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, n_groups, rep); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, n_groups, rep); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d, n_groups, rep); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, d, n_groups, rep); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, d, n_groups, rep); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, d, n_groups, rep); break;
                case 7: hipLaunchKernelGGL(k<7>, dim3(grid), dim3(256), 0, 0, d, n_groups, rep); break;
                case 8: hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, d, n_groups, rep); break;
                case 9: hipLaunchKernelGGL(k<9>, dim3(grid), dim3(256), 0, 0, d, n_groups, rep); break;
                case 10: hipLaunchKernelGGL(k<10>, dim3(grid), dim3(256), 0, 0, d, n_groups, rep); break;
                case 11: hipLaunchKernelGGL(k<11>, dim3(grid), dim3(256), 0, 0, d, n_groups, rep); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-30s %8.3f ms  %7.1f GB/s\n", names[mode], best, bytes / best / 1e6);
    }
    return 0;
}
