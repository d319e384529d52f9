// dec/text_len.h - name middles and text lengths
// Part of rfq_decode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
// ---- text
__device__ __forceinline__ uint32_t dec_digits(uint32_t v) { uint32_t n = 1; while (v >= 10) { v /= 10; n++; } return n; }
__device__ __forceinline__ uint32_t dec_put(uint8_t* dst, uint32_t v) { const uint32_t n = dec_digits(v); for (uint32_t k = 0; k < n; k++) { dst[n - 1 - k] = (uint8_t)('0' + v % 10); v /= 10; } return n; }
// ':' and the decimal digits of v at buf[k ..) of a row of `cap` bytes in LDS: the bytes written.  A sequencer's lane / tile / x / y are below 10^8: two 4-digit halves, split
// into digit pairs and digits by multiplies (x * 5243 >> 19 = x / 100 below 10000, y * 103 >> 10 = y / 10 below 100, both pairs of a half in one register), leading zeros
// shifted out, ':' and seven digits in ONE 8-byte store.  (dec_put's loop - a division and a byte store per digit, ~22 instructions each with the quarter-rate multiplies, and
// every byte store an eight-way bank conflict at 32 bytes per lane - was most of k_dec_textlen2's 3.6 M issued VALU instructions per SE: the kernel ran in 0.49 ms beside the
// list chain on configs[2].)  Larger values, or a row with less than nine bytes left: the loop.
__device__ __forceinline__ uint32_t mid_put(uint8_t* buf, uint32_t k, uint32_t cap, uint32_t v) {
    if (v >= 100000000u || k + 9u > cap) { buf[k] = ':'; return 1u + dec_put(buf + k + 1u, v); }
    const uint32_t hi = v / 10000u, lo = v - hi * 10000u;
    const uint32_t a = mul24(hi, 5243u) >> 19, b = hi - mul24(a, 100u), c = mul24(lo, 5243u) >> 19, d = lo - mul24(c, 100u);
    const uint32_t P = a | (b << 16), Q = c | (d << 16);
    const uint32_t pt = ((P * 103u) >> 10) & 0x000F000Fu, po = P - pt * 10u, qt = ((Q * 103u) >> 10) & 0x000F000Fu, qo = Q - qt * 10u;
    const unsigned long long dg = (unsigned long long)(pt | (po << 8)) | ((unsigned long long)(qt | (qo << 8)) << 32);      // the eight digits, the most significant in byte 0
    const uint32_t lz = dg ? (uint32_t)(__ffsll((long long)dg) - 1) >> 3 : 7u, n = 8u - lz;
    const unsigned long long asc = (dg | 0x3030303030303030ull) >> (8u * lz);
    LdsU8 w; w.a = (asc << 8) | 0x3Aull; *(LdsU8*)(buf + k) = w;
    if (n == 8u) buf[k + 8u] = (uint8_t)(asc >> 56);
    return 1u + n;
}
struct DName { uint32_t n1, n2, st, lane, tile, x, y; };
__device__ __forceinline__ DName dec_name_parts(const uint8_t* cp, const DChunk& d, const DevHeader* D, const uint32_t* xv, const uint32_t* yv, uint32_t r) {
    const uint32_t fl = d.flags, hf = D->flags; DName m;
    m.n1 = cp[d.o_n1lens + ((fl & C_NAME1_LEN_SAME) ? 0u : r)];
    m.n2 = (hf & H_NAME2) ? cp[d.o_n2lens + ((fl & C_NAME2_LEN_SAME) ? 0u : r)] : 0u;
    m.st = cp[d.o_stlens + ((fl & C_STRAND_LEN_SAME) ? 0u : r)];
    const uint32_t xy = (fl & C_PE_INTERLEAVED) ? r / 2 : r;
    m.lane = (hf & H_LANE) ? cp[d.o_lanes + ((fl & C_LANE_SAME) ? 0u : xy)] : 0u;
    m.tile = (hf & H_TILE) ? ld_u16(cp + d.o_tiles + 2 * (size_t)((fl & C_TILE_SAME) ? 0u : xy)) : 0u;
    m.x = (hf & H_X) ? xv[d.rbase + xy] : 0u; m.y = (hf & H_Y) ? yv[d.rbase + xy] : 0u;
    return m;
}
// text bytes of every read; tin[g] = (bytes into out1, bytes into out2, 0, 0)
__device__ __forceinline__ uint32_t dec_textlen_one(const uint8_t* __restrict__ img, const DChunk& d, const DevHeader* __restrict__ D, const DReadTab& R,
                                                    const uint32_t* __restrict__ xv, const uint32_t* __restrict__ yv, int split, uint32_t r, bool& second, uint8_t* buf /* 40 bytes of LDS, 8-aligned: mine */) {
    const uint8_t* cp = img + d.off; const uint32_t g = d.rbase + r, hf = D->flags;
    const DName m = dec_name_parts(cp, d, D, xv, yv, r);
    // the digits go to an LDS row and leave as five 8-byte stores (a local array indexed by a running count lives in scratch: 48 bytes of it, and the
    // row went out byte by byte - VERDICT r3)
    uint32_t k = 0;                                                  // ":255:65535:4294967295:4294967295" is 32 bytes
    { unsigned long long* z = (unsigned long long*)buf; z[0] = z[1] = z[2] = z[3] = z[4] = 0ull; }
    if (hf & H_LANE) k += mid_put(buf, k, 39u, m.lane);                // (byte 39 is the length's)
    if (hf & H_TILE) k += mid_put(buf, k, 39u, m.tile);
    if (hf & H_X) k += mid_put(buf, k, 39u, m.x);
    if (hf & H_Y) k += mid_put(buf, k, 39u, m.y);
    buf[39] = (uint8_t)k;
    { const unsigned long long* z = (const unsigned long long*)buf; unsigned long long* mp = (unsigned long long*)(R.mid + (size_t)g * 40);
      const unsigned long long a0 = z[0], a1 = z[1], a2 = z[2], a3 = z[3], a4 = z[4]; mp[0] = a0; mp[1] = a1; mp[2] = a2; mp[3] = a3; mp[4] = a4; }
    const uint32_t nl = m.n1 + m.n2 + k;
    const uint32_t len = R.len[g]; const uint32_t text = nl + 1 + len + 1 + m.st + 1 + len + 1;
    second = split && (r & 1u);
    U4 t; t.a = second ? 0u : text; t.b = second ? text : 0u; t.c = 0; t.d = 0;
    R.tin[g] = t;
    return text;
}
__global__ void k_dec_textlen(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, const DevHeader* __restrict__ D, DReadTab R,
                              const uint32_t* __restrict__ xv, const uint32_t* __restrict__ yv, int split, DecStatus* st) {
    const DChunk d = CH[blockIdx.y]; const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= d.reads) return;                  // block-uniform
    uint32_t text = 0; bool second = false;
    __shared__ unsigned long long s_mid[256 * 5];                     // a 40-byte row per thread (blockDim.x <= 256)
    if (r < d.reads) text = dec_textlen_one(img, d, D, R, xv, yv, split, r, second, (uint8_t*)(s_mid + 5u * threadIdx.x));
    // 64-bit totals: the per-read prefix sums that place the text are 32-bit, the host refuses a batch that would wrap them
    // (one atomic per block, spread over 64 slots: same-address atomics from every wave would serialise at ~11 ns each)
    __shared__ unsigned long long s_t[2][4];
    const unsigned long long s1 = wave_sum<unsigned long long>(second ? 0ull : (unsigned long long)text), s2 = wave_sum<unsigned long long>(second ? (unsigned long long)text : 0ull);
    if (lane_id() == 0) { s_t[0][wave_id()] = s1; s_t[1][wave_id()] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0, b = 0; for (uint32_t i = 0; i < (blockDim.x >> 6); i++) { a += s_t[0][i]; b += s_t[1][i]; }
        const uint32_t slot = (blockIdx.y * 7u + blockIdx.x) & 63u;
        if (a) atomicAdd((unsigned long long*)&st->text_slots[0][slot], a);
        if (b) atomicAdd((unsigned long long*)&st->text_slots[1][slot], b);
    }
}
// fused path: the same per read - formatted middle, text bytes - with the text prefix made CHUNK-LOCAL on the spot (a workgroup per chunk, 256 reads per step, one block
// scan of the (out1, out2) byte pair per step) and the chunk's totals left for one small scan over the chunks; no per-read tin / tp arrays, no batch-wide scan.  A read keeps the
// offset of ITS output only, and the length of its middle beside it (tpl: 8 bytes per read); the middle's row is 32 bytes (E3_MIDROW), not 40.
__global__ void __launch_bounds__(256) k_dec_textlen2(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, const DevHeader* __restrict__ D, const uint32_t* __restrict__ len_i,
                                                      uint8_t* __restrict__ mid, const uint32_t* __restrict__ xv, const uint32_t* __restrict__ yv, int split,
                                                      uint2* __restrict__ tpl, U4* __restrict__ ctext, DecStatus* st) {
    __shared__ unsigned long long s_mid[256 * 4];                     // a 32-byte row per thread
    const uint32_t c = blockIdx.x; const DChunk d = CH[c]; const uint8_t* cp = img + d.off; const uint32_t hf = D->flags;
    // (out1 bytes | out2 bytes << 32 for the block scan: a STEP's 256 reads are far below 4 GiB; the chunk's running totals are kept as two 64-bit sums - a chunk of
    // 4 GiB of text or more in one output (a large -k, a crafted image) must reach the host as what it is: it refuses the range, RFQ_RANGE_TOO_BIG; packed, the out1
    // half carried into the out2 half and the host saw small totals - ADVICE r5)
    unsigned long long carry = 0, sum1 = 0, sum2 = 0;
    for (uint32_t r0 = 0; r0 < d.reads; r0 += blockDim.x) {           // block-uniform
        const uint32_t r = r0 + threadIdx.x; unsigned long long mine = 0; uint32_t k = 0; bool second = false;
        if (r < d.reads) {
            const size_t g = (size_t)d.rbase + r;
            const DName m = dec_name_parts(cp, d, D, xv, yv, r);
            uint8_t* buf = (uint8_t*)(s_mid + 4u * threadIdx.x);
            { unsigned long long* z = (unsigned long long*)buf; z[0] = z[1] = z[2] = z[3] = 0ull; }
            if (hf & H_LANE) k += mid_put(buf, k, E3_MIDROW, m.lane);
            if (hf & H_TILE) k += mid_put(buf, k, E3_MIDROW, m.tile);
            if (hf & H_X) k += mid_put(buf, k, E3_MIDROW, m.x);
            if (hf & H_Y) k += mid_put(buf, k, E3_MIDROW, m.y);
            { const unsigned long long* z = (const unsigned long long*)buf; unsigned long long* mp = (unsigned long long*)(mid + g * E3_MIDROW);
              const unsigned long long a0 = z[0], a1 = z[1], a2 = z[2], a3 = z[3]; mp[0] = a0; mp[1] = a1; mp[2] = a2; mp[3] = a3; }
            const uint32_t len = len_i[g]; const uint32_t text = m.n1 + m.n2 + k + 1 + len + 1 + m.st + 1 + len + 1;
            second = split && (r & 1u);
            mine = second ? ((unsigned long long)text << 32) : (unsigned long long)text;
        }
        unsigned long long tot; const unsigned long long ex = carry + block_excl_sum<unsigned long long>(mine, &tot);
        if (r < d.reads) tpl[(size_t)d.rbase + r] = make_uint2(second ? (uint32_t)(ex >> 32) : (uint32_t)ex, k);
        sum1 += tot & 0xFFFFFFFFull; sum2 += tot >> 32;
        carry = (sum1 & 0xFFFFFFFFull) | (sum2 << 32);                   // (the two halves wrap on their own; the totals below tell the host when they did)
    }
    if (threadIdx.x == 0) {
        U4 t; t.a = (uint32_t)sum1; t.b = (uint32_t)sum2; t.c = 0; t.d = 0; ctext[c] = t;
        // 64-bit totals of the range (the text offsets are 32-bit: the host refuses a range that would wrap them); spread over 64 slots
        const uint32_t slot = (c * 7u) & 63u;
        if (sum1) atomicAdd((unsigned long long*)&st->text_slots[0][slot], sum1);
        if (sum2) atomicAdd((unsigned long long*)&st->text_slots[1][slot], sum2);
    }
}
