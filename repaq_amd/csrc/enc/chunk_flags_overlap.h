// enc/chunk_flags_overlap.h - per-chunk flag words, interleave test, the overlap search, stored prefixes
// Part of rfq_encode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
// =============================================================== per-chunk analysis (RfqCodec::encodeChunk pass 1, src/rfqcodec.cpp:181-287)
struct Layout {                  // byte offsets of every section inside one chunk image (RfqChunk::write order, src/rfqchunk.cpp:230-311)
    uint32_t off_readlens, off_n1lens, off_n2lens, off_stlens, off_lanes, off_tiles, off_x, off_y, off_n1, off_n2, off_st, off_seq, off_qual, off_ov, off_npos;
    uint32_t total, msize, seq_size, qual_size, npos_size, n1_size, n2_size, st_size, x_size, y_size, n_reads, flags;
};
__device__ __forceinline__ uint32_t name2_len_of(const Text& T, const ReadTab& R, uint32_t g) { return line_len(T, g, 0) - R.name2_off[g]; }

// Every read of a chunk is compared with the chunk's read 0 (src/rfqcodec.cpp:220-250) and, in a PE chunk under a header that supports interleaving, every
// odd read with its mate (:233-263).  The per-read verdicts are AND / MIN-combined per chunk:
//   cbits[c]  bits 0-7  readLen / name1Len / name2Len / strandLen / strand / lane / tile / name1 equal to read 0's      (starts as all ones)
//             bit 8     name2 equal to read 0's, every read;  bit 9  the same over the even reads only (what counts while the chunk stays interleaved)
//   cfail[c]  (first odd read whose mate test fails) << 1 | (0: the name2 rule failed, 1: only lane / tile / x / y differ)  (starts as all ones)
//   eq2[g]    name2 of read g equal to read 0's - only looked at for chunks whose mate test fails somewhere (the order-dependent rule of Q12)
// Two producers: g2_parse inside k_gather2 (the tile gather has the names staged) and k_chunk_flags_a (byte-wise gather path); k_chunk_flags_b turns them
// into the flag word.
#define CF_ALL 0x3FFu
// Pass A (byte-wise gather path) — grid (blocks, n_chunks): a wave takes 64 consecutive reads of the chunk, stages their names row by row in LDS with
// coalesced loads, and every lane compares its read with the chunk's read 0 (row 64) and, for odd reads of a PE chunk, with its mate (the previous row).
__global__ void k_chunk_flags_a(Text T, ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, int is_pe, uint32_t* __restrict__ cbits, uint32_t* __restrict__ cfail) {
    __shared__ uint8_t s_names[4 * 65 * NAME_STRIDE];
    const uint32_t c = blockIdx.y, f = C.first[c], e = C.first[c + 1];
    const int l = lane_id(), w = wave_id(); const uint32_t wpb = blockDim.x >> 6;
    uint8_t* rows = s_names + (size_t)w * 65 * NAME_STRIDE;
    const bool can0 = is_pe && D->support_interleaved; const uint32_t dpos = D->name2_diff_pos, dch = D->name2_diff_char;
    // read 0 of the chunk: name row 64, scalars in registers
    const uint32_t nl0 = line_len(T, f, 0); const uint8_t* nm0g = line_ptr(T, f, 0);
    const uint32_t n1l0 = R.name1_len[f], n2o0 = R.name2_off[f], n2l0 = nl0 - n2o0, len0 = R.len[f], stl0 = line_len(T, f, 2);
    const uint8_t* st0 = line_ptr(T, f, 2); const uint8_t lane0 = R.lane[f]; const uint16_t tile0 = R.tile[f];
    { const uint32_t take = nl0 < NAME_CAP ? nl0 : NAME_CAP; for (uint32_t i = (uint32_t)l; i < take; i += 64) rows[64 * NAME_STRIDE + i] = nm0g[i]; }
    const uint8_t* nm0 = nl0 <= NAME_CAP ? rows + 64 * NAME_STRIDE : nm0g;
    uint32_t bits = CF_ALL, fail = 0xFFFFFFFFu;
    for (uint32_t gb = f + (blockIdx.x * wpb + (uint32_t)w) * 64u; gb < e; gb += gridDim.x * wpb * 64u) {      // wave-uniform
        const uint32_t g = gb + (uint32_t)l; const bool v = g < e;
        uint32_t nb = 0, nl = 0, stb = 0, stl = 0; int s = 0;
        if (v) { uint32_t r; read_loc(T, g, s, r); const uint32_t* p = t_lo(T, s) + 4 * (size_t)r; nb = p[0]; nl = p[1] - 1 - nb; stb = p[2]; stl = p[3] - 1 - stb; }
        wave_lds_sync();                                                     // rows are private to the wave: previous group's rows are no longer read
        stage_name_rows(T, rows, nb, nl, s, l);
        wave_lds_sync();
        if (v) {
            const uint8_t* nm = nl <= NAME_CAP ? rows + l * NAME_STRIDE : t_fq(T, s) + nb;
            const uint32_t n1l = R.name1_len[g], n2o = R.name2_off[g], n2l = nl - n2o;
            const uint32_t rel = g - f;
            uint32_t b = 0;
            if (R.len[g] == len0) b |= 1u << 0;
            if (n1l == n1l0) b |= 1u << 1;
            if (n2l == n2l0) b |= 1u << 2;
            if (stl == stl0) b |= 1u << 3;
            if (bytes_eq(st0, stl0, t_fq(T, s) + stb, stl)) b |= 1u << 4;
            if (R.lane[g] == lane0) b |= 1u << 5;
            if (R.tile[g] == tile0) b |= 1u << 6;
            if (bytes_eq(nm0, n1l0, nm, n1l)) b |= 1u << 7;
            const bool e2 = bytes_eq(nm0 + n2o0, n2l0, nm + n2o, n2l);
            if (e2) b |= 1u << 8;
            if (e2 || (rel & 1u)) b |= 1u << 9;
            bits &= b;
            R.eq2[g] = e2 ? 1 : 0;
            if (can0 && (rel & 1u)) {                                        // mate = previous row (groups start at even reads)
                const uint32_t m = g - 1; const uint32_t mnl = line_len(T, m, 0), mo = R.name2_off[m];
                const uint8_t* mn = (mnl <= NAME_CAP ? rows + (l - 1) * NAME_STRIDE : line_ptr(T, m, 0)) + mo;
                const bool fa = !name2_eq_replaced(mn, mnl - mo, nm + n2o, n2l, dpos, dch);
                const bool fb = R.lane[m] != R.lane[g] || R.tile[m] != R.tile[g] || R.x[m] != R.x[g] || R.y[m] != R.y[g];
                if (fa || fb) { const uint32_t key = (rel << 1) | (fa ? 0u : 1u); if (key < fail) fail = key; }
            }
        }
    }
    bits = wave_and(bits); fail = wave_min(fail);
    if (l == 0) { if (bits != CF_ALL) atomicAnd(&cbits[c], bits); if (fail != 0xFFFFFFFFu) atomicMin(&cfail[c], fail); }
}
// Pass B — one wave per chunk: the flag word; name2Same with the order-dependent rule of src/rfqcodec.cpp:233-250 (Q12) - odd reads do not count while
// the chunk is still interleaved - from the accumulated bits, read by read only for a chunk whose mate test fails somewhere.
// assumed (may be null): the orientation the gather has already used for chunk c's mates; redo[c] = 1 where it turns out wrong (k_gather2 runs again there)
__global__ void k_chunk_flags_b(ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, int is_pe, const uint32_t* __restrict__ cbits, const uint32_t* __restrict__ cfail,
        uint32_t* __restrict__ redo) {
    const uint32_t c = blockIdx.x, f = C.first[c], e = C.first[c + 1]; const int l = lane_id();
    const bool can0 = is_pe && D->support_interleaved;
    const uint32_t acc = cbits[c], bits = acc & 0xFFu, fail = cfail[c];
    const bool failed = can0 && fail != 0xFFFFFFFFu; const uint32_t frel = fail >> 1; const bool kind_a = !(fail & 1u);
    uint32_t n2same = can0 ? (acc >> 9) & 1u : (acc >> 8) & 1u;
    if (failed) {                                                            // wave-uniform
        n2same = 1;
        for (uint32_t g = f + (uint32_t)l; g < e; g += 64) {
            const uint32_t rel = g - f;
            const bool counts = (rel < frel) ? !(rel & 1u) : (rel == frel ? kind_a : true);
            if (counts && !R.eq2[g]) n2same = 0;
        }
        n2same = wave_and(n2same);
    }
    if (l == 0) {
        const bool il = can0 && !failed;
        uint32_t fl = 0;
        if (il) fl |= C_PE_INTERLEAVED;
        if (bits & (1u << 0)) fl |= C_READ_LEN_SAME;
        if (bits & (1u << 1)) fl |= C_NAME1_LEN_SAME;
        if (bits & (1u << 2)) fl |= C_NAME2_LEN_SAME;
        if (bits & (1u << 3)) fl |= C_STRAND_LEN_SAME;
        if (bits & (1u << 4)) fl |= C_STRAND_SAME;
        if (bits & (1u << 5)) fl |= C_LANE_SAME;
        if (bits & (1u << 6)) fl |= C_TILE_SAME;
        if (bits & (1u << 7)) fl |= C_NAME1_SAME;
        if (n2same) fl |= C_NAME2_SAME;
        C.flags[c] = fl; C.il[c] = il ? 1u : 0u;
        if (redo) redo[c] = (can0 && failed) ? 1u : 0u;                      // the gather took the mates of every chunk for interleaved
    }
}

// RfqCodec::overlap (src/rfqcodec.cpp:1391-1438) for one pair per wave: lane = candidate overlap length.
// r1 = R1 as in the file, r2 = R2 as in the file (its reverse complement is formed on the fly).
__device__ __forceinline__ int wave_overlap(const uint8_t* __restrict__ r1, int len1, const uint8_t* __restrict__ r2, int len2) {
    const int l = lane_id(); const int minlen = len1 < len2 ? len1 : len2;
    for (int base = 12; base <= minlen; base += 64) {          // forward: R1 tail == RC(R2) head
        const int o = base + l; bool ok = o <= minlen;
        if (ok) for (int i = 0; i < o; i++) if (r1[len1 - o + i] != comp_base(r2[len2 - 1 - i])) { ok = false; break; }
        const unsigned long long b = __ballot(ok);
        if (b) return base + (__ffsll((long long)b) - 1);
    }
    for (int base = 12; base <= minlen; base += 64) {          // backward: RC(R2) tail == R1 head
        const int o = base + l; bool ok = o <= minlen;
        if (ok) for (int i = 0; i < o; i++) if (comp_base(r2[o - 1 - i]) != r1[i]) { ok = false; break; }
        const unsigned long long b = __ballot(ok);
        if (b) return -(base + (__ffsll((long long)b) - 1));
    }
    return 0;
}
// The same search in 2-bit space, ONE PAIR PER LANE.  A wave packs its 64 pairs into LDS rows - R1 as it is, R2 already reverse-
// complemented (RC2[i] = comp(R2[len2-1-i]): 16 bases taken from the END of R2, byte-reversed, complement codes) - as 2 bits per base
// (G 0, A 1, T 2, C 3, anything else 0) plus one "is N" bit per base.  RfqCodec::overlap compares characters: R1's are compared as
// they stand, RC2's are in {A,C,G,T,N} (Read::changeToReverseComplement maps everything else to N), so two bases are equal iff their
// codes and their N bits are equal - except a base of R1 outside A/C/G/T/N, which equals nothing (such pairs, and reads longer than
// the rows, take wave_overlap above).  Every lane then filters ITS pair's candidates o = 12, 13, ...: a candidate passes when the
// first 12 bases of its window equal the 12-base head of the other read - all window starts of the row at once, as bit-string
// arithmetic on the row held in registers; the few that pass are verified in full (codes and N bits) by the same lane.
// Forward before backward, smallest o first (src/rfqcodec.cpp:1391-1438).  The former wave-per-pair search cost ~600
// wave-instructions per pair, the per-candidate filter (one 64-bit window per 16 candidates) with wave-wide verification ~35.
// Row geometry by the longest read a launch has to hold (template parameter CAPB: 256, or 160 for the common short-read files).  The rows are most of what a wave
// needs of the CU - LDS decides how many waves are resident, and the search is a chain of LDS round trips that other waves hide: 160-base rows (9.2 KB per wave) let a CU hold
// sixteen waves where 256-base rows (13.3 KB) allow twelve, and the unrolled filter stops at ten dwords instead of sixteen.
//   code row: CAPB / 4 bytes + 4 of slack for the word behind the last; an ODD number of dwords, so that lanes reading their own rows at one offset hit different banks
//   N-bit row: CAPB / 8 bytes + slack, odd dwords again
#define OV2_CROW_OF(capb) ((capb) / 4u + 4u)
#define OV2_NROW_OF(capb) ((capb) == 256u ? 36u : (capb) / 8u + 8u)
#define OV2_FILTER 8              // bases of the head the candidate filter compares (any number <= 12, the smallest o: what passes is verified in full)
__device__ __forceinline__ uint32_t bfe_u32(uint32_t v, uint32_t off, uint32_t wid) { return (v >> off) & ((1u << wid) - 1u); }
// 16 bytes at base + off (any alignment); bytes outside [0, n) read as 0
static __device__ __noinline__ uint4 ld16_edge(const uint8_t* __restrict__ base, long long off, uint64_t n) {
    uint32_t w[4] = { 0, 0, 0, 0 };
    for (int b = 0; b < 16; b++) { const long long a = off + b; if (a >= 0 && (uint64_t)a < n) w[b >> 2] |= (uint32_t)base[a] << (8 * (b & 3)); }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
// four bases -> (four 2-bit codes in one byte, four N bits, "a byte that is neither A/C/G/T nor N" flags as 0xFF per byte)
__device__ __forceinline__ void ov2_pack_r1(uint32_t w, uint32_t& code, uint32_t& nbits, uint32_t& bad) {
    const uint32_t idx = (w >> 1) & 0x03030303u;
    const uint32_t ok = eq_bytes_full(__builtin_amdgcn_perm(0u, 0x47544341u, idx), w);
    code = ((__builtin_amdgcn_perm(0u, 0x00020301u, idx) & ok) * 0x01041040u) >> 24;
    nbits = 0; bad = 0;
    if (ok != 0xFFFFFFFFu) { const uint32_t isn = eq_bytes_full(w, 0x4E4E4E4Eu); nbits = ((isn & 0x01010101u) * 0x01020408u) >> 24; bad = ~ok & ~isn; }
}
// the complement's codes (Read::changeToReverseComplement: either case of A/C/G/T, anything else becomes N)
__device__ __forceinline__ void ov2_pack_rc(uint32_t w, uint32_t& code, uint32_t& nbits) {
    const uint32_t u = w & 0xDFDFDFDFu, idx = (u >> 1) & 0x03030303u;
    const uint32_t ok = eq_bytes_full(__builtin_amdgcn_perm(0u, 0x47544341u, idx), u);
    code = ((__builtin_amdgcn_perm(0u, 0x03010002u, idx) & ok) * 0x01041040u) >> 24;      // [A,C,T,G] -> codes of T,G,A,C
    nbits = ((~ok & 0x01010101u) * 0x01020408u) >> 24;
}
// 256 pairs per block, 64 per wave.  Only pairs of interleaved chunks are examined (src/rfqcodec.cpp:371-386)
// The search needs nothing but the text and its line table, so it runs for EVERY pair as soon as the index exists - on the second stream, beside
// the read table, the cut, the header and the chunk flags (those are latency-bound, this is VALU-bound) - and leaves the raw offset (0 = none)
// in ovraw; k_overlap_apply takes them over for the chunks that turn out to be interleaved under a header with BIT_ENCODE_PE_BY_OVERLAP.
// LOOSE: the rows are not packed from the text but copied from the loose slots k_gather2 has left (the same codes, R2 already reverse-complemented,
// valid for the pairs of interleaved chunks - the only ones whose result is used): lengths and slots come from the quality prefix pq, the
// "R1 holds a byte outside A/C/G/T/N" verdict from rflag.  The search then runs behind the gather, beside the position coder.
struct OvLoose { const uint32_t* pq; const uint32_t* lpk; const uint16_t* lnb; const uint8_t* rflag; };
template <bool LOOSE, uint32_t CAPB = 256u> __global__ void __launch_bounds__(256) k_overlap(Text T, OvLoose Z, int16_t* __restrict__ ovraw, uint32_t n_pairs) {
    constexpr uint32_t OV2_CAP = CAPB, OV2_CROW = OV2_CROW_OF(CAPB), OV2_NROW = OV2_NROW_OF(CAPB), OV2_WAVE_BYTES = 128u * (OV2_CROW + OV2_NROW); constexpr int OV2_ND = (int)(CAPB / 16u);
    static_assert((OV2_CROW / 4u) % 2u == 1u && (OV2_NROW / 4u) % 2u == 1u && OV2_WAVE_BYTES % 16u == 0u, "row geometry");
    // (+4: a verification step reads 9 bytes from a byte offset inside the last row)
    __shared__ uint32_t s_rows[4 * (OV2_WAVE_BYTES / 4) + 4]; __shared__ uint32_t s_bad[4][2];
    const int l = lane_id(), w = wave_id();
    uint8_t* const c1 = (uint8_t*)(s_rows + (size_t)w * (OV2_WAVE_BYTES / 4)); uint8_t* const c2 = c1 + 64u * OV2_CROW;
    uint8_t* const n1 = c2 + 64u * OV2_CROW; uint8_t* const n2 = n1 + 64u * OV2_NROW;
    for (uint32_t p0 = (blockIdx.x * 4u + (uint32_t)w) * 64u; p0 < n_pairs; p0 += gridDim.x * 256u) {       // wave-uniform
        const uint32_t p = p0 + (uint32_t)l; int len1 = -1, len2 = 0; uint32_t q1 = 0, q2 = 0; int s1 = 0, s2 = 0; uint32_t ld1 = 0, ld2 = 0;
        if (p < n_pairs) {
            const uint32_t g = 2u * p;
            if (LOOSE) { const uint32_t a = Z.pq[g], b = Z.pq[g + 1], c_ = Z.pq[g + 2]; len1 = (int)(b - a); len2 = (int)(c_ - b); ld1 = (a >> 4) + g;
                    ld2 = (b >> 4) + g + 1u; }
            else { uint32_t r; read_loc(T, g, s1, r); const uint32_t* pa = t_lo(T, s1) + 4 * (size_t)r; q1 = pa[1]; len1 = (int)(pa[2] - 1u - q1);
                              read_loc(T, g + 1, s2, r); const uint32_t* pb = t_lo(T, s2) + 4 * (size_t)r; q2 = pb[1]; len2 = (int)(pb[2] - 1u - q2); }
        }
        const bool slow = len1 >= 0 && ((uint32_t)len1 > OV2_CAP || (uint32_t)len2 > OV2_CAP), fast = len1 >= 0 && !slow;
        const int mx = wave_max(fast ? (len1 > len2 ? len1 : len2) : 0);
        if (l < 2) s_bad[w][l] = 0;
        // the rows are OR-ed together from 16-base pieces below: start from zero (the previous round's rows are no longer read)
        wave_lds_sync();
        { uint4* z = (uint4*)c1; for (uint32_t i = (uint32_t)l; i < OV2_WAVE_BYTES / 16u; i += 64u) z[i] = make_uint4(0, 0, 0, 0); }
        wave_lds_sync();
        // ---- pack: task t = (row, ALIGNED 32-byte group of the text that holds part of the row's sequence line); rows 0..63 R1, 64..127 RC2.
        // Consecutive lanes take consecutive groups of one line: every load is an aligned dwordx4 (a dwordx4 at an odd address - one per
        // 16 bases of the line itself - keeps the texture addresser busy for hundreds of cycles).  A group's 32 bases land at an arbitrary
        // base position of the row: their codes (64 bits) and N bits (32 bits) are shifted into place and OR-ed into the row.  (16-byte
        // tasks cost 150 instructions each, 90 of them per task and not per byte: row look-up, masks, atomics.)
        if (LOOSE) {
            // every lane copies its own pair's two slots into its two rows, four dwords of each per round: the loads of a round are all in flight
            // together (a task list dealt out over the wave - a shuffled row look-up and one load per step - was a chain of twenty round trips)
            const uint32_t nd_ = ((uint32_t)mx + 15u) >> 4;
            uint32_t* const r1w = (uint32_t*)(c1 + (uint32_t)l * OV2_CROW); uint32_t* const r2w = (uint32_t*)(c2 + (uint32_t)l * OV2_CROW);
            uint16_t* const m1w = (uint16_t*)(n1 + (uint32_t)l * OV2_NROW); uint16_t* const m2w = (uint16_t*)(n2 + (uint32_t)l * OV2_NROW);
            for (uint32_t j0 = 0; j0 < nd_; j0 += 4u) {
                uint32_t va[4], vb[4]; uint16_t ma[4], mb[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) {
                    const uint32_t j = j0 + u; const bool o1 = fast && j < nd_ && 16u * j < (uint32_t)len1, o2 = fast && j < nd_ && 16u * j < (uint32_t)len2;
                    va[u] = o1 ? Z.lpk[ld1 + j] : 0u; ma[u] = o1 ? Z.lnb[ld1 + j] : (uint16_t)0; vb[u] = o2 ? Z.lpk[ld2 + j] : 0u;
                            mb[u] = o2 ? Z.lnb[ld2 + j] : (uint16_t)0;
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) { const uint32_t j = j0 + u; if (fast && j < nd_) { r1w[j] = va[u]; m1w[j] = ma[u]; r2w[j] = vb[u]; m2w[j] = mb[u]; } }
            }
        } else {
        const uint32_t G = ((uint32_t)mx + 31u + 31u) >> 5, ntasks = 128u * G, ginv = G ? (65536u + G - 1u) / G : 0u;   // t / G == (t * ginv) >> 16 for t < 4096, G <= 17
            const uint32_t meta1 = (uint32_t)(len1 < 0 ? 0 : (len1 > 0xFFFF ? 0xFFFF : len1)) | ((uint32_t)s1 << 16) | (fast ? 1u << 17 : 0u);
            const uint32_t meta2 = (uint32_t)(len2 < 0 ? 0 : (len2 > 0xFFFF ? 0xFFFF : len2)) | ((uint32_t)s2 << 16);
            for (uint32_t t0 = 0; t0 < ntasks; t0 += 256u) {
                uint32_t v[4][8]; uint32_t row[4]; int L[4], pos0[4]; bool on[4], edge[4]; const uint8_t* src[4]; uint32_t at[4], lim[4];
    #pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t t = t0 + 64u * (uint32_t)u + (uint32_t)l; row[u] = t < ntasks ? (t * ginv) >> 16 : 0u; const uint32_t j = t - row[u] * G;
                    const int srcl = (int)(row[u] & 63u); const bool second = row[u] >= 64u;
                    const uint32_t ma = __shfl(meta1, srcl), mb = __shfl(meta2, srcl), o1 = __shfl(q1, srcl), o2 = __shfl(q2, srcl);
                    const uint32_t mm = second ? mb : ma; const bool f = (ma >> 17) & 1u;
                    L[u] = (int)(mm & 0xFFFFu); const uint32_t q = second ? o2 : o1, m = q & 31u;
                    at[u] = (q & ~31u) + 32u * j;                              // the group's offset in its stream
                    on[u] = t < ntasks && f && at[u] < q + (uint32_t)L[u];
                    const int z = (int)((mm >> 16) & 1u); src[u] = t_fq(T, z); lim[u] = t_n(T, z);
                    // base position (in the row) of the group's first byte once the row's orientation is applied: R1 as it stands, R2 back to front
                    pos0[u] = second ? L[u] - 32 * (int)j + (int)m - 32 : 32 * (int)j - (int)m;
                    edge[u] = on[u] && (unsigned long long)at[u] + 32ull > (unsigned long long)lim[u];
                }
    #pragma unroll
                for (int u = 0; u < 4; u++) {
    #pragma unroll
                    for (int i = 0; i < 8; i++) v[u][i] = 0;
                    if (on[u] && !edge[u]) { const uint4 x = *(const uint4*)(src[u] + at[u]), y = *(const uint4*)(src[u] + at[u] + 16u);
                                             v[u][0] = x.x; v[u][1] = x.y; v[u][2] = x.z; v[u][3] = x.w; v[u][4] = y.x; v[u][5] = y.y; v[u][6] = y.z; v[u][7] = y.w; }
                }
                if (__any(edge[0] || edge[1] || edge[2] || edge[3])) {
    #pragma unroll
                    for (int u = 0; u < 4; u++) if (edge[u]) { const uint4 x = ld16_edge(src[u], (long long)at[u], (uint64_t)lim[u]), y = ld16_edge(src[u],
                            (long long)at[u] + 16, (uint64_t)lim[u]);
                                                               v[u][0] = x.x; v[u][1] = x.y; v[u][2] = x.z; v[u][3] = x.w; v[u][4] = y.x; v[u][5] = y.y; v[u][6] = y.z;
                                                                       v[u][7] = y.w; }
                }
    #pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (!on[u]) continue;
                    const bool second = row[u] >= 64u; const uint32_t pr = row[u] & 63u;
                    // 32 codes, 32 N bits, 32 "neither A/C/G/T nor N" bits - byte b of the (re-oriented) group at bit b
                    unsigned long long cw = 0; uint32_t nw = 0, badb = 0;
                    if (!second) {
    #pragma unroll
                        for (int i = 0; i < 8; i++) { uint32_t c, nb, bd; ov2_pack_r1(v[u][i], c, nb, bd); cw |= (unsigned long long)c << (8 * i); nw |= nb << (4 * i);
                                if (bd) badb |= (((bd & 0x01010101u) * 0x01020408u) >> 24) << (4 * i); }
                    } else {
    #pragma unroll
                        for (int i = 0; i < 8; i++) { uint32_t c, nb; ov2_pack_rc(bswap32(v[u][7 - i]), c, nb); cw |= (unsigned long long)c << (8 * i);
                                nw |= nb << (4 * i); }
                    }
                    // keep the bases whose position lies inside the read: drop the `lo` leading ones and everything from `hi` on, shift into place
                    const int lo = pos0[u] < 0 ? -pos0[u] : 0, hi = L[u] - pos0[u] < 32 ? L[u] - pos0[u] : 32;
                    if (hi <= lo) continue;
                    const uint32_t nk = (uint32_t)(hi - lo);                   // 1..32 bases kept
                    const uint32_t km = nk >= 32u ? 0xFFFFFFFFu : (1u << nk) - 1u;
                    cw = (cw >> (2 * lo)) & (nk >= 32u ? ~0ull : (1ull << (2u * nk)) - 1ull); nw = (nw >> lo) & km;
                    if ((badb >> lo) & km) atomicOr(&s_bad[w][pr >> 5], 1u << (pr & 31u));
                    const uint32_t p = (uint32_t)(pos0[u] + lo);
                    uint32_t* const crow = (uint32_t*)((second ? c2 : c1) + pr * OV2_CROW) + ((2u * p) >> 5);
                            uint32_t* const nrow = (uint32_t*)((second ? n2 : n1) + pr * OV2_NROW) + (p >> 5);
                    const uint32_t cs = (2u * p) & 31u, ns = p & 31u;
                    const unsigned long long cv = cw << cs; const uint32_t ctop = cs ? (uint32_t)(cw >> (64u - cs)) : 0u;
                    const unsigned long long nv = (unsigned long long)nw << ns;
                    if ((uint32_t)cv) atomicOr(&crow[0], (uint32_t)cv);
                    if ((uint32_t)(cv >> 32)) atomicOr(&crow[1], (uint32_t)(cv >> 32));
                    if (ctop) atomicOr(&crow[2], ctop);
                    if ((uint32_t)nv) atomicOr(&nrow[0], (uint32_t)nv);
                    if ((uint32_t)(nv >> 32)) atomicOr(&nrow[1], (uint32_t)(nv >> 32));
                }
            }
        }
        wave_lds_sync();
        const bool bad = fast && (LOOSE ? (p < n_pairs && Z.rflag[2u * p] != 0) : ((s_bad[w][l >> 5] >> (l & 31)) & 1u) != 0);
        const bool go = fast && !bad; const int minlen = len1 < len2 ? len1 : len2;
        const uint8_t* const r1c = c1 + (uint32_t)l * OV2_CROW; const uint8_t* const r2c = c2 + (uint32_t)l * OV2_CROW;
        int ov = 0; bool done = !go || minlen < 12;
        const uint32_t head1 = lds_get4(r1c, 0) & 0xFFFFFFu, head2 = lds_get4(r2c, 0) & 0xFFFFFFu;
        const uint32_t nd = ((uint32_t)mx + 15u) >> 4;          // dwords of a code row in use (16 bases each), wave-uniform
#pragma unroll 1
        for (int dir = 0; dir < 2; dir++) {                     // 0: R1 tail == RC2 head (+o), 1: RC2 tail == R1 head (-o)
            const uint8_t* const wc = dir ? r2c : r1c; const int wl = dir ? len2 : len1; const uint32_t head = dir ? head1 : head2;
            if (!__any(!done)) break;
            // the filter, ALL window starts of the row at once: base i of the row starts a candidate (o = wl - i) when the OV2_FILTER bases
            // from i on equal the head of the other read.  With the row as a bit string (2 bits per base), X_k = (row >> 2k) ^ (head's base k
            // in every 2-bit group) has a zero group at i iff base i + k matches; OR over k leaves a zero group exactly at the starts that
            // pass.  3 instructions per 16 candidates and head base (funnel shift, xor, or) instead of 7 per candidate; 8 bases let a
            // random start through once in 65536 - 0.3 extra verifications per 64 pairs.
            uint32_t W[OV2_ND + 1], Dm[OV2_ND];
#pragma unroll
            for (int d = 0; d < OV2_ND + 1; d++) W[d] = (uint32_t)d <= nd ? ((const uint32_t*)wc)[d] : 0u;
#pragma unroll
            for (int d = 0; d < OV2_ND; d++) Dm[d] = 0u;
#pragma unroll
            for (int k = 0; k < OV2_FILTER; k++) {
                const uint32_t rep = ((head >> (2 * k)) & 3u) * 0x55555555u;
#pragma unroll
                for (int d = 0; d < OV2_ND; d++) if ((uint32_t)d < nd) {
                    const uint32_t sk = k ? (uint32_t)(((((unsigned long long)W[d + 1]) << 32) | W[d]) >> (2 * k)) : W[d];
                    Dm[d] |= sk ^ rep;
                }
            }
            // the starts that pass are verified by their own lane, in ascending o = descending start: the window row from base wl - o on
            // against the head of the other row, 32 bases (64 code bits) or 64 N bits per step - byte-granular 8-byte LDS reads + a
            // sub-byte funnel shift.  (The whole wave used to verify ONE candidate at a time: ~27 rounds of ~45 instructions for 64 pairs.)
            const int i_lo = wl - minlen, i_hi = wl - 12;       // starts that exist for this pair (12 <= o <= minlen)
#pragma unroll
            for (int d = 0; d < OV2_ND; d++) {
                if ((uint32_t)d >= nd) { Dm[d] = 0u; continue; }
                const int a0 = i_lo - 16 * d, a1 = i_hi - 16 * d + 1;            // valid starts of this dword: [a0, a1)
                const int e0 = a0 < 0 ? 0 : (a0 > 16 ? 16 : a0), e1 = a1 < 0 ? 0 : (a1 > 16 ? 16 : a1);
                const unsigned long long below1 = (1ull << (2 * e1)) - 1ull, below0 = (1ull << (2 * e0)) - 1ull;
                Dm[d] = done ? 0u : (~(Dm[d] | (Dm[d] >> 1)) & 0x55555555u & (uint32_t)(below1 & ~below0));
            }
            const uint8_t* const wn = (dir ? n2 : n1) + (uint32_t)l * OV2_NROW;    // window row's N bits; the other row: codes oc, N bits on
            const uint8_t* const oc = dir ? r1c : r2c; const uint8_t* const on_ = (dir ? n1 : n2) + (uint32_t)l * OV2_NROW;
            const uint32_t nch = ((uint32_t)mx + 31u) >> 5, nch2 = ((uint32_t)mx + 63u) >> 6;          // wave-uniform step counts
            for (;;) {
                int cand = -1;
#pragma unroll
                for (int d = OV2_ND - 1; d >= 0; d--) if ((uint32_t)d < nd && cand < 0 && Dm[d]) cand = 16 * d + ((31 - __clz((int)Dm[d])) >> 1);
                if (!__any(cand >= 0)) break;                    // wave-uniform
                const uint32_t pa = cand >= 0 ? (uint32_t)cand : 0u, o = cand >= 0 ? (uint32_t)(wl - cand) : 0u;
                unsigned long long diff = 0;
                // 64 bits of a row from any bit offset: three ALIGNED dwords and two funnel shifts (rows start on dwords; the word behind a row's last is the next row's or
                // the array's slack, and masked off).  An 8-byte LDS read at a byte offset - or at a dword that is not an 8-byte boundary: every other lane's row - is served one lane
                // per clock (rfq_common.h, lds_get16f): it was most of this loop.
                auto bits64 = [](const uint8_t* row, uint32_t bit) -> unsigned long long {
                    const uint32_t* q = (const uint32_t*)row + (bit >> 5); const uint32_t sh = bit & 31u; const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
                    return ((unsigned long long)lds_funnel(w2, w1, sh) << 32) | lds_funnel(w1, w0, sh);
                };
                auto word64 = [](const uint8_t* row, uint32_t c) -> unsigned long long { const uint32_t* q = (const uint32_t*)row + 2u * c; const uint32_t w0 = q[0], w1 = q[1];
                        return ((unsigned long long)w1 << 32) | w0; };
                for (uint32_t c = 0; c < nch; c++) {
                    if (32u * c >= o) continue;
                    const unsigned long long x = bits64(wc, 2u * (pa + 32u * c));
                    const uint32_t nb = o - 32u * c;
                    diff |= (x ^ word64(oc, c)) & (nb >= 32u ? ~0ull : (1ull << (2u * nb)) - 1ull);
                }
                for (uint32_t c = 0; c < nch2; c++) {
                    if (64u * c >= o) continue;
                    const unsigned long long x = bits64(wn, pa + 64u * c);
                    const uint32_t nb = o - 64u * c;
                    diff |= (x ^ word64(on_, c)) & (nb >= 64u ? ~0ull : (1ull << nb) - 1ull);
                }
                if (cand >= 0) {
                    if (diff == 0) { done = true; ov = dir ? -(int)o : (int)o;
#pragma unroll
                        for (int d = 0; d < OV2_ND; d++) Dm[d] = 0u; }
                    else {
#pragma unroll
                        for (int d = 0; d < OV2_ND; d++) if (d == (cand >> 4)) Dm[d] &= ~(1u << (2 * (cand & 15)));
                    }
                }
            }
        }
        // reads longer than a row, or an R1 holding a character outside A/C/G/T/N: the byte-wise search, one pair at a time
        unsigned long long sm = __ballot(slow || bad);
        // (the byte-wise search reads the text)
        if (LOOSE && sm && len1 >= 0) { uint32_t r; read_loc(T, 2u * p, s1, r); q1 = t_lo(T, s1)[4 * (size_t)r + 1]; read_loc(T, 2u * p + 1u, s2, r);
                q2 = t_lo(T, s2)[4 * (size_t)r + 1]; }
        while (sm) {
            const int j = __ffsll((long long)sm) - 1; sm &= sm - 1;
            const uint8_t* a = t_fq(T, __shfl(s1, j)) + __shfl(q1, j); const uint8_t* b = t_fq(T, __shfl(s2, j)) + __shfl(q2, j);
            const int r = wave_overlap(a, __shfl(len1, j), b, __shfl(len2, j));
            if (l == j) ov = r;
        }
        if (len1 >= 0) ovraw[p] = (int16_t)(ov > 32767 ? 0 : (ov < -32767 ? 0 : ov));      // (beyond +-127 - shift the clamp of k_overlap_apply makes it 0 anyway)
    }
}
// the clamp of src/rfqcodec.cpp:376-383 and the stored length of the mate, for the pairs of interleaved chunks (k_overlap found the offsets)
__global__ void k_overlap_apply(ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, const int16_t* __restrict__ ovraw, int8_t* __restrict__ ovb, uint32_t n_pairs) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs || !(D->flags & H_PE_OVERLAP)) return;
    const uint32_t g = 2u * p;
    if (!C.il[R.chunk[g]]) return;
    const int shift = D->overlap_shift; int ov = ovraw[p];
    if (ov + shift > 127) ov = 0;
    if (ov + shift < -127) ov = 0;
    ovb[p] = (int8_t)(ov + shift); R.stored[g + 1] = R.len[g + 1] - (uint32_t)(ov < 0 ? -ov : ov);
}
__global__ void k_pv_in(Text T, ReadTab R, U4* __restrict__ v, uint32_t n_reads) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_reads) { U4 t; t.a = R.name1_len[g]; t.b = name2_len_of(T, R, g); t.c = line_len(T, g, 2); t.d = R.stored[g]; v[g] = t; }
}
// which: bit 0 = qbase (needs the quality prefix only), bit 1 = sbase (needs the stored-base prefix, i.e. the overlaps)
__global__ void k_chunk_bases(ReadTab R, ChunkTab C, uint32_t n_chunks, int which) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_chunks) { const uint32_t f = C.first[c]; if (which & 1) C.qbase[c] = ((uint64_t)R.pq[f] & ~63ull) + 64ull * c;
            if (which & 2) C.sbase[c] = ((uint64_t)R.pv[f].d & ~63ull) + 64ull * c; }
}

// Tile path: overlap clamp (src/rfqcodec.cpp:376-383), stored lengths and the per-read prefix of (name1, name2, strand, stored) in ONE launch, a workgroup
// per chunk - the prefix restarts in every chunk, so nothing crosses workgroups.  (It was k_overlap_apply -> k_pv_in -> a three-launch U4 scan over the
// batch -> k_chunk_bases: six launches in a row on the second stream, 2.7 GB of traffic, 0.5 ms of latency in front of the sequence packer.)
__global__ void __launch_bounds__(256) k_chunk_prefix(Text T, ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, const int16_t* __restrict__ ovraw,
        int8_t* __restrict__ ovb) {
    const uint32_t c = blockIdx.x, f = C.first[c], e = C.first[c + 1], tid = threadIdx.x;
    const bool enc = C.il[c] != 0 && (D->flags & H_PE_OVERLAP); const int shift = D->overlap_shift;
    // the three name-piece prefixes only where the chunk stores a piece per read (k_assemble_names' test): a sequencer's names leave none, and the line table, the
    // parsed name lengths and the 16-byte prefix entries are then neither read nor written - the stored-base prefix alone (R.sd) is what k_seqpack needs
    const uint32_t fl = C.flags[c];
    const bool pieces = !(fl & C_NAME1_SAME) || ((D->flags & H_NAME2) && !(fl & C_NAME2_SAME)) || !(fl & C_STRAND_SAME);
    U4 carry; carry.a = carry.b = carry.c = carry.d = 0;
    for (uint32_t base = f; base < e; base += 1024u) {                     // block-uniform
        U4 v[4]; U4 acc; acc.a = acc.b = acc.c = acc.d = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t g = base + 4u * tid + (uint32_t)i; v[i].a = v[i].b = v[i].c = v[i].d = 0;
            if (g < e) {
                uint32_t st = R.len[g];
                if (enc && ((g - f) & 1u)) {
                    int ov = ovraw[g >> 1];
                    if (ov + shift > 127) ov = 0;
                    if (ov + shift < -127) ov = 0;
                    ovb[g >> 1] = (int8_t)(ov + shift); st -= (uint32_t)(ov < 0 ? -ov : ov); R.stored[g] = st;
                }
                if (pieces) { int s_; uint32_t r_; read_loc(T, g, s_, r_); const uint4 lo4 = *(const uint4*)(t_lo(T, s_) + 4 * (size_t)r_);
                        v[i].a = R.name1_len[g]; v[i].b = (lo4.y - 1u - lo4.x) - R.name2_off[g]; v[i].c = lo4.w - 1u - lo4.z; }
                v[i].d = st;
                acc = acc + v[i];
            }
        }
        U4 tot; U4 run = carry + block_excl_sum<U4>(acc, &tot);
#pragma unroll
        for (int i = 0; i < 4; i++) { const uint32_t g = base + 4u * tid + (uint32_t)i; if (g < e) { if (pieces) R.pv[g] = run; R.sd[g] = run.d; run = run + v[i]; } }
        carry = carry + tot;
    }
    if (tid == 0) { C.ptot[c] = carry; C.sbase[c] = C.qbase[c]; }          // (the tight streams are laid out like the qualities: stored <= len)
}
// byte-wise path: the totals from the batch-wide prefix
__global__ void k_chunk_ptot(ReadTab R, ChunkTab C, uint32_t n_chunks) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_chunks) C.ptot[c] = R.pv[C.first[c + 1]] - R.pv[C.first[c]];
}
