// enc/gather_tiles.h - tile gather k_gather2 (names parsed, match masks, bases packed where they stand), sequence packer, stream plan
// Part of rfq_encode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
// =============================================================== gather, second formulation (fast path) + sequence packer
// What round 2's kernel timeline left: k_gather's tile loop spends half its instructions on per-tile bookkeeping (fit test, seven LDS tables, four
// barriers) and needs two LDS output tiles, which caps the tile at 32 reads.  k_gather2 has NO output tile and no fit test:
//   * a tile is a fixed number K of reads (K = 64, 32, ... chosen by the host so that K records always fit the staged-text buffer);
//   * qualities go from the staged text straight to qcat with byte-granular 16-byte stores (the lanes of one read are neighbours, so a wave's
//     stores still cover contiguous runs), counted from the registers they pass through;
//   * bases are 2-bit packed (+ one "is N" bit each) where they stand - in stored orientation (a mate reverse-complemented) but untrimmed - into a per-read slot of a
//   LOOSE array:
//     read g (batch order) owns the dwords Ld(g) = (pq[g] >> 4) + g ... of `lpk` (16 codes each; G 0, A 1, T 2, C 3, anything else 0,
//     src/rfqcodec.cpp:590-604) and the same u16 slots of `lnb`.  No stored-base prefix and no overlap result is needed here:
//     k_seqpack applies them (overlap trim, compaction to the chunk's tight 2-bit stream + N bit mask).
#define G2_CAP 23552u             // staged text of a tile (64 x 357-byte records are 22.9 KB)
#define G2_SE_OK 1                // 0: never take the single-end instantiation (A/B on the box: tools/build_variant.sh)
#define G2_CNT 256u               // replicated quality counters (see QualCount)
struct __attribute__((packed, aligned(1))) GU16g { uint32_t a, b, c, d; };
// four bases -> four 2-bit codes (exact upper-case A/C/G/T, anything else 0), four "is N" bits, four "neither" bits
__device__ __forceinline__ void pack4_codes(uint32_t w, uint32_t& code, uint32_t& nb, uint32_t& bad) {
    const uint32_t idx = (w >> 1) & 0x03030303u;
    const uint32_t ok = eq_bytes_full(__builtin_amdgcn_perm(0u, 0x47544341u, idx), w);
    code = ((__builtin_amdgcn_perm(0u, 0x00020301u, idx) & ok) * 0x01041040u) >> 24;
    nb = 0; bad = 0;
    if (ok != 0xFFFFFFFFu) { const uint32_t isn = eq_bytes_full(w, 0x4E4E4E4Eu); nb = ((isn & 0x01010101u) * 0x01020408u) >> 24;
            bad = (((~ok & ~isn) & 0x01010101u) * 0x01020408u) >> 24; }
}
// the same for a base of a reverse-complemented mate (the four bytes are already in reversed order): Read::changeToReverseComplement
// (src/read.cpp:77-115) maps either case of A/C/G/T to the upper-case complement and everything else to N
__device__ __forceinline__ void pack4_codes_rc(uint32_t w, uint32_t& code, uint32_t& nb) {
    const uint32_t u = w & 0xDFDFDFDFu, idx = (u >> 1) & 0x03030303u;
    const uint32_t ok = eq_bytes_full(__builtin_amdgcn_perm(0u, 0x47544341u, idx), u);
    code = ((__builtin_amdgcn_perm(0u, 0x03010002u, idx) & ok) * 0x01041040u) >> 24;      // [A,C,T,G] -> codes of T,G,A,C
    nb = ((~ok & 0x01010101u) * 0x01020408u) >> 24;
}
// 16 bases that are all upper-case A/C/G/T (nearly every group of a sequencer's file) -> their 16 codes; false when a byte is anything else (the
// exact per-byte forms above then decide).  The letters are looked up back from the 2-bit index and compared with one xor: ten VALU instructions
// per four bases instead of seventeen (k_gather2 is VALU-bound: 1.85 G wave instructions on configs[2], 3.0 of its 3.5 ms).
__device__ __forceinline__ bool pack16_fast(const uint32_t (&w)[4], uint32_t& code) {
    uint32_t diff = 0; code = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t idx = (w[i] >> 1) & 0x03030303u;
        diff |= __builtin_amdgcn_perm(0u, 0x47544341u, idx) ^ w[i];
        code |= ((__builtin_amdgcn_perm(0u, 0x00020301u, idx) * 0x01041040u) >> 24) << (8 * i);
    }
    return diff == 0;
}
__device__ __forceinline__ uint32_t g2_rev2x16(uint32_t v) {                // the sixteen 2-bit fields of v in reverse order
    v = bswap32(v); v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4); return ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
}
struct G2Geo { uint32_t a00, a01, end0, end1, base1; };                         // a tile's text spans: 16-aligned begin and end per stream, LDS offset of stream 1's span
// my read: lengths, LDS offsets of its quality / sequence line, chunk-relative quality position, loose slot
struct G2Read { bool on, rc; uint32_t len, qsrc, ssrc, qpos, ld, gi; };
// ---- what a tile needs before its text can be requested.  A tile BOUNDARY (where the text of tile k starts in each stream, its first quality position)
// is a scalar load issued three tiles ahead, the lines of my read in the next tile are requested a tile ahead: the tile's only round trip at its start
// is the text's own (they used to be two: boundaries, then text + lines).  (The text itself cannot be requested a tile ahead: into registers it costs 24
// VGPRs the kernel does not have at six waves per SIMD - it spills at 80 as it is -, into a second LDS buffer it costs resident workgroups.)
// tile boundary: line-table entry of its first read in each stream, quality prefix of that read
struct G2Bound { uint32_t l0, l1, q; };
struct G2MRaw { uint4 lo4; uint32_t pg; };
#ifdef RFQ_SIMT_EMULATION
__device__ __forceinline__ uint32_t ld_uniform(const uint32_t* p) { return *p; }
#else
// a load whose address is the same in every lane, from memory no kernel in flight writes: constant address space -> s_load_dword, the value in an SGPR
__device__ __forceinline__ uint32_t ld_uniform(const uint32_t* p) { return *(const __attribute__((address_space(4))) uint32_t*)(uintptr_t)p; }
#endif
__device__ __forceinline__ G2Bound g2_bound(const Text& T, bool two, const uint32_t* __restrict__ pq, uint32_t r) {   // r: uniform; even when `two`
    G2Bound b;
    if (two) { const size_t k = 4 * (size_t)(r >> 1); b.l0 = ld_uniform(T.lo[0] + k); b.l1 = ld_uniform(T.lo[1] + k); }
    else { b.l0 = ld_uniform(T.lo[0] + 4 * (size_t)r); b.l1 = 0u; }
    b.q = ld_uniform(pq + r);
    return b;
}
__device__ __forceinline__ G2Geo g2_geo(const G2Bound& b, const G2Bound& e, bool two) {
    G2Geo g; g.a00 = b.l0 & ~15u; g.end0 = e.l0; g.a01 = two ? b.l1 & ~15u : 0u; g.end1 = two ? e.l1 : 0u;
    g.base1 = two ? (((g.end0 - g.a00 + 15u) & ~15u) + 16u) : 0u;
    return g;
}
// thread tid's groups of a tile: group i = tid + 256 k of the spans laid end to end (stream 0's n0 groups, then stream 1's); its place in LDS: i, or one
// group further on for stream 1 (base1).  Only a stream's very last group may reach past the caller's buffer: it is not requested here but copied byte by
// byte when the tile is put down.
// LDS-DMA of a tile's spans to buf4 (global_load_lds_dwordx4: every lane names its own 16 global bytes, a wave's 64 groups land contiguously)
__device__ __forceinline__ void g2_stage1(const uint8_t* __restrict__ fq, uint32_t n, uint32_t a0, uint32_t end, uint4* l4, uint32_t tid) {
    const uint32_t nb = end - a0, ng = (nb + 15u) / 16u;
    const uint8_t* src = fq + a0;
    const uint32_t nfull = (uint64_t)a0 + 16ull * ng <= (uint64_t)n ? ng : ng - 1u;      // (only a stream's very last group may reach past the buffer)
    for (uint32_t i = tid; i < nfull; i += blockDim.x)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * (size_t)i),
                (__attribute__((address_space(3))) void*)(l4 + (i - (tid & 63u))), 16, 0, 0);
    if (nfull < ng && tid == 0) { uint8_t* const bytes = (uint8_t*)l4;
            for (uint32_t k = 0; k < 16 && a0 + 16 * nfull + k < n; k++) bytes[16 * nfull + k] = src[16 * (size_t)nfull + k]; }
}
__device__ __forceinline__ void g2_stage(const Text& T, bool two, const G2Geo& g, uint4* buf4, uint32_t tid) {
    g2_stage1(T.fq[0], T.n[0], g.a00, g.end0, buf4, tid);
    if (two) g2_stage1(T.fq[1], T.n[1], g.a01, g.end1, buf4 + g.base1 / 16, tid);
}
__device__ __forceinline__ G2MRaw g2_mraw(const Text& T, const uint32_t* __restrict__ pq, uint32_t cur, uint32_t j, uint32_t cnt) {
    G2MRaw r; r.lo4 = make_uint4(0, 0, 0, 0); r.pg = 0;
    // starts of the read's four lines
    if (j < cnt) { const uint32_t gi = cur + j; int s_; uint32_t r_; read_loc(T, gi, s_, r_); r.lo4 = *(const uint4*)(t_lo(T, s_) + 4 * (size_t)r_); r.pg = pq[gi]; }
    return r;
}
__device__ __forceinline__ G2Read g2_read(const Text& T, const G2MRaw& r, const G2Geo& g, uint32_t f, uint32_t pq0, bool il, uint32_t cur, uint32_t j, uint32_t cnt) {
    G2Read m; m.on = j < cnt; m.rc = false; m.len = m.qsrc = m.ssrc = m.qpos = m.ld = 0; m.gi = cur + j;
    if (m.on) {
        const uint32_t gi = cur + j; int s_; uint32_t r_; read_loc(T, gi, s_, r_);
        const uint32_t lb = s_ ? g.base1 : 0u, a = s_ ? g.a01 : g.a00;
        m.len = r.lo4.z - 1u - r.lo4.y; m.ssrc = lb + (r.lo4.y - a); m.qsrc = lb + (r.lo4.w - a);
        m.qpos = r.pg - pq0; m.ld = (r.pg >> 4) + gi;
        m.rc = il && ((gi - f) & 1u);
    }
    return m;
}
// my share (groups part, part + P, ...) of my read's sequence line: 16 bases per step -> one dword of codes + 16 N bits into the read's loose slot, in STORED
// orientation (an interleaved chunk's mate reverse-complemented, src/rfqcodec.cpp:371-407) but untrimmed: k_seqpack skips what the overlap with R1 implies
__device__ __forceinline__ void g2_bases(const uint8_t* s_text, const G2Read& m, uint32_t part, uint32_t P, uint32_t* __restrict__ lpk, uint16_t* __restrict__ lnb,
        uint8_t* __restrict__ rflag, uint8_t* __restrict__ rn) {
    const uint32_t ng = (m.len + 15u) >> 4;
    for (uint32_t gi = part; gi < ng; gi += P) {
        uint32_t w[4], code = 0, nbits = 0;
        const uint32_t nv0 = m.len - 16u * gi;                             // valid bases of this step
        // A last, partial step of a line of >= 16 bases takes the 16 bases that END the line (that begin it, for a mate stored back to front) and shifts the ones it owns down:
        // every byte it looks at is a base of the read.  (Bytes outside the line would fail the all-ACGT test in some lane of nearly every wave - and a wave runs the exact path if
        // any of its lanes does; they used to be made 'A' by a loop over the bytes, ~20 instructions each: two hundred for the last step of a 150-base line, in every wave that
        // holds one - k_gather2 3.64 -> 3.54 ms.)  Only reads of fewer than 16 bases still take that loop.
        const bool tail = nv0 < 16u && m.len >= 16u; const uint32_t tsh = tail ? 16u - nv0 : 0u;
        auto blank = [&](uint32_t from, uint32_t to) { for (uint32_t k = from; k < to; k++) { uint32_t& x = w[k >> 2]; const uint32_t sh = 8u * (k & 3u);
                x = (x & ~(0xFFu << sh)) | (0x41u << sh); } };
        if (!m.rc) {
            lds_get16(s_text, m.ssrc + (tail ? m.len - 16u : 16u * gi), w);
            if (nv0 < 16u && !tail) blank(nv0, 16u);
            if (!pack16_fast(w, code)) {                                    // (an N, a lower-case or any other byte among the 16)
                uint32_t bad = 0; code = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) { uint32_t c4, n4, b4; pack4_codes(w[i], c4, n4, b4); code |= c4 << (8 * i); nbits |= n4 << (4 * i); bad |= b4 << (4 * i); }
                // a byte outside A/C/G/T/N: it equals nothing in RfqCodec::overlap (k_overlap's byte-wise path)
                if (bad) rflag[m.gi] = 1;
            }
        } else {
            lds_get16(s_text, tail ? m.ssrc : m.ssrc + m.len - 16u * gi - 16u, w);   // the 16 file bases that END at len - 16 gi
            if (nv0 < 16u && !tail) blank(0u, 16u - nv0);
            if (pack16_fast(w, code)) code = ~g2_rev2x16(code);             // reverse complement in 2-bit space: the fields back to front, G 0 <-> C 3, A 1 <-> T 2
            else {
                const uint32_t x0 = bswap32(w[3]), x1 = bswap32(w[2]), x2 = bswap32(w[1]), x3 = bswap32(w[0]); w[0] = x0; w[1] = x1; w[2] = x2; w[3] = x3; code = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) { uint32_t c4, n4; pack4_codes_rc(w[i], c4, n4); code |= c4 << (8 * i); nbits |= n4 << (4 * i); }
            }
        }
        code >>= 2u * tsh; nbits >>= tsh;
        if (nv0 < 16u) { code &= (1u << (2u * nv0)) - 1u; nbits &= (1u << nv0) - 1u; }   // (what lies outside the line is not the read's)
        lpk[m.ld + gi] = code; lnb[m.ld + gi] = (uint16_t)nbits;
        if (nbits) rn[m.gi] = 1;                                           // (the read holds an N: whoever reads the loose slots' N bits asks this first - one read in a few hundred does)
    }
}
// my share (groups part, part + P, ...) of my read's two lines: qualities -> qcat, bases -> the loose slot
__device__ __forceinline__ void g2_compose(const uint8_t* s_text, const G2Read& m, uint32_t part, uint32_t P, uint8_t* qd, uint32_t* __restrict__ lpk,
        uint16_t* __restrict__ lnb, uint8_t* __restrict__ rflag, uint8_t* __restrict__ rn, QualCount& qc) {
    if (!m.on) return;
    {
        // ---- qualities: text -> qcat (an interleaved chunk's mate back to front), counted on the way
        const uint32_t n = m.len; uint8_t* const o = qd + m.qpos; const bool rc = m.rc;
        if (n >= 16u) {
            const uint32_t ng = (n + 15u) >> 4;
            for (uint32_t gi = part; gi < ng; gi += P) {
                // the last group ends exactly at n: its first `dup` bytes repeat the group before
                uint32_t p0 = 16u * gi, dup = 0; if (p0 + 16u > n) { dup = p0 + 16u - n; p0 = n - 16u; }
                uint32_t w[4]; lds_get16(s_text, rc ? m.qsrc + n - p0 - 16u : m.qsrc + p0, w);
                if (rc) { const uint32_t x0 = bswap32(w[3]), x1 = bswap32(w[2]), x2 = bswap32(w[1]), x3 = bswap32(w[0]); w[0] = x0; w[1] = x1; w[2] = x2; w[3] = x3; }
                { GU16g v; v.a = w[0]; v.b = w[1]; v.c = w[2]; v.d = w[3]; *(GU16g*)(o + p0) = v; }
                if (!dup) qc.group(m.qpos + p0, w[0], w[1], w[2], w[3]);
                else for (uint32_t k = dup; k < 16u; k++) qc(m.qpos + p0 + k, (uint8_t)(w[k >> 2] >> (8u * (k & 3u))));
            }
        } else for (uint32_t i = part; i < n; i += P) { const uint8_t q = s_text[rc ? m.qsrc + n - 1u - i : m.qsrc + i]; o[i] = q; qc(m.qpos + i, q); }
    }
    g2_bases(s_text, m, part, P, lpk, lnb, rflag, rn);
}
// FastqMeta::parse + RfqCodec::encodeChunk's pass 1 (src/fastqmeta.cpp:22-80, src/rfqcodec.cpp:220-263) for the reads of the tile k_gather2 has staged:
// the name line is in LDS already, so the text is not fetched a third time for the names (the separate read-table pass cost 8.1 GB / 1.9 ms
// on configs[2]).  ONE WAVE of the workgroup per tile - a different one every tile, so that the extra work spreads over the SIMDs - A LANE PER READ,
// and no loop over the name's bytes:
//   * the name's first 64 bytes become a 64-bit colon mask and a 64-bit space mask (four 16-byte LDS reads, SWAR byte equality);
//   * the parse is a function of those masks: the reference's loop stops at the first space or the seventh colon, whichever comes first; the fields are
//     the digits between colons 3|4, 4|5, 5|6, 6|7, and a space that ends the name part early takes over the field it closes (restated below);
//   * a field of up to eight digits is converted from one 8-byte LDS read (SWAR: pairs, then fours);
//   * the comparisons with the chunk's read 0 (staged once per workgroup: G2Ref) run 16 bytes per step; an odd read meets its mate's fields through
//     a shift by one lane.  The verdicts are accumulated per lane (G2Acc, see CF_ALL) and leave the workgroup as one atomicAnd / atomicMin per wave.
// A name whose first 64 bytes hold neither a space nor seven colons, a field with a sign / white space / more than eight characters: the byte-wise
// dev_parse_name / dev_atoi decide.  What this replaced, on configs[2] (k_gather2 alone: 3.5 ms): a lane per read walking its name byte by byte,
// 7.1 ms - one wave in a chain of dependent LDS reads, three waiting at the barrier; four lanes per read on 16 bytes each, every wave, 4.9 ms - ~600
// instructions per wave and tile, most of them the same work four times over.
#define G2_REFN 256u              // bytes of read 0's name kept in LDS (a longer one is compared from global memory)
#define G2_REFS 128u              // ... of its strand line
// read 0 of the chunk: lengths, parsed fields, where its name / strand line start in the text
struct G2Ref { uint32_t nl, n1l, n2o, len, stl, lane, tile, nb, tb; int s; };
struct G2Acc { uint32_t bits, fail; };
// n bytes at LDS offsets a and b of tx: are they equal?  16 bytes per step; a length that is not a multiple of 16 ends with a group moved back to end at
// n (>= 16 bytes) or with one masked group (< 16).  Every lane of the wave must call it (the loop runs while any lane has bytes left); `on` = mine count.
__device__ __forceinline__ bool lane_bytes_eq(const uint8_t* tx, uint32_t a, uint32_t b, uint32_t n, bool on) {
    bool eq = true;
    for (uint32_t o = 0; __any(on && eq && o < n); o += 16u) {
        if (on && eq && o < n) {
            uint32_t p0 = o, v = n - o; if (v < 16u && n >= 16u) { p0 = n - 16u; v = 16u; }
            uint32_t x[4], y[4]; lds_get16(tx, a + p0, x); lds_get16(tx, b + p0, y);
            unsigned long long dl = (((unsigned long long)(x[1] ^ y[1])) << 32) | (x[0] ^ y[0]), dh = (((unsigned long long)(x[3] ^ y[3])) << 32) | (x[2] ^ y[2]);
            if (v < 16u) { dl &= v >= 8u ? ~0ull : (1ull << (8u * v)) - 1ull; dh &= v > 8u ? (1ull << (8u * (v - 8u))) - 1ull : 0ull; }
            if (dl | dh) eq = false;
        }
    }
    return eq;
}
// the common sizes without a loop: n <= 32 bytes as one or two 16-byte groups (the second moved back to end at n; one masked group below 16)
__device__ __forceinline__ bool lane_bytes_eq32(const uint8_t* tx, uint32_t a, uint32_t b, uint32_t n) {
    uint32_t x[4], y[4]; lds_get16(tx, a, x); lds_get16(tx, b, y);
    unsigned long long dl = (((unsigned long long)(x[1] ^ y[1])) << 32) | (x[0] ^ y[0]), dh = (((unsigned long long)(x[3] ^ y[3])) << 32) | (x[2] ^ y[2]);
    if (n < 16u) { dl &= n >= 8u ? ~0ull : (1ull << (8u * n)) - 1ull; dh &= n > 8u ? (1ull << (8u * (n - 8u))) - 1ull : 0ull; }
    else { const uint32_t t = n - 16u; lds_get16(tx, a + t, x); lds_get16(tx, b + t, y); dl |= (((unsigned long long)(x[1] ^ y[1])) << 32) | (x[0] ^ y[0]);
            dh |= (((unsigned long long)(x[3] ^ y[3])) << 32) | (x[2] ^ y[2]); }
    return (dl | dh) == 0ull;
}
// the same against read 0's bytes [off0, off0 + n): LDS (offset ro of tx) when read 0's line fits the part of it kept there, else global memory
__device__ __forceinline__ bool g2_eq_ref(const uint8_t* tx, uint32_t a, uint32_t ro, uint32_t cap, const uint8_t* g0, uint32_t off0, uint32_t len0, uint32_t n,
        bool on) {
    const bool slow = on && len0 > cap, big = on && !slow && n > 32u; bool eq = true;
    if (on && !slow && !big && n) eq = lane_bytes_eq32(tx, a, ro + off0, n);
    if (__any(big)) { if (!lane_bytes_eq(tx, a, ro + off0, n, big)) eq = false; }   // (rare: wave-uniform)
    if (slow) for (uint32_t i = 0; i < n && eq; i++) if (tx[a + i] != g0[off0 + i]) eq = false;
    return eq;
}
// digits of tx[a, a + n) as glibc's atoi reads them: the common form - at most eight characters, the first neither white space nor a sign - from one
// 8-byte LDS read; anything else byte by byte
__device__ __forceinline__ uint32_t g2_atoi(const uint8_t* tx, uint32_t a, uint32_t n) {
    if (n == 0) return 0u;
    const unsigned long long w = lds_get8(tx, a); const uint32_t c0 = (uint32_t)w & 0xFFu;
    if (n > 8u || c0 == ' ' || (c0 >= 9u && c0 <= 13u) || c0 == '+' || c0 == '-') return (uint32_t)dev_atoi(tx + a, n);
    const unsigned long long x = w ^ 0x3030303030303030ull;                             // a digit's byte is now its value 0 .. 9
    const unsigned long long nd = (((x & 0x7F7F7F7F7F7F7F7Full) + 0x7676767676767676ull) | x) & 0x8080808080808080ull;   // 0x80 in every byte that is not a digit
    uint32_t m = nd ? (uint32_t)(__ffsll((long long)nd) - 1) >> 3 : 8u; if (m > n) m = n;      // leading digits: atoi stops at the first other byte
    if (m == 0) return 0u;
    const unsigned long long X = x << (8u * (8u - m));                                  // last digit in byte 7, zeros (leading zero digits) in front
    const uint32_t hi4 = (uint32_t)X, lo4 = (uint32_t)(X >> 32);                        // four digits each, the most significant one in the lowest byte
    // pairs: d0 d1 -> 10 d0 + d1 (no carry between bytes: <= 99)
    const uint32_t uh = ((hi4 << 3) + (hi4 << 1) + (hi4 >> 8)) & 0x00FF00FFu, ul = ((lo4 << 3) + (lo4 << 1) + (lo4 >> 8)) & 0x00FF00FFu;
    const uint32_t vh = mul24(uh & 0xFFu, 100u) + (uh >> 16), vl = mul24(ul & 0xFFu, 100u) + (ul >> 16);
    return mul24(vh, 10000u) + vl;                                                      // (24-bit multiplies run at full rate, v_mul_lo_u32 at a quarter)
}
__device__ __forceinline__ uint32_t ctz64_or64(unsigned long long m) { return m ? (uint32_t)(__ffsll((long long)m) - 1) : 64u; }
__device__ __forceinline__ void g2_parse(const Text& T, const ReadTab& R, const uint8_t* tx, uint32_t refn, uint32_t refs, const G2Geo& g, const G2Ref& r0, uint32_t f,
        uint32_t cur, uint32_t cnt,
                                         bool can0, uint32_t dpos, uint32_t dch, G2Acc& acc) {
    const uint32_t l = (uint32_t)lane_id(); const bool on = l < cnt; const uint32_t gi = cur + l;
    uint32_t nsrc = 0, nl = 0, sl = 0, tsrc = 0, tl = 0;
    if (on) {
        int s_; uint32_t r_; read_loc(T, gi, s_, r_);
        const uint4 lo4 = *(const uint4*)(t_lo(T, s_) + 4 * (size_t)r_);
        const uint32_t lb = s_ ? g.base1 : 0u, a = s_ ? g.a01 : g.a00;
        nsrc = lb + (lo4.x - a); nl = lo4.y - 1u - lo4.x; sl = lo4.z - 1u - lo4.y; tsrc = lb + (lo4.z - a); tl = lo4.w - 1u - lo4.z;
    }
    // ---- colon / space masks of the name's first 64 bytes (what lies behind the name is read too - it is inside the tile or its slack - and masked off)
    unsigned long long Cm = 0, Sm = 0;
    {
        uint32_t c[4], sp_[4];
#pragma unroll
        for (int p = 0; p < 4; p++) { uint32_t w[4]; lds_get16(tx, nsrc + 16u * (uint32_t)p, w); const uint4 q = make_uint4(w[0], w[1], w[2], w[3]);
                c[p] = eq_mask16c(q, 0x3A3A3A3Au); sp_[p] = eq_mask16c(q, 0x20202020u); }
        const unsigned long long keep = !on ? 0ull : (nl >= 64u ? ~0ull : (1ull << nl) - 1ull);
        Cm = ((((unsigned long long)(c[2] | (c[3] << 16))) << 32) | (c[0] | (c[1] << 16))) & keep;
        Sm = ((((unsigned long long)(sp_[2] | (sp_[3] << 16))) << 32) | (sp_[0] | (sp_[1] << 16))) & keep;
    }
    // ---- the parse as a function of the masks (src/fastqmeta.cpp:22-80: the loop stops at the first space or at the seventh colon; at a colon
    // numbered 4 .. 7 and at a space behind colon 4 .. 6 the digits since the previous colon become lane / tile / x / y)
    const uint32_t sp = ctz64_or64(Sm);
    unsigned long long cb = sp < 64u ? Cm & ((1ull << sp) - 1ull) : Cm;                // colons in front of the first space
    const uint32_t k = (uint32_t)__popcll(cb);
    uint32_t cpos[8];
#pragma unroll
    for (int i = 1; i <= 7; i++) { cpos[i] = ctz64_or64(cb); cb &= cb - 1ull; }
    const bool at7 = cpos[7] < 64u, at_sp = !at7 && sp < 64u;                         // where the loop stops (inside these 64 bytes)
    const bool undecided = on && !at7 && !at_sp && nl > 64u;                           // the stop, if any, lies further on
    uint32_t ok = 0, n1l = nl, n2o = nl, lane_v = 0, tile_v = 0, x_v = 0, y_v = 0;
    if (on && (at7 || (at_sp && k >= 4u))) {
        ok = 1; n2o = at7 ? cpos[7] : sp;
        const bool k4 = at_sp && k == 4u, k5 = at_sp && k == 5u;
        n1l = k4 ? cpos[4] : cpos[3];                                                  // cstart - 1: the colon in front of the lane field
        const uint32_t ls = (k4 ? cpos[4] : cpos[3]) + 1u, le = k4 ? sp : cpos[4];
        lane_v = g2_atoi(tx, nsrc + ls, le - ls) & 0xFFu;                              // (uint8_t)
        if (k >= 5u) { const uint32_t ts = (k5 ? cpos[5] : cpos[4]) + 1u, te = k5 ? sp : cpos[5]; tile_v = g2_atoi(tx, nsrc + ts, te - ts) & 0xFFFFu; }   // (uint16_t)
        if (k >= 6u) x_v = g2_atoi(tx, nsrc + cpos[5] + 1u, cpos[6] - cpos[5] - 1u);
        if (at7) y_v = g2_atoi(tx, nsrc + cpos[6] + 1u, cpos[7] - cpos[6] - 1u);
        else if (k == 6u) y_v = g2_atoi(tx, nsrc + cpos[6] + 1u, sp - cpos[6] - 1u);
    }
    if (__any(undecided)) {                                                            // (rare: wave-uniform)
        if (undecided) { const Meta m = dev_parse_name(tx + nsrc, nl); ok = m.ok; n1l = m.name1_len; n2o = m.name2_off; lane_v = m.lane; tile_v = m.tile; x_v = m.x;
                y_v = m.y; }
    }
    if (on) { R.name1_len[gi] = n1l; R.name2_off[gi] = n2o; R.x[gi] = x_v; R.y[gi] = y_v; R.tile[gi] = (uint16_t)tile_v; R.lane[gi] = (uint8_t)lane_v;
            R.ok[gi] = (uint8_t)ok; }
    // ---- against read 0 of the chunk
    const uint32_t n2l = nl - n2o, n2l0 = r0.nl - r0.n2o;
    const uint8_t* g0n = t_fq(T, r0.s) + r0.nb; const uint8_t* g0s = t_fq(T, r0.s) + r0.tb;
    const bool st_eq = g2_eq_ref(tx, tsrc, refs, G2_REFS, g0s, 0u, r0.stl, tl, on && tl == r0.stl);
    const bool n1_eq = g2_eq_ref(tx, nsrc, refn, G2_REFN, g0n, 0u, r0.nl, n1l, on && n1l == r0.n1l);
    const bool n2_eq = g2_eq_ref(tx, nsrc + n2o, refn, G2_REFN, g0n, r0.n2o, r0.nl, n2l, on && n2l == n2l0);
    // ---- an odd read and its mate (the lane in front: tiles start at even reads and hold whole pairs)
    const uint32_t pn = wave_shr1(nsrc, 0u), pnl = wave_shr1(nl, 0u), pn2o = wave_shr1(n2o, 0u), plane = wave_shr1(lane_v, 0u), ptile = wave_shr1(tile_v,
            0u), px = wave_shr1(x_v, 0u), py = wave_shr1(y_v, 0u);
    const uint32_t rel = gi - f; const bool odd = on && can0 && (rel & 1u);
    bool fa = false;                                                                   // (R1's name2 with [dpos] = dch) != R2's name2   (src/rfqcodec.cpp:237-245)
    if (__any(odd)) {
        // byte dpos apart, the names must be equal; at dpos the mate's byte - or dch in its place - must be mine
        const uint32_t pn2l = pnl - pn2o; const bool same_len = odd && pn2l == n2l;
        if (odd && !same_len) fa = true;
        const uint32_t ma = pn + pn2o, mb = nsrc + n2o; const bool small = same_len && n2l <= 16u, big = same_len && !small;
        if (small && n2l) {                                                            // one group: the mate's bytes with [dpos] patched, against mine
            uint32_t x[4], y[4]; lds_get16(tx, ma, x); lds_get16(tx, mb, y);
            if (dch != 0u && dpos < n2l) {
                const uint32_t sh = 8u * (dpos & 3u);
#pragma unroll
                for (int q = 0; q < 4; q++) if ((dpos >> 2) == (uint32_t)q) x[q] = (x[q] & ~(0xFFu << sh)) | (dch << sh);      // (static indices: no scratch)
            }
            unsigned long long dl = (((unsigned long long)(x[1] ^ y[1])) << 32) | (x[0] ^ y[0]), dh = (((unsigned long long)(x[3] ^ y[3])) << 32) | (x[2] ^ y[2]);
            if (n2l < 16u) { dl &= n2l >= 8u ? ~0ull : (1ull << (8u * n2l)) - 1ull; dh &= n2l > 8u ? (1ull << (8u * (n2l - 8u))) - 1ull : 0ull; }
            if (dl | dh) fa = true;
        }
        if (__any(big)) {                                                              // (rare: wave-uniform) in front of dpos, at dpos, behind it
            if (!lane_bytes_eq(tx, ma, mb, dpos < n2l ? dpos : n2l, big)) fa = true;
            if (big && dpos < n2l) { if ((dch != 0u ? dch : (uint32_t)tx[ma + dpos]) != (uint32_t)tx[mb + dpos]) fa = true; }
            if (!lane_bytes_eq(tx, ma + dpos + 1u, mb + dpos + 1u, n2l > dpos + 1u ? n2l - dpos - 1u : 0u, big && dpos + 1u < n2l)) fa = true;
        }
    }
    if (on) {
        uint32_t b = 0;
        if (sl == r0.len) b |= 1u << 0;
        if (n1l == r0.n1l) b |= 1u << 1;
        if (n2l == n2l0) b |= 1u << 2;
        if (tl == r0.stl) b |= 1u << 3;
        if (tl == r0.stl && st_eq) b |= 1u << 4;
        if (lane_v == r0.lane) b |= 1u << 5;
        if (tile_v == r0.tile) b |= 1u << 6;
        if (n1l == r0.n1l && n1_eq) b |= 1u << 7;
        const bool e2 = n2l == n2l0 && n2_eq;
        if (e2) b |= 1u << 8;
        if (e2 || (rel & 1u)) b |= 1u << 9;
        acc.bits &= b;
        R.eq2[gi] = e2 ? 1 : 0;
        if (odd) {
            const bool fb = plane != lane_v || ptile != tile_v || px != x_v || py != y_v;
            if (fa || fb) { const uint32_t key = (rel << 1) | (fa ? 0u : 1u); if (key < acc.fail) acc.fail = key; }
        }
    }
}
// ---- MASKS mode (files with at most four coded quality values - a NovaSeq-binned file has three, or four with the table's 0xFF entry): no quality bytes leave the kernel.  A lane turns its 16
// bytes into one 16-bit match mask per value and ORs them, shifted to their chunk position, into bit planes of the tile in LDS (ds_or, no return); after the
// barrier the planes leave as whole 32-bit words - coalesced, plain stores - and are counted on the way (popcount per segment, last match: what QualCount
// did with two LDS atomics per coded byte).  A word that straddles two tiles of a workgroup is carried to the next tile; one that straddles two workgroups is
// OR-ed into global memory by both (k_mask_bounds has zeroed those words).  A byte that is neither the major value nor a coded one (rare: the header's table
// comes from chunk 0) goes to qcat at its position, its bit into the exception plane (global atomicOr on a plane zeroed per batch).
// The planes of a batch: the plane of coded value j (its index in the header's table, j < 4) at planes + j * pstride (u32 words; chunk c's words start at
// qbase[c] >> 5), the exception plane at index 4.  The `nd` most frequent values (DevHeader::dense) are DENSE: built in LDS and stored whole.  The others -
// on a NovaSeq-binned file '#', which only N bases carry, and the 0xFF entry the reference appends to the table of a file whose N bases have no quality of
// their own (src/rfqheader.cpp:214-230) - and the exceptions are RARE: their planes stay all-zero between batches, bits are OR-ed in where there is one, rare[c]
// remembers the chunks that have any, and k_rare_cleanup zeroes those again behind the coder.  (Three LDS planes of a 64-read tile would cost the kernel its
// sixth resident workgroup - 10 % - on 150-base reads; two fit.)
#define G2_PLANES 5u
#define G2_PLANE_EXC 4u
#define G2_RARE_LIST 255u          // words of rare planes a chunk may touch before the cleanup zeroes its whole extent instead
// pw: words per LDS plane; nd: dense planes; rare: [n_chunks][1 + G2_RARE_LIST]: count, then (plane << 28 | word of the chunk)
struct G2Planes { uint32_t* planes; uint64_t pstride; uint32_t pw, nd; uint32_t* rare; };
__device__ __forceinline__ void g2_rare_or(uint32_t* __restrict__ gpl, uint64_t pstride, uint32_t* __restrict__ rare_c, uint32_t plane, uint32_t pos) {
    const uint32_t old = atomicOr(&gpl[(size_t)plane * pstride + (pos >> 5)], 1u << (pos & 31u));
    // the word's first bit: remember the word
    if (old == 0u) { const uint32_t k = atomicAdd(rare_c, 1u); if (k < G2_RARE_LIST) rare_c[1u + k] = (plane << 28) | (pos >> 5); }
}
__device__ __forceinline__ void g2_quals_masks(const uint8_t* tx, const G2Read& m, uint32_t part, uint32_t P, uint32_t* pl, uint32_t pw, uint32_t wbase, uint32_t nd,
                                               uint32_t pat0, uint32_t pat1, uint32_t pat2, uint32_t patm, const DevHeader* __restrict__ D, uint8_t* qd, uint32_t* __restrict__ gpl, uint64_t pstride,
                                               uint32_t* __restrict__ segm_c, int* __restrict__ segc_c, uint32_t n_seg, uint32_t* __restrict__ rare_c) {
    if (!m.on) return;
    const uint32_t n = m.len, ng = (n + 15u) >> 4; const bool rc = m.rc;
    for (uint32_t gi = part; gi < ng; gi += P) {
        // (a last, partial group reads past the line - in front of it, for a reversed mate - and masks those bits off)
        const uint32_t p0 = 16u * gi, nv = n - p0;
        uint32_t w[4]; lds_get16(tx, rc ? m.qsrc + n - p0 - 16u : m.qsrc + p0, w);
        if (rc) { const uint32_t x0 = bswap32(w[3]), x1 = bswap32(w[2]), x2 = bswap32(w[1]), x3 = bswap32(w[0]); w[0] = x0; w[1] = x1; w[2] = x2; w[3] = x3; }
        const uint32_t valid = nv >= 16u ? 0xFFFFu : (1u << nv) - 1u;
        const uint4 q = make_uint4(w[0], w[1], w[2], w[3]);
        const uint32_t bit = m.qpos + p0 - wbase, wi = bit >> 5, sh = bit & 31u;
        uint32_t known = eq_mask16c(q, patm);
        auto plane = [&](uint32_t d, uint32_t pat) {
            const uint32_t mv = eq_mask16c(q, pat); known |= mv;
            const unsigned long long x = (unsigned long long)(mv & valid) << sh;
            if ((uint32_t)x) atomicOr(&pl[d * pw + wi], (uint32_t)x);
            if ((uint32_t)(x >> 32)) atomicOr(&pl[d * pw + wi + 1u], (uint32_t)(x >> 32));
        };
        if (nd > 0u) plane(0u, pat0);                                       // (nd is the same for every lane)
        if (nd > 1u) plane(1u, pat1);
        if (nd > 2u) plane(2u, pat2);
        uint32_t rest = ~known & valid;
        // (rare) neither the major value nor a dense one: a rare coded value, or one the header's table does not know
        while (rest) {
            const uint32_t k = (uint32_t)__ffs((int)rest) - 1u; rest &= rest - 1u;
            const uint32_t ww = k < 8u ? (k < 4u ? w[0] : w[1]) : (k < 12u ? w[2] : w[3]), pos = m.qpos + p0 + k, b = (ww >> (8u * (k & 3u))) & 0xFFu;
            const uint32_t j = D->stream_of[b];
            if (j < 4u && j < D->n_normal) { const size_t si = (size_t)j * n_seg + pos / PC_SEG_POS; g2_rare_or(gpl, pstride, rare_c, j, pos); atomicAdd(&segm_c[si], 1u);
                    atomicMax(&segc_c[si], (int)pos); }
            else { qd[pos] = (uint8_t)b; g2_rare_or(gpl, pstride, rare_c, G2_PLANE_EXC, pos); atomicAdd(&segm_c[(size_t)EXC_SLOT * n_seg + pos / PC_SEG_POS], 1u); }
        }
    }
}
// the tile's planes -> global words [gw0, gw0 + nw) of each plane, counted per coder segment; the LDS planes are left zeroed, a word that the next tile of
// this workgroup continues (carry) stays behind in s_carry.  or_first / or_last: that word is shared with another workgroup.
__device__ __forceinline__ void g2_flush_masks(uint32_t* pl, uint32_t pw, uint32_t nd, uint32_t dense3 /* the dense planes' streams, a byte each */, uint32_t* s_carry, uint32_t* __restrict__ gpl, uint64_t pstride,
                                               uint32_t gw0, uint32_t nw, bool carry, bool or_first, bool or_last,
                                               size_t seg_index0 /* (c * MAX_STREAMS) * n_seg */, uint32_t n_seg, uint32_t* __restrict__ segm, int* __restrict__ segc) {
    const uint32_t tid = threadIdx.x, seg0 = (gw0 << 5) / PC_SEG_POS;
    for (uint32_t d = 0; d < nd; d++) {                                     // (uniform)
        const uint32_t v = (dense3 >> (8u * d)) & 0xFFu;                   // LDS plane d holds coded value v
        uint32_t c01 = 0; int l0 = -1, l1 = -1;
        for (uint32_t i = tid; i < nw; i += blockDim.x) {
            const uint32_t x = pl[d * pw + i]; pl[d * pw + i] = 0u;
            const bool last = i + 1u == nw;
            if (last && carry) { s_carry[d] = x; continue; }
            const uint32_t gw = gw0 + i; uint32_t* const dst = gpl + (size_t)v * pstride + gw;
            if ((i == 0u && or_first) || (last && or_last)) { if (x) atomicOr(dst, x); } else *dst = x;
            if (x) { const uint32_t sg = ((gw << 5) / PC_SEG_POS) - seg0; const int lp = (int)((gw << 5) + 31u - (uint32_t)__clz((int)x));
                     c01 += (uint32_t)__popc(x) << (16u * sg); if (sg) { if (lp > l1) l1 = lp; } else if (lp > l0) l0 = lp; }
        }
        if (!carry && tid == 0) s_carry[d] = 0u;
        c01 = wave_sum(c01); l0 = wave_max(l0); l1 = wave_max(l1);
        if ((tid & 63u) == 0 && c01) {
            const size_t si = seg_index0 + (size_t)v * n_seg + seg0;
            if (c01 & 0xFFFFu) { atomicAdd(&segm[si], c01 & 0xFFFFu); atomicMax(&segc[si], l0); }
            if ((c01 >> 16) && seg0 + 1u < n_seg) { atomicAdd(&segm[si + 1u], c01 >> 16); atomicMax(&segc[si + 1u], l1); }
        }
    }
}
// the words of the batch's planes that two workgroups of k_gather2<true> OR into: zeroed (per = reads per workgroup, as the gather computes it)
__global__ void k_mask_bounds(const uint32_t* __restrict__ pq, const uint32_t* __restrict__ first, const uint64_t* __restrict__ qbase, uint32_t* __restrict__ planes,
        uint64_t pstride, const DevHeader* __restrict__ D, uint32_t nd, uint32_t bx,
                              const uint32_t* __restrict__ only) {
    const uint32_t c = blockIdx.x; if (only && !only[c]) return;
    const uint32_t f = first[c], e = first[c + 1], pq0 = pq[f];
    uint32_t per = ((e - f) + bx - 1) / bx; per = (per + 1u) & ~1u;
    for (uint32_t b = 1u + threadIdx.x; b < bx; b += blockDim.x) {
        const uint32_t gs = f + b * per; if (gs >= e) break;
        const uint32_t w = (uint32_t)(qbase[c] >> 5) + ((pq[gs] - pq0) >> 5);
        for (uint32_t d = 0; d < nd; d++) planes[(size_t)D->dense[d] * pstride + w] = 0u;
    }
}
// phase 1: every chunk, mates taken for interleaved wherever the header allows it (the names that decide are parsed in this very pass), names parsed and
// compared; phase 2: only the chunks k_chunk_flags_b marked in `only` - their interleave test failed somewhere - once more with the mates as they stand.
// Dynamic LDS: [text4 x 16 bytes of staged text, slack included][read 0's name and strand line][MASKS: three planes of M.pw words].
// PE = false: single-end input - no second stream, no mates, nothing stored back to front: the instantiation drops those paths and the values they keep alive
// PAIRED = 0 / 1 / 2: the input's pairing (RFQ_SE / RFQ_PE_TWO_FILES / RFQ_PE_INTERLEAVED) as a compile-time constant - every read_loc, every "which stream" select and, for
// single-end input, the mates' whole path fold away; -1: taken from the argument
template <bool MASKS, int PAIRED = -1> __global__ void __launch_bounds__(256, 6) k_gather2(Text T_, ReadTab R, const uint32_t* __restrict__ first, const uint64_t* __restrict__ qbase,
                                                 const DevHeader* __restrict__ D, uint8_t* __restrict__ qcat, uint32_t* __restrict__ lpk, uint16_t* __restrict__ lnb, uint8_t* __restrict__ rflag,
                                                 uint8_t* __restrict__ rn, uint32_t* __restrict__ segm, int* __restrict__ segc, uint32_t n_seg, uint32_t kshift,
                                                 uint32_t* __restrict__ cbits, uint32_t* __restrict__ cfail, const uint32_t* __restrict__ only, uint32_t text4, G2Planes M) {
    Text T = T_; if (PAIRED >= 0) T.paired = PAIRED;
    constexpr bool PE = PAIRED != 0;
    RFQ_DYN_SHARED(uint4, g2_lds);
    __shared__ uint32_t sh[MASKS ? 1 : G2_CNT]; __shared__ int sh_last[MASKS ? 1 : G2_CNT]; __shared__ uint8_t s_slot[MASKS ? 16 : 256];
            __shared__ uint32_t s_r0[8], s_carry[4];
    const uint32_t REFN = (text4 - 1u) * 16u, REFS = REFN + G2_REFN + 16u;      // (byte offsets from the tile's first byte)
    uint32_t* const pl = (uint32_t*)(g2_lds + text4 + (G2_REFN + G2_REFS + 32u) / 16u);
    const uint32_t c = blockIdx.y;
    const bool redo = only != nullptr;
    if (redo && !only[c]) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t* __restrict__ pq = R.pq;
    const uint32_t nn_s = D->n_normal < NPOS_SLOT ? D->n_normal : NPOS_SLOT, nslot = nn_s + 1u;
    uint32_t nrep = 1;
    if (!MASKS) {
        for (uint32_t i = tid; i < G2_CNT; i += blockDim.x) { sh[i] = 0; sh_last[i] = -1; }
        while (nrep < 16u && 4u * nrep * nslot <= G2_CNT) nrep *= 2u;
        for (uint32_t i = tid; i < 256; i += blockDim.x) { const uint32_t j = D->stream_of[i]; s_slot[i] = (uint8_t)(j < nn_s ? j : nn_s); }
    } else {
        for (uint32_t i = tid; i < M.nd * M.pw; i += blockDim.x) pl[i] = 0u;
        if (tid < 4u) s_carry[tid] = 0u;
    }
    const uint32_t f = first[c], e = first[c + 1];
    const bool two = PE && T.paired == 1, can0 = PE && T.paired != 0 && D->support_interleaved != 0, il = can0 && !redo;
    const uint32_t dpos = D->name2_diff_pos, dch = D->name2_diff_char;
    uint8_t* const qd = qcat + qbase[c]; const uint32_t pq0 = pq[f];
    uint32_t per = ((e - f) + gridDim.x - 1) / gridDim.x; per = (per + 1u) & ~1u;       // whole pairs per workgroup
    const uint32_t gs = f + blockIdx.x * per, ge = gs + per < e ? gs + per : e;
    const uint32_t K = 1u << kshift, pshift = 8u - kshift, P = 1u << pshift;              // K reads per tile, P threads per read
    // thread -> (read of the tile, part of it): the even reads first, then the odd ones - an interleaved chunk's mates (odd, reverse-complemented) and
    // their R1 take different paths through the base packer, and a wave that holds both runs both
    const uint32_t jj = tid >> pshift, j = ((jj << 1) & (K - 1u)) | (jj >> (kshift - 1u)), part = tid & (P - 1u);
    QualCount qc; qc.cnt = sh; qc.last = sh_last; qc.slot = s_slot; qc.major = D->major & 0xFFu; qc.seg0 = 0; qc.nslot = nslot; qc.rep = tid & (nrep - 1u);
            qc.hot_ok = D->stream_of[D->major & 0xFFu] == 0xFF;
    // planes built in LDS, and whose they are
    const uint32_t nd = MASKS ? M.nd : 0u, dense3 = (uint32_t)D->dense[0] | ((uint32_t)D->dense[1] << 8) | ((uint32_t)D->dense[2] << 16);
    const uint32_t pat0 = (uint32_t)D->normal[dense3 & 0xFFu] * 0x01010101u, pat1 = (uint32_t)D->normal[(dense3 >> 8) & 0xFFu] * 0x01010101u, pat2 = (uint32_t)D->normal[(dense3 >> 16) & 0xFFu] * 0x01010101u, patm = (D->major & 0xFFu) * 0x01010101u;
    uint32_t* const gpl = M.planes + (MASKS ? (size_t)(qbase[c] >> 5) : (size_t)0);        // the chunk's words of plane 0
    uint4* const buf4 = g2_lds + 1; const uint8_t* const tx = (const uint8_t*)buf4;
    G2Ref r0 = {}; G2Acc acc; acc.bits = CF_ALL; acc.fail = 0xFFFFFFFFu;
    const bool parse = !redo && gs < ge;                                    // block-uniform
    if (parse) {
        // read 0 of the chunk: the first bytes of its name and strand lines into LDS, its name parsed by one lane
        uint32_t r_; read_loc(T, f, r0.s, r_); const uint32_t* p = t_lo(T, r0.s) + 4 * (size_t)r_;
        const uint32_t p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
        r0.nb = p0; r0.nl = p1 - 1u - p0; r0.len = p2 - 1u - p1; r0.tb = p2; r0.stl = p3 - 1u - p2;
        uint8_t* const wr = (uint8_t*)buf4;
        if (tid < r0.nl) wr[REFN + tid] = t_fq(T, r0.s)[r0.nb + tid];      // (G2_REFN = the workgroup's 256 threads)
        if (tid < G2_REFS && tid < r0.stl) wr[REFS + tid] = t_fq(T, r0.s)[r0.tb + tid];
        __syncthreads();
        if (tid == 0) {
            const Meta m0 = r0.nl <= G2_REFN ? dev_parse_name(tx + REFN, r0.nl) : dev_parse_name(t_fq(T, r0.s) + r0.nb, r0.nl);
            s_r0[0] = m0.name1_len; s_r0[1] = m0.name2_off; s_r0[2] = m0.lane; s_r0[3] = m0.tile;
        }
    }
    __syncthreads();
    if (parse) { r0.n1l = s_r0[0]; r0.n2o = s_r0[1]; r0.lane = s_r0[2]; r0.tile = s_r0[3]; }
    uint32_t tix = blockIdx.x + blockIdx.y;                                 // (which wave parses: another one every tile, and not the same one in every workgroup)
    // boundaries b0 .. b2 of tiles t, t + 1, t + 2 are here, b3 is requested; my read's lines in tile t are in `mr` (requested a tile ago)
    const uint32_t ntile = gs < ge ? (ge - gs + K - 1u) >> kshift : 0u;
    auto tile_at = [&](uint32_t t_) -> uint32_t { const uint32_t x = gs + (t_ << kshift); return x < ge ? x : ge; };
    G2Bound b0 = g2_bound(T, two, pq, tile_at(0)), b1 = g2_bound(T, two, pq, tile_at(1)), b2 = g2_bound(T, two, pq, tile_at(2));
    G2MRaw mr = g2_mraw(T, pq, gs, j, ntile ? tile_at(1) - gs : 0u);
    if (ntile) g2_stage(T, two, g2_geo(b0, b1, two), buf4, tid);
    for (uint32_t t = 0; t < ntile; t++, tix++) {                          // block-uniform
        const uint32_t cur = tile_at(t), cnt = tile_at(t + 1u) - cur;
        const G2Geo g = g2_geo(b0, b1, two);                                // (its text was requested before the previous tile's flush)
        const G2Read m = g2_read(T, mr, g, f, pq0, il, cur, j, cnt);
        const uint32_t qbeg = b0.q - pq0, qend = b1.q - pq0;               // the tile's quality positions (chunk-relative)
        { const uint32_t ncur = tile_at(t + 1u); mr = g2_mraw(T, pq, ncur, j, tile_at(t + 2u) - ncur); }
        const G2Bound b3 = g2_bound(T, two, pq, tile_at(t + 3u));
        __syncthreads();                                                    // (drains the LDS-DMA)
        qc.seg0 = qbeg / PC_SEG_POS;
        // (wave-uniform: this tile's parsing wave)
        if (parse && (uint32_t)wave_id() == (tix & 3u)) g2_parse(T, R, tx, REFN, REFS, g, r0, f, cur, cnt, can0, dpos, dch, acc);
        if (MASKS) {
            if (tid < nd && s_carry[tid]) atomicOr(&pl[tid * M.pw], s_carry[tid]);       // the word the tile in front left unfinished
            g2_quals_masks(tx, m, part, P, pl, M.pw, qbeg & ~31u, nd, pat0, pat1, pat2, patm, D, qd, gpl, M.pstride, segm + (size_t)c * MAX_STREAMS * n_seg,
                    segc + (size_t)c * MAX_STREAMS * n_seg, n_seg, M.rare + (size_t)c * (1u + G2_RARE_LIST));
            g2_bases(tx, m, part, P, lpk, lnb, rflag, rn);
        } else g2_compose(tx, m, part, P, qd, lpk, lnb, rflag, rn, qc);
        __syncthreads();                                                    // the text is free for the next tile; the tile's counts / planes are complete
        if (t + 1u < ntile) g2_stage(T, two, g2_geo(b1, b2, two), buf4, tid);  // the next tile's text is on its way while the planes / counters of this one leave
        if (MASKS) {
            const uint32_t gw0 = qbeg >> 5, nw = ((qend + 31u) >> 5) - gw0; const bool last_tile = cur + cnt >= ge;
            g2_flush_masks(pl, M.pw, nd, dense3, s_carry, gpl, M.pstride, gw0, nw, !last_tile && (qend & 31u) != 0u, cur == gs && gs > f && (qbeg & 31u) != 0u,
                    last_tile && ge < e && (qend & 31u) != 0u,
                           (size_t)c * MAX_STREAMS * n_seg, n_seg, segm, segc);
        } else qual_flush(sh, sh_last, nrep, nslot, qc.seg0, c, nn_s, segm, segc, n_seg);
        b0 = b1; b1 = b2; b2 = b3;
    }
    if (parse) {
        const uint32_t bits = wave_and(acc.bits), fail = wave_min(acc.fail);
        if ((tid & 63u) == 0) { if (bits != CF_ALL) atomicAnd(&cbits[c], bits); if (fail != 0xFFFFFFFFu) atomicMin(&cfail[c], fail); }
    }
}
// phase 2 of the gather re-counts the qualities of the chunks it repeats: their per-(stream, segment) entries back to "nothing seen"
__device__ __forceinline__ uint32_t dense_mask_of(const DevHeader* __restrict__ D, uint32_t nd) { uint32_t m = 0;
        for (uint32_t d = 0; d < nd; d++) m |= 1u << D->dense[d]; return m; }
__device__ __forceinline__ void k_rare_zero_chunk(uint32_t* __restrict__ planes, uint64_t pstride, uint32_t dense_mask, const uint32_t* __restrict__ pq,
        const uint32_t* __restrict__ first, const uint64_t* __restrict__ qbase, uint32_t c) {
    const uint32_t nw = (pq[first[c + 1]] - pq[first[c]] + 31u) >> 5; const size_t w0 = (size_t)(qbase[c] >> 5);
    for (uint32_t v = 0; v < G2_PLANES; v++) if (!((dense_mask >> v) & 1u)) for (uint32_t i = threadIdx.x; i < nw; i += blockDim.x) planes[(size_t)v * pstride + w0 + i] = 0u;
}
// (xplane: the planes of MASKS mode, or null - the chunk's words of the rare planes are zeroed: the repeat sets them afresh)
__global__ void k_gather_redo_reset(const uint32_t* __restrict__ only, uint32_t* __restrict__ segm, int* __restrict__ segc, uint32_t n_seg,
                                    uint32_t* __restrict__ xplane, uint64_t pstride, const DevHeader* __restrict__ D, uint32_t nd, const uint32_t* __restrict__ pq, const uint32_t* __restrict__ first, const uint64_t* __restrict__ qbase) {
    const uint32_t c = blockIdx.x; if (!only[c]) return;
    const size_t k = (size_t)c * MAX_STREAMS * n_seg;
    for (uint32_t i = threadIdx.x; i < MAX_STREAMS * n_seg; i += blockDim.x) { segm[k + i] = 0u; segc[k + i] = -1; }
    // (rare[] lies behind the planes)
    if (xplane) { k_rare_zero_chunk(xplane, pstride, dense_mask_of(D, nd), pq, first, qbase, c);
            if (threadIdx.x == 0) xplane[(size_t)G2_PLANES * pstride + (size_t)c * (1u + G2_RARE_LIST)] = 0u; }
}
// behind the coder: the rare planes all-zero again (chunks that set bits in them are marked in rare[])
__global__ void k_rare_cleanup(uint32_t* __restrict__ rare, uint32_t* __restrict__ planes, uint64_t pstride, const DevHeader* __restrict__ D, uint32_t nd,
        const uint32_t* __restrict__ pq, const uint32_t* __restrict__ first, const uint64_t* __restrict__ qbase) {
    const uint32_t c = blockIdx.x; uint32_t* const rc = rare + (size_t)c * (1u + G2_RARE_LIST); const uint32_t n = rc[0]; if (!n) return;
    if (n <= G2_RARE_LIST) { const size_t w0 = (size_t)(qbase[c] >> 5); for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { const uint32_t e = rc[1u + i];
            planes[(size_t)(e >> 28) * pstride + w0 + (e & 0x0FFFFFFFu)] = 0u; } }
    else k_rare_zero_chunk(planes, pstride, dense_mask_of(D, nd), pq, first, qbase, c);
    __syncthreads();
    if (threadIdx.x == 0) rc[0] = 0u;
}
// 16 consecutive codes / N bits of a loose slot from base index b on (b + 16 may pass the slot's end: the caller masks)
__device__ __forceinline__ uint32_t loose_codes(const uint32_t* __restrict__ lpk, uint32_t ld, uint32_t b) {
    const uint32_t d = ld + (b >> 4), sh = 2u * (b & 15u); const uint32_t lo = lpk[d];
    return sh ? (uint32_t)(((((unsigned long long)lpk[d + 1]) << 32) | lo) >> sh) : lo;
}
__device__ __forceinline__ uint32_t loose_nbits(const uint16_t* __restrict__ lnb, uint32_t ld, uint32_t b) {
    const uint32_t d = ld + (b >> 4), sh = b & 15u; const uint32_t lo = lnb[d];
    return (sh ? ((((uint32_t)lnb[d + 1]) << 16) | lo) >> sh : lo) & 0xFFFFu;
}
// Loose slots -> the chunk's tight streams: spk = 2-bit stored bases, 16 per dword, dword k of chunk c at (sbase[c] >> 4) + k - the bytes of the
// image's sequence section (RfqChunk::write copies them) - and snm = one "is N" bit per stored base at the same u16 index (the N-position
// coder's match mask).  A read's stored bases are slot[skip, skip + keep) (skip: what the overlap with R1 implies for a mate,
// src/rfqcodec.cpp:376-407) and go to tight positions sd .. sd + keep; the read OWNS the tight dwords whose first base is one of its own, and
// what is left of its last one comes from the read(s) behind it.  A workgroup takes R consecutive reads at a time, in two phases:
//   1  a lane per read: stored prefix, slot, skip -> LDS, and the read's index into s_own[] for every dword it owns (LDS stores, no search);
//   2  a lane per tight dword, consecutive lanes = consecutive dwords: owner from s_own[], its data from LDS, 16 codes + 16 N bits fetched from
//      the slot(s) with a funnel shift, stored.  Loads and stores are coalesced, two memory round trips per R reads, ~50 instructions per dword.
// What the four earlier forms cost on configs[2] (210 M dwords), and why: a lane per dword with the reads found by bisecting stored prefixes (in LDS /
// in the wave's lanes by shuffles) and a reverse complement in 2-bit space, 2.3 - 2.8 ms: ~600 instructions per dword; four lanes per read, one
// dword per round trip, 1.9 ms: the waves' chains of dependent round trips; a lane per read with all its loads up front, 2.3 ms: 64 scattered
// 4-byte (2-byte) stores per wave instruction - 420 M write requests at the L2's request rate (ablation: 1.0 of the 1.5 ms were the stores).
// N counts per coder segment, the chunk's N total and N map are left as k_gather leaves them.
#define SP_OWN 4096u              // tight dwords of one step (the host sizes R by the longest read: R * (max_len / 16 + 1) <= SP_OWN)
#define SP_EXTRA 8u               // reads behind the step's last whose LDS entries the last dword's tail may need (beyond: global memory)
#define SP_U 3                    // tight dwords per thread whose loads are in flight together (1 .. 3 measure the same beside the coder, 4 and more cost the stage 0.15 ms: registers)
struct __attribute__((packed, aligned(4))) SpU8 { uint32_t a, b; };
struct __attribute__((packed, aligned(2))) SpU4 { uint32_t a; };
__global__ void __launch_bounds__(256) k_seqpack(const uint32_t* __restrict__ pq, const uint32_t* __restrict__ sd, const U4* __restrict__ ptot,
        const uint32_t* __restrict__ first, const uint32_t* __restrict__ ilv, const int8_t* __restrict__ ovb,
                                                 const DevHeader* __restrict__ D, const uint64_t* __restrict__ sbase, const uint32_t* __restrict__ lpk, const uint16_t* __restrict__ lnb, const uint8_t* __restrict__ rn,
                                                 uint32_t* __restrict__ spk, uint16_t* __restrict__ snm, uint32_t* __restrict__ ncount, uint32_t* __restrict__ nmap, uint32_t* __restrict__ segm, int* __restrict__ segc, uint32_t n_seg,
                                                 uint32_t rshift) {
    __shared__ uint32_t s_sd[256 + SP_EXTRA + 1], s_ld[256 + SP_EXTRA], s_sk[256 + SP_EXTRA]; __shared__ uint8_t s_own[SP_OWN];
    // "this read holds an N" (k_gather2 leaves it): the N bits of the others' slots are all zero and are not fetched - half of the packer's loads (a probe that fetched none
    // of them: the phase beside the coder 3.31 -> 3.12 ms, profiles/r06_x_probe_no_n.txt)
    __shared__ uint8_t s_hn[256 + SP_EXTRA + 8];
    const uint32_t c = blockIdx.y, f = first[c], e = first[c + 1], tid = threadIdx.x;
    const uint32_t ps0 = sd[f], S = ptot[c].d;
    const bool il = ilv[c] != 0, enc = il && (D->flags & H_PE_OVERLAP); const int shift = D->overlap_shift;
    uint32_t* const ok = spk + (size_t)(sbase[c] >> 4); uint16_t* const on = snm + (size_t)(sbase[c] >> 4);
    const uint32_t nshift = nmap_shift(S); uint32_t* const nm = nmap + (size_t)c * NMAP_WORDS;
    const size_t nsi = ((size_t)c * MAX_STREAMS + NPOS_SLOT) * n_seg;
    uint32_t nsum = 0;
    auto skip_of = [&](uint32_t g) -> uint32_t {                            // leading bases of read g's slot that are not stored
        if (enc && ((g - f) & 1u)) { const int ov = (int)ovb[g >> 1] - shift; if (ov > 0) return (uint32_t)ov; }
        return 0u;
    };
    const uint32_t R = 1u << rshift;
    for (uint32_t r0 = f + blockIdx.x * R; r0 < e; r0 += gridDim.x * R) {   // block-uniform
        const uint32_t nr = e - r0 < R ? e - r0 : R, nx = e - r0 < R + SP_EXTRA ? e - r0 : R + SP_EXTRA;    // my reads; reads with LDS entries
        // ---- phase 1
        for (uint32_t t = tid; t <= nx; t += blockDim.x) {
            const uint32_t g = r0 + t; s_sd[t] = g < e ? sd[g] - ps0 : S;  // (sd[e] belongs to the next chunk)
            if (t < nx) { s_ld[t] = (pq[g] >> 4) + g; s_sk[t] = skip_of(g); s_hn[t] = rn[g]; } else s_hn[t] = 0;
        }
        __syncthreads();
        const uint32_t kbase = (s_sd[0] + 15u) >> 4, kend = (s_sd[nr] + 15u) >> 4;    // the step's dwords: those whose first base belongs to one of my reads
        if (tid < nr) { const uint32_t ka = (s_sd[tid] + 15u) >> 4, kb = (s_sd[tid + 1] + 15u) >> 4; for (uint32_t k = ka; k < kb; k++) s_own[k - kbase] = (uint8_t)tid; }
        __syncthreads();
        // ---- phase 2, SP_U dwords per thread at a time: every load a dword needs - sixteen codes and N bits from its owner's slot and from the slot of the
        // read behind it, which finishes a read's last dword - is requested before the first one is used (a dword at a time, the step was a chain of a dozen
        // round trips: 1.29 ms for a kernel with 0.47 ms of instructions)
        const uint32_t ndw = kend - kbase;
        for (uint32_t i0 = 0; i0 < ndw; i0 += blockDim.x * SP_U) {
            struct Dw { uint32_t k, need, t1, jj, sh, sh2, take2, n, n2; unsigned long long c, c2; } q[SP_U]; uint32_t nres[SP_U];
#pragma unroll
            for (int u = 0; u < SP_U; u++) nres[u] = 0;
#pragma unroll
            for (int u = 0; u < SP_U; u++) {
                Dw& x = q[u]; x.need = 0; x.take2 = 0; x.k = x.t1 = x.jj = x.sh = x.sh2 = x.n = x.n2 = 0; x.c = x.c2 = 0;
                const uint32_t i = i0 + (uint32_t)u * blockDim.x + tid;
                if (i < ndw) {
                    const uint32_t k = kbase + i, j = s_own[i];
                    const uint32_t B = 16u * k, si = B - s_sd[j], need = S - B < 16u ? S - B : 16u, av = s_sd[j + 1] - B, t1 = need < av ? need : av;
                    const uint32_t b0 = s_sk[j] + si, d = s_ld[j] + (b0 >> 4);
                    x.k = k; x.need = need; x.t1 = t1; x.sh = b0 & 15u; x.jj = j + 1u;
                    { const SpU8 v = *(const SpU8*)(lpk + d); x.c = (((unsigned long long)v.b) << 32) | v.a; x.n = s_hn[j] ? ((const SpU4*)(lnb + d))->a : 0u; }
                    if (t1 < need && j + 1u < nx) {                          // the read behind: its LDS entries are there
                        const uint32_t avail = s_sd[j + 2] - s_sd[j + 1]; x.jj = j + 2u;
                        if (avail) {
                            const uint32_t s2 = s_sk[j + 1], d2 = s_ld[j + 1] + (s2 >> 4); x.sh2 = s2 & 15u; x.take2 = avail < need - t1 ? avail : need - t1;
                            const SpU8 v = *(const SpU8*)(lpk + d2); x.c2 = (((unsigned long long)v.b) << 32) | v.a; x.n2 = s_hn[j + 1] ? ((const SpU4*)(lnb + d2))->a : 0u;
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < SP_U; u++) {
                const Dw& x = q[u];
                if (!x.need) continue;
                const uint32_t k = x.k, B = 16u * k, need = x.need;
                unsigned long long acc = (uint32_t)(x.c >> (2u * x.sh)) & (x.t1 >= 16u ? 0xFFFFFFFFu : (1u << (2u * x.t1)) - 1u);
                uint32_t nacc = ((x.n >> x.sh) & 0xFFFFu) & ((1u << x.t1) - 1u);
                uint32_t filled = x.t1, jj = x.jj;
                if (x.take2) {
                    acc |= (unsigned long long)((uint32_t)(x.c2 >> (2u * x.sh2)) & (x.take2 >= 16u ? 0xFFFFFFFFu : (1u << (2u * x.take2)) - 1u)) << (2u * filled);
                    nacc |= (((x.n2 >> x.sh2) & 0xFFFFu) & ((1u << x.take2) - 1u)) << filled; filled += x.take2;
                }
                while (filled < need) {                                         // (rare) reads of a few bases in a row, or reads beyond the step's LDS entries
                    uint32_t a, b, l2, s2;
                    if (jj < nx) { a = s_sd[jj]; b = s_sd[jj + 1]; l2 = s_ld[jj]; s2 = s_sk[jj]; }
                    else { const uint32_t gg = r0 + jj; a = sd[gg] - ps0; b = gg + 1u < e ? sd[gg + 1] - ps0 : S; l2 = (pq[gg] >> 4) + gg; s2 = skip_of(gg); }
                    const uint32_t avail = b - a;
                    if (avail) {
                        const uint32_t take = avail < need - filled ? avail : need - filled;
                        acc |= (unsigned long long)(loose_codes(lpk, l2, s2) & (take >= 16u ? 0xFFFFFFFFu : (1u << (2u * take)) - 1u)) << (2u * filled);
                        if (jj < nx ? s_hn[jj] : rn[r0 + jj]) nacc |= (loose_nbits(lnb, l2, s2) & ((1u << take) - 1u)) << filled;
                        filled += take;
                    }
                    jj++;
                }
                ok[k] = (uint32_t)acc; on[k] = (uint16_t)nacc; nres[u] = nacc;
            }
            // the N of these dwords into the coder's segment counters, the chunk's N map and total.  A wave's dwords are neighbours - nearly always one segment, one bit of
            // the map: ONE atomic of each kind per wave, by the first lane that has an N (files with runs of N - the configs[4] shape - had every such dword send three
            // atomics to the same few words); lanes in another segment / at another bit send their own.
#pragma unroll
            for (int u = 0; u < SP_U; u++) {
                const uint32_t na = q[u].need ? nres[u] : 0u;
                const unsigned long long hn = __ballot(na != 0u);
                if (!hn) continue;                                              // wave-uniform
                const uint32_t B = 16u * q[u].k, sg = B / PC_SEG_POS, mb = (B >> 12) >> nshift, n = (uint32_t)__popc(na);
                const int lead = __ffsll((long long)hn) - 1;
                const uint32_t sg0 = (uint32_t)__shfl((int)sg, lead), mb0 = (uint32_t)__shfl((int)mb, lead);
                const bool mine = na != 0u && sg == sg0; const int last = na ? (int)(B + 31u - (uint32_t)__clz((int)na)) : -1;
                const uint32_t sum = wave_sum<uint32_t>(mine ? n : 0u); const int mx = wave_max<int>(mine ? last : -1);
                if (lane_id() == lead) { atomicAdd(&segm[nsi + sg0], sum); atomicMax(&segc[nsi + sg0], mx); atomicOr(&nm[mb0 >> 5], 1u << (mb0 & 31u)); }
                if (na != 0u && !mine) { atomicAdd(&segm[nsi + sg], n); atomicMax(&segc[nsi + sg], last); }
                if (na != 0u && mb != mb0) atomicOr(&nm[mb >> 5], 1u << (mb & 31u));
                nsum += n;
            }
        }
        __syncthreads();                                                    // (the LDS tables are rewritten by the next step)
    }
    nsum = wave_sum(nsum);
    if (lane_id() == 0 && nsum) atomicAdd(&ncount[c], nsum);
}
// general path: the byte-wise k_gather left the stored bases as bytes in scat (and counted their N); the same tight streams from those
__global__ void __launch_bounds__(256) k_packbytes(const U4* __restrict__ pv, const uint32_t* __restrict__ first, const uint64_t* __restrict__ sbase,
        const uint8_t* __restrict__ scat,
                                                   uint32_t* __restrict__ spk, uint16_t* __restrict__ snm) {
    const uint32_t c = blockIdx.y, f = first[c], e = first[c + 1];
    const uint32_t S = pv[e].d - pv[f].d, ndw = (S + 15u) >> 4;              // (byte-wise path: the prefix runs over the whole batch)
    const uint4* const src = (const uint4*)(scat + sbase[c]);               // (chunk bases are 64-byte aligned and padded)
    uint32_t* const ok = spk + (size_t)(sbase[c] >> 4); uint16_t* const on = snm + (size_t)(sbase[c] >> 4);
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < ndw; k += gridDim.x * blockDim.x) {
        const uint4 v = src[k]; const uint32_t w[4] = { v.x, v.y, v.z, v.w };
        uint32_t code = 0, nbits = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { uint32_t c4, n4, b4; pack4_codes(w[i], c4, n4, b4); code |= c4 << (8 * i); nbits |= n4 << (4 * i); }
        const uint32_t nv = S - 16u * k;
        if (nv < 16u) { code &= (1u << (2u * nv)) - 1u; nbits &= (1u << nv) - 1u; }
        ok[k] = code; on[k] = (uint16_t)nbits;
    }
}

// scratch capacity of every stream of a chunk: a value with k matches in len positions codes to at most
// k + len/128 + 3*len/16384 bytes (one byte per token, +1 for each gap > 128, +3 for each gap > 16384).
// which: 1 = the quality-value and exception streams (arena `scratch`, chunk total -> ctotal), 2 = the N-position stream (its own arena: it is
// planned later, when the sequence packer has counted the N; chunk total -> ctotal_n), 3 = both
__global__ void k_stream_plan(ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, uint64_t* __restrict__ ctotal, uint64_t* __restrict__ ctotal_n, uint32_t n_chunks,
        const uint32_t* __restrict__ segm, uint32_t n_seg, int which) {
    // one wave per chunk: lane j plans slot j (slots 64 / 65 by lanes 0 / 1 afterwards); offsets by a wave scan
    const uint32_t c = blockIdx.x; const int l = lane_id();
    if (c >= n_chunks) return;
    const uint32_t f = C.first[c], e = C.first[c + 1];
    const uint32_t len = R.pq[e] - R.pq[f];
    const uint32_t nn = D->n_normal < NPOS_SLOT ? D->n_normal : NPOS_SLOT;
    const bool bycol = (D->flags & H_QUAL_BY_COL) && !(D->flags & H_DONT_QUAL);
    const size_t k = (size_t)c * MAX_STREAMS;
    if (which & 1) {
        // occurrences of a stream's value in the chunk = its per-segment match counts (k_gather), summed
        auto occ = [&](uint32_t j) -> uint32_t { uint32_t t = 0; const uint32_t* p = segm + ((size_t)c * MAX_STREAMS + j) * n_seg;
                for (uint32_t s_ = 0; s_ < n_seg; s_++) t += p[s_]; return t; };
        const uint32_t ex = bycol ? occ(EXC_SLOT) : 0u;
        const uint32_t pad = PC_SEG_PAD * pc_n_seg(len);
        uint32_t cap = (bycol && (uint32_t)l < nn) ? occ((uint32_t)l) + len / 128 + 3 * (len / 16384) + 16 + pad : 0u;
        const uint32_t al = (cap + 15u) & ~15u;
        const uint32_t incl = wave_incl_sum(al);
        C.scap[k + l] = cap; C.soff[k + l] = incl - al; C.ssize[k + l] = 0;
        const uint32_t run64 = wave_last(incl);
        if (l == 0) { const uint32_t cape = bycol ? 5 * ex + 16 + pad : 0u, ale = (cape + 15u) & ~15u;
                      C.scap[k + EXC_SLOT] = cape; C.soff[k + EXC_SLOT] = run64; C.ssize[k + EXC_SLOT] = 0; ctotal[c] = (uint64_t)run64 + ale; }
    }
    if ((which & 2) && l == 0) {
        const uint32_t slen = C.ptot[c].d, pads = PC_SEG_PAD * pc_n_seg(slen);
        const uint32_t capn = (D->flags & H_N_POS) ? C.ncount[c] + slen / 128 + 3 * (slen / 16384) + 16 + pads : 0u;
        C.scap[k + NPOS_SLOT] = capn; C.soff[k + NPOS_SLOT] = 0; C.ssize[k + NPOS_SLOT] = 0; ctotal_n[c] = (uint64_t)((capn + 15u) & ~15u);
    }
}
