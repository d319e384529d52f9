// enc/pos_coder.h - position coder: a wave per (chunk, value streams, segment)
// Part of rfq_encode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
// =============================================================== position coder (encodeSingleQualByCol, src/rfqcodec.cpp:625-710)
// One wave codes one (chunk, stream).  A step covers 4096 positions: lane l owns the 64 positions [4096*step + 64*l, +64)
// as a u64 match mask.  Closed form of the reference's state machine for a maximal streak of matches [a, E]:
//   position a          gap token, d = a - (previous match or -1): 1 byte (d <= 128), 2 bytes (d <= 16384) or 4 bytes
//   position 1 if a==0  gap token 0x00                                   (the `cur > 1` rule, Q3)
//   positions a+b+32k   run token 0xC0 | (min(32, E - i + 1) - 1), b = (a == 0 ? 2 : 1)
// Every token but the streak-start gap is one byte, so byte offsets need only the previous-match distance (a max-scan)
// and the streak start (last zero position + 1, another max-scan); run lengths look at most 31 positions ahead.
enum { PC_MATCH = 0, PC_EXCEPT = 1 };

struct Raw64 { uint4 v[4]; };
__device__ __forceinline__ Raw64 pc_load_raw(const uint8_t* __restrict__ B, uint32_t len, uint32_t p0) {
    Raw64 r; const uint4 z = make_uint4(0, 0, 0, 0);
    if (p0 < len) { const uint4* p = (const uint4*)(B + p0); r.v[0] = p[0]; r.v[1] = p[1]; r.v[2] = p[2]; r.v[3] = p[3]; }
    else { r.v[0] = z; r.v[1] = z; r.v[2] = z; r.v[3] = z; }
    return r;
}
__device__ __forceinline__ uint64_t pc_mask_of(const Raw64& r, uint32_t len, uint32_t p0, int mode, uint32_t q, const DevHeader* __restrict__ D,
        const uint8_t* exc_tab = nullptr) {
    if (p0 >= len) return 0ull;
    uint64_t m = 0;
    const uint4* p = r.v;
    if (mode == PC_MATCH) {
        const uint32_t pat = q * 0x01010101u;
#pragma unroll
        for (int k = 0; k < 4; k++) m |= (uint64_t)eq_mask16c(p[k], pat) << (16 * k);
    } else {
        // exception = neither the major value nor any normal value.  Few values: union of byte-equality masks;
        // many values: 256-bit membership set held in four u64 (no table loads either way).
        const uint32_t nn = D->n_normal;
        if (nn <= 8) {
            uint64_t known = 0; const uint32_t pm = (D->major & 0xFFu) * 0x01010101u;
#pragma unroll
            for (int k = 0; k < 4; k++) known |= (uint64_t)eq_mask16c(p[k], pm) << (16 * k);
            for (uint32_t j = 0; j < nn; j++) { const uint32_t pj = (uint32_t)D->normal[j] * 0x01010101u;
#pragma unroll
                for (int k = 0; k < 4; k++) known |= (uint64_t)eq_mask16c(p[k], pj) << (16 * k); }
            m = ~known;
        } else {
            // many values: exc_tab = the header's 256-entry "is an exception" table in LDS (its 64 words lie in 64 banks: any 64 byte reads are
            // conflict-free); one read per position.  (Rebuilding a 256-bit set from the header in every call - a 256-step scalar loop - and
            // testing it with 64-bit selects and shifts cost ~3600 instructions per step, eight times a value stream's.)
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 w = p[k]; const uint32_t ww[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const uint32_t e = exc_tab[(ww[t >> 2] >> (8 * (t & 3))) & 0xFFu];
                    if (k < 2) lo |= e << (16 * k + t); else hi |= e << (16 * (k - 2) + t);
                }
            }
            m = ((uint64_t)hi << 32) | lo;
        }
    }
    if (len - p0 < 64) m &= (1ull << (len - p0)) - 1ull;
    return m;
}
__device__ __forceinline__ uint32_t ones_from(uint64_t m, int s) {          // length of the run of ones starting at bit s (bit s is set)
    const uint64_t inv = ~(m >> s);                                             // zero-extended: a zero appears within 64 - s bits unless s == 0 and m is all ones
    return inv ? (uint32_t)(__ffsll((long long)inv) - 1) : 64u;
}
// Token generator for one lane's 64-position word: calls sink.put(byte) for every token byte, in stream order.
//   m        match mask of the word, p0 its first position
//   prev_in  last match position before the word (-1: none), zero_in  last non-match position before it (-1: none)
//   after    matches continuing right after the word (leading ones of the next word, <= 64)
struct PackSink {                       // counts, and keeps the first 8 bytes in a register (most words code to <= 8 bytes)
    uint64_t pk = 0; uint32_t n = 0;
    __device__ __forceinline__ void put(uint32_t b) { if (n < 8) pk |= (uint64_t)(b & 0xFFu) << (8 * n); n++; }
};
struct StoreSink {
    uint8_t* p;
    __device__ __forceinline__ void put(uint32_t b) { *p++ = (uint8_t)b; }
};
template <class Sink> __device__ __forceinline__ void pc_gen_tokens(uint64_t m, uint32_t p0, int prev_in, int zero_in, uint32_t after, Sink& sink) {
    uint64_t mm = m; int prev = prev_in;
    while (mm) {
        const int s = __ffsll((long long)mm) - 1; const uint32_t run = ones_from(mm, s); const int e = s + (int)run - 1;
        const int abs_s = (int)p0 + s, abs_e = (int)p0 + e;
        const int a = s > 0 ? abs_s : zero_in + 1;                          // start of the streak this run belongs to
        const uint32_t aft = (e == 63) ? after : 0u;
        if (a == abs_s) {                                                    // streak starts here: gap token
            const int d = abs_s - prev; const uint32_t v = (uint32_t)(d - 1);
            if (d <= 128) sink.put(v);
            else if (d <= 16384) { sink.put((v >> 8) | 0x80u); sink.put(v); }
            else { sink.put((v >> 24) | 0xE0u); sink.put(v >> 16); sink.put(v >> 8); sink.put(v); }
            if (a == 0 && run >= 2) sink.put(0);                             // position 1 of a streak starting at 0 (`cur > 1`, Q3)
        }
        const int b0 = a + (a == 0 ? 2 : 1);
        int i = b0; if (i < abs_s) i += ((abs_s - i + 31) / 32) * 32;
        for (; i <= abs_e; i += 32) { int rem = abs_e - i + 1 + (int)aft; if (rem > 32) rem = 32; sink.put(0xC0u | (uint32_t)(rem - 1)); }
        prev = abs_e;
        mm = (run >= 64u - (uint32_t)s) ? 0ull : (mm & ~(((1ull << run) - 1ull) << s));
    }
}
// The first 16 token bytes of a word in four registers (a word of a stream that takes up to a quarter of the positions codes to that): bytes are shifted
// in from the top, finish() moves them down to byte 0.  (PackSink kept 8: on a NovaSeq-binned file one lane in thirty overflowed it, so nearly every wave
// generated its tokens a second time, straight to memory.)
__device__ __forceinline__ uint32_t pc_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> sh); }   // sh < 32
struct PackSink16 {
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, n = 0;
    __device__ __forceinline__ void put(uint32_t b) { w0 = pc_alignbit(w1, w0, 8); w1 = pc_alignbit(w2, w1, 8); w2 = pc_alignbit(w3, w2, 8); w3 = (w3 >> 8) | (b << 24);
            n++; }
    __device__ __forceinline__ void finish() {                              // (n <= 16)
        const uint32_t k = 16u - n;
        if (k & 8u) { w0 = w2; w1 = w3; w2 = 0; w3 = 0; }
        if (k & 4u) { w0 = w1; w1 = w2; w2 = w3; w3 = 0; }
        const uint32_t sh = 8u * (k & 3u);
        if (k >= 16u) { w0 = w1 = w2 = w3 = 0; }
        else { w0 = pc_alignbit(w1, w0, sh); w1 = pc_alignbit(w2, w1, sh); w2 = pc_alignbit(w3, w2, sh); w3 >>= sh; }
    }
};
struct __attribute__((packed, aligned(1))) GPc8 { uint32_t a, b; };
struct __attribute__((packed, aligned(1))) GPc4 { uint32_t a; };
struct __attribute__((packed, aligned(1))) GPc2 { uint16_t a; };
// n <= 16 finished bytes to p: at most five stores of 8, 8, 4, 2, 1 bytes (a byte loop ran as long as the wave's longest lane)
__device__ __forceinline__ void pc_store16(uint8_t* p, const PackSink16& k) {
    uint32_t a0 = k.w0, a1 = k.w1, a2 = k.w2, a3 = k.w3; const uint32_t n = k.n;
    if (n & 16u) { GPc8 v; v.a = a0; v.b = a1; *(GPc8*)p = v; v.a = a2; v.b = a3; *(GPc8*)(p + 8) = v; return; }
    if (n & 8u) { GPc8 v; v.a = a0; v.b = a1; *(GPc8*)p = v; p += 8; a0 = a2; a1 = a3; }
    if (n & 4u) { GPc4 v; v.a = a0; *(GPc4*)p = v; p += 4; a0 = a1; }
    if (n & 2u) { GPc2 v; v.a = (uint16_t)a0; *(GPc2*)p = v; p += 2; a0 >>= 16; }
    if (n & 1u) *p = (uint8_t)a0;
}
// May pc_gen_fast code this word?  Not when it holds position 0 of the stream or continues a streak that started there (the `cur > 1` rule), or holds a run
// of 32 or more.
__device__ __forceinline__ bool pc_word_is_plain(uint64_t m, uint32_t p0, int zero_in) {
    if (p0 == 0u && (m & 1ull)) return false;
    uint64_t x = m & (m >> 1); x &= x >> 2; x &= x >> 4; x &= x >> 8; x &= x >> 16;
    if (x) return false;
    if ((m & 1ull) && zero_in < 0) return false;                             // (the continuation of a streak that starts at position 0)
    return true;
}
// pc_gen_tokens for such a word: a streak is a gap token and, from two positions on, ONE run token
template <class Sink> __device__ __forceinline__ void pc_gen_fast(uint64_t m, uint32_t p0, int prev_in, int zero_in, uint32_t after, Sink& sink) {
    uint64_t mm = m; int prev = prev_in;
    if ((m & 1ull) && zero_in + 1 != (int)p0) {
        // the word starts inside a streak (begun at zero_in + 1): it owes the run token that starts in its part, if one does (they start every 32 positions
        // behind the streak's second position; lead < 32: at most one)
        const uint32_t lead = (uint32_t)(__ffsll((long long)~m) - 1), b0 = (uint32_t)zero_in + 2u, i = b0 + (((p0 - b0) + 31u) & ~31u);
        if (i < p0 + lead) sink.put(0xC0u | (p0 + lead - i - 1u));
        prev = (int)(p0 + lead) - 1; mm = (m >> lead) << lead;
    }
    while (mm) {
        // (run < 32)
        const uint32_t s = (uint32_t)(__ffsll((long long)mm) - 1); const uint64_t t = mm >> s; const uint32_t run = (uint32_t)(__ffsll((long long)~t) - 1);
        const int abs_s = (int)(p0 + s); const int d = abs_s - prev; const uint32_t v = (uint32_t)(d - 1);
        if (d <= 128) sink.put(v);
        else if (d <= 16384) { sink.put((v >> 8) | 0x80u); sink.put(v & 0xFFu); }
        else { sink.put((v >> 24) | 0xE0u); sink.put((v >> 16) & 0xFFu); sink.put((v >> 8) & 0xFFu); sink.put(v & 0xFFu); }
        if (run >= 2u) { uint32_t rem = run - 1u + (s + run == 64u ? after : 0u); if (rem > 32u) rem = 32u; sink.put(0xC0u | (rem - 1u)); }
        prev = abs_s + (int)run - 1;
        mm = s + run >= 64u ? 0ull : (t >> run) << (s + run);
    }
}
// A (chunk, stream) is cut into segments of PC_SEG_STEPS steps (32768 positions) coded by independent waves: a wave's steps are
// a dependent chain at memory latency, so the kernel's run time is that of its longest chain (256 steps with one wave per stream;
// 32-step segments measured 1.15 ms, 8-step segments 0.93 ms, 4-step segments 0.96 ms).  What a segment needs to start:
//   * the last match before it        k_gather left every segment's last match in segc: the nearest earlier segment that has one
//   * the last non-match before it    a short look-back over the bytes in front of the segment (almost always the byte right there)
//   * where its bytes go              its own slot of the stream's scratch area, sized from the match counts k_gather left in segm
// so ONE launch codes everything (the summary pass that used to read the qualities a first time is gone); k_assemble joins the slots.
// One wave codes up to PC_G streams of the SAME buffer over the same segment: the 4096 raw bytes of a step are loaded once and turned
// into one match mask per stream.
#define PC_G 2                    // (4 when the mask coder read quality bytes: one load of a step for four streams; on match planes two waves of two streams each are 0.04 ms faster than one of four, and half the code)
struct PcStream {
    bool on; int mode; uint32_t q;          // PC_MATCH value q, or PC_EXCEPT
    uint64_t m_cur, m_next;                 // masks of the current and the next step (lane's 64 positions)
    int prev_carry, zero_carry;             // last match / last non-match before the current step
    uint32_t outpos; uint8_t* out; uint32_t room;
};
// one stream, one step of 4096 positions: the tokens of the lanes' words to the stream's slot, carries updated.  B: the bytes an exception record quotes (MODE PC_EXCEPT)
template <int MODE> __device__ __forceinline__ void pc_stream_step(PcStream& s, const uint8_t* __restrict__ B, uint32_t step, uint32_t p0, int l,
        unsigned long long below) {
    const uint64_t m = s.m_cur;
    const unsigned long long has1 = __ballot(m != 0);
    if (!has1) { s.zero_carry = (int)(step * 4096u + 4095u); return; }   // nothing to code in these 4096 positions
    const unsigned long long has0 = __ballot(~m != 0);
    // last match / last non-match before my word: the nearest earlier lane that has one (ballot + one permute), else the carry
    const int mylast = m ? (int)p0 + 63 - __clzll((long long)m) : -1;
    const int myzero = (~m) ? (int)p0 + 63 - __clzll((long long)~m) : -1;
    const unsigned long long b1 = has1 & below, b0m = has0 & below;
    const int src1 = b1 ? 63 - __clzll((long long)b1) : 0, src0 = b0m ? 63 - __clzll((long long)b0m) : 0;
    const int got1 = __shfl(mylast, src1), got0 = __shfl(myzero, src0);
    const int prev_in = b1 ? got1 : s.prev_carry, zero_in = b0m ? got0 : s.zero_carry;
    // matches continuing right after my word (for run lengths): leading ones of the next lane's word (the next step's first word
    // for lane 63 - also when that step belongs to the next segment)
    const uint32_t lead = (m == ~0ull) ? 64u : (uint32_t)(__ffsll((long long)~m) - 1);
    const uint32_t lead_n = (s.m_next == ~0ull) ? 64u : (uint32_t)(__ffsll((long long)~s.m_next) - 1);
    uint32_t after = __shfl_down(lead, 1u); const uint32_t after63 = __shfl(lead_n, 0);
    if (l == 63) after = after63;
    uint32_t bytes; PackSink16 ps;
    if (MODE == PC_EXCEPT) bytes = 5u * (uint32_t)__popcll(m);
    else {
        if (pc_word_is_plain(m, p0, zero_in)) pc_gen_fast(m, p0, prev_in, zero_in, after, ps); else pc_gen_tokens(m, p0, prev_in, zero_in, after, ps);
        bytes = ps.n; if (bytes <= 16u) ps.finish();
    }
    const uint32_t incl = wave_incl_sum(bytes);
    uint32_t o = s.outpos + incl - bytes;
    const uint32_t tot = wave_last(incl);
    if (s.outpos + tot <= s.room) {
        uint8_t* out = s.out;
        if (MODE == PC_EXCEPT) {
            uint64_t mm = m;
            while (mm) { const int b = __ffsll((long long)mm) - 1; mm &= mm - 1; out[o] = B[p0 + (uint32_t)b]; st_u32(out + o + 1, p0 + (uint32_t)b); o += 5; }
        } else if (bytes <= 16u) pc_store16(out + o, ps);
        else { StoreSink ss; ss.p = out + o; pc_gen_tokens(m, p0, prev_in, zero_in, after, ss); }   // dense word: regenerate straight to memory
    }
    s.outpos += tot;
    // carries: the last lane that has a match / a non-match in this step
    const int pl = __shfl(mylast, 63 - __clzll((long long)has1));
    if (pl > s.prev_carry) s.prev_carry = pl;
    if (has0) { const int zl = __shfl(myzero, 63 - __clzll((long long)has0)); if (zl > s.zero_carry) s.zero_carry = zl; }
}
// B must be 64-byte aligned and readable up to the next multiple of 64 past len.  Codes steps [step0, step1) of every active stream
// with its entry state; S[t].outpos ends as the segment's byte count (wave-uniform).  The bytes go to S[t].out[0..).
template <int MODE, int G, bool BITS = false> __device__ __forceinline__ void wave_pos_encode_group(const uint8_t* __restrict__ B, uint32_t len,
        const DevHeader* __restrict__ D,
                                                                           PcStream (&S)[G], uint32_t step0, uint32_t step1, const uint32_t* __restrict__ nmap, uint32_t nshift, const uint8_t* exc_tab) {
    const int l = lane_id();
    const unsigned long long below = l ? (~0ull >> (64 - l)) : 0ull;       // lanes before mine
    // software pipeline, two steps deep: raw bytes of step+2 are in flight while step is coded; a step's raw bytes become masks
    // only one step after they were requested, so the wave never waits on the load it has just issued
    const uint32_t q0 = step0 * 4096u + 64u * (uint32_t)l;                 // positions fit int32: a stream of one batch is < 4 GiB of text, i.e. < 2^31 bases
    // (N positions: a step whose bit in the chunk's N map is clear holds no match - its 4096 bytes are not even loaded)
    // BITS: B is not a byte per position but the match mask itself, one bit per position (the N-position stream reads k_seqpack's N mask):
    // a lane's 64 positions are one u64 (kept in v[0].x / .y)
    auto load = [&](uint32_t step_, uint32_t p_) -> Raw64 { if (nmap && !nmap_test(nmap, nshift, step_)) { Raw64 z;
            z.v[0] = z.v[1] = z.v[2] = z.v[3] = make_uint4(0, 0, 0, 0); return z; }
                                                           if (BITS) { Raw64 r; r.v[0] = r.v[1] = r.v[2] = r.v[3] = make_uint4(0, 0, 0, 0);
                                                                   if (p_ < len) { const uint2 w = ((const uint2*)B)[p_ >> 6]; r.v[0].x = w.x; r.v[0].y = w.y; } return r;
                                                                   }
                                                           return pc_load_raw(B, len, p_); };
    auto mask_of = [&](const Raw64& r_, uint32_t p_, uint32_t q_) -> uint64_t {
        if (!BITS) return pc_mask_of(r_, len, p_, MODE, q_, D, exc_tab);
        if (p_ >= len) return 0ull;
        uint64_t m_ = ((uint64_t)r_.v[0].y << 32) | r_.v[0].x; if (len - p_ < 64) m_ &= (1ull << (len - p_)) - 1ull; return m_; };
    const uint32_t nst = (len + 4095u) / 4096u;
    auto loadc = [&](uint32_t step_, uint32_t p_) -> Raw64 { return load(step_ < nst ? step_ : nst - 1u, step_ < nst ? p_ : len); };
    Raw64 raw_n = loadc(step0 + 1, q0 + 4096u);
    Raw64 r0 = loadc(step0, q0);
    if (MODE == PC_MATCH && step0 > 0) {
        // the last non-match in front of the segment: walk back step by step until every stream has met one (the first step back does it
        // unless a stream's value fills 4096 positions in a row)
        bool need[G]; bool any = false;
#pragma unroll
        for (int t = 0; t < G; t++) { need[t] = S[t].on; any = any || need[t]; }
        for (uint32_t sb = step0; any && sb > 0; ) {                        // wave-uniform
            sb--; const uint32_t pb = sb * 4096u + 64u * (uint32_t)l;
            const Raw64 rb = load(sb, pb);
            any = false;
#pragma unroll
            for (int t = 0; t < G; t++) {
                if (!need[t]) continue;
                const uint64_t z = ~mask_of(rb, pb, S[t].q);
                const unsigned long long h0 = __ballot(z != 0);
                if (h0) { const int v = z ? (int)pb + 63 - __clzll((long long)z) : -1; S[t].zero_carry = __shfl(v, 63 - __clzll((long long)h0)); need[t] = false; }
                else any = true;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < G; t++) if (S[t].on) { S[t].m_cur = mask_of(r0, q0, S[t].q); S[t].m_next = mask_of(raw_n, q0 + 4096u, S[t].q); S[t].outpos = 0; }
    raw_n = loadc(step0 + 2, q0 + 8192u);
    for (uint32_t step = step0; step < step1; step++) {
        const uint32_t p0 = step * 4096u + 64u * (uint32_t)l;
#pragma unroll
        for (int t = 0; t < G; t++) if (S[t].on) pc_stream_step<MODE>(S[t], B, step, p0, l, below);      // (wave-uniform)
#pragma unroll
        for (int t = 0; t < G; t++) if (S[t].on) { S[t].m_cur = S[t].m_next; S[t].m_next = mask_of(raw_n, p0 + 8192u, S[t].q); }
        raw_n = loadc(step + 3, p0 + 12288u);
    }
}
// 1-D grid of ceil(n_chunks / 8) * 8 * (n_qgroups + 2) * n_seg workgroups, one wave each.  Group g < n_qgroups holds the quality-value
// streams 4g .. 4g+3, group n_qgroups the exception stream, group n_qgroups + 1 the N-position stream (it reads the base buffer).
// si = (c * MAX_STREAMS + j) * n_seg + seg; segm[si] = matches in the segment, segc[si] = its last match (k_gather), segb[si] = bytes
// written here.  Streams whose value does not occur in the chunk (histogram) are skipped outright.
template <int MODE, int G, bool BITS = false> __device__ __forceinline__ void pc_run(const ReadTab& R, const ChunkTab& C, const DevHeader* __restrict__ D,
        const uint8_t* __restrict__ B, uint32_t len,
                            uint8_t* __restrict__ scratch, const uint64_t* __restrict__ cbase, uint32_t* __restrict__ segb, const int* __restrict__ segc, const uint32_t* __restrict__ segm, uint32_t n_seg,
                            uint32_t c, uint32_t seg, uint32_t j0, uint32_t jend, const uint32_t* __restrict__ nmap, DevStatus* st, const uint8_t* exc_tab = nullptr) {
    const uint32_t nshift = nmap ? nmap_shift(len) : 0u;
    const uint32_t nsteps = (len + 4095u) / 4096u, step0 = seg * PC_SEG_STEPS;
    const uint32_t step1 = step0 + PC_SEG_STEPS < nsteps ? step0 + PC_SEG_STEPS : nsteps;
    if (step0 >= nsteps) return;
    PcStream S[G]; size_t kk[G]; bool any = false;
#pragma unroll
    for (int t = 0; t < G; t++) {
        const uint32_t j = j0 + (uint32_t)t;
        S[t].on = false; kk[t] = 0;
        if (j >= jend) continue;
        const size_t k = (size_t)c * MAX_STREAMS + j; kk[t] = k;
        const uint32_t cap = C.scap[k];
        if (cap == 0) continue;                                            // stream not present
        const size_t s0i = k * n_seg;
        if (segm[s0i + seg] == 0) continue;                                // nothing to code in this segment: its byte count stays 0
        S[t].on = true; any = true;
        S[t].mode = MODE; S[t].q = j < NPOS_SLOT ? D->normal[j] : (uint32_t)'N';
        S[t].outpos = 0;
        // entry state: the nearest earlier segment that saw a match; the last non-match comes from the look-back (-1 for segment 0)
        int prev = -1;
        for (int s = (int)seg - 1; s >= 0 && prev < 0; s--) prev = segc[s0i + (uint32_t)s];
        S[t].prev_carry = prev; S[t].zero_carry = -1;
        // the segment's slot inside the stream's scratch area: after the slots of the earlier segments (capacities from their match counts)
        uint32_t off = 0;
        for (uint32_t s = 0; s < seg; s++) off += pc_seg_cap(MODE == PC_EXCEPT, segm[s0i + s], PC_SEG_POS);
        const uint32_t own = pc_seg_cap(MODE == PC_EXCEPT, segm[s0i + seg], len - seg * PC_SEG_POS < PC_SEG_POS ? len - seg * PC_SEG_POS : PC_SEG_POS);
        S[t].out = scratch + cbase[c] + C.soff[k] + off; S[t].room = off + own <= cap ? own : 0u;
    }
    if (!any) return;                                                      // wave-uniform
    wave_pos_encode_group<MODE, G, BITS>(B, len, D, S, step0, step1, nmap, nshift, exc_tab);
#pragma unroll
    for (int t = 0; t < G; t++) if (S[t].on && lane_id() == 0) {
        segb[kk[t] * n_seg + seg] = S[t].outpos;
        if (S[t].outpos > S[t].room) atomicOr(&st->err, (uint32_t)DE_CORRUPT);   // (would mean pc_seg_cap is wrong: nothing was written past the slot)
    }
}
// ---- the same over MATCH MASKS: k_gather2<true> leaves, for files with few coded quality values, one bit per position and value instead of the quality
// bytes (planes: value v's u64 of the chunk's positions [64 k, 64 k + 64) at bits[v][k]; the exception plane behind them).  A lane's 64 positions are one
// 8-byte load per stream and step - the byte form loads 64 bytes and compares them with every value (~120 instructions per stream and step) - and the
// gather writes 0.375 - 0.5 B per base instead of 1.
template <int MODE, int G> __device__ __forceinline__ void wave_pos_encode_planes(const uint8_t* __restrict__ qbytes, uint32_t len, PcStream (&S)[G],
        const unsigned long long* const (&bits)[G], uint32_t step0, uint32_t step1) {
    const int l = lane_id();
    const unsigned long long below = l ? (~0ull >> (64 - l)) : 0ull;
    const uint32_t nst = (len + 4095u) / 4096u, q0 = step0 * 4096u + 64u * (uint32_t)l;
    auto load = [&](int t, uint32_t step_, uint32_t p_) -> uint64_t {
        if (step_ >= nst || p_ >= len) return 0ull;
        uint64_t m_ = bits[t][p_ >> 6]; if (len - p_ < 64u) m_ &= (1ull << (len - p_)) - 1ull; return m_; };
    uint64_t ahead[G];                                                      // the masks of step + 2: requested two steps before they are coded
#pragma unroll
    for (int t = 0; t < G; t++) { ahead[t] = 0; if (S[t].on) { S[t].m_cur = load(t, step0, q0); S[t].m_next = load(t, step0 + 1u, q0 + 4096u);
            ahead[t] = load(t, step0 + 2u, q0 + 8192u); S[t].outpos = 0; } }
    if (MODE == PC_MATCH && step0 > 0) {
        // the last non-match in front of the segment: back step by step until every stream has met one
        bool need[G]; bool any = false;
#pragma unroll
        for (int t = 0; t < G; t++) { need[t] = S[t].on; any = any || need[t]; }
        for (uint32_t sb = step0; any && sb > 0; ) {                        // wave-uniform
            sb--; const uint32_t pb = sb * 4096u + 64u * (uint32_t)l;
            any = false;
#pragma unroll
            for (int t = 0; t < G; t++) {
                if (!need[t]) continue;
                const uint64_t z = ~load(t, sb, pb);
                const unsigned long long h0 = __ballot(z != 0);
                if (h0) { const int v = z ? (int)pb + 63 - __clzll((long long)z) : -1; S[t].zero_carry = __shfl(v, 63 - __clzll((long long)h0)); need[t] = false; }
                else any = true;
            }
        }
    }
    for (uint32_t step = step0; step < step1; step++) {
        const uint32_t p0 = step * 4096u + 64u * (uint32_t)l;
#pragma unroll
        for (int t = 0; t < G; t++) if (S[t].on) pc_stream_step<MODE>(S[t], qbytes, step, p0, l, below);      // (wave-uniform)
#pragma unroll
        for (int t = 0; t < G; t++) if (S[t].on) { S[t].m_cur = S[t].m_next; S[t].m_next = ahead[t]; ahead[t] = load(t, step + 3u, p0 + 12288u); }
    }
}
// planes: plane v of the batch at planes + v * pstride (u32 words); the chunk's words start at qbase >> 5.  Streams j0 .. jend - 1 of the quality values,
// or (MODE PC_EXCEPT, j0 = EXC_SLOT) the exception records from plane 3 and the bytes k_gather2 kept at the exceptions' positions in qcat.
template <int MODE, int G> __device__ __forceinline__ void pc_run_planes(const ChunkTab& C, const DevHeader* __restrict__ D, const uint32_t* __restrict__ planes,
        uint64_t pstride, const uint8_t* __restrict__ qbytes, uint32_t len,
                            uint8_t* __restrict__ scratch, const uint64_t* __restrict__ cbase, uint32_t* __restrict__ segb, const int* __restrict__ segc, const uint32_t* __restrict__ segm, uint32_t n_seg,
                            uint32_t c, uint32_t seg, uint32_t j0, uint32_t jend, DevStatus* st) {
    const uint32_t nsteps = (len + 4095u) / 4096u, step0 = seg * PC_SEG_STEPS;
    const uint32_t step1 = step0 + PC_SEG_STEPS < nsteps ? step0 + PC_SEG_STEPS : nsteps;
    if (step0 >= nsteps) return;
    PcStream S[G]; size_t kk[G]; const unsigned long long* bits[G]; bool any = false;
    const size_t w0 = (size_t)(C.qbase[c] >> 5);
#pragma unroll
    for (int t = 0; t < G; t++) {
        const uint32_t j = j0 + (uint32_t)t;
        S[t].on = false; kk[t] = 0; bits[t] = nullptr;
        if (j >= jend) continue;
        const size_t k = (size_t)c * MAX_STREAMS + j; kk[t] = k;
        const uint32_t cap = C.scap[k];
        if (cap == 0) continue;                                            // stream not present
        const size_t s0i = k * n_seg;
        if (segm[s0i + seg] == 0) continue;                                // nothing to code in this segment: its byte count stays 0
        S[t].on = true; any = true;
        S[t].mode = MODE; S[t].q = 0; S[t].outpos = 0;
        bits[t] = (const unsigned long long*)(planes + (size_t)(MODE == PC_EXCEPT ? G2_PLANE_EXC : j) * pstride + w0);
        int prev = -1;
        for (int s = (int)seg - 1; s >= 0 && prev < 0; s--) prev = segc[s0i + (uint32_t)s];
        S[t].prev_carry = prev; S[t].zero_carry = -1;
        uint32_t off = 0;
        for (uint32_t s = 0; s < seg; s++) off += pc_seg_cap(MODE == PC_EXCEPT, segm[s0i + s], PC_SEG_POS);
        const uint32_t own = pc_seg_cap(MODE == PC_EXCEPT, segm[s0i + seg], len - seg * PC_SEG_POS < PC_SEG_POS ? len - seg * PC_SEG_POS : PC_SEG_POS);
        S[t].out = scratch + cbase[c] + C.soff[k] + off; S[t].room = off + own <= cap ? own : 0u;
    }
    if (!any) return;                                                      // wave-uniform
    wave_pos_encode_planes<MODE, G>(qbytes, len, S, bits, step0, step1);
#pragma unroll
    for (int t = 0; t < G; t++) if (S[t].on && lane_id() == 0) {
        segb[kk[t] * n_seg + seg] = S[t].outpos;
        if (S[t].outpos > S[t].room) atomicOr(&st->err, (uint32_t)DE_CORRUPT);
    }
    (void)D;
}
// bytes of every stream of a chunk = sum of its segments' byte counts; one wave per chunk.  Also, for k_assemble's copy of every (stream, segment) piece into
// the image: segd[si] = where the piece goes inside the quality payload (behind the length words: the streams in header order, the exception records last;
// N-position stream: inside its own section) and segs[si] = where it lies in the stream's scratch area (the slots of the segments in front of it) - the
// pieces used to find both by walking over the streams and segments in front of them, ~70 loads for each of a chunk's (streams + 1) x segments pieces.
__global__ void k_pos_sizes(ChunkTab C, const DevHeader* __restrict__ D, const uint32_t* __restrict__ segb, const uint32_t* __restrict__ segm, uint32_t n_seg,
        uint32_t* __restrict__ segd, uint32_t* __restrict__ segs) {
    const uint32_t c = blockIdx.x; const int l = lane_id();
    const uint32_t nn = D->n_normal < NPOS_SLOT ? D->n_normal : NPOS_SLOT;
    uint32_t mine = 0;                                                      // lane j < 64: bytes of value stream j
    for (uint32_t j = (uint32_t)l; j < MAX_STREAMS; j += 64) {
        const size_t k = (size_t)c * MAX_STREAMS + j; uint32_t tot = 0, so = 0;
        if (C.scap[k]) for (uint32_t s = 0; s < n_seg; s++) { const size_t si = k * n_seg + s; segd[si] = tot; segs[si] = so; tot += segb[si];
                so += pc_seg_cap(j == EXC_SLOT, segm[si], PC_SEG_POS); }
        C.ssize[k] = tot; if (j < 64u) mine = j < nn ? tot : 0u;
    }
    // the streams' places in the payload: value streams in header order, then the exception records
    const uint32_t incl = wave_incl_sum(mine), base = incl - mine, total = wave_last(incl);
    if ((uint32_t)l < nn) { const size_t k = (size_t)c * MAX_STREAMS + (uint32_t)l; if (C.scap[k]) for (uint32_t s = 0; s < n_seg; s++) segd[k * n_seg + s] += base; }
    // (the lane that wrote them)
    if (l == (int)(EXC_SLOT - 64u)) { const size_t k = (size_t)c * MAX_STREAMS + EXC_SLOT;
            if (C.scap[k]) for (uint32_t s = 0; s < n_seg; s++) segd[k * n_seg + s] += total; }
}
// g0, gn: the groups this launch codes (the quality / exception groups run behind the gather, the N group behind the sequence packer)
// (six waves per SIMD: 80 VGPRs, 22 of the fattest path's spilled - the byte-stream / exception / N instantiations share the kernel -; uncapped the kernel took 106 and ran at four:
// the phase beside the packer 3.27 -> 3.15 ms, five waves 3.21, seven 3.20, eight 3.60: profiles/r06_zze_pos_coder_waves.txt)
__global__ void __launch_bounds__(64, 6) k_pos_coder(ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, const uint8_t* __restrict__ qcat, const uint16_t* __restrict__ snm,
                            uint8_t* __restrict__ scratch, const uint64_t* __restrict__ cbase, uint8_t* __restrict__ scratch_n, const uint64_t* __restrict__ cbase_n,
                            uint32_t* __restrict__ segb, const int* __restrict__ segc, const uint32_t* __restrict__ segm, uint32_t n_seg, uint32_t n_chunks,
                            uint32_t n_qgroups, uint32_t g0, uint32_t gn, DevStatus* st, const uint32_t* __restrict__ planes, uint64_t pstride) {
    // XCD-aware mapping: workgroup b runs on XCD b % 8 (observed dispatch order; a different placement only costs speed).  All
    // (group, segment) workgroups of chunk c are given ids congruent to c mod 8, so a chunk's data stays in ONE private L2.
    if (enc_arena_small(st)) return;                                       // (an arena the host sized in advance is too small: nothing is coded, the batch is repeated)
    const uint32_t b = blockIdx.x, xcd = b & 7u, idx = b >> 3, per_chunk = gn * n_seg;
    // (a chunk's workgroups group by group, not segment by segment: with two groups - quality streams and the usually empty exception stream -
    // alternating, every other workgroup returned at once and the coder ran at half speed: 4.7 instead of 2.6 ms, consecutive ids share a SIMD pattern)
    const uint32_t c = (idx / per_chunk) * 8u + xcd, rest = idx % per_chunk, grp = g0 + rest / n_seg, seg = rest % n_seg;
    if (c >= n_chunks) return;
    const uint32_t nn = D->n_normal < NPOS_SLOT ? D->n_normal : NPOS_SLOT, f = C.first[c], e = C.first[c + 1];   // (> 64 values: raw qualities, no streams)
    if (planes && grp <= n_qgroups) {                                      // (k_gather2<true> ran: match masks, not bytes)
        if (grp < n_qgroups) pc_run_planes<PC_MATCH, PC_G>(C, D, planes, pstride, nullptr, R.pq[e] - R.pq[f], scratch, cbase, segb, segc, segm, n_seg, c, seg, grp * PC_G,
                nn, st);
        else pc_run_planes<PC_EXCEPT, 1>(C, D, planes, pstride, qcat + C.qbase[c], R.pq[e] - R.pq[f], scratch, cbase, segb, segc, segm, n_seg, c, seg, EXC_SLOT,
                EXC_SLOT + 1, st);
    }
    else if (grp < n_qgroups) pc_run<PC_MATCH, PC_G>(R, C, D, qcat + C.qbase[c], R.pq[e] - R.pq[f], scratch, cbase, segb, segc, segm, n_seg, c, seg, grp * PC_G, nn,
            nullptr, st);
    else if (grp == n_qgroups) {
        __shared__ uint8_t s_exc[256];                                      // (a workgroup is one wave)
        for (uint32_t v = (uint32_t)lane_id(); v < 256u; v += 64u) s_exc[v] = D->is_exception[v] ? 1 : 0;
        wave_lds_sync();
        pc_run<PC_EXCEPT, 1>(R, C, D, qcat + C.qbase[c], R.pq[e] - R.pq[f], scratch, cbase, segb, segc, segm, n_seg, c, seg, EXC_SLOT, EXC_SLOT + 1, nullptr, st, s_exc);
    }
    else pc_run<PC_MATCH, 1, true>(R, C, D, (const uint8_t*)(snm + (size_t)(C.sbase[c] >> 4)), C.ptot[c].d, scratch_n, cbase_n, segb, segc, segm, n_seg, c, seg,
            NPOS_SLOT, NPOS_SLOT + 1, C.nmap + (size_t)c * NMAP_WORDS, st);
}
