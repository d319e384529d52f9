// dec/pos_streams.h - token-boundary automaton; position streams of the expanded path (summary, link, emit)
// Part of rfq_decode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
// ---- token-boundary automaton: state = bytes of the current token still to skip (0 = next byte starts a token).
// A byte's transition is s>0 ? s-1 : len(byte)-1; composition of 4-entry tables is associative -> wave scan.
// A table is four bytes, byte s = the state that follows state s: composing two tables is ONE v_perm_b32 (the first table's bytes select bytes of
// the second).  (Two bits per state in one byte - the form the segment summaries are stored in, fn_pack8 - made a composition ~28 instructions, and a
// step's wave scan composes nine times: most of what the position-list passes executed.)
__device__ __forceinline__ uint32_t fn_compose(uint32_t first, uint32_t then) { return __builtin_amdgcn_perm(0u, then, first); }    // (then o first)[s] = then[first[s]]
__device__ __forceinline__ uint32_t fn_apply(uint32_t F, uint32_t s) { return (F >> (8u * s)) & 3u; }
__device__ __forceinline__ uint32_t fn_of_len(uint32_t tok_len) { return 0x02010000u | (tok_len - 1u); }                            // s > 0 ? s - 1 : len - 1
__device__ __forceinline__ uint32_t fn_pack8(uint32_t F) { return (F & 3u) | ((F >> 6) & 0xCu) | ((F >> 12) & 0x30u) | ((F >> 18) & 0xC0u); }
__device__ __forceinline__ uint32_t fn_unpack8(uint32_t b) { return (b & 3u) | ((b & 0xCu) << 6) | ((b & 0x30u) << 12) | ((b & 0xC0u) << 18); }
// inclusive wave scan of transition tables: lane l ends with (table of lane 0) o ... o (its own).  Composition is associative, not commutative:
// the earlier lanes' table always goes first.  DPP row shifts + row broadcasts on the GPU (rfq_common.h), shuffles under the SIMT interpreter.
__device__ __forceinline__ uint32_t wave_scan_compose(uint32_t F) {
#ifdef RFQ_SIMT_EMULATION
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(F, (unsigned)d); if (l >= d) F = fn_compose(t, F); }
#else
#define RFQ_OP_COMPOSE(a, b) fn_compose((b), (a))
    RFQ_DPP_SCAN(F, RFQ_OP_COMPOSE, 0x03020100u)
#undef RFQ_OP_COMPOSE
#endif
    return F;
}
// returns the state BEFORE this lane's byte; carry = state after the wave's last byte
__device__ __forceinline__ uint32_t wave_token_states(uint32_t tok_len, bool valid, uint32_t& carry) {
    uint32_t f = valid ? fn_of_len(tok_len) : 0x03020100u;
    f = wave_scan_compose(f);
    const uint32_t after = fn_apply(f, carry);
    const uint32_t before = wave_shr1(after, carry);
    carry = wave_last(after);
    return before;
}
// decodeSingleQualByCol (src/rfqcodec.cpp:957-1007): one wave per (stream, chunk); writes q at every coded position.
// A step covers 256 stream bytes, 4 consecutive bytes per lane: the lane composes its 4 transition tables locally, ONE wave
// scan gives the automaton state in front of every lane, ONE sum-scan the position in front of it.
__device__ __forceinline__ uint32_t pos_tok_len(uint32_t b0) { return (b0 & 0x80u) == 0 ? 1u : ((b0 & 0x40u) == 0 ? 2u : ((b0 & 0x20u) == 0 ? 1u : 4u)); }
// aligned word at p, or its readable bytes when it straddles `lim` (the end of the image): no read ever leaves the caller's buffer
__device__ __forceinline__ uint32_t ld_word_lim(const uint8_t* p, const uint8_t* lim) {
    if (p + 4 <= lim) return *(const uint32_t*)p;
    uint32_t v = 0; for (int k = 0; k < 4; k++) if (p + k < lim) v |= (uint32_t)p[k] << (8 * k);
    return v;
}
// the 8 stream bytes from i0 on (bytes at or past slen read as 0): three aligned words + funnel shifts
struct PosStep { uint32_t w0, w1, w2; };
__device__ __forceinline__ PosStep pos_fetch(const uint8_t* __restrict__ sp, uint32_t slen, uint32_t i0, const uint8_t* lim) {
    PosStep r; r.w0 = r.w1 = r.w2 = 0;
    if (i0 < slen) {
        const uint8_t* p = (const uint8_t*)((uintptr_t)(sp + i0) & ~(uintptr_t)3);
        r.w0 = ld_word_lim(p, lim); r.w1 = ld_word_lim(p + 4, lim); r.w2 = ld_word_lim(p + 8, lim);
    }
    return r;
}
__device__ __forceinline__ unsigned long long pos_bytes8(const PosStep& r, const uint8_t* __restrict__ sp, uint32_t slen, uint32_t i0) {
    if (i0 >= slen) return 0ull;
    const uint32_t sh = (uint32_t)((uintptr_t)(sp + i0) & 3u) * 8u;
    const uint32_t lo = (uint32_t)((((unsigned long long)r.w1 << 32) | r.w0) >> sh), hi = (uint32_t)((((unsigned long long)r.w2 << 32) | r.w1) >> sh);
    unsigned long long v = ((unsigned long long)hi << 32) | lo;
    const uint32_t nv = slen - i0;                                           // valid bytes from i0
    if (nv < 8) v &= (1ull << (8 * nv)) - 1ull;
    return v;
}
// One step = 256 stream bytes, 4 per lane.  pos_front: the lane's bytes, their transition tables and Fin = the composed table of all
// bytes of the step up to and including the lane's (one wave scan).
struct PosFront { unsigned long long v; uint32_t bt[4], fn[4], Fin; };
#define POS_ID 0x03020100u
__device__ __forceinline__ PosFront pos_front(const PosStep& w, const uint8_t* __restrict__ sp, uint32_t slen, uint32_t i0, int l) {
    PosFront f; f.v = pos_bytes8(w, sp, slen, i0);
    // the four bytes' token lengths - 1 with ONE table look-up: a byte's top three bits select from (0, 0, 0, 0 | 1, 1, 0, 3)  (pos_tok_len: 0xxxxxxx 1, 10xxxxxx 2, 110xxxxx 1, 111xxxxx 4;
    // three compare-and-select chains per byte before)
    const uint32_t lm1 = __builtin_amdgcn_perm(0x03000101u, 0x00000000u, ((uint32_t)f.v >> 5) & 0x07070707u);
#pragma unroll
    for (int k = 0; k < 4; k++) { const bool valid = i0 + (uint32_t)k < slen; f.bt[k] = (uint32_t)(f.v >> (8 * k)) & 0xFFu;
            f.fn[k] = valid ? (0x02010000u | ((lm1 >> (8 * k)) & 0xFFu)) : POS_ID; }
    uint32_t F = fn_compose(fn_compose(fn_compose(f.fn[0], f.fn[1]), f.fn[2]), f.fn[3]);
    (void)l;
    f.Fin = wave_scan_compose(F);
    return f;
}
// positions covered by the tokens that START in the lane's 4 bytes when the automaton enters them in state st
__device__ __forceinline__ int pos_lane_adv(const PosFront& f, uint32_t slen, uint32_t i0, uint32_t st) {
    int a = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t b0 = f.bt[k]; const bool valid = i0 + (uint32_t)k < slen;
        const uint32_t b1 = (uint32_t)(f.v >> (8 * (k + 1))) & 0xFFu, b2 = (uint32_t)(f.v >> (8 * (k + 2))) & 0xFFu, b3 = (uint32_t)(f.v >> (8 * (k + 3))) & 0xFFu;
        if (valid && st == 0) {
            if ((b0 & 0x80u) == 0) a += (int)b0 + 1;
            else if ((b0 & 0x40u) == 0) a += (int)(((b0 & 0x3Fu) << 8) | b1) + 1;
            else if ((b0 & 0x20u) == 0) a += (int)(b0 & 0x1Fu) + 1;
            else a += (int)(((b0 & 0x1Fu) << 24) | (b1 << 16) | (b2 << 8) | b3) + 1;
        }
        if (valid) st = fn_apply(f.fn[k], st);
    }
    return a;
}
// decodeSingleQualByCol over the stream bytes [b0, b1) entered in automaton state `carry` with `last` = last position covered so far
__device__ __forceinline__ void wave_pos_decode(const uint8_t* __restrict__ sp, uint32_t slen, uint32_t b0_, uint32_t b1_, uint32_t carry, int last,
                                                uint8_t q, uint8_t* __restrict__ out, uint32_t out_len, const uint8_t* lim, int* tp) {
    // tp: 256 ints of LDS private to the wave.  A lane decodes four consecutive stream bytes, so in "store my k-th token" the 64
    // lanes hit 64 different cache lines (their tokens are ~4 gaps apart).  The single-position tokens of a step are therefore
    // compacted into tp in stream order and stored TRANSPOSED - lane l takes tokens l, l + 64, ... - so that one store
    // instruction covers neighbouring positions (k_dec_pos_emit 355 -> 310 us).  Staging the segment's bytes in LDS as well, so
    // that no load waits behind the stores, was measured too: no gain.
    const int l = lane_id();                                                 // positions < 2^31 (see the encoder)
    PosStep nxt = pos_fetch(sp, slen, b0_ + 4u * (uint32_t)l, lim);
    for (uint32_t base = b0_; base < b1_; base += 256) {
        const uint32_t i0 = base + 4u * (uint32_t)l;
        const PosStep cur = nxt;
        if (base + 256 < b1_) nxt = pos_fetch(sp, slen, i0 + 256u, lim);    // the next step's words are in flight while this one is decoded
        const PosFront f = pos_front(cur, sp, slen, i0, l);
        const uint32_t after = fn_apply(f.Fin, carry);                  // state after my 4 bytes
        uint32_t st = wave_shr1(after, carry);         // state in front of my first byte
        carry = wave_last(after);
        int adv[4]; uint32_t run[4]; bool start[4]; int lane_adv = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t b0 = f.bt[k]; const bool valid = i0 + (uint32_t)k < slen;
            const uint32_t b1 = (uint32_t)(f.v >> (8 * (k + 1))) & 0xFFu, b2 = (uint32_t)(f.v >> (8 * (k + 2))) & 0xFFu, b3 = (uint32_t)(f.v >> (8 * (k + 3))) & 0xFFu;
            start[k] = valid && st == 0; adv[k] = 0; run[k] = 0;
            if (start[k]) {
                if ((b0 & 0x80u) == 0) adv[k] = (int)b0 + 1;
                else if ((b0 & 0x40u) == 0) adv[k] = (int)(((b0 & 0x3Fu) << 8) | b1) + 1;
                else if ((b0 & 0x20u) == 0) { run[k] = (b0 & 0x1Fu) + 1; adv[k] = (int)run[k]; }
                else adv[k] = (int)(((b0 & 0x1Fu) << 24) | (b1 << 16) | (b2 << 8) | b3) + 1;
            }
            lane_adv += adv[k];
            if (valid) st = fn_apply(f.fn[k], st);
        }
        const int incl = wave_incl_sum(lane_adv);
        int end = last + incl - lane_adv;                                    // last covered position in front of my tokens
        uint32_t singles = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) if (start[k] && !run[k]) singles++;
        const uint32_t sincl = wave_incl_sum(singles); uint32_t so = sincl - singles; const uint32_t stot = wave_last(sincl);
        wave_lds_sync();                                                     // the previous step's tp is no longer read
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (!start[k]) continue;
            end += adv[k];
            if (run[k]) { for (uint32_t t = 0; t < run[k]; t++) { const int p = end - (int)run[k] + 1 + (int)t; if (p >= 0 && (uint32_t)p < out_len) out[p] = q; } }
            else tp[so++] = end;
        }
        wave_lds_sync();
        for (uint32_t j = (uint32_t)l; j < stot; j += 64) { const int p = tp[j]; if (p >= 0 && (uint32_t)p < out_len) out[p] = q; }
        last += wave_last(incl);
    }
}
// A position stream is decoded in SEGMENTS of POS_SEG bytes by independent waves (a serial walk of a 50 KB stream is ~200 dependent
// steps): k_dec_pos_sum reduces every segment to (transition table, positions covered per entry state), k_dec_pos_link walks those
// summaries (one thread per stream), k_dec_pos_emit decodes every segment from its now-known entry state and position.
#define POS_SEG 2048u
struct PosStream { const uint8_t* sp; uint32_t slen; uint8_t q; uint8_t* out; uint32_t out_len; };
// stream jj of chunk c: jj < nn = quality value stream, jj == nn = N positions.  slen = 0 when absent; corrupt length tables are flagged.
__device__ __forceinline__ PosStream pos_stream_of(const uint8_t* __restrict__ img, const DChunk& d, const DevHeader* __restrict__ D, const DReadTab& R, uint32_t c,
                                                   const uint64_t* __restrict__ qbase, const uint64_t* __restrict__ sbase, uint8_t* qdec, uint8_t* sdec, uint32_t jj, DecStatus* st) {
    PosStream s; s.sp = nullptr; s.slen = 0; s.q = 0; s.out = nullptr; s.out_len = 0;
    const uint32_t nn = D->n_normal, hf = D->flags, f = d.rbase; const uint8_t* cp = img + d.off;
    if (jj == nn) {
        if (!(hf & H_N_POS)) return s;
        s.sp = cp + d.o_npos; s.slen = d.npos_size; s.q = (uint8_t)'N'; s.out = sdec + sbase[c]; s.out_len = R.pv[f + d.reads].d - R.pv[f].d;
        return s;
    }
    if (jj > nn || jj >= NPOS_SLOT || (hf & H_DONT_QUAL) || !(hf & H_QUAL_BY_COL)) return s;
    if (4ull * nn > d.qual_size) { if (st && lane_id() == 0) atomicOr(&st->err, (uint32_t)DE_CORRUPT); return s; }
    const uint8_t* qp = cp + d.o_qual; uint64_t off = 4ull * nn;
    for (uint32_t i = 0; i < jj; i++) off += ld_u32(qp + 4 * i);
    const uint32_t sl = ld_u32(qp + 4 * jj);
    if (off + sl > d.qual_size) { if (st && lane_id() == 0) atomicOr(&st->err, (uint32_t)DE_CORRUPT); return s; }
    s.sp = qp + off; s.slen = sl; s.q = D->normal[jj]; s.out = qdec + qbase[c]; s.out_len = R.pq[f + d.reads] - R.pq[f];
    return s;
}
// grid (maxseg, nn + 1, n_chunks), one wave per segment
__global__ void k_dec_pos_sum(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, const DevHeader* __restrict__ D, DReadTab R,
                              const uint64_t* __restrict__ qbase, const uint64_t* __restrict__ sbase, uint8_t* __restrict__ qdec, uint8_t* __restrict__ sdec,
                              uint8_t* __restrict__ segF, int* __restrict__ segA, uint32_t* __restrict__ segN, uint32_t maxseg, DecStatus* st, uint64_t img_bytes,
                              uint32_t jj0, uint32_t nstr) {
    // grid (segments, streams jj0 .. jj0 + gridDim.y - 1, n_chunks); index arrays are [chunk][nstr][maxseg]
    const uint32_t g = blockIdx.x, jj = jj0 + blockIdx.y, c = blockIdx.z; const int l = lane_id(); const uint8_t* lim = img + img_bytes;
    const DChunk d = CH[c];
    const PosStream s = pos_stream_of(img, d, D, R, c, qbase, sbase, qdec, sdec, jj, g == 0 ? st : nullptr);
    if (g == 0 && l == 0) segN[(size_t)c * nstr + jj] = (s.slen + POS_SEG - 1) / POS_SEG;
    const uint32_t b0 = g * POS_SEG; if (b0 >= s.slen) return;
    const uint32_t b1 = b0 + POS_SEG < s.slen ? b0 + POS_SEG : s.slen;
    uint32_t Fcum = POS_ID; int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    PosStep nxt = pos_fetch(s.sp, s.slen, b0 + 4u * (uint32_t)l, lim);
    for (uint32_t base = b0; base < b1; base += 256) {
        const uint32_t i0 = base + 4u * (uint32_t)l;
        const PosStep cur = nxt;
        if (base + 256 < b1) nxt = pos_fetch(s.sp, s.slen, i0 + 256u, lim);
        const PosFront f = pos_front(cur, s.sp, s.slen, i0, l);
        const uint32_t Fex = wave_shr1(f.Fin, POS_ID);
        const uint32_t G = fn_compose(Fcum, Fex);                            // segment entry state -> state in front of my bytes
        a0 += pos_lane_adv(f, s.slen, i0, fn_apply(G, 0u)); a1 += pos_lane_adv(f, s.slen, i0, fn_apply(G, 1u));
        a2 += pos_lane_adv(f, s.slen, i0, fn_apply(G, 2u)); a3 += pos_lane_adv(f, s.slen, i0, fn_apply(G, 3u));
        Fcum = fn_compose(Fcum, wave_last(f.Fin));
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
    if (l == 0) {
        const size_t idx = ((size_t)c * nstr + jj) * maxseg + g;
        segF[idx] = (uint8_t)fn_pack8(Fcum); segA[4 * idx + 0] = a0; segA[4 * idx + 1] = a1; segA[4 * idx + 2] = a2; segA[4 * idx + 3] = a3;
    }
}
// one thread per (chunk, stream): entry state and entry position of every segment
__global__ void k_dec_pos_link(const uint8_t* __restrict__ segF, const int* __restrict__ segA, const uint32_t* __restrict__ segN,
                               uint8_t* __restrict__ segS, int* __restrict__ segP, uint32_t maxseg, uint32_t n_streams) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; if (t >= n_streams) return;
    const uint32_t n = segN[t]; uint32_t st = 0; int last = -1;
    for (uint32_t g = 0; g < n; g++) {
        const size_t idx = (size_t)t * maxseg + g;
        segS[idx] = (uint8_t)st; segP[idx] = last;
        last += segA[4 * idx + st]; st = fn_apply(fn_unpack8(segF[idx]), st);
    }
}
// grid (maxseg, nn + 1, n_chunks): normal quality streams -> qdec, N positions -> sdec
__global__ void k_dec_pos_emit(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, const DevHeader* __restrict__ D, DReadTab R,
                               const uint64_t* __restrict__ qbase, const uint64_t* __restrict__ sbase, uint8_t* __restrict__ qdec, uint8_t* __restrict__ sdec,
                               const uint8_t* __restrict__ segS, const int* __restrict__ segP, uint32_t maxseg, uint64_t img_bytes, uint32_t jj0, uint32_t nstr) {
    // grid (maxseg, streams jj0 .. jj0 + gridDim.y - 1, n_chunks): the quality streams and the N-position stream are launched apart
    const uint32_t g = blockIdx.x, jj = jj0 + blockIdx.y, c = blockIdx.z; const uint8_t* lim = img + img_bytes;
    const DChunk d = CH[c];
    const PosStream s = pos_stream_of(img, d, D, R, c, qbase, sbase, qdec, sdec, jj, nullptr);
    const uint32_t b0 = g * POS_SEG; if (b0 >= s.slen) return;
    const uint32_t b1 = b0 + POS_SEG < s.slen ? b0 + POS_SEG : s.slen;
    const size_t idx = ((size_t)c * nstr + jj) * maxseg + g;
    __shared__ int s_tp[256];                                           // (one wave per block)
    wave_pos_decode(s.sp, s.slen, b0, b1, segS[idx], segP[idx], s.q, s.out, s.out_len, lim, s_tp);
}
