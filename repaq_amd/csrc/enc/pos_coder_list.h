// enc/pos_coder_list.h - position coder for many value streams: work follows the coded positions
// Part of rfq_encode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
// =============================================================== position coder for MANY value streams (list form)
// k_pos_coder tests every position against every value: ~650 instructions per ACTIVE stream and 4096-position step - fine for the three or four streams of
// a NovaSeq-binned file, most of the encode at forty (old Illumina / BGI files: the configs[4] shape).  Here ONE wave codes ALL value streams of a (chunk,
// segment) and the work is proportional to the coded POSITIONS:
//   * per step every quality byte is looked up once in the header's value -> stream table;
//   * what kind of token a position gets is a property of the BYTE sequence, not of the stream: a stream holds one value, so "the previous match of my
//     stream is the position in front of me" is "my byte equals the byte in front of me".  One SWAR pass gives a lane the mask E of its 64 positions
//     that equal their predecessor; streak starts are the zeros of E, a position's distance from its streak start and the matches that follow it are
//     bit scans of E (chained through the lanes for runs that cross them);
//   * the coded positions are bucketed by stream in LDS (count, one prefix over the lanes per stream, scatter), every entry already carrying its kind -
//     gap token / the 0x00 of a streak that starts at position 0 (the `cur > 1` rule) / run token with its length / nothing;
//   * the tokens are written from the list, 64 entries per round whatever streams they belong to: the only thing an entry still needs is its stream's
//     previous match - the entry in front of it.
// Round 3's version of this idea found streak starts by a keyed max-scan over the list and run lengths by a 5-probe search in it: ~400 instructions per
// round of 64 entries, slower than k_pos_coder even at forty streams (4.4 against 3.2 ms).  Slots, capacities and byte counts are k_pos_coder's
// (pc_seg_cap, segb): k_assemble does not know which coder ran.  The exception records stay with k_pos_coder's exception group.
#define PL_LIST 4096u
// (LDS per wave decides how many of these one-wave workgroups a CU holds - the rounds are chains of LDS round trips, other waves are what hides them: the
// list is 16 bits per entry + a byte for its stream, a stream's state one 16-byte record)
struct PlStream { uint32_t outpos, room; unsigned long long out; };          // bytes written so far, the slot's size, where the slot is
struct PlLds {
    // entries, stream after stream: position in the step (12 bits) | code << 12 - 0 no token, 1 gap token (a streak starts), 2 the 0x00
    uint16_t list[PL_LIST];
                                             // of a streak that starts at position 0, 3 + v: run token 0xC0 | v for v <= 11, 15: run token, length to be counted from the
                                             // list
    uint16_t off[NPOS_SLOT + 2];             // where a stream's part of the list starts
    int prev[NPOS_SLOT];                     // the stream's last match so far (-1: none)
    PlStream str[NPOS_SLOT];
    uint8_t tab[256], on[NPOS_SLOT], after;  // after: matches that follow the step's last position (for a run token there)
};
// bit k of the result: byte k of (w, 64 bytes) equals byte k - 1 (byte 0: pb)
__device__ __forceinline__ unsigned long long pl_eq_prev(const uint32_t (&w)[16], uint32_t pb) {
    uint32_t lo = 0, hi = 0, carry = pb << 24;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const uint32_t a = w[i], b = w[i + 1];
        const uint32_t sa = (a << 8) | (carry >> 24), sb_ = (b << 8) | (a >> 24); carry = b;      // the bytes in front
        const uint32_t m = eq_mask8(a ^ sa, b ^ sb_, 0u);                                        // zero bytes of the xors
        if (i < 8) lo |= m << (4 * i); else hi |= m << (4 * (i - 8));
    }
    return ((unsigned long long)hi << 32) | lo;
}
#define PL_WAVES 3                 // waves per SIMD the list coder is compiled for (A/B on the box: tools/ab_macro.sh)
__global__ void __launch_bounds__(64, PL_WAVES) k_pos_coder_list(ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, const uint8_t* __restrict__ qcat,
        uint8_t* __restrict__ scratch, const uint64_t* __restrict__ cbase,
                                                       uint32_t* __restrict__ segb, const int* __restrict__ segc, const uint32_t* __restrict__ segm, uint32_t n_seg, uint32_t n_chunks, DevStatus* st) {
    __shared__ PlLds S;
    // [stream][lane]: matches among the lane's 64 positions, then the lane's next free entry in the stream's part (n_normal x 64 u16: dynamic)
    RFQ_DYN_SHARED(uint16_t, pl_base);
    if (enc_arena_small(st)) return;
    const uint32_t bid = blockIdx.x, xcd = bid & 7u, idx = bid >> 3;       // (a chunk's workgroups on one XCD, as in k_pos_coder)
    const uint32_t c = (idx / n_seg) * 8u + xcd, seg = idx % n_seg;
    if (c >= n_chunks) return;
    const int l = lane_id();
    const uint32_t nn = D->n_normal < NPOS_SLOT ? D->n_normal : NPOS_SLOT, f = C.first[c], e = C.first[c + 1];
    const uint8_t* __restrict__ B = qcat + C.qbase[c]; const uint32_t len = R.pq[e] - R.pq[f];
    const uint32_t nsteps = (len + 4095u) / 4096u, step0 = seg * PC_SEG_STEPS, step1 = step0 + PC_SEG_STEPS < nsteps ? step0 + PC_SEG_STEPS : nsteps;
    if (step0 >= nsteps) return;
    for (uint32_t v = (uint32_t)l; v < 256u; v += 64u) { const uint32_t j = D->stream_of[v]; S.tab[v] = (uint8_t)(j < nn ? j : 0xFFu); }
    {   // a lane per stream: is it there, where it stands, where its bytes go (pc_run's entry state)
        const uint32_t j = (uint32_t)l; bool on = false;
        if (j < nn) {
            const size_t k = (size_t)c * MAX_STREAMS + j, s0i = k * n_seg; const uint32_t cap = C.scap[k];
            on = cap != 0 && segm[s0i + seg] != 0;
            int prev = -1; for (int s_ = (int)seg - 1; s_ >= 0 && prev < 0; s_--) prev = segc[s0i + (uint32_t)s_];
            uint32_t off = 0; for (uint32_t s_ = 0; s_ < seg; s_++) off += pc_seg_cap(false, segm[s0i + s_], PC_SEG_POS);
            const uint32_t own = pc_seg_cap(false, segm[s0i + seg], len - seg * PC_SEG_POS < PC_SEG_POS ? len - seg * PC_SEG_POS : PC_SEG_POS);
            S.prev[j] = prev; PlStream ps; ps.outpos = 0; ps.room = off + own <= cap ? own : 0u;
            ps.out = (unsigned long long)(uintptr_t)(scratch + cbase[c] + C.soff[k] + off); S.str[j] = ps;
        }
        S.on[l] = on ? 1 : 0;
        if (!__any(on)) return;
    }
    wave_lds_sync();
    // (values whose stream has nothing in this segment: not looked at again)
    for (uint32_t v = (uint32_t)l; v < 256u; v += 64u) { const uint32_t j = S.tab[v]; if (j != 0xFFu && !S.on[j]) S.tab[v] = 0xFFu; }
    wave_lds_sync();
    // the byte in front of the segment and how far it is from the start of its streak (the segment may begin inside one)
    uint32_t carry_byte = 0x100u; uint32_t carry_R = 0;                     // (0x100: no byte in front - it equals nothing)
    if (step0 > 0) {
        const uint32_t sb0 = step0 * 4096u; carry_byte = B[sb0 - 1u];
        uint32_t p = sb0 - 1u; while (p > 0 && B[p - 1u] == (uint8_t)carry_byte) p--;          // (every lane walks the same bytes)
        carry_R = sb0 - 1u - p;
    }
    const uint32_t inc = (l & 1) ? 0x10000u : 1u;
    Raw64 ahead = pc_load_raw(B, len, step0 * 4096u + 64u * (uint32_t)l);   // (a step's bytes are requested one step before they are looked at)
    for (uint32_t step = step0; step < step1; step++) {
        const uint32_t sb = step * 4096u, p0 = sb + 64u * (uint32_t)l;
        const uint32_t nv = p0 >= len ? 0u : (len - p0 < 64u ? len - p0 : 64u);
        const Raw64 r = ahead;
        if (step + 1u < step1) ahead = pc_load_raw(B, len, p0 + 4096u);
        const uint32_t w[16] = { r.v[0].x, r.v[0].y, r.v[0].z, r.v[0].w, r.v[1].x, r.v[1].y, r.v[1].z, r.v[1].w, r.v[2].x, r.v[2].y, r.v[2].z, r.v[2].w, r.v[3].x, r.v[3].y, r.v[3].z, r.v[3].w };
        const unsigned long long vmask = nv >= 64u ? ~0ull : ((1ull << nv) - 1ull);
        // ---- the stream of each of my 64 positions: 64 independent table reads, kept packed in registers (0xFF: none); Cm: my coded positions
        uint32_t sw[16]; unsigned long long Cm = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t x = w[i];
            const uint32_t s0_ = S.tab[x & 0xFFu], s1_ = S.tab[(x >> 8) & 0xFFu], s2_ = S.tab[(x >> 16) & 0xFFu], s3_ = S.tab[x >> 24];
            sw[i] = s0_ | (s1_ << 8) | (s2_ << 16) | (s3_ << 24);
            Cm |= (unsigned long long)(((s0_ != 0xFFu) ? 1u : 0u) | ((s1_ != 0xFFu) ? 2u : 0u) | ((s2_ != 0xFFu) ? 4u : 0u) | ((s3_ != 0xFFu) ? 8u : 0u)) << (4 * i);
        }
        Cm &= vmask;
        // ---- E: my positions that equal the position in front; Rin: how far the position in front of my first is from the start of its streak
        const uint32_t lastb = nv ? (w[15] >> 24) : 0x100u;
        const uint32_t pb = wave_shr1(nv == 64u ? lastb : 0x100u, carry_byte);
        unsigned long long E = (pb > 0xFFu) ? (pl_eq_prev(w, 0u) & ~1ull) : pl_eq_prev(w, pb);
        E &= vmask; if (p0 == 0u) E &= ~1ull;
        const bool hz = (~E & vmask) != 0ull || nv < 64u;                   // my positions do not all continue one streak
        const uint32_t ztop = (~E & vmask) ? (uint32_t)(63 - __clzll((long long)(~E & vmask))) : 0u;
        uint32_t tailR = hz ? (nv ? nv - 1u - ztop : 0u) : 0u, Rin = 0;
        for (;;) {                                                          // (one pass unless a streak covers whole lanes)
            Rin = wave_shr1(tailR, carry_R);
            const uint32_t t2 = hz ? tailR : Rin + 64u;
            const bool ch = t2 != tailR; tailR = t2;
            if (!__any(ch)) break;
        }
        // matches that follow my last position (a run token counts up to 31 of them): the head of the next lane's E, for lane 63 the next step's first bytes
        uint32_t ext;
        {
            // my leading positions that continue the streak in front
            const uint32_t hd = (E & 1ull) ? ((~E & vmask) ? (uint32_t)(__ffsll((long long)(~E & vmask)) - 1) : nv) : 0u;
            ext = (uint32_t)__shfl_down((int)hd, 1u);
            if (l == 63) { ext = 0; const uint32_t nb_ = sb + 4096u; if (nv == 64u) { while (ext < 31u && nb_ + ext < len && B[nb_ + ext] == (uint8_t)lastb) ext++;
                    } S.after = (uint8_t)ext; }
            if (nv < 64u) ext = 0;
        }
        // ---- count: my positions per stream (fire-and-forget 32-bit atomics on the u16 pairs of neighbouring lanes)
        for (uint32_t j = 0; j < nn; j++) pl_base[j * 64u + l] = 0;
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 64; k++) { const uint32_t j = (sw[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                if ((Cm >> k) & 1ull) atomicAdd((uint32_t*)&pl_base[j * 64u + (l & ~1)], inc); }
        wave_lds_sync();
        // ---- a prefix over the lanes per stream: where my entries of the stream go
        uint32_t tot = 0;
        for (uint32_t j0 = 0; j0 < nn; j0 += 4u) {                         // (wave-uniform; four streams at a time: their LDS reads are in flight together)
            uint32_t cnt[4], incl[4];
#pragma unroll
            // (a stream that is not `on` has no entries: its counts are zero)
            for (uint32_t u = 0; u < 4u; u++) cnt[u] = j0 + u < nn ? pl_base[(j0 + u) * 64u + l] : 0u;
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) incl[u] = wave_incl_sum<uint32_t>(cnt[u]);
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) if (j0 + u < nn) { if (l == 0) S.off[j0 + u] = (uint16_t)tot; pl_base[(j0 + u) * 64u + l] = (uint16_t)(incl[u] - cnt[u]);
                    tot += wave_last(incl[u]); }
        }
        if (l == 0) { S.off[nn] = (uint16_t)tot; S.off[nn + 1] = (uint16_t)tot; }
        wave_lds_sync();
        // ---- scatter the entries into the list (returning atomics, independent of one another), each with its code.
        // The codes of a lane's 64 positions as four bit masks - no work per position: a streak starts at the zeros of E (code 1); a run token stands at the
        // first continuing position of a run (R == 1: E set, the bit below clear; the lane's head continues the streak in front: where Rin + k is a multiple
        // of 32) - code 3 when the run ends there, 15 (counted from the list, rare) when it goes on.  Lanes with a run of 33 or more in them, and the step that
        // holds position 0 of the chunk (the `cur > 1` rule), take the exact per-position form.
        bool slow = sb == 0u;
        { unsigned long long x = E & (E >> 1); x &= x >> 2; x &= x >> 4; x &= x >> 8; x &= x >> 16; if (x) slow = true; }      // 32 consecutive ones in E
        // my leading positions that continue the streak in front
        const uint32_t hd_ = (E & 1ull) ? ((~E & vmask) ? (uint32_t)(__ffsll((long long)(~E & vmask)) - 1) : nv) : 0u;
        if (hd_ && Rin + hd_ >= 32u) slow = true;
        unsigned long long M3 = 0, M15 = 0;
        {
            unsigned long long T3 = E & ~(E << 1) & ~1ull;                  // R == 1 inside the lane
            if ((E & 1ull) && (Rin & 31u) == 0u) T3 |= 1ull;                // my first position: R = Rin + 1
            const unsigned long long En = (E >> 1) | ((ext ? 1ull : 0ull) << 63);      // the position behind continues
            M3 = T3 & ~En; M15 = T3 & En;
        }
        if (__any(slow)) {                                                  // (rare: wave-uniform) every position by the book
#pragma unroll 1
            for (uint32_t k = 0; k < 64u; k++) {
                if (!((Cm >> k) & 1ull)) continue;
                const uint32_t j = (uint32_t)S.tab[B[p0 + k]];
                uint32_t kind = 1u, val = 0u;                               // 1: the streak starts here - gap token
                if ((E >> k) & 1ull) {
                    const unsigned long long zb = ~E & (k ? ((2ull << k) - 1ull) : 1ull);            // zeros of E at or below k
                    const uint32_t Rk = zb ? k - (uint32_t)(63 - __clzll((long long)zb)) : Rin + k + 1u;     // my distance from the start of my streak
                    const uint32_t p = p0 + k; kind = 0u;
                    int t;
                    // (p == Rk: the streak starts at position 0 of the chunk)
                    if (p == Rk) { if (Rk == 1u) { kind = 2u; t = -1; } else t = (int)Rk - 2; } else t = (int)Rk - 1;
                    if (kind == 0u && t >= 0 && (t & 31) == 0) {
                        const unsigned long long up = (k < 63u) ? (E >> (k + 1u)) : 0ull;             // the positions behind me that continue
                        const uint32_t on_ = (k < 63u) ? ((~up) ? (uint32_t)(__ffsll((long long)~up) - 1) : 64u) : 0u;
                        uint32_t L = 1u + (on_ > 63u - k ? 63u - k : on_);
                        if (k + L == 64u) L += ext;
                        if (L > 32u) L = 32u;
                        kind = 3u; val = L - 1u;
                    }
                }
                const uint32_t code = kind < 3u ? kind : (val <= 11u ? 3u + val : 15u);
                const uint32_t old_ = atomicAdd((uint32_t*)&pl_base[j * 64u + (l & ~1)], inc); const uint32_t at = S.off[j] + ((l & 1) ? old_ >> 16 : old_ & 0xFFFFu);
                S.list[at] = (uint16_t)((64u * (uint32_t)l + k) | (code << 12));
            }
        } else {
            const uint32_t m1lo = (uint32_t)~E, m1hi = (uint32_t)(~E >> 32), m3lo = (uint32_t)M3, m3hi = (uint32_t)(M3 >> 32), m15lo = (uint32_t)M15, m15hi = (uint32_t)(M15 >> 32), clo = (uint32_t)Cm, chi = (uint32_t)(Cm >> 32);
#pragma unroll
            for (int k = 0; k < 64; k++) {
                const uint32_t sh = (uint32_t)k & 31u;
                if (!(((k < 32 ? clo : chi) >> sh) & 1u)) continue;
                const uint32_t j = (sw[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                const uint32_t code = (((k < 32 ? m1lo : m1hi) >> sh) & 1u) + 3u * (((k < 32 ? m3lo : m3hi) >> sh) & 1u) + 15u * (((k < 32 ? m15lo : m15hi) >> sh) & 1u);
                const uint32_t old_ = atomicAdd((uint32_t*)&pl_base[j * 64u + (l & ~1)], inc); const uint32_t at = S.off[j] + ((l & 1) ? old_ >> 16 : old_ & 0xFFFFu);
                S.list[at] = (uint16_t)((64u * (uint32_t)l + (uint32_t)k) | (code << 12));
            }
        }
        wave_lds_sync();
        // ---- tokens: stream after stream, 64 entries of its part of the list per round.  The stream's state - previous match, bytes written - is the same
        // for every lane (scalar registers); an entry's previous match is the entry in front of it (a shift by one lane, the round in front by its last lane).
        for (uint32_t j = 0; j < nn; j++) {                                // (wave-uniform)
            const uint32_t b0 = uni32(S.off[j]), b1 = uni32(S.off[j + 1]);
            if (b0 == b1) continue;
            int prevp = (int)uni32((uint32_t)S.prev[j]);
            const PlStream ps = S.str[j]; uint32_t outpos = uni32(ps.outpos); const uint32_t room = uni32(ps.room);
                    uint8_t* const outp = (uint8_t*)(uintptr_t)uni64(ps.out);
            for (uint32_t r0 = b0; r0 < b1; r0 += 64u) {                    // (wave-uniform)
                const uint32_t i = r0 + (uint32_t)l; const bool valid = i < b1;
                const uint32_t en = valid ? (uint32_t)S.list[i] : 0u, pos = en & 0xFFFu, code = en >> 12;
                const int p = (int)(sb + pos), pp = wave_shr1(p, prevp);
                uint32_t nb = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
                if (valid && code == 1u) {
                    const uint32_t d = (uint32_t)(p - pp), v = d - 1u;
                    if (d <= 128u) { nb = 1; t0 = v; } else if (d <= 16384u) { nb = 2; t0 = (v >> 8) | 0x80u; t1 = v & 0xFFu; } else { nb = 4; t0 = (v >> 24) | 0xE0u;
                            t1 = (v >> 16) & 0xFFu; t2 = (v >> 8) & 0xFFu; t3 = v & 0xFFu; }
                } else if (valid && code >= 2u && code < 15u) { nb = 1; t0 = code == 2u ? 0u : (0xC0u | (code - 3u)); }
                // (rare) a run token that covers 13 .. 32 matches: they are the entries behind me at consecutive positions
                if (__any(valid && code == 15u)) {
                    if (valid && code == 15u) {
                        uint32_t L = 1;
#pragma unroll
                        for (uint32_t stp = 16; stp >= 1; stp >>= 1) { const uint32_t k = L - 1u + stp;
                                if (i + k < b1 && ((uint32_t)S.list[i + k] & 0xFFFu) == pos + k) L += stp; }
                        if (L < 32u && i + L == b1 && pos + L == 4096u) L += S.after;
                        if (L > 32u) L = 32u;
                        nb = 1; t0 = 0xC0u | (L - 1u);
                    }
                }
                const uint32_t incl = wave_incl_sum<uint32_t>(nb), o = outpos + incl - nb;
                if (nb && o + nb <= room) { uint8_t* op = outp + o; op[0] = (uint8_t)t0; if (nb >= 2u) op[1] = (uint8_t)t1; if (nb == 4u) { op[2] = (uint8_t)t2;
                        op[3] = (uint8_t)t3; } }
                outpos += wave_last(incl);
                const uint32_t nlast = b1 - r0 < 64u ? b1 - r0 - 1u : 63u;   // the round's last entry
                prevp = wave_read(p, nlast);
            }
            if (l == 0) { S.str[j].outpos = outpos; S.prev[j] = prevp; }
        }
        carry_byte = wave_last(nv == 64u ? lastb : 0x100u); carry_R = wave_last(tailR);
        wave_lds_sync();
    }
    if ((uint32_t)l < nn && S.on[l]) {
        segb[((size_t)c * MAX_STREAMS + (uint32_t)l) * n_seg + seg] = S.str[l].outpos;
        if (S.str[l].outpos > S.str[l].room) atomicOr(&st->err, (uint32_t)DE_CORRUPT);
    }
}
