// dec/pos_lists.h - fused path: position lists + cell index (sum2, link2, off, list); exception records, RLE, prefill, coordinates
// Part of rfq_decode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
// ================================================================== fused path: no expanded qualities / bases in HBM
// (by-column and raw-quality files whose reads and exception lists fit a tile).  The position streams are turned into POSITION LISTS: one
// u32 per coded position, in stream order, all streams of all chunks in one arena (a position belongs to at most one stream, so a list is a
// few percent of the bases).  The emitter prefills a tile's qualities with the major value in LDS, scatters the list entries that fall into
// the tile, unpacks the tile's bases LDS -> LDS from the packed bytes and scatters the N list: no qdec / sdec, no prefill, unpack or
// one-line-per-token scatter kernels.  Three light passes build the lists, one wave per POS2_SEG-byte segment of a stream (256-byte steps):
//   k_dec_pos_sum2   per segment and entry state of the token automaton: exit state, positions advanced, positions emitted
//   k_dec_pos_link2  per stream, a wave scan over those summaries: entry state / entry position / entry list index of every segment
//   k_dec_pos_list   decodes every segment from its now-known entry and writes its positions; records for every POS2_CELL positions the
//                    index of the first list entry at or beyond the cell (the emitter starts there)
// bytes of a stream per wave (segb / 256 steps of 4 bytes per lane): the host takes POS2_SEG_BIG for files whose largest stream is 32 KB or more - a NovaSeq-binned file's three streams of
// 40 - 70 KB per chunk: half the waves, each setting up once for eight steps, dec:streams 1.30 -> 1.22 ms on configs[2] - and POS2_SEG for many short streams (forty quality values: the
// configs[4] shape is 6 % slower with the larger segments; 512 and 4096 lose on both - profiles/r06_ze_pos2seg.txt)
#define POS2_SEG 1024u
#define POS2_SEG_BIG 2048u
#define POS2_CELL 1024u
// A list entry is the LOW 16 BITS of a coded position (round 5; u32 before: 1.17 GB written and read back per 8 GB of text).  The emitter asks the cell index for the
// entries of [tile start, tile start + tile + a cell) - a window far below 65536 positions - so an entry e is position q0 + ((e - q0) & 0xFFFF), q0 = the tile's first position.
typedef uint16_t plist_t;           // (32-bit entries were A/B'd on one box - profiles/r05_ab.txt: 0.15 ms slower in the emitter, twice the list traffic - and are gone)
// the position an entry stands for, seen from a window that starts at position w0 (0xFFFFFFFF: none - `e` is the 32-bit register an entry was loaded into, or its "no entry" preset)
__device__ __forceinline__ uint32_t plist_pos(uint32_t e, uint32_t w0) { return e == 0xFFFFFFFFu ? e : w0 + ((e - w0) & 0xFFFFu); }
struct PosSrc { const uint8_t* sp; uint32_t slen; uint8_t q; };
// stream jj of a chunk: jj < nn = quality value stream, jj == nn = N positions.  slen = 0 when absent; corrupt length tables are flagged.
__device__ __forceinline__ PosSrc pos_src_of(const uint8_t* __restrict__ img, const DChunk& d, const DevHeader* __restrict__ D, uint32_t jj, DecStatus* st) {
    PosSrc s; s.sp = nullptr; s.slen = 0; s.q = 0;
    const uint32_t nn = D->n_normal, hf = D->flags; const uint8_t* cp = img + d.off;
    if (jj == nn) { if (hf & H_N_POS) { s.sp = cp + d.o_npos; s.slen = d.npos_size; s.q = (uint8_t)'N'; } return s; }
    if (jj > nn || jj >= NPOS_SLOT || (hf & H_DONT_QUAL) || !(hf & H_QUAL_BY_COL)) return s;
    if (4ull * nn > d.qual_size) { if (st && lane_id() == 0) atomicOr(&st->err, (uint32_t)DE_CORRUPT); return s; }
    const uint8_t* qp = cp + d.o_qual; uint64_t off = 4ull * nn;
    for (uint32_t i = 0; i < jj; i++) off += ld_u32(qp + 4 * i);
    const uint32_t sl = ld_u32(qp + 4 * jj);
    if (off + sl > d.qual_size) { if (st && lane_id() == 0) atomicOr(&st->err, (uint32_t)DE_CORRUPT); return s; }
    s.sp = qp + off; s.slen = sl; s.q = D->normal[jj];
    return s;
}
// tokens that START in the lane's 4 bytes when the automaton enters them in state st: positions advanced (adv) and positions emitted (cnt:
// one per gap token, the run length per run token)
__device__ __forceinline__ void pos_lane_adv_cnt(const PosFront& f, uint32_t slen, uint32_t i0, uint32_t st, int& adv, int& cnt) {
    adv = 0; cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t b0 = f.bt[k]; const bool valid = i0 + (uint32_t)k < slen;
        const uint32_t b1 = (uint32_t)(f.v >> (8 * (k + 1))) & 0xFFu, b2 = (uint32_t)(f.v >> (8 * (k + 2))) & 0xFFu, b3 = (uint32_t)(f.v >> (8 * (k + 3))) & 0xFFu;
        if (valid && st == 0) {
            if ((b0 & 0x80u) == 0) { adv += (int)b0 + 1; cnt++; }
            else if ((b0 & 0x40u) == 0) { adv += (int)(((b0 & 0x3Fu) << 8) | b1) + 1; cnt++; }
            else if ((b0 & 0x20u) == 0) { adv += (int)(b0 & 0x1Fu) + 1; cnt += (int)(b0 & 0x1Fu) + 1; }
            else { adv += (int)(((b0 & 0x1Fu) << 24) | (b1 << 16) | (b2 << 8) | b3) + 1; cnt++; }
        }
        if (valid) st = fn_apply(f.fn[k], st);
    }
}
__device__ __forceinline__ int sel4(const int (&v)[4], uint32_t t) { return t == 0 ? v[0] : (t == 1 ? v[1] : (t == 2 ? v[2] : v[3])); }
// The same for ALL four entry states at once: the tokens that start at each of the lane's bytes are decoded once, a backward pass chains them
// (a token that starts at byte k is followed by the one at k + its length), and entry state s - s bytes to skip - reads the chain at byte s.
// (pos_lane_adv_cnt four times over was 60 % of the summary kernel's instructions.)
__device__ __forceinline__ void pos_lane_adv_cnt4(const PosFront& f, uint32_t slen, uint32_t i0, int (&adv)[4], int (&cnt)[4]) {
    const uint32_t nv = i0 >= slen ? 0u : (slen - i0 < 4u ? slen - i0 : 4u);   // the lane's valid bytes
    int ca[4], cc[4]; uint32_t tl[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t b0 = f.bt[k];
        const uint32_t b1 = (uint32_t)(f.v >> (8 * (k + 1))) & 0xFFu, b2 = (uint32_t)(f.v >> (8 * (k + 2))) & 0xFFu, b3 = (uint32_t)(f.v >> (8 * (k + 3))) & 0xFFu;
        if ((b0 & 0x80u) == 0) { ca[k] = (int)b0 + 1; cc[k] = 1; tl[k] = 1; }
        else if ((b0 & 0x40u) == 0) { ca[k] = (int)(((b0 & 0x3Fu) << 8) | b1) + 1; cc[k] = 1; tl[k] = 2; }
        else if ((b0 & 0x20u) == 0) { ca[k] = (int)(b0 & 0x1Fu) + 1; cc[k] = ca[k]; tl[k] = 1; }
        else { ca[k] = (int)(((b0 & 0x1Fu) << 24) | (b1 << 16) | (b2 << 8) | b3) + 1; cc[k] = 1; tl[k] = 4; }
    }
#pragma unroll
    for (int k = 2; k >= 0; k--) {                                           // chain: byte k's token, then whatever starts behind it inside the lane
        const uint32_t nx = (uint32_t)k + tl[k];
        if (nx < nv) { const int a_ = nx == 1u ? ca[1] : (nx == 2u ? ca[2] : ca[3]), c_ = nx == 1u ? cc[1] : (nx == 2u ? cc[2] : cc[3]); ca[k] += a_; cc[k] += c_; }
    }
#pragma unroll
    for (int s = 0; s < 4; s++) { const bool on = (uint32_t)s < nv; adv[s] = on ? ca[s] : 0; cnt[s] = on ? cc[s] : 0; }
}
#define POS_ADV_MAX 0x3FFFFFFFull   // advances are summed saturating at this value (two saturated terms still fit an int)
__device__ __forceinline__ int pos_adv_sat(int v) { return v > (int)POS_ADV_MAX ? (int)POS_ADV_MAX : v; }
// grid (ceil(maxseg / 4), streams, n_chunks) x 256 threads: one wave per segment; index arrays are [chunk][nstr][maxseg]; segA[8 * idx + s] =
// positions advanced, segA[8 * idx + 4 + s] = positions emitted for entry state s
__global__ void k_dec_pos_sum2(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, const DevHeader* __restrict__ D,
                               uint8_t* __restrict__ segF, int* __restrict__ segA, uint32_t* __restrict__ segN, uint32_t maxseg, DecStatus* st, uint64_t img_bytes, uint32_t jj0, uint32_t nstr,
                               uint32_t segb) {
    const uint32_t g = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave_id(), jj = jj0 + blockIdx.y, c = blockIdx.z; const int l = lane_id();
            const uint8_t* lim = img + img_bytes;
    const DChunk d = CH[c];
    const PosSrc s = pos_src_of(img, d, D, jj, g == 0 ? st : nullptr);
    if (g == 0 && l == 0) segN[(size_t)c * nstr + jj] = (s.slen + segb - 1) / segb;
    const uint32_t b0 = g * segb; if (b0 >= s.slen) return;
    const uint32_t b1 = b0 + segb < s.slen ? b0 + segb : s.slen;
    uint32_t Fcum = POS_ID; int a[4] = { 0, 0, 0, 0 }, n[4] = { 0, 0, 0, 0 };
    PosStep nxt = pos_fetch(s.sp, s.slen, b0 + 4u * (uint32_t)l, lim);
    for (uint32_t base = b0; base < b1; base += 256u) {                    // (wave-uniform)
        const uint32_t i0 = base + 4u * (uint32_t)l;
        const PosStep w = nxt;
        if (base + 256u < b1) nxt = pos_fetch(s.sp, s.slen, i0 + 256u, lim);
        const PosFront f = pos_front(w, s.sp, s.slen, i0, l);
        const uint32_t Fex = wave_shr1(f.Fin, POS_ID);
        const uint32_t G = fn_compose(Fcum, Fex);                          // segment entry state -> state in front of my bytes
        int la[4], lc[4]; pos_lane_adv_cnt4(f, s.slen, i0, la, lc);         // the lane's tokens for each state in front of its bytes
#pragma unroll
        for (int e = 0; e < 4; e++) { const uint32_t t_ = fn_apply(G, e); a[e] += sel4(la, t_); n[e] += sel4(lc, t_); }
        Fcum = fn_compose(Fcum, wave_last(f.Fin));
    }
    // (a lane's four steps advance less than 4 x 2^30 positions: no wrap as 32 bits; the wave's sum is taken in 64 bits and SATURATED at POS_ADV_MAX - a corrupt
    // stream advances up to 2^29 per token, a wrapped sum went negative, slipped through k_dec_pos_link2's "beyond the chunk" test and aliased 16-bit list entries:
    // ADVICE r5.  Saturated sums stay sums for every position a chunk the fused path takes can hold - rfq_decode.hip sends larger chunks to the expanded path.)
#pragma unroll
    for (int e = 0; e < 4; e++) { const unsigned long long w64 = wave_sum<unsigned long long>((unsigned long long)(uint32_t)a[e]); a[e] = (int)(w64 > POS_ADV_MAX ? POS_ADV_MAX : w64);
            n[e] = wave_sum(n[e]); }
    if (l == 0) { const size_t idx = ((size_t)c * nstr + jj) * maxseg + g; segF[idx] = (uint8_t)fn_pack8(Fcum);
#pragma unroll
                  for (int e = 0; e < 4; e++) { segA[8 * idx + e] = a[e]; segA[8 * idx + 4 + e] = n[e]; } }
}
// one wave per (chunk, stream): entry state / position / list index of every segment by a scan over (transition table, advance and count
// per entry state): x then y is (y.F o x.F, s -> x.a[s] + y.a[x.F[s]]); also the stream's number of list entries
struct PosLink { uint32_t F; int a[4], n[4]; };
__device__ __forceinline__ PosLink poslink_then(const PosLink& x, const PosLink& y) {   // x first, then y
    PosLink r; r.F = fn_compose(x.F, y.F);
#pragma unroll
    for (int s = 0; s < 4; s++) { const uint32_t t = fn_apply(x.F, s); r.a[s] = pos_adv_sat(x.a[s] + sel4(y.a, t)); r.n[s] = x.n[s] + sel4(y.n, t); }
    return r;
}
__device__ __forceinline__ PosLink poslink_shfl_up(const PosLink& v, unsigned dd) {
    PosLink u; u.F = __shfl_up(v.F, dd);
#pragma unroll
    for (int s = 0; s < 4; s++) { u.a[s] = __shfl_up(v.a[s], dd); u.n[s] = __shfl_up(v.n[s], dd); }
    return u;
}
__global__ void k_dec_pos_link2(const uint8_t* __restrict__ segF, const int* __restrict__ segA, const uint32_t* __restrict__ segN, uint8_t* __restrict__ segS,
        int* __restrict__ segP,
                                uint32_t* __restrict__ segK, uint32_t* __restrict__ nent, uint32_t maxseg, uint32_t n_streams,
                                const DChunk* __restrict__ CH, uint32_t nstr, DecStatus* st) {
    const uint32_t t = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave_id(); if (t >= n_streams) return;
    const int l = lane_id(); const uint32_t n = segN[t];
    uint32_t cs = 0; int cp = -1; uint32_t ck = 0;                          // state / last covered position / list entries in front of the block of 64 segments
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t g = base + (uint32_t)l; const size_t idx = (size_t)t * maxseg + g;
        PosLink me; me.F = POS_ID;
#pragma unroll
        for (int s = 0; s < 4; s++) { me.a[s] = 0; me.n[s] = 0; }
        if (g < n) { me.F = fn_unpack8(segF[idx]);
#pragma unroll
                     for (int s = 0; s < 4; s++) { me.a[s] = segA[8 * idx + s]; me.n[s] = segA[8 * idx + 4 + s]; } }
        PosLink inc = me;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const PosLink up = poslink_shfl_up(inc, (unsigned)dd); if (l >= dd) inc = poslink_then(up, inc); }
        PosLink ex = poslink_shfl_up(inc, 1u);
        if (l == 0) { ex.F = POS_ID;
#pragma unroll
                      for (int s = 0; s < 4; s++) { ex.a[s] = 0; ex.n[s] = 0; } }
        if (g < n) { segS[idx] = (uint8_t)(fn_apply(ex.F, cs)); segP[idx] = pos_adv_sat(cp + sel4(ex.a, cs)); segK[idx] = ck + (uint32_t)sel4(ex.n, cs); }
        const uint32_t Fl = wave_last(inc.F); int al[4], nl[4];
#pragma unroll
        for (int s = 0; s < 4; s++) { al[s] = wave_last(inc.a[s]); nl[s] = wave_last(inc.n[s]); }
        cp = pos_adv_sat(cp + sel4(al, cs)); ck += (uint32_t)sel4(nl, cs); cs = fn_apply(Fl, cs);
    }
    if (l == 0) {
        nent[t] = ck;
        // 16-bit list entries (plist_t) are unambiguous while a stream's positions stay within 32 K of the chunk's own extent - they do, unless the image codes positions its
        // length table does not cover (the reference's two-byte lengths of reads > 65535 bases, App. C; corrupt images): such ranges take the expanded path
        const unsigned long long ext = CH[t / nstr].bases;
        if (cp >= 0 && (unsigned long long)cp >= ext + 32768ull) atomicOr(&st->err, (uint32_t)DE_E3_RETRY);
    }
}
// exclusive prefix of the streams' entry counts (one workgroup; n_streams is some thousands) -> where each list starts in the arena; the total
// goes to st->list_need (the host grows the arena and repeats k_dec_pos_list when it did not fit)
__global__ void k_dec_pos_off(const uint32_t* __restrict__ nent, unsigned long long* __restrict__ loff, uint32_t n_streams, DecStatus* st) {
    // every thread a run of consecutive streams (summed, one block scan, re-walked): no barrier per 256 streams
    const uint32_t K = (n_streams + blockDim.x - 1) / blockDim.x, i0 = threadIdx.x * K, i1 = i0 + K < n_streams ? i0 + K : n_streams;
    unsigned long long acc = 0;
    for (uint32_t i = i0; i < i1; i++) acc += nent[i];
    unsigned long long tot; unsigned long long run = block_excl_sum<unsigned long long>(acc, &tot);
    for (uint32_t i = i0; i < i1; i++) { loff[i] = run; run += nent[i]; }
    if (threadIdx.x == 0) st->list_need = tot;
}
// decodeSingleQualByCol (src/rfqcodec.cpp:957-1007) for one segment from its entry (state, last covered position, list index): the positions
// it codes go to plist[loff + k ...] in stream order; cellidx[cell] = index (within the stream's list) of the first entry >= cell * POS2_CELL
__global__ void k_dec_pos_list(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, const DevHeader* __restrict__ D,
                               const uint8_t* __restrict__ segS, const int* __restrict__ segP, const uint32_t* __restrict__ segK, const unsigned long long* __restrict__ loff,
                               plist_t* __restrict__ plist, unsigned long long cap, uint32_t* __restrict__ cellidx, uint32_t maxseg, uint32_t ncell, uint64_t img_bytes, uint32_t jj0, uint32_t nstr, const DecStatus* st,
                               uint32_t segb) {
    if (st->list_need > cap) return;                                      // (uniform) the arena is too small: the host repeats the pass
    const uint32_t g = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave_id(), jj = jj0 + blockIdx.y, c = blockIdx.z; const int l = lane_id();
            const uint8_t* lim = img + img_bytes;
    const DChunk d = CH[c];
    const PosSrc s = pos_src_of(img, d, D, jj, nullptr);
    const uint32_t b0 = g * segb; if (b0 >= s.slen) return;
    const size_t t = (size_t)c * nstr + jj, idx = t * maxseg + g;
    uint32_t carry = segS[idx]; int last = segP[idx]; uint32_t k0 = segK[idx];
    plist_t* const out = plist + loff[t]; uint32_t* const cells = cellidx + t * ncell;
    const uint32_t b1 = b0 + segb < s.slen ? b0 + segb : s.slen;
    PosStep nxt = pos_fetch(s.sp, s.slen, b0 + 4u * (uint32_t)l, lim);
    for (uint32_t base = b0; base < b1; base += 256u) {                    // (wave-uniform) a step = 256 bytes; state, position and list index carry over
        const uint32_t i0 = base + 4u * (uint32_t)l;
        const PosStep w = nxt;
        if (base + 256u < b1) nxt = pos_fetch(s.sp, s.slen, i0 + 256u, lim);
        const PosFront f = pos_front(w, s.sp, s.slen, i0, l);
        const uint32_t Fex = wave_shr1(f.Fin, POS_ID);
        uint32_t st0 = fn_apply(Fex, carry);                          // state in front of my first byte
        int adv[4]; uint32_t run[4]; bool start[4]; int lane_adv = 0, lane_cnt = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t bb = f.bt[k]; const bool valid = i0 + (uint32_t)k < s.slen;
            const uint32_t b1_ = (uint32_t)(f.v >> (8 * (k + 1))) & 0xFFu, b2_ = (uint32_t)(f.v >> (8 * (k + 2))) & 0xFFu, b3_ = (uint32_t)(f.v >> (8 * (k + 3))) & 0xFFu;
            start[k] = valid && st0 == 0; adv[k] = 0; run[k] = 0;
            if (start[k]) {
                if ((bb & 0x80u) == 0) adv[k] = (int)bb + 1;
                else if ((bb & 0x40u) == 0) adv[k] = (int)(((bb & 0x3Fu) << 8) | b1_) + 1;
                else if ((bb & 0x20u) == 0) { run[k] = (bb & 0x1Fu) + 1; adv[k] = (int)run[k]; }
                else adv[k] = (int)(((bb & 0x1Fu) << 24) | (b1_ << 16) | (b2_ << 8) | b3_) + 1;
                lane_cnt += run[k] ? (int)run[k] : 1;
            }
            lane_adv += adv[k];
            if (valid) st0 = fn_apply(f.fn[k], st0);
        }
        const int ia = wave_incl_sum(lane_adv), ic = wave_incl_sum(lane_cnt);
        int end = last + ia - lane_adv; uint32_t k = k0 + (uint32_t)(ic - lane_cnt);   // last covered position / list index in front of my tokens
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (!start[q]) continue;
            const int prev = end; end += adv[q];
            const int lo = run[q] ? end - (int)run[q] + 1 : end;
            int pp = prev;                                                   // the position of list entry k - 1 (-1: none)
            for (int p = lo; p <= end; p++, k++) {
                out[k] = (plist_t)p;
                uint32_t c0 = pp < 0 ? 0u : (uint32_t)pp / POS2_CELL + 1u; const uint32_t c1 = (uint32_t)p / POS2_CELL;
                for (; c0 <= c1 && c0 < ncell; c0++) cells[c0] = k;
                pp = p;
            }
        }
        last += wave_last(ia); k0 += (uint32_t)wave_last(ic); carry = fn_apply((uint32_t)wave_last(f.Fin), carry);
    }
}

// exception records (q, u32 LE position) after the streams (src/rfqcodec.cpp:1034-1043); raw copy when DONT_ENCODE_QUAL (:905-910)
__global__ void k_dec_except(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, const DevHeader* __restrict__ D, DReadTab R,
                             const uint64_t* __restrict__ qbase, uint8_t* __restrict__ qdec) {
    const uint32_t c = blockIdx.y, nn = D->n_normal, hf = D->flags;
    const DChunk d = CH[c]; const uint8_t* qp = img + d.off + d.o_qual; const uint32_t f = d.rbase;
    const uint32_t len = R.pq[f + d.reads] - R.pq[f]; uint8_t* dst = qdec + qbase[c];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, NT = gridDim.x * blockDim.x;
    if (hf & H_DONT_QUAL) { for (uint32_t i = t; i < d.qual_size && i < len; i += NT) dst[i] = qp[i]; return; }
    if (!(hf & H_QUAL_BY_COL) || 4ull * nn > d.qual_size) return;
    uint64_t off = 4ull * nn;
    for (uint32_t i = 0; i < nn; i++) off += ld_u32(qp + 4 * i);
    if (off > d.qual_size) return;
    const uint32_t nrec = (uint32_t)((d.qual_size - off) / 5);
    for (uint32_t i = t; i < nrec; i += NT) { const uint8_t* r = qp + off + 5ull * i; const uint32_t pos = ld_u32(r + 1); if (pos < len) dst[pos] = r[0]; }
}
// decodeQualByRunLenCoding (src/rfqcodec.cpp:919-955): the legacy run-length quality coding (v0.5.1 never writes it; such
// images take the materialising path).  One byte per run: bit 0 clear = the major value, run = (byte >> 1) + 1 (majorQualNumBits is 7,
// src/rfqheader.cpp:255-257); bit 0 set = the value whose "bit" code is byte & mask, run = (byte >> (8 - n)) + 1 with n = normalQualNumBits
// (computeNormalQualBits, :117-128); code -> value is mBit2QualTable (makeQualBitTable, :103-115: entry i of the header's table has code 0, 1,
// 3, 5, ...; codes the table does not list read its zeroed entries).  The reference re-reads the buffer until every quality is out.
// grid (1, n_chunks): the workgroup walks the chunk's bytes 256 at a time, run starts by a block scan.
__global__ void k_dec_rle(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, const DevHeader* __restrict__ D, DReadTab R,
                          const uint64_t* __restrict__ qbase, uint8_t* __restrict__ qdec) {
    __shared__ uint8_t s_b2q[256]; __shared__ uint32_t s_carry;
    const uint32_t c = blockIdx.y; const DChunk d = CH[c]; const uint8_t* qp = img + d.off + d.o_qual; const uint32_t f = d.rbase;
    const uint32_t len = R.pq[f + d.reads] - R.pq[f]; uint8_t* dst = qdec + qbase[c];
    const uint32_t bins = D->bytes[16]; int mx = (int)bins * 2 - 3; if (mx < 1) mx = 1;
    const uint32_t nq = mx >= 64 ? 1u : mx >= 32 ? 2u : mx >= 16 ? 3u : mx >= 8 ? 4u : mx >= 4 ? 5u : mx >= 2 ? 6u : 7u, mask = (1u << (8u - nq)) - 1u;
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_b2q[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bins; i += blockDim.x) s_b2q[(uint8_t)(i ? 2u * i - 1u : 0u)] = D->bytes[17 + i];
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    if (d.qual_size == 0 || len == 0) return;                             // (the reference would spin for ever on an empty buffer: the prefill stays)
    for (uint32_t rounds = 0; ; rounds++) {                                // block-uniform
        for (uint32_t b0 = 0; b0 < d.qual_size; b0 += blockDim.x) {
            const uint32_t i = b0 + threadIdx.x; uint32_t run = 0, q = 0;
            if (i < d.qual_size) { const uint32_t e = qp[i]; if ((e & 1u) == 0) { q = 0; run = (e >> 1) + 1u; } else { q = e & mask; run = (e >> (8u - nq)) + 1u; } }
            uint32_t tot; const uint32_t ex = block_excl_sum<uint32_t>(run, &tot);
            const uint32_t start = s_carry + ex; const uint8_t v = s_b2q[q];
            for (uint32_t p = start; p < start + run && p < len; p++) dst[p] = v;
            __syncthreads();
            if (threadIdx.x == 0) s_carry += tot;
            __syncthreads();
            if (s_carry >= len) return;
        }
    }
}
// quality prefill with the major value (src/rfqcodec.cpp:1089)
__global__ void k_dec_fill(uint8_t* __restrict__ p, uint64_t n, const DevHeader* __restrict__ D) {
    const uint32_t v = D->major & 0xFFu; const uint4 q = make_uint4(v * 0x01010101u, v * 0x01010101u, v * 0x01010101u, v * 0x01010101u);
    uint4* p4 = (uint4*)p; const uint64_t n4 = n / 16;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) p4[i] = q;
    if (blockIdx.x == 0 && threadIdx.x < (n & 15u)) p[n4 * 16 + threadIdx.x] = (uint8_t)v;
}

// decodeCoords (src/rfqcodec.cpp:1332-1389): one wave per (axis, chunk)
__global__ void k_dec_coords(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, const DevHeader* __restrict__ D, uint32_t* __restrict__ xv,
        uint32_t* __restrict__ yv) {
    const uint32_t axis = blockIdx.x, c = blockIdx.y;
    if (!(D->flags & (axis ? H_Y : H_X))) return;
    const DChunk d = CH[c]; const uint8_t* sp = img + d.off + (axis ? d.o_y : d.o_x) + 4; const uint32_t slen = axis ? d.y_size : d.x_size;
    const uint32_t num = (d.flags & C_PE_INTERLEAVED) ? d.reads / 2 : d.reads;
    uint32_t* out = (axis ? yv : xv) + d.rbase;
    const int l = lane_id(); uint32_t carry = 0, cur = 1000u, produced = 0;
    // (a step's bytes - mine and the two behind it, a token's tail - are requested two steps before they are decoded: the steps are one dependent chain, and a load in
    // it costs a round trip to memory per 64 bytes of the stream)
    auto fetch = [&](uint32_t base_) -> uint32_t { const uint32_t i_ = base_ + (uint32_t)l; return i_ < slen ? (uint32_t)sp[i_] : 0u; };
    uint32_t w1 = fetch(0u), w2 = fetch(64u);
    for (uint32_t base = 0; base < slen; base += 64) {
        const uint32_t i = base + (uint32_t)l; const bool valid = i < slen;
        const uint32_t b0 = w1; w1 = w2; w2 = fetch(base + 128u);
        // the two bytes behind mine: the next lanes', for the last two lanes the next step's first (bytes past the stream read as 0)
        const uint32_t n0 = wave_read(w1, 0u), n1 = wave_read(w1, 1u);
        uint32_t b1 = (uint32_t)__shfl_down((int)b0, 1u), b2 = (uint32_t)__shfl_down((int)b0, 2u);
        if (l == 63) { b1 = n0; b2 = n1; } else if (l == 62) b2 = n0;
        const uint32_t tl = (b0 & 0x80u) == 0 ? 2u : ((b0 & 0xE0u) == 0xE0u ? 3u : 1u);
        const uint32_t before = wave_token_states(tl, valid, carry);
        const bool start = valid && before == 0;
        uint32_t cnt = 0, isabs = 0, val = 0;                      // val: absolute value, or the +diff
        if (start) {
            if ((b0 & 0x80u) == 0) { isabs = 1; val = (b0 << 8) | b1; cnt = 1; }
            else if ((b0 & 0x40u) == 0) { val = (b0 & 0x3Fu) + 1; cnt = 1; }
            else if ((b0 & 0x20u) == 0) { val = 0; cnt = (b0 & 0x1Fu) + 1; }
            else { isabs = 1; val = ((b0 & 0x1Fu) << 16) | (b1 << 8) | b2; cnt = 1; }
        }
        // segmented prefix: value after this token = last absolute at or before it + diffs since
        uint32_t v = val, a = isabs;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t tv = __shfl_up(v, (unsigned)dd), ta = __shfl_up(a, (unsigned)dd); if (l >= dd && !a) { v += tv; a = ta; } }
        const uint32_t value = a ? v : cur + v;
        const uint32_t incl = wave_incl_sum(cnt); const uint32_t o = produced + incl - cnt;
        if (start) for (uint32_t k = 0; k < cnt; k++) if (o + k < num) out[o + k] = value;
        produced += wave_last(incl); cur = wave_last(value);
    }
}
