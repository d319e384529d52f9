// enc/gather_bytes.h - byte-wise gather (reads that do not fit a tile) + the counters both gathers share
// Part of rfq_encode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
// =============================================================== gather (RfqCodec::encodeChunk pass 2, src/rfqcodec.cpp:371-407)
// qcat = full-length qualities in chunk order (R2 reversed when interleaved); scat = stored bases (R2 reverse-complemented and
// overlap-trimmed when interleaved).  Also builds the chunk's quality histogram and N count.
//
// Byte-granular global accesses cost ~30-40 cycles per wave instruction on gfx950 (measured: the byte-copy version of this kernel
// ran at 0.9 TB/s), so a workgroup stages the CONTIGUOUS text of up to 128 consecutive reads in LDS with aligned 16 B/lane loads,
// does all byte shuffling (line extraction, reversal, complement, trimming) from LDS, and writes the two output tiles — also
// contiguous — with aligned 16 B/lane stores.
#define GT_READS 32
#define GT_CAP 13312u             // staged text of a tile (32 x 357-byte records are 11.4 KB)
#define GT_OCAP 5632u             // output tile, qualities and stored bases each (LDS: 13.4 + 2 x 5.7 + counters 4 + tables 1.2 = 30 KB, five blocks per CU)
// Where the N bases of a chunk are, at the granularity of the position coder's 4096-base steps (256 bits per chunk; chunks of more
// than 256 steps fold 2^shift steps into a bit): the N-position coder skips the steps - nearly all of them - that hold no N.
#define PC_SEG_STEPS 8u          // position-coder segment = 8 steps of 4096 positions
#define PC_SEG_POS (PC_SEG_STEPS * 4096u)
#define PC_SEG_PAD 40u           // per-segment slack reserved in a stream's scratch (see pc_seg_cap)
// Bytes reserved for ONE segment of a stream inside the stream's scratch area (16-byte aligned).  MATCH: every token but a gap
// token is one byte per match; gaps > 128 (> 16384) positions cost one (three) more and at most seglen/128 + 1 (seglen/16384 + 1)
// of them end inside the segment; + the `cur > 1` token.  EXCEPT: five bytes per record.  The sum over a stream's segments stays
// below the stream capacity of k_stream_plan (which adds PC_SEG_PAD per segment to the whole-stream bound).
__device__ __forceinline__ uint32_t pc_seg_cap(bool except, uint32_t cnt, uint32_t seglen) {
    const uint32_t c = except ? 5u * cnt + 24u : cnt + seglen / 128u + 3u * (seglen / 16384u) + 24u;
    return (c + 15u) & ~15u;
}
__device__ __forceinline__ uint32_t pc_n_seg(uint32_t len) { return ((len + 4095u) / 4096u + PC_SEG_STEPS - 1u) / PC_SEG_STEPS; }
#define NMAP_WORDS 8u
__device__ __forceinline__ uint32_t nmap_shift(uint32_t n_bases) { const uint32_t steps = (n_bases + 4095u) / 4096u; uint32_t sh = 0;
        while ((steps >> sh) > 32u * NMAP_WORDS) sh++; return sh; }
__device__ __forceinline__ void nmap_mark(uint32_t* m, uint32_t shift, uint32_t pos) { const uint32_t b = (pos >> 12) >> shift; atomicOr(&m[b >> 5], 1u << (b & 31u)); }
__device__ __forceinline__ bool nmap_test(const uint32_t* m, uint32_t shift, uint32_t step) { const uint32_t b = step >> shift; return (m[b >> 5] >> (b & 31u)) & 1u; }
// Quality / N counters of the gather's flush: group() takes 16 packed bytes, operator() one byte.  They keep what the position coder
// needs to start any of its 32768-position segments without a pass of its own: how often each coded value occurs in the segment (the size
// of the segment's slot in the stream's scratch area; summed over the segments, the stream's capacity) and where it occurs last (the
// "previous match" of the segments after it).  A tile holds < 32768 positions, i.e. parts of at most two segments.  Counters live in LDS per
// (replica, segment of the tile, slot): slot = the value's stream, or the last slot for an exception value; the lanes of a wave are spread
// over the replicas (a NovaSeq-binned file has three coded values: without replicas every lane's atomic hits one of six words).
struct QualCount {
    uint32_t* cnt; int* last;            // LDS [nrep][2][nslot]
    const uint8_t* slot;                 // LDS [256]: value -> slot
    uint32_t major; uint32_t seg0, nslot, rep; bool hot_ok;   // hot_ok: the major value has no stream of its own (it has one when it is also the N quality)
    __device__ __forceinline__ void one(uint32_t p, uint32_t q) { const uint32_t i = (rep * 2u + (((p / PC_SEG_POS) - seg0) & 1u)) * nslot + slot[q];
            atomicAdd(&cnt[i], 1u); atomicMax(&last[i], (int)p); }
    __device__ __forceinline__ void operator()(uint32_t p, uint8_t q) { if (!(hot_ok && q == major)) one(p, q); }
    __device__ __forceinline__ void word(uint32_t p, uint32_t w, uint32_t pat) {
        if (hot_ok && w == pat) return;                                     // four major values (72 % of the words of a NovaSeq-binned file)
        uint32_t rest = ~(hot_ok ? eq_mask4(w, pat) : 0u) & 0xFu;
        while (rest) { const int k = __ffs((int)rest) - 1; rest &= rest - 1; one(p + (uint32_t)k, (w >> (8 * k)) & 0xFFu); }
    }
    __device__ __forceinline__ void group(uint32_t p, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
        const uint32_t pat = major * 0x01010101u;
        word(p, w0, pat); word(p + 4u, w1, pat); word(p + 8u, w2, pat); word(p + 12u, w3, pat);
    }
};
struct NCount {                          // p = chunk-relative position of the byte / of the group's first byte (a group never crosses a 4096 boundary)
    uint32_t n; uint32_t* nmap; uint32_t shift; uint32_t* segm; int* segc;   // segm / segc: the N-position stream's per-segment entries of the chunk
    __device__ __forceinline__ void operator()(uint32_t p, uint8_t b) { if (b == 'N') { n++; nmap_mark(nmap, shift, p); atomicAdd(&segm[p / PC_SEG_POS], 1u);
            atomicMax(&segc[p / PC_SEG_POS], (int)p); } }
    __device__ __forceinline__ void group(uint32_t p, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
        // bit 3 is set in 'N' and in none of A / C / G / T: a group without it holds no N (anything else with the bit takes the exact test)
        if (!((w0 | w1 | w2 | w3) & 0x08080808u)) return;
        const uint32_t pat = (uint32_t)'N' * 0x01010101u;
        const uint32_t mk = eq_mask4(w0, pat) | (eq_mask4(w1, pat) << 4) | (eq_mask4(w2, pat) << 8) | (eq_mask4(w3, pat) << 12);
        if (mk) { const uint32_t k = (uint32_t)__popc(mk); n += k; nmap_mark(nmap, shift, p); atomicAdd(&segm[p / PC_SEG_POS], k);
                atomicMax(&segc[p / PC_SEG_POS], (int)(p + 31u - (uint32_t)__clz((int)mk))); }
    }
};
// the tile's counters (summed over the replicas) -> the coder's per-(stream, segment) tables; the counters are left zeroed.  One thread per
// (counter, replica) - nrep * 2 * nslot <= 256 of them, a counter's replicas in neighbouring lanes - and a butterfly over the replicas: the
// serial walk over 16 replicas by eight threads was a chain of 32 dependent LDS round trips at the end of every tile
__device__ __forceinline__ void qual_flush(uint32_t* cnt, int* last, uint32_t nrep, uint32_t nslot, uint32_t seg0, uint32_t c, uint32_t nn, uint32_t* __restrict__ segm,
        int* __restrict__ segc, uint32_t n_seg) {
    const uint32_t nitem = 2u * nslot;
    for (uint32_t t = threadIdx.x; (t & ~63u) < nrep * nitem; t += blockDim.x) {   // (wave-uniform bound: a wave none of whose lanes has a counter is done)
        const uint32_t r = t & (nrep - 1u), i = t / nrep;                 // nrep is a power of two <= 16
        uint32_t n = 0; int lp = -1;
        if (i < nitem) { const uint32_t k = r * nitem + i; n = cnt[k]; lp = last[k]; cnt[k] = 0; last[k] = -1; }
        for (uint32_t d = 1; d < nrep; d <<= 1) { n += (uint32_t)__shfl_xor((int)n, (int)d); const int o = __shfl_xor(lp, (int)d); if (o > lp) lp = o; }
        if (r == 0 && i < nitem && n) {
            const uint32_t sl = i % nslot, seg = seg0 + i / nslot, j = sl < nn ? sl : (uint32_t)EXC_SLOT;
            if (seg < n_seg) { const size_t si = ((size_t)c * MAX_STREAMS + j) * n_seg + seg; atomicAdd(&segm[si], n); if (j != EXC_SLOT) atomicMax(&segc[si], lp); }
        }
    }
}

// One piece of the gather tile: 16-byte groups [g0, g1) of the piece's ceil(n / 16), copied from the staged text to an LDS output tile.  Both
// sides are byte-granular ds_read_b128 / ds_write_b128; the last group of a piece >= 16 bytes is moved back to end exactly at n, a piece
// < 16 bytes is stored as 8 + 4 + 2 + 1 (k_dec_emit's emit_copy, the other way round).  rev: the piece is emitted back to front (an
// interleaved chunk's mate, src/read.cpp:77-115) and, SEQ, complemented.
struct __attribute__((packed, aligned(1))) GLdsW8 { uint32_t a, b; };
struct __attribute__((packed, aligned(1))) GLdsW2 { uint16_t a; };
template <bool SEQ> __device__ __forceinline__ void gather_copy(uint8_t* o, const uint8_t* text, uint32_t src, uint32_t n, uint32_t g0, uint32_t g1, bool rev) {
    for (uint32_t g = g0; g < g1; g++) {
        uint32_t p0 = 16u * g; const bool small = n < 16u;
        if (p0 + 16u > n && !small) p0 = n - 16u;
        uint32_t w[4];
        lds_get16(text, rev ? src + n - p0 - 16u : src + p0, w);
        if (rev) {
            const uint32_t x0 = bswap32(w[3]), x1 = bswap32(w[2]), x2 = bswap32(w[1]), x3 = bswap32(w[0]); w[0] = x0; w[1] = x1; w[2] = x2; w[3] = x3;
            if (SEQ) { w[0] = comp4(w[0]); w[1] = comp4(w[1]); w[2] = comp4(w[2]); w[3] = comp4(w[3]); }
        }
        uint8_t* q = o + p0;
        if (!small) { LdsU16 v; v.a = w[0]; v.b = w[1]; v.c = w[2]; v.d = w[3]; *(LdsU16*)q = v; }
        else {
            if (n & 8u) { GLdsW8 v; v.a = w[0]; v.b = w[1]; *(GLdsW8*)q = v; q += 8; w[0] = w[2]; w[1] = w[3]; }
            if (n & 4u) { LdsU4 v; v.a = w[0]; *(LdsU4*)q = v; q += 4; w[0] = w[1]; }
            if (n & 2u) { GLdsW2 v; v.a = (uint16_t)w[0]; *(GLdsW2*)q = v; q += 2; w[0] >>= 16; }
            if (n & 1u) *q = (uint8_t)w[0];
        }
    }
}
// LDS tile -> global [gbeg, gend) (positions relative to gbase, which is 64-byte aligned; the tile sits at LDS offset gbeg & 15): aligned
// 16-byte stores; every byte goes through the counter - count.group for an aligned group, count(pos, byte) for the edge bytes
template <class Count> __device__ __forceinline__ void flush_count(const uint4* lds4, uint8_t* gbase, uint32_t gbeg, uint32_t gend, Count& count) {
    if (gend <= gbeg) return;
    const uint8_t* lds = (const uint8_t*)lds4; const uint32_t a0 = gbeg & ~15u;
    const uint32_t first_full = (gbeg + 15u) & ~15u, last_full = gend & ~15u;
    if (first_full < last_full) { const uint32_t ng = (last_full - first_full) / 16u, g0 = (first_full - a0) / 16u;
        for (uint32_t i = threadIdx.x; i < ng; i += blockDim.x) { const uint4 v = lds4[g0 + i]; *(uint4*)(gbase + first_full + 16u * i) = v;
                count.group(first_full + 16u * i, v.x, v.y, v.z, v.w); } }
    const uint32_t he = first_full < gend ? first_full : gend;
    for (uint32_t x = gbeg + threadIdx.x; x < he; x += blockDim.x) { const uint8_t b = lds[x - a0]; gbase[x] = b; count(x, b); }
    if (last_full >= first_full) for (uint32_t x = last_full + threadIdx.x; x < gend; x += blockDim.x) { const uint8_t b = lds[x - a0]; gbase[x] = b; count(x, b); }
}
__global__ void k_gather(Text T, ReadTab R, ChunkTab C, const int8_t* __restrict__ ovb, const DevHeader* __restrict__ D,
                         uint8_t* __restrict__ qcat, uint8_t* __restrict__ scat, uint32_t* __restrict__ segm, int* __restrict__ segc, uint32_t n_seg) {
    __shared__ uint4 s_text4[GT_CAP / 16 + 6]; __shared__ uint4 s_qo4[GT_OCAP / 16 + 2], s_so4[GT_OCAP / 16 + 2];
    __shared__ uint32_t s_qsrc[GT_READS], s_ssrc[GT_READS], s_len[GT_READS], s_skip[GT_READS], s_keep[GT_READS], s_qdst[GT_READS + 1], s_sdst[GT_READS + 1];
    __shared__ uint8_t s_rc[GT_READS]; __shared__ uint32_t s_nx[GT_READS];
    __shared__ uint32_t sh[512]; __shared__ int sh_last[512]; __shared__ uint8_t s_slot[256]; __shared__ uint32_t s_n, s_cnt;
    uint8_t* s_text = (uint8_t*)(s_text4 + 1);                            // 16 bytes of slack in front: reversed 16-byte fetches may start before a line
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 512; i += blockDim.x) { sh[i] = 0; sh_last[i] = -1; }
    const uint32_t nn_s = D->n_normal < NPOS_SLOT ? D->n_normal : NPOS_SLOT, nslot = nn_s + 1u;        // stream slots + one for the exception values
    uint32_t nrep = 1; while (nrep < 16u && 4u * nrep * nslot <= 512u) nrep *= 2u;                     // replicas that fit the 512 counters
    for (uint32_t i = tid; i < 256; i += blockDim.x) { const uint32_t j = D->stream_of[i]; s_slot[i] = (uint8_t)(j < nn_s ? j : nn_s); }
    if (tid == 0) s_n = 0;
    const uint32_t c = blockIdx.y, f = C.first[c], e = C.first[c + 1];
    const bool il = C.il[c] != 0; const bool enc = il && (D->flags & H_PE_OVERLAP); const int shift = D->overlap_shift;
    uint8_t* qd = qcat + C.qbase[c]; uint8_t* sd = scat + C.sbase[c];
    const uint32_t pq0 = R.pq[f], ps0 = R.pv[f].d;
    const bool two = T.paired == 1; const uint32_t upr = T.upr;
    uint32_t per = ((e - f) + gridDim.x - 1) / gridDim.x; per = (per + 1u) & ~1u;       // whole pairs per workgroup
    const uint32_t gs = f + blockIdx.x * per; const uint32_t ge = gs + per < e ? gs + per : e;
    QualCount qc; qc.cnt = sh; qc.last = sh_last; qc.slot = s_slot; qc.major = D->major & 0xFFu; qc.seg0 = 0; qc.nslot = nslot; qc.rep = tid & (nrep - 1u);
            qc.hot_ok = D->stream_of[D->major & 0xFFu] == 0xFF;
    NCount nc; nc.n = 0; nc.nmap = C.nmap + (size_t)c * NMAP_WORDS; nc.shift = nmap_shift(R.pv[e].d - ps0);
    nc.segm = segm + ((size_t)c * MAX_STREAMS + NPOS_SLOT) * n_seg; nc.segc = segc + ((size_t)c * MAX_STREAMS + NPOS_SLOT) * n_seg;
    uint32_t cur = gs;
    // what a tile's fit test and tables need of its (up to GT_READS) candidate reads, + the end sentinel: ONE round of global loads.  The
    // round for the NEXT tile is issued while this tile's text is being staged (its start is known as soon as this tile's read count
    // is), so that a tile's chain holds one memory latency - the staging - not two.
    uint32_t a0[2] = { 0, 0 }, na0[2] = { 0, 0 };                          // 16-aligned global begin of each stream's span
    bool fits = false, nfits = false; int m_st = 0, nm_st = 0;
            uint32_t m_p1 = 0, m_p3 = 0, m_nx = 0, m_len = 0, m_qdst = 0, m_sdst = 0, nm_p1 = 0, nm_p3 = 0, nm_nx = 0, nm_len = 0, nm_qdst = 0, nm_sdst = 0;
    int m_ov = 0, nm_ov = 0; bool m_rc = false, nm_rc = false, have = false;
#define GATHER_META_LOAD(from)                                                                                                         \
    { nfits = false; nm_st = 0; nm_p1 = nm_p3 = nm_nx = nm_len = nm_qdst = nm_sdst = 0; nm_ov = 0; nm_rc = false;                      \
      if (two) { na0[0] = T.lo[0][4 * (size_t)((from) >> 1)] & ~15u; na0[1] = T.lo[1][4 * (size_t)((from) >> 1)] & ~15u; }              \
      else { na0[0] = T.lo[0][4 * (size_t)(from)] & ~15u; na0[1] = 0; }                                                                  \
      if (tid <= GT_READS && (from) + tid <= ge) {                                                                                    \
          const uint32_t g = (from) + tid;                                                                                            \
          nm_qdst = R.pq[g] - pq0; nm_sdst = R.pv[g].d - ps0;             /* (valid for the sentinel too) */                           \
          if (tid < GT_READS && g < ge) {                                                                                             \
              uint32_t rr; read_loc(T, g, nm_st, rr);                                                                                 \
              const uint32_t* p = t_lo(T, nm_st) + 4 * (size_t)rr;                                                                    \
              nm_p1 = p[1]; nm_p3 = p[3]; nm_nx = p[4];                    /* p[4]: start of the record after mine, in my stream */      \
              nm_len = R.len[g]; nm_rc = il && ((g - f) & 1u);                                                                        \
              if (nm_rc && enc) nm_ov = (int)ovb[g >> 1] - shift;                                                                     \
              uint32_t need;                                                                                                          \
              if (two) { const uint32_t recs = (tid + 2) >> 1; const size_t r1 = (size_t)((from) >> 1) + recs;                        \
                         need = ((T.lo[0][4 * r1] - na0[0] + 15u) & ~15u) + 16u + (T.lo[1][4 * r1] - na0[1]); }                       \
              else need = nm_nx - na0[0];                                                                                             \
              /* ... and the output tiles: the qualities (never fewer than the stored bases) of everything up to the end of my read / pair */ \
              const uint32_t qend = R.pq[upr == 2 ? (g | 1u) + 1u : g + 1u] - pq0;                                                    \
              nfits = need + 16u <= GT_CAP && (qend - (R.pq[(from)] - pq0)) + 16u <= GT_OCAP;                                         \
          } } }
    while (cur < ge) {                                                   // block-uniform
        if (!have) GATHER_META_LOAD(cur)
        a0[0] = na0[0]; a0[1] = na0[1]; fits = nfits; m_st = nm_st; m_p1 = nm_p1; m_p3 = nm_p3; m_nx = nm_nx; m_len = nm_len; m_qdst = nm_qdst; m_sdst = nm_sdst;
                m_ov = nm_ov; m_rc = nm_rc; have = false;
        if (tid < GT_READS) s_nx[tid] = m_nx;
        // the candidates are the first GT_READS threads: wave 0 counts them (no barrier is needed in front: every wave read the previous tile's count four barriers ago)
        if (tid < 64) {
            const unsigned long long fb = __ballot(fits);
            if (tid == 0) s_cnt = (uint32_t)__popcll(fb);
        }
        __syncthreads();
        const uint32_t cnt = s_cnt;
        if (cnt == 0) {
            // a single read (pair) larger than the tile: byte-wise copy straight from global memory (rare: reads > ~28 kb)
            for (uint32_t g = cur; g < cur + upr && g < ge; g++) {
                const uint32_t len = R.len[g]; const uint8_t* sq = line_ptr(T, g, 1); const uint8_t* ql = line_ptr(T, g, 3);
                const bool rc = il && ((g - f) & 1u); int ov = 0; if (rc && enc) ov = (int)ovb[g >> 1] - shift;
                const uint32_t skip = ov > 0 ? (uint32_t)ov : 0u, keep = len - (uint32_t)(ov < 0 ? -ov : ov);
                uint8_t* qo = qd + (R.pq[g] - pq0); uint8_t* so = sd + (R.pv[g].d - ps0);
                // (its positions may span many coder segments: non-major bytes go straight to the global tables)
                for (uint32_t i = tid; i < len; i += blockDim.x) { const uint8_t q = rc ? ql[len - 1 - i] : ql[i]; qo[i] = q;
                    if (!(qc.hot_ok && q == qc.major)) { const uint32_t pp = R.pq[g] - pq0 + i, sg = pp / PC_SEG_POS;
                           const uint32_t j = D->is_exception[q] ? (uint32_t)EXC_SLOT : (uint32_t)D->stream_of[q];
                           if ((j < NPOS_SLOT || j == EXC_SLOT) && sg < n_seg) { const size_t si = ((size_t)c * MAX_STREAMS + j) * n_seg + sg; atomicAdd(&segm[si], 1u);
                                   if (j != EXC_SLOT) atomicMax(&segc[si], (int)pp); } } }
                for (uint32_t i = tid; i < keep; i += blockDim.x) { const uint32_t j = i + skip; const uint8_t b = rc ? comp_base(sq[len - 1 - j]) : sq[j]; so[i] = b;
                        nc(R.pv[g].d - ps0 + i, b); }
            }
            cur += upr; __syncthreads(); continue;
        }
        // ---- per-read metadata -> LDS (from the registers loaded above)
        uint32_t span_end[2] = { 0, 0 };
        if (two) { span_end[0] = s_nx[cnt - 2]; span_end[1] = s_nx[cnt - 1]; }   // cnt is even for two files: the last pair's records end the spans
        else span_end[0] = s_nx[cnt - 1];
        const uint32_t base1 = two ? (((span_end[0] - a0[0] + 15u) & ~15u) + 16u) : 0u;   // LDS offset of stream 1's span
        if (tid < cnt) {
            const uint32_t lb = m_st ? base1 : 0u;
            s_ssrc[tid] = lb + (m_p1 - a0[m_st]); s_qsrc[tid] = lb + (m_p3 - a0[m_st]); s_len[tid] = m_len; s_rc[tid] = m_rc ? 1 : 0;
            s_skip[tid] = m_ov > 0 ? (uint32_t)m_ov : 0u; s_keep[tid] = m_len - (uint32_t)(m_ov < 0 ? -m_ov : m_ov);
        }
        if (tid <= cnt) { s_qdst[tid] = m_qdst; s_sdst[tid] = m_sdst; }
        // ---- stage the spans: aligned 16-byte loads (the very last group of a stream may not be fully inside the buffer)
        for (int st = 0; st < (two ? 2 : 1); st++) {
            const uint32_t nb = span_end[st] - a0[st]; const uint32_t ng = (nb + 15) / 16; const uint32_t lb = st ? base1 : 0u;
            const uint8_t* src = t_fq(T, st) + a0[st];
            // LDS-DMA (global_load_lds_dwordx4): every lane names its own 16 global bytes, a wave's 64 groups land contiguously at a
            // wave-uniform LDS address - no staging registers, no ds_write pass; everything is in flight until the barrier.  Only
            // the very last group of a stream may reach past the buffer: it is copied byte-wise.
            const uint32_t nfull = (uint64_t)a0[st] + 16ull * ng <= (uint64_t)t_n(T, st) ? ng : ng - 1u;
            uint4* const l4 = s_text4 + 1 + lb / 16;
            for (uint32_t i = tid; i < nfull; i += blockDim.x)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * (size_t)i),
                        (__attribute__((address_space(3))) void*)(l4 + (i - (tid & 63u))), 16, 0, 0);
            if (nfull < ng && tid == 0) for (uint32_t k = 0; k < 16 && a0[st] + 16 * nfull + k < t_n(T,
                    st); k++) s_text[lb + 16 * nfull + k] = src[16 * (size_t)nfull + k];
        }
        if (cur + cnt < ge) { GATHER_META_LOAD(cur + cnt) have = true; }      // the next tile's round of loads, in flight beside the staging
        __syncthreads();
        // ---- compose: one thread = a quarter of one piece (32 reads x {qualities, stored bases} x 4), text tile -> output tiles, LDS to LDS
        const uint32_t q_beg = s_qdst[0], q_end = s_qdst[cnt], s_beg = s_sdst[0], s_end = s_sdst[cnt];
        {
            const uint32_t j = tid % GT_READS, part = tid / GT_READS, quarter = part & 3u; const bool seq = part >= 4u;
            if (j < cnt) {
                const uint32_t len = s_len[j]; const bool rc = s_rc[j] != 0;
                uint32_t n, src; uint8_t* o;
                if (!seq) { n = len; src = s_qsrc[j]; o = (uint8_t*)s_qo4 + (q_beg & 15u) + (s_qdst[j] - q_beg); }
                // stored bases of a mate: RC(R2)[skip, skip + keep) = R2[len - skip - keep, len - skip) back to front
                else { n = s_keep[j]; src = s_ssrc[j] + (rc ? len - s_skip[j] - n : 0u); o = (uint8_t*)s_so4 + (s_beg & 15u) + (s_sdst[j] - s_beg); }
                const uint32_t ng = (n + 15u) >> 4, per4 = (ng + 3u) >> 2, gb = quarter * per4, ge_ = gb + per4 < ng ? gb + per4 : ng;
                if (gb < ge_) { if (seq) gather_copy<true>(o, s_text, src, n, gb, ge_, rc); else gather_copy<false>(o, s_text, src, n, gb, ge_, rc); }
            }
        }
        __syncthreads();
        // ---- flush the two tiles with aligned 16-byte stores; the same pass counts (histogram, per-segment tables, N map)
        qc.seg0 = q_beg / PC_SEG_POS;
        flush_count(s_qo4, qd, q_beg, q_end, qc);
        flush_count(s_so4, sd, s_beg, s_end, nc);
        __syncthreads();
        qual_flush(sh, sh_last, nrep, nslot, qc.seg0, c, nn_s, segm, segc, n_seg);       // (the next tile's counting starts three barriers from here)
        cur += cnt;
    }
    const uint32_t nn = wave_sum(nc.n);
    if (lane_id() == 0 && nn) atomicAdd(&s_n, nn);
    __syncthreads();
    if (tid == 0 && s_n) atomicAdd(&C.ncount[c], s_n);
}
