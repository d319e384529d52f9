// dec/chunk_walk.h - chunk descriptors, RfqChunk::read per chunk, the index-less walk (exact, guess and verify), verification
// Part of rfq_decode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
#include "../rfq_common.h"

struct DChunk {                  // one parsed chunk (offsets relative to the chunk start)
    uint64_t off;                // byte offset of the chunk in the image
    uint32_t reads, flags, seq_size, qual_size, npos_size, x_size, y_size;
    uint32_t o_readlens, o_n1lens, o_n2lens, o_stlens, o_lanes, o_tiles, o_x, o_y, o_n1, o_n2, o_st, o_seq, o_qual, o_ov, o_npos, total;
    uint32_t n1_size, n2_size, st_size;
    uint32_t rbase;              // reads in earlier chunks of the range being decoded
    uint32_t rbase_abs;          // reads in earlier chunks of the image (rbase is re-based per range by k_dec_rebase)
    uint64_t bases;              // sum of the chunk's read lengths (64-bit: a corrupt length table must not wrap the 32-bit prefix sums)
    uint32_t max_len, nrec;      // longest read of the chunk; exception records behind its quality streams (0xFFFFFFFF: not looked at)
    uint32_t max_one, pad_;      // its longest single quality stream
};
#define ET_N1CAP 3072u            // staged name1 / name2 / strand pieces of an emitter's tile (k_dec_emit3 hands a range whose pieces are larger over to the expanded path)
#define ET_N2CAP 1024u
#define ET_STCAP 1024u
struct DecStatus {
    uint32_t err, n_chunks, max_reads, overflow;
    uint64_t total_reads, consumed, total_bases, total_stored, text1, text2;
    uint32_t last_flags, pad;
    uint32_t max_stream, max_npos;   // largest quality section / N-position section of any chunk (bound the position streams)
    uint64_t text_slots[2][64];      // partial sums of the text bytes per output stream (k_dec_textlen)
    uint64_t base_slots[16];         // partial sums of the read lengths of all chunks (parse_chunk)
    uint32_t max_len, max_bases;     // longest read / largest chunk (bases, clamped to 2^32 - 1) of the image
    uint32_t max_nrec, max_one;      // most exception records of any chunk / longest single quality stream (by-column quality payloads)
    unsigned long long list_need;    // fused path: entries of all position lists (k_dec_pos_off)
    uint32_t per_read_pieces, piece_avg;  // some chunk stores name1 / name2 / strand per read; the largest average size of a per-read name2 / strand piece over the
                                     // chunks, as a fraction of its tile capacity in 1/256 (k_dec_emit3: the host sizes its tiles by it, a tile that still does not fit
                                     // asks for the expanded path)
    uint32_t piece_n1, pad4;              // the same for name1, in bytes per read (rounded up): k_dec_emit3 has a second instantiation with a large name1 tile
};

// sum of n bytes by one wave (wave-uniform result)
__device__ __forceinline__ uint32_t wave_sum_bytes(const uint8_t* __restrict__ p, uint32_t n) {
    uint32_t acc = 0;
    for (uint32_t i = (uint32_t)lane_id(); i < n; i += 64) acc += p[i];
    return wave_sum(acc);
}
// RfqChunk::read for the chunk at byte k (wave-cooperative: length arrays are summed by the whole wave).
// Returns 0 = ok, 1 = clean end of image (short tail / mReads == 0), 2 = corrupt.
__device__ __forceinline__ int parse_chunk(const uint8_t* __restrict__ img, uint64_t n, uint64_t k, uint32_t hf, uint32_t rlb, DChunk& d) {
    if (n - k < 18) return 1;
    const uint8_t* p = img + k;
    d.off = k;
    d.reads = ld_u32(p + 4); d.flags = ld_u16(p + 8); d.seq_size = ld_u32(p + 10); d.qual_size = ld_u32(p + 14);
    if (d.reads == 0) return 1;
    const uint64_t left = n - k; uint64_t q = 18;
    d.npos_size = 0; if (hf & H_N_POS) { if (left < q + 4) return 2; d.npos_size = ld_u32(p + q); q += 4; }
    const uint32_t s = d.reads, fl = d.flags; const uint32_t h = (fl & C_PE_INTERLEAVED) ? s / 2 : s;
    d.o_readlens = (uint32_t)q; q += (uint64_t)((fl & C_READ_LEN_SAME) ? 1u : s) * rlb;
    if (q > left) return 2;
    {   // sum of the read lengths, 64-bit (the per-read prefix sums that place bases and qualities are 32-bit; the host refuses a batch that would wrap them)
        const uint8_t* lp = p + d.o_readlens; unsigned long long sum = 0;
        auto rl = [&](uint32_t r) -> uint32_t { const uint8_t* x = lp + (size_t)r * rlb; return rlb == 1 ? x[0] : (rlb == 2 ? ld_u16(x) : ld_u32(x)); };
        uint32_t mx = 0;
        if (fl & C_READ_LEN_SAME) { mx = rl(0); sum = (unsigned long long)mx * s; }
        else { for (uint32_t r = (uint32_t)lane_id(); r < s; r += 64) { const uint32_t v = rl(r); sum += v; if (v > mx) mx = v; } sum = wave_sum<unsigned long long>(sum);
                mx = wave_max(mx); }
        d.bases = sum; d.max_len = mx; d.nrec = 0; d.max_one = 0; d.pad_ = 0;
    }
#define RFQ_LENARR(OFF, SIZE, LENFLAG, SAMEFLAG) { \
        const uint32_t m_ = (fl & (LENFLAG)) ? 1u : s; OFF = (uint32_t)q; if (q + m_ > left) return 2; \
        uint32_t sum_ = (fl & (LENFLAG)) ? (uint32_t)p[q] : wave_sum_bytes(p + q, m_); \
        if ((fl & (LENFLAG)) && !(fl & (SAMEFLAG))) { sum_ *= s; } \
        SIZE = sum_; q += m_; }
    RFQ_LENARR(d.o_n1lens, d.n1_size, C_NAME1_LEN_SAME, C_NAME1_SAME)
    d.o_n2lens = (uint32_t)q; d.n2_size = 0;
    if (hf & H_NAME2) RFQ_LENARR(d.o_n2lens, d.n2_size, C_NAME2_LEN_SAME, C_NAME2_SAME)
    RFQ_LENARR(d.o_stlens, d.st_size, C_STRAND_LEN_SAME, C_STRAND_SAME)
#undef RFQ_LENARR
    d.o_lanes = (uint32_t)q; if (hf & H_LANE) q += (fl & C_LANE_SAME) ? 1u : h;
    d.o_tiles = (uint32_t)q; if (hf & H_TILE) q += 2ull * ((fl & C_TILE_SAME) ? 1u : h);
    d.x_size = 0; d.y_size = 0;
    d.o_x = (uint32_t)q; if (hf & H_X) { if (q + 4 > left) return 2; d.x_size = ld_u32(p + q); q += 4ull + d.x_size; }
    if (q > left) return 2;
    d.o_y = (uint32_t)q; if (hf & H_Y) { if (q + 4 > left) return 2; d.y_size = ld_u32(p + q); q += 4ull + d.y_size; }
    d.o_n1 = (uint32_t)q; q += d.n1_size;
    d.o_n2 = (uint32_t)q; if (hf & H_NAME2) q += d.n2_size;
    d.o_st = (uint32_t)q; q += d.st_size;
    d.o_seq = (uint32_t)q; q += d.seq_size;
    d.o_qual = (uint32_t)q; q += d.qual_size;
    d.o_ov = (uint32_t)q; if ((fl & C_PE_INTERLEAVED) && (hf & H_PE_OVERLAP)) q += s / 2;
    d.o_npos = (uint32_t)q; if (hf & H_N_POS) q += d.npos_size;
    if (q > left || q > 0xFFFFFFFFull) return 2;
    d.total = (uint32_t)q;
    return 0;
}
// One wave walks the image chunk by chunk (each chunk's extent depends on its own length arrays; the reader ignores mSize).
__global__ void k_dec_walk(const uint8_t* __restrict__ img, uint64_t n, uint64_t start, const DevHeader* __restrict__ D, DChunk* __restrict__ out, uint32_t cap,
        DecStatus* st, int final) {
    const uint32_t hf = D->flags, rlb = D->read_len_bytes; const int l = lane_id();
    uint64_t k = start; uint32_t c = 0, maxr = 0, maxs = 0, maxn = 0, lastfl = 0, maxl = 0, maxb = 0; uint64_t rb = 0, tb = 0; uint32_t err = 0, ovf = 0;
    if (rlb != 1 && rlb != 2 && rlb != 4) err = DE_CORRUPT;
    while (!err) {
        DChunk d; const int rc = parse_chunk(img, n, k, hf, rlb, d);
        if (rc == 1) break;
        if (rc == 2) { if (final) err = DE_CORRUPT; break; }            // not final: the chunk continues in the caller's next batch
        d.rbase = (uint32_t)rb; d.rbase_abs = (uint32_t)rb;
        if (c < cap) { if (l == 0) out[c] = d; } else ovf = 1;
        if (d.reads > maxr) maxr = d.reads;
        if (d.qual_size > maxs) maxs = d.qual_size;
        if (d.npos_size > maxn) maxn = d.npos_size;
        lastfl = d.flags; rb += d.reads; k += d.total; c++; tb += d.bases;
        if (d.max_len > maxl) maxl = d.max_len;
        { const uint32_t b32 = d.bases > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)d.bases; if (b32 > maxb) maxb = b32; }
        if (rb > 0xFFFFFFF0ull) { err = DE_CORRUPT; break; }
    }
    if (l == 0) { st->max_len = maxl; st->max_bases = maxb; st->base_slots[0] = tb; st->err |= err; st->n_chunks = c; st->max_reads = maxr; st->total_reads = rb;
            st->consumed = k; st->last_flags = lastfl; st->overflow = ovf; st->max_stream = maxs; st->max_npos = maxn; }
}
// The chain from a chunk index the caller supplied (rfq_decode_args.h_chunk_off): the read counts of all chunks are fetched in
// parallel (one workgroup, 256 chunks per round, running read base by a block scan) - no dependent load per chunk.  k_dec_parse
// verifies every extent exactly as it does behind the speculative walk.
__global__ void k_dec_table(const uint8_t* __restrict__ img, uint64_t n, const uint64_t* __restrict__ off, uint32_t nch_, DChunk* __restrict__ out, DecStatus* st) {
    // (nch_ == ~0: the table was made on the device - k_dec_gw_* below - and so was its length; a table that failed there leaves pad set)
    const uint32_t nch = nch_ == 0xFFFFFFFFu ? st->n_chunks : nch_;
    if (nch_ == 0xFFFFFFFFu && (st->pad || st->overflow)) return;
    __shared__ uint32_t s_bad, s_maxr;
    if (threadIdx.x == 0) { s_bad = 0; s_maxr = 0; }
    __syncthreads();
    // every thread a run of consecutive chunks: their read counts summed, one block scan, the run re-walked with its base
    const uint32_t K = (nch + blockDim.x - 1) / blockDim.x, c0 = threadIdx.x * K, c1 = c0 + K < nch ? c0 + K : nch;
    auto reads_of = [&](uint32_t c, bool& bad) -> uint32_t {
        const uint64_t k = off[c], e = off[c + 1];
        if (e > n || k + 18 > e || e - k > 0xFFFFFFFFull) { bad = true; return 0u; }
        const uint32_t r = ld_u32(img + k + 4); if (r == 0) bad = true;
        return r;
    };
    unsigned long long acc = 0; bool anybad = false; uint32_t mx = 0;
    for (uint32_t c = c0; c < c1; c++) { bool bad = false; const uint32_t r = reads_of(c, bad); if (bad) anybad = true; else { acc += r; if (r > mx) mx = r; } }
    unsigned long long tot; unsigned long long run = block_excl_sum<unsigned long long>(acc, &tot);
    for (uint32_t c = c0; c < c1; c++) {
        bool bad = false; const uint32_t r = reads_of(c, bad);
        if (!bad) { out[c].off = off[c]; out[c].total = (uint32_t)(off[c + 1] - off[c]); out[c].rbase = (uint32_t)run; out[c].reads = r; run += r; }
    }
    if (mx) atomicMax(&s_maxr, mx);
    if (anybad) atomicOr(&s_bad, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t rb = tot; uint32_t bad = s_bad | (rb > 0xFFFFFFF0ull ? 1u : 0u);
        st->n_chunks = nch; st->max_reads = s_maxr; st->total_reads = rb; st->consumed = off[nch]; st->overflow = 0; st->pad = bad;
        st->last_flags = (!bad && nch) ? ld_u16(img + off[nch - 1] + 8) : 0u;
    }
}
// ---- the chunk starts of an image that comes without an index (a .rfq file has none: RfqChunk::read walks it, src/rfqchunk.cpp:161-228).
// The one-wave chain above is one dependent memory round trip per chunk - 2.0 ms for the 3360 chunks of configs[2].  Guess-and-verify instead:
// the image is cut into up to GW_SEGS segments of ~16 chunks (sized from the first chunk); k_dec_gw_find tests every byte offset of a
// window at each segment's start for "a chunk header whose mSize chain leads to another plausible header" and keeps the lowest; k_dec_gw_walk
// walks each segment from its candidate to the next segment's (the same one-read-per-chunk chain, a wave per segment, ~16 hops); k_dec_gw_stitch
// checks that every walk lands exactly on the next candidate and concatenates the lists into a chunk index, which k_dec_table / k_dec_parse
// then treat like a caller's: every extent is parsed and verified in full.  Anything that does not add up (a foreign writer, chunk sizes that
// differ wildly, a corrupt image) sets pad, and the host falls back to the chain.
#define GW_SEGS 1024u
#define GW_LCAP 256u              // chunk starts a segment's walk may record
struct GwGeo { uint64_t first, seglen, win; uint32_t nseg; };
// true chunk size from mSize: repaq's writers store mSize = true size - Delta(flags) (accounting bug Q1, a pure function of header and chunk flags)
__device__ __forceinline__ long long gw_total(uint32_t ms, uint32_t s, uint32_t fl, uint32_t hf) {
    const uint32_t h = (fl & C_PE_INTERLEAVED) ? s / 2 : s; long long total = (long long)ms;
    if (hf & H_LANE) total += (fl & C_LANE_SAME) ? 1 : (long long)h;
    if (!(hf & H_TILE)) total -= (fl & C_TILE_SAME) ? 2 : 2ll * h;
    if (!(hf & H_NAME2)) total -= (fl & C_NAME2_LEN_SAME) ? 1 : (long long)s;
    return total;
}
// a plausible chunk header at byte o?  (lite: the fields alone; else also that its size leads to the image's end or another plausible header)
__device__ __forceinline__ bool gw_plausible(const uint8_t* __restrict__ img, uint64_t n, uint64_t o, uint32_t hf, bool lite, uint64_t* next) {
    if (n - o < 18) return false;
    const LdsU16 hd = *(const LdsU16*)(img + o);
    const uint32_t ms = hd.a, s = hd.b, fl = hd.c & 0xFFFFu, seqsz = (hd.c >> 16) | (hd.d << 16), qualsz = (hd.d >> 16) | ((uint32_t)ld_u16(img + o + 16) << 16);
    if (s == 0 || s > 0x1000000u || fl >= 0x1000u) return false;
    const long long total = gw_total(ms, s, fl, hf);
    if (total < 18 || (unsigned long long)total > n - o || (unsigned long long)seqsz + qualsz + 18ull > (unsigned long long)total) return false;
    if (next) *next = o + (uint64_t)total;
    if (lite) return true;
    const uint64_t o2 = o + (uint64_t)total;
    if (n - o2 < 18) return true;                                           // the image ends here (or with a tail too short to be a chunk)
    if (ld_u32(img + o2 + 4) == 0) return true;                             // mReads == 0: a clean end
    return gw_plausible(img, n, o2, hf, true, nullptr);
}
// the header at o is plausible in its fields but the chunk runs past the end of the image (the tail of a range that does not end the image)
__device__ __forceinline__ bool gw_cut_by_end(const uint8_t* __restrict__ img, uint64_t n, uint64_t o, uint32_t hf) {
    if (n - o < 18) return true;
    const LdsU16 hd = *(const LdsU16*)(img + o);
    const uint32_t ms = hd.a, s = hd.b, fl = hd.c & 0xFFFFu;
    if (s == 0 || s > 0x1000000u || fl >= 0x1000u) return false;
    const long long total = gw_total(ms, s, fl, hf);
    return total >= 18 && (unsigned long long)total > n - o;
}
__device__ __forceinline__ GwGeo gw_geo(const uint8_t* __restrict__ img, uint64_t n, uint64_t start, uint32_t hf, uint32_t max_seg) {
    GwGeo g; g.first = 0; g.seglen = n - start; g.win = 0; g.nseg = 1;
    uint64_t nx = 0;
    if (start < n && gw_plausible(img, n, start, hf, true, &nx)) {
        g.first = nx - start;
        const uint64_t want = 16ull * g.first; uint64_t ns = (n - start) / (want ? want : 1ull);
        if (ns < 1) ns = 1; if (ns > GW_SEGS) ns = GW_SEGS; if (ns > max_seg) ns = max_seg;     // (max_seg: what the host sized its grids for)
        g.nseg = (uint32_t)ns; g.seglen = (n - start + ns - 1) / ns; g.win = 2ull * g.first < g.seglen ? 2ull * g.first : g.seglen;
    }
    return g;
}
// cand[k] = the lowest plausible chunk start in the window at the head of segment k (k >= 1; cand[0] = start); ~0 when there is none
__global__ void k_dec_gw_find(const uint8_t* __restrict__ img, uint64_t n, uint64_t start, const DevHeader* __restrict__ D, unsigned long long* __restrict__ cand,
        uint32_t max_seg) {
    const uint32_t hf = D->flags; const GwGeo g = gw_geo(img, n, start, hf, max_seg);
    const uint32_t k = blockIdx.y + 1u; if (k >= g.nseg) return;
    const uint64_t g0 = start + (uint64_t)k * g.seglen;
    // first the two bytes that are almost never right by chance - read count < 2^24, flags < 0x1000 (bytes 6 .. 9 of a header): 1 offset in 4096 passes.
    // A thread tests 16 consecutive offsets from two 16-byte loads (the 19 bytes they look at); two such groups per round, their loads in flight
    // together.  (One dword load per offset - a wave instruction for 67 useful bytes, a thread's offsets one dependent round trip after the other - was
    // 171 us for the 135 MB of the bench image's windows.)
    const uint64_t stride = 16ull * gridDim.x * blockDim.x;
    for (uint64_t i = 16ull * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x); i < g.win; i += 2ull * stride) {
        uint32_t d[2][8];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint64_t ii = i + (uint64_t)u * stride; const bool in = ii < g.win && g0 + ii + 6 + 32 <= n;      // (loads without a branch around them)
            const uint8_t* p = img + (in ? g0 + ii : g0) + 6;
            const LdsU16 x = *(const LdsU16*)p, y = *(const LdsU16*)(p + 16);
            d[u][0] = x.a; d[u][1] = x.b; d[u][2] = x.c; d[u][3] = x.d; d[u][4] = y.a; d[u][5] = y.b; d[u][6] = y.c; d[u][7] = y.d;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint64_t ii = i + (uint64_t)u * stride;
            if (ii >= g.win) continue;
            if (g0 + ii + 6 + 32 > n) {                                       // the image's last bytes: offset by offset
                for (uint32_t j = 0; j < 16 && ii + j < g.win; j++) { const uint64_t o = g0 + ii + j;
                        if (o + 18 <= n && !(((const LdsU4*)(img + o + 6))->a & 0xF000FF00u) && gw_plausible(img, n, o, hf, false, nullptr)) atomicMin(&cand[k],
                        (unsigned long long)o); }
                continue;
            }
            uint32_t hit = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) { const uint32_t w = (uint32_t)((((unsigned long long)d[u][(j >> 2) + 1] << 32) | d[u][j >> 2]) >> (8 * (j & 3)));
                    if (!(w & 0xF000FF00u)) hit |= 1u << j; }
            while (hit) { const int j = __ffs((int)hit) - 1; hit &= hit - 1; const uint64_t o = g0 + ii + (uint32_t)j;
                    if (ii + (uint32_t)j < g.win && o + 18 <= n && gw_plausible(img, n, o, hf, false, nullptr)) atomicMin(&cand[k], (unsigned long long)o); }
        }
    }
}
// a wave per segment: the chain from its candidate up to the next segment that has one; list[k][..] = the chunk starts met, land[k] = where it stopped
__global__ void k_dec_gw_walk(const uint8_t* __restrict__ img, uint64_t n, uint64_t start, const DevHeader* __restrict__ D, const unsigned long long* __restrict__ cand,
                              unsigned long long* __restrict__ list, uint32_t* __restrict__ cnt, unsigned long long* __restrict__ land, uint32_t* __restrict__ bad, uint32_t max_seg, int final) {
    const uint32_t hf = D->flags; const GwGeo g = gw_geo(img, n, start, hf, max_seg);
    const uint32_t k = blockIdx.x; if (k >= g.nseg) return;
    uint64_t o = k == 0 ? start : cand[k];
    if (o == ~0ull) { if (lane_id() == 0) { cnt[k] = 0; land[k] = ~0ull; } return; }
    uint64_t stop = n; for (uint32_t m = k + 1; m < g.nseg; m++) if (cand[m] != ~0ull) { stop = cand[m]; break; }
    uint32_t c = 0, b = 0, ended = 0;
    while (o < stop) {
        uint64_t nx = 0;
        if (n - o < 18 || ld_u32(img + o + 4) == 0) { ended = 1; break; }   // end of the image
        if (!gw_plausible(img, n, o, hf, true, &nx)) {
            // a range that does not end the image may end inside a chunk: a header whose fields hold but whose size leads past the end stops the chain cleanly
            if (!final && gw_cut_by_end(img, n, o, hf)) { ended = 1; break; }
            b = 1; break;
        }
        if (c < GW_LCAP) { if (lane_id() == 0) list[(size_t)k * GW_LCAP + c] = o; } else { b = 1; break; }
        c++; o = nx;
    }
    if (o >= n || n - o < 18) ended = 1;
    if (lane_id() == 0) { cnt[k] = c | (ended << 31); land[k] = o; if (b) atomicOr(bad, 1u); }     // (bit 31: the chain ended in this segment)
}
// every walk must land on the next candidate; the lists, concatenated, are the chunk index (off[0 .. n_chunks], the last entry = where the chain ended)
__global__ void __launch_bounds__(1024) k_dec_gw_stitch(const uint8_t* __restrict__ img, uint64_t n, uint64_t start, const DevHeader* __restrict__ D,
        const unsigned long long* __restrict__ cand,
                                const unsigned long long* __restrict__ list, const uint32_t* __restrict__ cnt, const unsigned long long* __restrict__ land, const uint32_t* __restrict__ bad,
                                uint64_t* __restrict__ off, uint32_t cap, DecStatus* st, uint32_t max_seg) {
    __shared__ uint32_t s_base[GW_SEGS + 1], s_cnt[GW_SEGS]; __shared__ unsigned long long s_cand[GW_SEGS], s_land[GW_SEGS]; __shared__ uint32_t s_fail;
            __shared__ unsigned long long s_end;
    const uint32_t hf = D->flags; const GwGeo g = gw_geo(img, n, start, hf, max_seg);
    for (uint32_t k = threadIdx.x; k < g.nseg; k += blockDim.x) { s_cand[k] = k == 0 ? (unsigned long long)start : cand[k]; s_cnt[k] = cnt[k]; s_land[k] = land[k]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t fail = *bad, tot = 0; unsigned long long expect = start; bool open = true;      // expect: where the next walk must begin
        for (uint32_t k = 0; k < g.nseg; k++) {
            s_base[k] = tot;
            const unsigned long long ck = s_cand[k];
            if (ck == ~0ull) continue;
            if (!open) { fail = 1; break; }                                  // a candidate behind the end of the chain
            if (ck != expect) { fail = 1; break; }
            tot += s_cnt[k] & 0x7FFFFFFFu; expect = s_land[k];
            if (s_cnt[k] >> 31) open = false;                                // the chain has ended
        }
        s_base[g.nseg] = tot; s_fail = fail; s_end = expect;
        // (the first header itself is not plausible: let the chain decide)
        if (g.first == 0 && start < n && n - start >= 18 && ld_u32(img + start + 4) != 0) s_fail = 1;
    }
    __syncthreads();
    const uint32_t tot = s_base[g.nseg];
    if (s_fail || tot > cap) { if (threadIdx.x == 0) { st->pad = s_fail ? 1u : 0u; st->overflow = (!s_fail && tot > cap) ? 1u : 0u; st->n_chunks = tot; } return; }
    // (a wave per segment: its list is a handful of entries; sixteen waves - with four, a wave copied 50 segments one dependent load -> store after the other: 50 of the
    // kernel's 55 us)
    for (uint32_t k = threadIdx.x >> 6; k < g.nseg; k += blockDim.x >> 6) { const uint32_t c = s_cnt[k] & 0x7FFFFFFFu; if (s_cand[k] == ~0ull) continue;
            for (uint32_t i = threadIdx.x & 63u; i < c; i += 64u) off[s_base[k] + i] = list[(size_t)k * GW_LCAP + i]; }
    if (threadIdx.x == 0) { off[tot] = s_end; st->n_chunks = tot; st->pad = 0; st->overflow = 0; }
}
// a range of chunks is decoded as a batch of its own: its reads count from 0
__global__ void k_dec_rebase(DChunk* __restrict__ CH, uint32_t n, uint32_t base) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) CH[c].rbase = CH[c].rbase_abs - base;
}
// one wave per speculated chunk: full parse + verification of the extent
// (launched right behind the walk, before the host knows how many chunks it found: a fixed grid starting at chunk `first`, blocks past
// the walk's count - read from the status words - leave at once; nothing runs when the walk itself gave up or overflowed its table)
__global__ void k_dec_parse(const uint8_t* __restrict__ img, uint64_t n, const DevHeader* __restrict__ D, DChunk* __restrict__ CH, DecStatus* st, uint32_t first) {
    const uint32_t c = first + blockIdx.x; const uint32_t hf = D->flags, rlb = D->read_len_bytes;
    if (c >= st->n_chunks || st->overflow || st->pad) return;             // (pad: the chain / the caller's table already failed - its entries are not to be trusted)
    const uint64_t k = CH[c].off; const uint32_t want = CH[c].total, rbase = CH[c].rbase, reads = CH[c].reads;
    DChunk d; const int rc = (rlb == 1 || rlb == 2 || rlb == 4) ? parse_chunk(img, n, k, hf, rlb, d) : 2;
    if (rc != 0 || d.total != want || d.reads != reads) { if (lane_id() == 0) atomicOr(&st->pad, 1u); return; }
    d.rbase = rbase; d.rbase_abs = rbase;
    // what the host sizes its passes from - longest stream, exception records - stays with the chunk; k_dec_summary reduces it (thousands of waves
    // raising the same few maxima with atomics, all at once, were 130 of this kernel's 164 us)
    if ((hf & H_QUAL_BY_COL) && !(hf & H_DONT_QUAL) && 4ull * D->n_normal <= d.qual_size) {
        const uint8_t* qp = img + d.off + d.o_qual; uint64_t off = 4ull * D->n_normal; uint32_t mo = 0;
        for (uint32_t i = 0; i < D->n_normal; i++) { const uint32_t sl = ld_u32(qp + 4 * i); off += sl; if (sl > mo) mo = sl; }
        if (off <= d.qual_size) { d.nrec = (uint32_t)((d.qual_size - off) / 5); d.max_one = mo; }
    }
    if (lane_id() == 0) CH[c] = d;
}
// maxima / sums over the parsed chunks [first, first + count) -> the status words (one workgroup; the parse wrote every chunk's own values)
__global__ void k_dec_summary(const DChunk* __restrict__ CH, DecStatus* st, uint32_t first, uint32_t count) {
    if (st->overflow || st->pad) return;
    const uint32_t end = first + count < st->n_chunks ? first + count : st->n_chunks;
    uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, pr = 0, pa = 0, p1 = 0; unsigned long long sum = 0;
    for (uint32_t c = first + threadIdx.x; c < end; c += blockDim.x) {
        const DChunk& d = CH[c];
        if ((d.flags & (C_NAME1_SAME | C_NAME2_SAME | C_STRAND_SAME)) != (C_NAME1_SAME | C_NAME2_SAME | C_STRAND_SAME)) pr = 1;
        if (d.reads) {                                                     // per-read pieces: bytes per read against the emitter's tile capacity for that piece
            if (!(d.flags & C_NAME1_SAME)) { const uint32_t v = (d.n1_size + d.reads - 1) / d.reads; if (v > p1) p1 = v; }
            if (!(d.flags & C_NAME2_SAME)) { const uint32_t v = (uint32_t)(((unsigned long long)d.n2_size * 256ull / d.reads + ET_N2CAP - 1) / ET_N2CAP);
                    if (v > pa) pa = v; }
            if (!(d.flags & C_STRAND_SAME)) { const uint32_t v = (uint32_t)(((unsigned long long)d.st_size * 256ull / d.reads + ET_STCAP - 1) / ET_STCAP);
                    if (v > pa) pa = v; }
        }
        if (d.qual_size > m0) m0 = d.qual_size;
        if (d.npos_size > m1) m1 = d.npos_size;
        if (d.max_len > m2) m2 = d.max_len;
        const uint32_t b32 = d.bases > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)d.bases; if (b32 > m3) m3 = b32;
        if (d.nrec > m4) m4 = d.nrec;
        sum += d.bases;
    }
    uint32_t m5 = 0; for (uint32_t c = first + threadIdx.x; c < end; c += blockDim.x) if (CH[c].max_one > m5) m5 = CH[c].max_one;
    m0 = wave_max(m0); m1 = wave_max(m1); m2 = wave_max(m2); m3 = wave_max(m3); m4 = wave_max(m4); m5 = wave_max(m5); sum = wave_sum<unsigned long long>(sum);
    if (__any(pr != 0) && lane_id() == 0) atomicOr(&st->per_read_pieces, 1u);
    pa = wave_max(pa); if (pa && lane_id() == 0) atomicMax(&st->piece_avg, pa);
    p1 = wave_max(p1); if (p1 && lane_id() == 0) atomicMax(&st->piece_n1, p1);
    if (lane_id() == 0) { atomicMax(&st->max_stream, m0); atomicMax(&st->max_npos, m1); atomicMax(&st->max_len, m2); atomicMax(&st->max_bases, m3);
            atomicMax(&st->max_nrec, m4); atomicMax(&st->max_one, m5);
                          atomicAdd((unsigned long long*)&st->base_slots[0], sum); }
}
