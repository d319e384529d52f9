// rfq_decode_kernels.h — gfx950 kernels of the RFQ -> FASTQ decode path (included by rfq_decode.hip only).
//
//   walk     k_dec_walk                 RfqChunk::read for every chunk: section offsets (the reader ignores mSize,
//                                       src/rfqchunk.cpp:161-228), running read base
//   table    k_dec_readtab + scans      per-read lengths, overlap values, stored lengths, name/strand piece lengths
//   streams  k_dec_unpack, k_dec_pos, k_dec_except, k_dec_coords
//                                       decodeSeqQual / decodeSingleQualByCol / decodeQualByCol / decodeCoords
//                                       (src/rfqcodec.cpp:826-1047,1332-1389)
//   text     k_dec_textlen + scan, k_dec_emit
//                                       decodeChunk's per-read loop + Read::toString (src/rfqcodec.cpp:1141-1254, src/read.cpp:170)
#pragma once
#include "rfq_common.h"

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

struct DReadTab {
    uint32_t* len; uint32_t* chunk; int32_t* ov; U4* pvin; U4* pv; uint32_t* pq; U4* tin; U4* tp;
    uint8_t* mid;                // [g][40]: the formatted ":lane:tile:x:y" middle of the name (<= 4+6+11+11 bytes); mid[g*40+39] = its length
};
__device__ __forceinline__ uint32_t dec_read_len(const uint8_t* cp, const DChunk& d, uint32_t rlb, uint32_t r) {
    const uint8_t* p = cp + d.o_readlens + (size_t)((d.flags & C_READ_LEN_SAME) ? 0u : r) * rlb;
    return rlb == 1 ? p[0] : (rlb == 2 ? ld_u16(p) : ld_u32(p));
}
// grid (ceil(max_reads/256), n_chunks)
__global__ void k_dec_readtab(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, const DevHeader* __restrict__ D, DReadTab R, DecStatus* st) {
    const DChunk d = CH[blockIdx.y]; const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= d.reads) return;
    const uint8_t* cp = img + d.off; const uint32_t g = d.rbase + r, fl = d.flags, hf = D->flags;
    const uint32_t len = dec_read_len(cp, d, D->read_len_bytes, r);
    U4 v;
    v.a = (fl & C_NAME1_SAME) ? 0u : cp[d.o_n1lens + ((fl & C_NAME1_LEN_SAME) ? 0u : r)];
    v.b = ((hf & H_NAME2) && !(fl & C_NAME2_SAME)) ? cp[d.o_n2lens + ((fl & C_NAME2_LEN_SAME) ? 0u : r)] : 0u;
    v.c = (fl & C_STRAND_SAME) ? 0u : cp[d.o_stlens + ((fl & C_STRAND_LEN_SAME) ? 0u : r)];
    int ov = 0; uint32_t stored = len;
    if ((fl & C_PE_INTERLEAVED) && (hf & H_PE_OVERLAP) && (r & 1u)) {
        ov = (int)(int8_t)cp[d.o_ov + r / 2] - D->overlap_shift;
        const uint32_t a = (uint32_t)(ov < 0 ? -ov : ov);
        const uint32_t prevlen = dec_read_len(cp, d, D->read_len_bytes, r - 1);
        if (a > len || a > prevlen) { atomicOr(&st->err, (uint32_t)DE_CORRUPT); ov = 0; } else stored = len - a;
    }
    v.d = stored;
    R.len[g] = len; R.chunk[g] = blockIdx.y; R.ov[g] = ov; R.pvin[g] = v;
}
// aligned bases of each chunk inside the concatenated quality / stored-sequence buffers
__global__ void k_dec_bases(const DChunk* __restrict__ CH, DReadTab R, uint64_t* __restrict__ qbase, uint64_t* __restrict__ sbase, uint32_t n_chunks) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_chunks) { const uint32_t f = CH[c].rbase; qbase[c] = ((uint64_t)R.pq[f] & ~63ull) + 64ull * c; sbase[c] = ((uint64_t)R.pv[f].d & ~63ull) + 64ull * c; }
}

// 2-bit unpack (src/rfqcodec.cpp:833-853): grid (blocks, n_chunks).  One thread turns 4 packed bytes into 16 bases and stores them
// as one aligned uint4 (the chunk's base in sdec is 64-byte aligned); byte stores cost ~30 cycles per wave instruction.
__device__ __forceinline__ uint32_t ld_word_lim(const uint8_t* p, const uint8_t* lim);
__global__ void k_dec_unpack(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, DReadTab R, const uint64_t* __restrict__ sbase, uint8_t* __restrict__ sdec,
        uint64_t img_bytes) {
    // packed byte -> its four bases without a table: two shift-and-mask steps spread the four 2-bit codes over four bytes, v_perm_b32
    // looks them up in the 4-entry G A T C table (an LDS table cost a bank-conflicted read per byte and a fill per block)
    auto unpack4v = [](uint32_t b) -> uint32_t {
        const uint32_t y = (b | (b << 12)) & 0x000F000Fu, idx = (y | (y << 6)) & 0x03030303u;
        return __builtin_amdgcn_perm(0u, 0x43544147u, idx);
    };
    const DChunk d = CH[blockIdx.y]; const uint32_t f = d.rbase;
    const uint32_t n = R.pv[f + d.reads].d - R.pv[f].d;          // stored bases of the chunk
    const uint8_t* src = img + d.off + d.o_seq; uint8_t* dst = sdec + sbase[blockIdx.y]; const uint8_t* lim = img + img_bytes;
    const uint32_t ngroups = (n + 15) / 16, NT = gridDim.x * blockDim.x;
    // a group's four packed bytes sit at any phase: two aligned words + a funnel shift; four groups per thread in flight
    const uint32_t ph = (uint32_t)((uintptr_t)src & 3u); const uint8_t* sa = src - ph; const uint32_t sh = ph * 8u;
    for (uint32_t g0 = blockIdx.x * blockDim.x + threadIdx.x; g0 < ngroups; g0 += 4 * NT) {
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t gi = g0 + (uint32_t)u * NT; lo[u] = hi[u] = 0; if (gi < ngroups) { lo[u] = ld_word_lim(sa + 4 * (size_t)gi, lim);
                if (ph) hi[u] = ld_word_lim(sa + 4 * (size_t)gi + 4, lim); } }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t gi = g0 + (uint32_t)u * NT; if (gi >= ngroups) continue;
            const uint32_t pk = ph ? (uint32_t)((((uint64_t)hi[u] << 32) | lo[u]) >> sh) : lo[u];
            uint32_t w[4];
#pragma unroll
            // beyond mSeqBuf the 'N' prefill of allSeq stays (src/rfqcodec.cpp:1088)
            for (int k = 0; k < 4; k++) { const uint32_t i = 4 * gi + (uint32_t)k; w[k] = i < d.seq_size ? unpack4v((pk >> (8 * k)) & 0xFFu) : 0x4E4E4E4Eu; }
            if (16 * gi + 16 <= n) *(uint4*)(dst + 16 * (size_t)gi) = make_uint4(w[0], w[1], w[2], w[3]);
            else for (uint32_t p = 16 * gi; p < n; p++) dst[p] = (uint8_t)(w[(p >> 2) & 3u] >> (8 * (p & 3u)));
        }
    }
}

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
#pragma unroll
    for (int k = 0; k < 4; k++) { const bool valid = i0 + (uint32_t)k < slen; f.bt[k] = (uint32_t)(f.v >> (8 * k)) & 0xFFu;
            f.fn[k] = valid ? fn_of_len(pos_tok_len(f.bt[k])) : POS_ID; }
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
#define POS2_SEG 1024u            // bytes of a stream per wave: POS2_SEG / 256 steps of 4 bytes per lane
#define POS2_CELL 1024u
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
// grid (ceil(maxseg / 4), streams, n_chunks) x 256 threads: one wave per segment; index arrays are [chunk][nstr][maxseg]; segA[8 * idx + s] =
// positions advanced, segA[8 * idx + 4 + s] = positions emitted for entry state s
__global__ void k_dec_pos_sum2(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, const DevHeader* __restrict__ D,
                               uint8_t* __restrict__ segF, int* __restrict__ segA, uint32_t* __restrict__ segN, uint32_t maxseg, DecStatus* st, uint64_t img_bytes, uint32_t jj0, uint32_t nstr) {
    const uint32_t g = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave_id(), jj = jj0 + blockIdx.y, c = blockIdx.z; const int l = lane_id();
            const uint8_t* lim = img + img_bytes;
    const DChunk d = CH[c];
    const PosSrc s = pos_src_of(img, d, D, jj, g == 0 ? st : nullptr);
    if (g == 0 && l == 0) segN[(size_t)c * nstr + jj] = (s.slen + POS2_SEG - 1) / POS2_SEG;
    const uint32_t b0 = g * POS2_SEG; if (b0 >= s.slen) return;
    const uint32_t b1 = b0 + POS2_SEG < s.slen ? b0 + POS2_SEG : s.slen;
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
#pragma unroll
    for (int e = 0; e < 4; e++) { a[e] = wave_sum(a[e]); n[e] = wave_sum(n[e]); }
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
    for (int s = 0; s < 4; s++) { const uint32_t t = fn_apply(x.F, s); r.a[s] = x.a[s] + sel4(y.a, t); r.n[s] = x.n[s] + sel4(y.n, t); }
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
                                uint32_t* __restrict__ segK, uint32_t* __restrict__ nent, uint32_t maxseg, uint32_t n_streams) {
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
        if (g < n) { segS[idx] = (uint8_t)(fn_apply(ex.F, cs)); segP[idx] = cp + sel4(ex.a, cs); segK[idx] = ck + (uint32_t)sel4(ex.n, cs); }
        const uint32_t Fl = wave_last(inc.F); int al[4], nl[4];
#pragma unroll
        for (int s = 0; s < 4; s++) { al[s] = wave_last(inc.a[s]); nl[s] = wave_last(inc.n[s]); }
        cp += sel4(al, cs); ck += (uint32_t)sel4(nl, cs); cs = fn_apply(Fl, cs);
    }
    if (l == 0) nent[t] = ck;
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
                               uint32_t* __restrict__ plist, unsigned long long cap, uint32_t* __restrict__ cellidx, uint32_t maxseg, uint32_t ncell, uint64_t img_bytes, uint32_t jj0, uint32_t nstr, const DecStatus* st) {
    if (st->list_need > cap) return;                                      // (uniform) the arena is too small: the host repeats the pass
    const uint32_t g = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave_id(), jj = jj0 + blockIdx.y, c = blockIdx.z; const int l = lane_id();
            const uint8_t* lim = img + img_bytes;
    const DChunk d = CH[c];
    const PosSrc s = pos_src_of(img, d, D, jj, nullptr);
    const uint32_t b0 = g * POS2_SEG; if (b0 >= s.slen) return;
    const size_t t = (size_t)c * nstr + jj, idx = t * maxseg + g;
    uint32_t carry = segS[idx]; int last = segP[idx]; uint32_t k0 = segK[idx];
    uint32_t* const out = plist + loff[t]; uint32_t* const cells = cellidx + t * ncell;
    const uint32_t b1 = b0 + POS2_SEG < s.slen ? b0 + POS2_SEG : s.slen;
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
                out[k] = (uint32_t)p;
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
    for (uint32_t base = 0; base < slen; base += 64) {
        const uint32_t i = base + (uint32_t)l; const bool valid = i < slen;
        const uint32_t b0 = valid ? sp[i] : 0u;
        const uint32_t tl = (b0 & 0x80u) == 0 ? 2u : ((b0 & 0xE0u) == 0xE0u ? 3u : 1u);
        const uint32_t before = wave_token_states(tl, valid, carry);
        const bool start = valid && before == 0;
        uint32_t cnt = 0, isabs = 0, val = 0;                      // val: absolute value, or the +diff
        if (start) {
            if ((b0 & 0x80u) == 0) { isabs = 1; val = (b0 << 8) | (i + 1 < slen ? sp[i + 1] : 0u); cnt = 1; }
            else if ((b0 & 0x40u) == 0) { val = (b0 & 0x3Fu) + 1; cnt = 1; }
            else if ((b0 & 0x20u) == 0) { val = 0; cnt = (b0 & 0x1Fu) + 1; }
            else { isabs = 1; val = ((b0 & 0x1Fu) << 16) | ((i + 1 < slen ? sp[i + 1] : 0u) << 8) | (i + 2 < slen ? sp[i + 2] : 0u); cnt = 1; }
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

// ---- text
__device__ __forceinline__ uint32_t dec_digits(uint32_t v) { uint32_t n = 1; while (v >= 10) { v /= 10; n++; } return n; }
__device__ __forceinline__ uint32_t dec_put(uint8_t* dst, uint32_t v) { const uint32_t n = dec_digits(v); for (uint32_t k = 0; k < n; k++) { dst[n - 1 - k] = (uint8_t)('0' + v % 10); v /= 10; } return n; }
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
    if (hf & H_LANE) { buf[k++] = ':'; k += dec_put(buf + k, m.lane); }
    if (hf & H_TILE) { buf[k++] = ':'; k += dec_put(buf + k, m.tile); }
    if (hf & H_X) { buf[k++] = ':'; k += dec_put(buf + k, m.x); }
    if (hf & H_Y) { buf[k++] = ':'; k += dec_put(buf + k, m.y); }
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
// ---- text emission (name re-assembly src/rfqcodec.cpp:1157-1231, overlap re-expansion :865-897, implied N :1093-1100, RC of odd
// reads :1248-1252, Read::toString src/read.cpp:170).
// One wave writes one read's four lines.  w = destination, sb / qb = stored bases / qualities addressed so that sb[sp], qb[qp] are
// the read's first stored base / quality (either global memory or the LDS copies of a tile).
struct EmitRead {
    uint32_t len, n1, n2, stl, mid, sp, qp, prevlen; int ov; bool rc, patch;
    const uint8_t *n1p, *n2p, *stp, *mp;
};
__device__ __forceinline__ void emit_one(uint8_t* w, const EmitRead& e, const uint8_t* sb, const uint8_t* qb, bool implied_n, uint32_t nq,
                                         uint32_t dpos, uint32_t dch, int l) {
    for (uint32_t i = (uint32_t)l; i < e.n1; i += 64) w[i] = e.n1p[i];
    if ((uint32_t)l < e.mid) w[e.n1 + (uint32_t)l] = e.mp[l];
    uint8_t* w2 = w + e.n1 + e.mid;
    for (uint32_t i = (uint32_t)l; i < e.n2; i += 64) w2[i] = (e.patch && i == dpos) ? (uint8_t)dch : e.n2p[i];
    if (l == 0) w2[e.n2] = '\n';
    uint8_t* ws = w2 + e.n2 + 1; uint8_t* wst = ws + e.len + 1; uint8_t* wq = wst + e.stl + 1;
    const uint32_t len = e.len; const int ov = e.ov;
    for (uint32_t k = (uint32_t)l; k < len; k += 64) {
        const uint32_t p = e.rc ? len - 1 - k : k;                     // position in interleaved orientation
        uint8_t b;
        if (ov > 0) b = p < (uint32_t)ov ? sb[e.sp - (uint32_t)ov + p] : sb[e.sp + p - (uint32_t)ov];
        else if (ov < 0) { const uint32_t keep = len - (uint32_t)(-ov); b = p < keep ? sb[e.sp + p] : sb[e.sp - e.prevlen + (p - keep)]; }
        else b = sb[e.sp + p];
        const uint8_t q = qb[e.qp + p];
        if (implied_n && q == nq) b = 'N';
        ws[k] = e.rc ? comp_base(b) : b; wq[k] = q;
    }
    for (uint32_t i = (uint32_t)l; i < e.stl; i += 64) wst[i] = e.stp[i];
    if (l == 0) { ws[len] = '\n'; wst[e.stl] = '\n'; wq[len] = '\n'; }
}
// global [gbeg, gend) -> LDS so that LDS offset == (global address & 15) + (addr - gbeg): aligned 16-byte loads; the last group is
// fetched byte-wise when it would cross `glimit` (end of the allocation's valid bytes)
__device__ __forceinline__ void stage_span(uint4* lds4, const uint8_t* gbase, uint64_t gbeg, uint64_t gend, uint64_t glimit) {
    const uint64_t a0 = gbeg & ~15ull; const uint32_t ng = (uint32_t)((gend - a0 + 15) / 16);
    uint8_t* lds = (uint8_t*)lds4;
    for (uint32_t i = threadIdx.x; i < ng; i += blockDim.x) {
        const uint64_t ga = a0 + 16ull * i;
        if (ga + 16 <= glimit) lds4[i] = *(const uint4*)(gbase + ga);
        else for (uint32_t k = 0; k < 16 && ga + k < glimit; k++) lds[16 * i + k] = gbase[ga + k];
    }
}
// Several spans at once: all their loads are in flight together, so the tile's six small spans cost ONE memory latency instead of six.
struct StageSpan { const uint8_t* g; uint64_t a0; uint32_t ng; uint4* l; uint64_t lim; };
__device__ __forceinline__ StageSpan make_span(uint4* lds4, const uint8_t* gbase, uint64_t gbeg, uint64_t gend, uint64_t glimit, bool on) {
    StageSpan s; s.g = gbase; s.a0 = gbeg & ~15ull; s.ng = on ? (uint32_t)((gend - s.a0 + 15) / 16) : 0u; s.l = lds4; s.lim = glimit; return s;
}
// Span by span, every thread taking groups tid, tid + blockDim, ... of each: which span a load belongs to is then known at compile
// time (the earlier "one flat index space" form spent ~60 VALU instructions per group on selecting the span's base / limit /
// destination, ~300 per wave and tile in a VALU-bound kernel).  UMAX = groups per thread the caller's capacities allow for the
// span (a slower loop covers anything beyond).  A span that ends >= 16 bytes before its buffer's limit loads without per-group
// limit tests.
// byte-wise near the buffer's limit
static __device__ __noinline__ void stage_span_slow(const uint8_t* g, uint64_t a0, uint32_t ng, uint4* l, uint64_t lim, uint32_t from) {
    for (uint32_t i = threadIdx.x + from; i < ng; i += blockDim.x) {
        const uint64_t ga = a0 + 16ull * i; uint32_t w[4] = { 0, 0, 0, 0 };
        for (uint32_t b = 0; b < 16 && ga + b < lim; b++) w[b >> 2] |= (uint32_t)g[ga + b] << (8 * (b & 3));
        l[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// The aligned body of a span goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: each lane names its own 16 global bytes, the
// wave's 64 groups land contiguously at a wave-uniform LDS address): no staging registers, no ds_write pass, nothing to wait for
// until the barrier - staging through registers made this VALU- and register-bound kernel spill.  U = groups per thread the
// caller's capacities allow (a slower loop covers anything beyond, and a span that ends < 16 bytes before its buffer's limit).
template <int U> __device__ __forceinline__ void span_dma(const StageSpan& sp) {
    const bool inside = sp.a0 + 16ull * sp.ng <= sp.lim;                     // block-uniform
    const uint8_t* const gp = sp.g + sp.a0; const uint32_t w0 = threadIdx.x & ~63u;
    if (inside) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = threadIdx.x + (uint32_t)u * blockDim.x;
            if (i < sp.ng) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + 16u * i),
                                                            (__attribute__((address_space(3))) void*)(sp.l + (w0 + (uint32_t)u * blockDim.x)), 16, 0, 0);
        }
        if (sp.ng > (uint32_t)U * blockDim.x) stage_span_slow(sp.g, sp.a0, sp.ng, sp.l, sp.lim, (uint32_t)U * blockDim.x);   // (never with the tile sizes above)
    } else stage_span_slow(sp.g, sp.a0, sp.ng, sp.l, sp.lim, 0u);
}
// The same by ONE wave (lane l of it): R rounds of 64 groups.  A tile's small spans are dealt out one per wave - a wave then runs the
// address arithmetic and the issue of its own span only.
template <int R> __device__ __forceinline__ void span_dma_wave(const StageSpan& sp, int l) {
    const bool inside = sp.a0 + 16ull * sp.ng <= sp.lim;                     // wave-uniform
    const uint8_t* const gp = sp.g + sp.a0;
    uint32_t done = 0;
    if (inside) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t i = (uint32_t)l + 64u * (uint32_t)r;
            if (i < sp.ng) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + 16u * i),
                                                            (__attribute__((address_space(3))) void*)(sp.l + 64u * (uint32_t)r), 16, 0, 0);
        }
        done = 64u * (uint32_t)R;
    }
    for (uint32_t i = (uint32_t)l + done; i < sp.ng; i += 64u) {              // (near the buffer's limit, or beyond R rounds: byte-wise)
        const uint64_t ga = sp.a0 + 16ull * i; uint32_t w[4] = { 0, 0, 0, 0 };
        for (uint32_t b = 0; b < 16 && ga + b < sp.lim; b++) w[b >> 2] |= (uint32_t)sp.g[ga + b] << (8 * (b & 3));
        sp.l[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// the emit tile's six spans: two big ones (UB groups per thread) and four small ones (one group per thread)
template <int UB> __device__ __forceinline__ void stage_spans6(const StageSpan (&sp)[6]) {
    span_dma<UB>(sp[0]); span_dma<UB>(sp[1]); span_dma<1>(sp[2]); span_dma<1>(sp[3]); span_dma<1>(sp[4]); span_dma<1>(sp[5]);
}
// LDS tile -> global [gbeg, gend): the tile sits at LDS offset (gbeg & 15) so body groups are aligned on both sides
__device__ __forceinline__ void flush_span(const uint4* lds4, uint8_t* gbase, uint64_t gbeg, uint64_t gend) {
    if (gend <= gbeg) return;
    const uint8_t* lds = (const uint8_t*)lds4; const uint64_t a0 = gbeg & ~15ull;
    const uint64_t first_full = (gbeg + 15) & ~15ull, last_full = gend & ~15ull;
    if (first_full < last_full) { const uint32_t ng = (uint32_t)((last_full - first_full) / 16), g0 = (uint32_t)((first_full - a0) / 16);
        for (uint32_t i = threadIdx.x; i < ng; i += blockDim.x) *(uint4*)(gbase + first_full + 16ull * i) = lds4[g0 + i]; }
    const uint64_t he = first_full < gend ? first_full : gend;
    for (uint64_t x = gbeg + threadIdx.x; x < he; x += blockDim.x) gbase[x] = lds[x - a0];
    if (last_full >= first_full) for (uint64_t x = last_full + threadIdx.x; x < gend; x += blockDim.x) gbase[x] = lds[x - a0];
}
// complement of four bases drawn from {A,C,G,T,N} - the only bytes the decoder itself puts into its base buffer (2-bit unpack,
// N positions): A<->T is x ^ 0x15, C<->G is x ^ 0x04, N stays (Read::changeToReverseComplement, src/read.cpp:77-115, on that alphabet)
__device__ __forceinline__ uint32_t comp4_acgtn(uint32_t w) {
    const uint32_t b1 = (w >> 1) & 0x01010101u, b3 = (w >> 3) & 0x01010101u;
    const uint32_t cg = b1 & ~b3, at = b1 ^ 0x01010101u;
    return w ^ (cg * 0x04u + at * 0x15u);
}
// One piece of the emit tile: 16-byte groups [g0, g1) of the piece's ceil(n / 16), copied from the LDS source pool to the LDS output
// tile.  Both sides are byte-granular ds_read_b128 / ds_write_b128 (LDS runs in unaligned access mode), so a group is simply bytes
// [16g, 16g + 16) of the piece; the last group of a piece >= 16 bytes is moved back to end exactly at n (it rewrites a few bytes of
// its predecessor with the same values), a piece < 16 bytes is stored as 8 + 4 + 2 + 1.  No head / tail edge cases per word - the
// destination-aligned form spent most of its instructions there.  SEQ: bases (complement for a reversed piece, implied N where
// the quality equals the header's N quality); REV: the piece may be emitted back to front; PAT: one byte of the piece is replaced
// (the mate's differing name character).
struct __attribute__((packed, aligned(1))) LdsW8 { uint32_t a, b; };
struct __attribute__((packed, aligned(1))) LdsW2 { uint16_t a; };
template <bool SEQ, bool REV, bool PAT>
__device__ __forceinline__ void emit_copy(uint8_t* o, const uint8_t* pool, uint32_t src, uint32_t n, uint32_t g0, uint32_t g1, bool rev_,
                                          uint32_t qsrc, bool implied_n, uint32_t nq, int pat, uint32_t dch) {
    const bool rev = REV && rev_;
    for (uint32_t g = g0; g < g1; g++) {
        uint32_t p0 = 16u * g; const bool small = n < 16u;
        if (p0 + 16u > n && !small) p0 = n - 16u;
        uint32_t w[4];
        // bytes [p0, p0 + 16) of the piece: forward from src + p0; reversed they are the 16 source bytes ENDING at src + n - p0
        lds_get16(pool, rev ? src + n - p0 - 16u : src + p0, w);
        if (REV && rev) { const uint32_t x0 = bswap32(w[3]), x1 = bswap32(w[2]), x2 = bswap32(w[1]), x3 = bswap32(w[0]); w[0] = x0; w[1] = x1; w[2] = x2; w[3] = x3; }
        if (SEQ) {
            if (rev) { w[0] = comp4_acgtn(w[0]); w[1] = comp4_acgtn(w[1]); w[2] = comp4_acgtn(w[2]); w[3] = comp4_acgtn(w[3]); }
            if (implied_n) {
                uint32_t qw[4]; lds_get16(pool, rev ? qsrc + n - p0 - 16u : qsrc + p0, qw);
                if (rev) { const uint32_t x0 = bswap32(qw[3]), x1 = bswap32(qw[2]), x2 = bswap32(qw[1]), x3 = bswap32(qw[0]); qw[0] = x0; qw[1] = x1; qw[2] = x2;
                        qw[3] = x3; }
#pragma unroll
                for (int i = 0; i < 4; i++) { const uint32_t mk = eq_bytes_full(qw[i], (nq & 0xFFu) * 0x01010101u); w[i] = (w[i] & ~mk) | (0x4E4E4E4Eu & mk); }
            }
        }
        if (PAT && pat >= (int)p0 && pat < (int)p0 + 16) { const int b = pat - (int)p0; const uint32_t sh = 8u * (uint32_t)(b & 3); uint32_t& x = w[b >> 2];
                x = (x & ~(0xFFu << sh)) | ((dch & 0xFFu) << sh); }
        uint8_t* q = o + p0;
        if (!small) { LdsU16 v; v.a = w[0]; v.b = w[1]; v.c = w[2]; v.d = w[3]; *(LdsU16*)q = v; }
        else {
            if (n & 8u) { LdsW8 v; v.a = w[0]; v.b = w[1]; *(LdsW8*)q = v; q += 8; w[0] = w[2]; w[1] = w[3]; }
            if (n & 4u) { LdsU4 v; v.a = w[0]; *(LdsU4*)q = v; q += 4; w[0] = w[1]; }
            if (n & 2u) { LdsW2 v; v.a = (uint16_t)w[0]; *(LdsW2*)q = v; q += 2; w[0] >>= 16; }
            if (n & 1u) *q = (uint8_t)w[0];
        }
    }
}
#define ET_READS 32
#define EM_ROW 17                 // words per read in s_meta: 16 used + 1 pad, so that lanes reading the same field of consecutive reads hit 32 different banks
#define ET_OCAP 12288u            // output tile bytes (split: half per stream): 32 records of 357 bytes are 11.4 KB
#define ET_SCAP 5632u             // staged qualities / stored bases
__global__ void k_dec_emit(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, const DevHeader* __restrict__ D, DReadTab R,
                           const uint64_t* __restrict__ qbase, const uint64_t* __restrict__ sbase,
                           const uint8_t* __restrict__ qdec, const uint8_t* __restrict__ sdec, uint64_t qdec_bytes, uint64_t sdec_bytes, uint64_t img_bytes, int split,
                           uint8_t* __restrict__ out1, uint64_t cap1, uint8_t* __restrict__ out2, uint64_t cap2, DecStatus* st) {
    __shared__ uint4 s_out4[ET_OCAP / 16 + 4];
    // staged sources in ONE pool (a piece is addressed by a byte offset into it): qualities | stored bases | name middles | name1 | name2 |
    // strand pieces (region starts in uint4 units)
#define EG_Q 0
#define EG_S (EG_Q + ET_SCAP / 16 + 4)
#define EG_MID (EG_S + ET_SCAP / 16 + 4)
#define EG_N1 (EG_MID + ET_READS * 40 / 16 + 4)
#define EG_N2 (EG_N1 + ET_N1CAP / 16 + 4)
#define EG_ST (EG_N2 + ET_N2CAP / 16 + 4)
#define EG_END (EG_ST + ET_STCAP / 16 + 4)
    __shared__ uint4 s_src4[EG_END];
#define s_q4 (s_src4 + EG_Q)
#define s_s4 (s_src4 + EG_S)
#define s_mid4 (s_src4 + EG_MID)
#define s_n14 (s_src4 + EG_N1)
#define s_n24 (s_src4 + EG_N2)
#define s_st4 (s_src4 + EG_ST)
    // scalars of the tile's reads: two buffers, the next tile's are fetched while this one's sources are staged
    __shared__ uint32_t s_cnt; __shared__ __attribute__((aligned(16))) uint32_t s_meta2[2][(ET_READS + 1) * EM_ROW];
    const uint32_t c = blockIdx.y; const DChunk d = CH[c]; const uint8_t* cp = img + d.off;
    const uint32_t fl = d.flags, hf = D->flags, f = d.rbase; const bool il = (fl & C_PE_INTERLEAVED) != 0;
    const bool implied_n = !(hf & H_N_POS); const uint32_t nq = D->n_base_qual, dpos = D->name2_diff_pos, dch = D->name2_diff_char;
    const uint32_t wpb = blockDim.x >> 6; const int l = lane_id(); const uint32_t tid = threadIdx.x;
    const U4 pv0 = R.pv[f]; const uint32_t pq0 = R.pq[f];
    const uint64_t qg0 = qbase[c], sg0 = sbase[c];                        // chunk bases inside qdec / sdec
    uint32_t per = (d.reads + gridDim.x - 1) / gridDim.x; per = (per + 1u) & ~1u;
    const uint32_t rs = blockIdx.x * per; const uint32_t re = rs + per < d.reads ? rs + per : d.reads;
    const uint32_t ocap = split ? ET_OCAP / 2 : ET_OCAP;
    // scalars of the <= ET_READS reads from `from` on (+1 end sentinel) -> `mrow`: one parallel round of global loads by the first
    // ET_READS + 1 threads.  It runs for tile t+1 inside the staging phase of tile t (same wait as the staged sources), so that a
    // tile's chain is one global-load latency, not two; holding them in registers across the compose phase instead was tried
    // and cost a resident block per CU
#define EMIT_META_VARS U4 tp_, pv_; uint32_t pq_ = 0, len_ = 0, ov_ = 0, pl_ = 0, n1_ = 0, n2_ = 0, sl_ = 0, md_ = 0, r_ = 0; bool odd_ = false; tp_.a = tp_.b = 0; pv_.a = pv_.b = pv_.c = pv_.d = 0;
#define EMIT_META_LOAD(from)                                                                                                          \
        { r_ = (from) + tid; const uint32_t g_ = f + r_; odd_ = (r_ & 1u) != 0;                                                       \
          tp_ = R.tp[g_]; pv_ = R.pv[g_]; pq_ = R.pq[g_];                                                                             \
          if (r_ < re) {                                                                                                              \
              len_ = R.len[g_]; ov_ = (uint32_t)R.ov[g_]; pl_ = odd_ ? R.len[g_ - 1] : 0u;                                            \
              n1_ = cp[d.o_n1lens + ((fl & C_NAME1_LEN_SAME) ? 0u : r_)];                                                             \
              n2_ = (hf & H_NAME2) ? cp[d.o_n2lens + ((fl & C_NAME2_LEN_SAME) ? 0u : r_)] : 0u;                                       \
              sl_ = cp[d.o_stlens + ((fl & C_STRAND_LEN_SAME) ? 0u : r_)]; md_ = R.mid[(size_t)g_ * 40 + 39];                         \
          } }
#define EMIT_META_STORE(mrow)                                                                                                         \
        { uint32_t* m = (mrow) + EM_ROW * tid;                                                                                        \
          m[12] = tp_.a; m[13] = tp_.b; m[14] = pv_.d - pv0.d; m[15] = pq_ - pq0;          /* prefix values (also valid for the sentinel) */ \
          m[7] = pv_.a - pv0.a; m[8] = pv_.b - pv0.b; m[9] = pv_.c - pv0.c;                                                           \
          if (r_ < re) { m[0] = (split && odd_) ? tp_.b : tp_.a; m[1] = len_; m[2] = ov_; m[3] = pl_; m[4] = n1_; m[5] = n2_; m[6] = sl_;   \
                         m[11] = md_; m[10] = n1_ + md_ + n2_ + 1; } }                     /* ":lane:tile:x:y" bytes; offset of the sequence line */
    uint32_t cur = rs; uint32_t pb = 0;
    { EMIT_META_VARS if (cur < re && tid <= ET_READS && cur + tid <= re) { EMIT_META_LOAD(cur) EMIT_META_STORE(s_meta2[0]) } }
    __syncthreads();
    while (cur < re) {                                                       // block-uniform
        uint32_t* const s_meta = s_meta2[pb]; uint32_t* const s_next = s_meta2[pb ^ 1u];
        const uint32_t g0 = f + cur;
        // ---- phase 2: how many reads fit (from LDS)
        const uint32_t* mb = s_meta;                                          // entry 0 = first read of the tile
#define EMIT_FITS(me) ((me[12] - mb[12]) + 16u <= ocap && (me[13] - mb[13]) + 16u <= ocap && (me[15] - mb[15]) + 48u <= ET_SCAP && (me[14] - mb[14]) + 48u <= ET_SCAP \
                && ((fl & C_NAME1_SAME) || (me[7] - mb[7]) + 32u <= ET_N1CAP) && ((fl & C_NAME2_SAME) || (me[8] - mb[8]) + 32u <= ET_N2CAP)                          \
                && ((fl & C_STRAND_SAME) || (me[9] - mb[9]) + 32u <= ET_STCAP))
        // the usual case - all ET_READS candidates (or all that are left) fit - is one test every thread makes for itself on the same
        // LDS words: no vote, no barrier.  Only a tile of unusually long reads goes through the per-candidate vote.
        const uint32_t all = re - cur < ET_READS ? re - cur : ET_READS;
        uint32_t cnt;
        { const uint32_t* ma = s_meta + EM_ROW * all; cnt = EMIT_FITS(ma) ? all : 0xFFFFFFFFu; }
        if (cnt == 0xFFFFFFFFu) {                                          // block-uniform
        bool fits = false;
        if (tid < ET_READS && cur + tid < re) {
            uint32_t mm = (tid + 2u) & ~1u; if (cur + mm > re) mm = re - cur;  // whole pairs (a lone last read of an SE chunk is fine)
            const uint32_t* me = s_meta + EM_ROW * mm;
            fits = EMIT_FITS(me);
        }
        if (tid < 64) { const unsigned long long fb = __ballot(fits); if (l == 0) s_cnt = (uint32_t)__popcll(fb); }    // ET_READS <= 64: wave 0 holds every candidate
        __syncthreads();
        cnt = s_cnt; if (cur + cnt > re) cnt = re - cur;
        }
#undef EMIT_FITS
        const bool tiled = cnt > 0;
        if (!tiled) { cnt = 2; if (cur + cnt > re) cnt = re - cur; }      // oversized read / pair: straight to global memory, byte-wise
        const uint32_t g1 = g0 + cnt; const uint32_t* me = s_meta + EM_ROW * cnt;
        U4 tp0, tp1; tp0.a = mb[12]; tp0.b = mb[13]; tp1.a = me[12]; tp1.b = me[13];
        const uint32_t q0 = mb[15], s0 = mb[14];
        const uint64_t qa = qg0 + q0, qe = qg0 + me[15], sa = sg0 + s0, se = sg0 + me[14];
        // ---- phase 3: stage the tile's sources with aligned 16-byte loads.  name1 / name2 / strand: one copy when the chunk stores
        // them once, else the contiguous run of the tile's reads
        const uint64_t ib = d.off;                                         // global byte offsets inside the image
        const uint32_t a7 = (fl & C_NAME1_SAME) ? 0u : mb[7], a8 = (fl & C_NAME2_SAME) ? 0u : mb[8], a9 = (fl & C_STRAND_SAME) ? 0u : mb[9];
        const uint64_t n1a = ib + d.o_n1 + a7, n1e = (fl & C_NAME1_SAME) ? n1a + d.n1_size : ib + d.o_n1 + me[7];
        const uint64_t n2a = ib + d.o_n2 + a8, n2e = (fl & C_NAME2_SAME) ? n2a + d.n2_size : ib + d.o_n2 + me[8];
        const uint64_t sta = ib + d.o_st + a9, ste = (fl & C_STRAND_SAME) ? sta + d.st_size : ib + d.o_st + me[9];
        const bool n1l = tiled && n1e - n1a + 32 <= ET_N1CAP, n2l = tiled && n2e - n2a + 32 <= ET_N2CAP, stl_ = tiled && ste - sta + 32 <= ET_STCAP;
        {
            const bool nextm = cur + cnt < re && tid <= ET_READS && cur + cnt + tid <= re;
            EMIT_META_VARS
            if (nextm) EMIT_META_LOAD(cur + cnt)
            // +1: 16 readable bytes in front
            const StageSpan sp[6] = { make_span(s_q4 + 1, qdec, qa, qe, qdec_bytes, tiled), make_span(s_s4 + 1, sdec, sa, se, sdec_bytes, tiled),
                                      make_span(s_mid4, R.mid, (uint64_t)g0 * 40, (uint64_t)g1 * 40, ~0ull >> 1, tiled),
                                      make_span(s_n14, img, n1a, n1e, img_bytes, n1l), make_span(s_n24, img, n2a, n2e, img_bytes, n2l), make_span(s_st4, img, sta, ste,
                                              img_bytes, stl_) };
            stage_spans6<(int)((ET_SCAP / 16 + 4 + 255) / 256)>(sp);                 // groups per thread at 256 threads
            if (nextm) EMIT_META_STORE(s_next)
        }
        __syncthreads();
        // ---- phase 4: compose the tile's text in LDS
        const uint8_t* q_l = (const uint8_t*)(s_q4 + 1) + (qa & 15ull); const uint8_t* s_l = (const uint8_t*)(s_s4 + 1) + (sa & 15ull);
        const uint8_t* m_l = (const uint8_t*)s_mid4 + (((uint64_t)g0 * 40) & 15ull);
        uint8_t* oA = (uint8_t*)s_out4 + (tp0.a & 15u); uint8_t* oB = (uint8_t*)s_out4 + ET_OCAP / 2 + (tp0.b & 15u);
        if (tiled && n1l && n2l && stl_) {
            // piece-parallel: for each kind of piece one flat loop over (read j, destination word k) - every thread copies whole words
            uint8_t* const out = (uint8_t*)s_out4;
            const uint32_t qoff = 16u + (uint32_t)(qa & 15ull), soff = 16u + (uint32_t)(sa & 15ull), moff = (uint32_t)(((uint64_t)g0 * 40) & 15ull);
            const uint32_t n1off = (uint32_t)(n1a & 15ull), n2off = (uint32_t)(n2a & 15ull), stoff = (uint32_t)(sta & 15ull);
            const uint32_t recA = (tp0.a & 15u) - tp0.a, recB = ET_OCAP / 2 + (tp0.b & 15u) - tp0.b;       // + at = LDS offset of a record
            // one thread = one PIECE of one read (8 slots x ET_READS reads; quality and sequence are cut in two halves of whole 16-byte
            // groups): the set-up (offsets, lengths, alignment) is paid once per piece and the copy itself is a short loop over aligned
            // 16-byte destination groups with the next group's source words already in flight
            const uint8_t* const pool = (const uint8_t*)s_src4;
            for (uint32_t slot = tid; slot < 8u * ET_READS; slot += blockDim.x) {
                const uint32_t rs_ = slot;
                // 0,1 quality halves; 2,3 sequence halves; 4 borrowed part; 5 name1; 6 middle + newlines; 7 name2 + strand
                const uint32_t j = rs_ % ET_READS, kind = rs_ / ET_READS;
                if (j >= cnt) continue;
                const uint32_t* m = s_meta + EM_ROW * j; const bool odd = ((cur + j) & 1u) != 0, to2 = split && odd, rc = il && odd;
                const uint32_t rec = (to2 ? recB : recA) + m[0], len = m[1], mid = m[11];
                if (kind == 6) {                                              // the four newlines; capacity check
                    const uint32_t e0 = m[10] - 1, e1 = e0 + 1 + len, e2 = e1 + 1 + m[6], e3 = e2 + 1 + len;
                    out[rec + e0] = '\n'; out[rec + e1] = '\n'; out[rec + e2] = '\n'; out[rec + e3] = '\n';
                    if ((uint64_t)m[0] + e3 + 1 > (to2 ? cap2 : cap1)) atomicOr(&st->err, 1u << 31);
                }
                for (int sub = 0; sub < (kind == 7 ? 2 : 1); sub++) {         // (slot 7 copies two short pieces)
                    uint32_t n, dst, src, qsrc = 0; bool rev = false; int pat = -1, half = -1;   // pat: piece offset of the byte to patch (name2)
                    const uint32_t qs = 16u * EG_Q + qoff + (m[15] - q0);
                    // quality (back to front for an RC mate)
                    if (kind <= 1) { n = len; dst = rec + m[10] + len + 1 + m[6] + 1; src = qs; rev = rc; half = (int)kind; }
                    else if (kind <= 4) {
                        // sequence: interleaved-orientation positions p in [0, xa) come from sA + p, p in [xa, len) from sB + (p - xa) (the
                        // part a negative overlap borrowed from the mate); an RC mate emits complemented, back to front
                        const int ov = (int)m[2]; const uint32_t xa = ov < 0 ? len - (uint32_t)(-ov) : len; const uint32_t sp = m[14] - s0;
                        const bool partb = kind == 4; const uint32_t p0 = partb ? xa : 0u;
                        n = partb ? len - xa : xa;
                        src = 16u * EG_S + soff + (partb ? sp - m[3] : (ov > 0 ? sp - (uint32_t)ov : sp));
                        dst = rec + m[10] + (rc ? len - p0 - n : p0); rev = rc; qsrc = qs + p0; half = partb ? -1 : (int)kind - 2;
                    }
                    else if (kind == 5) { n = m[4]; dst = rec; src = 16u * EG_N1 + n1off + ((fl & C_NAME1_SAME) ? 0u : m[7] - a7); }
                    else if (kind == 6) { n = mid; dst = rec + m[4]; src = 16u * EG_MID + moff + 40u * j; }
                    else if (sub == 0) { n = m[5]; dst = rec + m[4] + mid; src = 16u * EG_N2 + n2off + ((fl & C_NAME2_SAME) ? 0u : m[8] - a8);
                                         if ((fl & C_NAME2_SAME) && rc && dch != 0 && dpos < n) pat = (int)dpos; }      // the mate's differing character
                    else { n = m[6]; dst = rec + m[10] + len + 1; src = 16u * EG_ST + stoff + ((fl & C_STRAND_SAME) ? 0u : m[9] - a9); }
                    uint32_t gb = 0, ge = (n + 15u) >> 4;                       // the piece's 16-byte groups; a half takes the first / the second part
                    if (half == 0) ge = (ge + 1u) >> 1; else if (half == 1) gb = (ge + 1u) >> 1;
                    uint8_t* const o = out + dst;
                    // the copy loop, specialised for what the piece can need (a wave holds two kinds: the tests below are nearly wave-uniform)
                    if (kind <= 1) emit_copy<false, true, false>(o, pool, src, n, gb, ge, rev, 0u, false, nq, -1, dch);
                    else if (kind <= 4) emit_copy<true, true, false>(o, pool, src, n, gb, ge, rev, qsrc, implied_n, nq, -1, dch);
                    else if (pat < 0) emit_copy<false, false, false>(o, pool, src, n, gb, ge, false, 0u, false, nq, -1, dch);
                    else emit_copy<false, false, true>(o, pool, src, n, gb, ge, false, 0u, false, nq, pat, dch);
                }
            }
        } else
        for (uint32_t j = (uint32_t)wave_id(); j < cnt; j += wpb) {
            const uint32_t r = cur + j, g = g0 + j; const uint32_t* m = s_meta + EM_ROW * j;
            const bool odd = (r & 1u) != 0; const bool to2 = split && odd;
            EmitRead e;
            e.len = m[1]; e.ov = (int)m[2]; e.prevlen = m[3]; e.n1 = m[4]; e.n2 = m[5]; e.stl = m[6];
            const uint32_t o7 = (fl & C_NAME1_SAME) ? 0u : m[7], o8 = (fl & C_NAME2_SAME) ? 0u : m[8], o9 = (fl & C_STRAND_SAME) ? 0u : m[9];
            e.n1p = n1l ? (const uint8_t*)s_n14 + (n1a & 15ull) + (o7 - a7) : cp + d.o_n1 + o7;
            e.n2p = n2l ? (const uint8_t*)s_n24 + (n2a & 15ull) + (o8 - a8) : cp + d.o_n2 + o8;
            e.stp = stl_ ? (const uint8_t*)s_st4 + (sta & 15ull) + (o9 - a9) : cp + d.o_st + o9;
            e.rc = il && odd; e.patch = (fl & C_NAME2_SAME) && il && odd && dch != 0;
            const uint64_t at = m[0]; const uint64_t cap = to2 ? cap2 : cap1;
            const uint64_t total = (uint64_t)e.n1 + e.n2 + 1 + e.len + 1 + e.stl + 1 + e.len + 1;   // + mid below
            if (tiled) {
                e.mp = m_l + 40u * j; e.mid = e.mp[39];
                e.sp = m[14] - s0; e.qp = m[15] - q0;
                if (at + total + e.mid > cap) { if (l == 0) atomicOr(&st->err, 1u << 31); continue; }
                uint8_t* w = to2 ? oB + ((uint32_t)at - tp0.b) : oA + ((uint32_t)at - tp0.a);
                emit_one(w, e, s_l, q_l, implied_n, nq, dpos, dch, l);
            } else {
                e.mp = R.mid + (size_t)g * 40; e.mid = e.mp[39];
                e.sp = m[14]; e.qp = m[15];
                if (at + total + e.mid > cap) { if (l == 0) atomicOr(&st->err, 1u << 31); continue; }
                emit_one((to2 ? out2 : out1) + at, e, sdec + sg0, qdec + qg0, implied_n, nq, dpos, dch, l);
            }
        }
        __syncthreads();
        // ---- phase 5: aligned 16-byte stores of the finished tile (no barrier after it: three barriers precede the next compose)
        if (tiled) {
            if (tp1.a <= cap1) flush_span(s_out4, out1, tp0.a, tp1.a);
            if (split && tp1.b <= cap2) flush_span(s_out4 + ET_OCAP / 32, out2, tp0.b, tp1.b);
        }
        cur += cnt; pb ^= 1u;
    }
#undef EMIT_META_VARS
#undef EMIT_META_LOAD
#undef EMIT_META_STORE
}

// =============================================================== text emission, third formulation (no output tile)
// What k_gather2 showed on the encode side holds here: the tile kernels spend half their instructions on per-tile bookkeeping and their LDS on an
// output tile whose only purpose is an aligned flush.  k_dec_emit3:
//   * a tile is a fixed number K of reads (64 when reads are <= 160 bases; the host halves K until K reads fit the quality tile), four lanes per read;
//   * the text goes from the LDS sources straight to its place in the output with byte-granular 16-byte stores (the four lanes of a read write
//     neighbouring groups of the same line); no output tile, no flush pass, no fit test;
//   * the bases are never expanded to a byte tile: a lane takes 16 codes from the staged 2-bit stream at any bit offset (8-byte LDS read + shift),
//     reverses / complements them in 2-bit space, looks the letters up with v_perm_b32 and patches N from a bit tile the N list was scattered into;
//   * the quality group of the same 16 positions is in registers at that moment (same lane), which is all the implied-N rule needs.
// Everything else - the lists' exact entry ranges from the cell index, wave-per-list rounds, the next tile's metadata requested a tile ahead - carries over from the tile
// emitter this kernel replaced.
#define E3_QCAP 10240u            // quality tile: K reads' qualities (64 x 160)
#define E3_N1BIG 13312u           // name1 tile of the second instantiation: 64 per-read names of 200 bytes (34 KB of LDS, four workgroups per CU)
struct __attribute__((packed, aligned(1))) GU16d { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(1))) GU8d { uint32_t a, b; };
struct __attribute__((packed, aligned(1))) GU4d { uint32_t a; };
struct __attribute__((packed, aligned(1))) GU2d { uint16_t a; };
// n bytes LDS -> global at any alignment: 16-byte groups g0, g0 + gs, ... (the last one moved back to end exactly at n); n < 16: 8 + 4 + 2 + 1 by the lane with g0 == 0
__device__ __forceinline__ void e3_copy(uint8_t* __restrict__ dst, const uint8_t* src, uint32_t n, uint32_t g0, uint32_t gs, int pat, uint32_t dch) {
    if (n >= 16u) {
        const uint32_t ng = (n + 15u) >> 4;
        for (uint32_t g = g0; g < ng; g += gs) {
            uint32_t p0 = 16u * g; if (p0 + 16u > n) p0 = n - 16u;
            uint32_t w[4]; lds_get16(src, p0, w);
            if (pat >= (int)p0 && pat < (int)p0 + 16) { const int b = pat - (int)p0; const uint32_t sh = 8u * (uint32_t)(b & 3); uint32_t& x = w[b >> 2];
                    x = (x & ~(0xFFu << sh)) | ((dch & 0xFFu) << sh); }
            GU16d v; v.a = w[0]; v.b = w[1]; v.c = w[2]; v.d = w[3]; *(GU16d*)(dst + p0) = v;
        }
    } else if (g0 == 0 && n) {
        uint32_t w[4]; lds_get16(src, 0, w);
        if (pat >= 0 && pat < 16) { const uint32_t sh = 8u * (uint32_t)(pat & 3); uint32_t& x = w[pat >> 2]; x = (x & ~(0xFFu << sh)) | ((dch & 0xFFu) << sh); }
        uint8_t* q = dst;
        if (n & 8u) { GU8d v; v.a = w[0]; v.b = w[1]; *(GU8d*)q = v; q += 8; w[0] = w[2]; w[1] = w[3]; }
        if (n & 4u) { GU4d v; v.a = w[0]; *(GU4d*)q = v; q += 4; w[0] = w[1]; }
        if (n & 2u) { GU2d v; v.a = (uint16_t)w[0]; *(GU2d*)q = v; q += 2; w[0] >>= 16; }
        if (n & 1u) *q = (uint8_t)w[0];
    }
}
__device__ __forceinline__ uint32_t e3_rev2x16(uint32_t v) { v = bswap32(v); v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
        return ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2); }
__device__ __forceinline__ uint32_t e3_rev1x16(uint32_t v) {
    v = ((v >> 8) & 0xFFu) | ((v & 0xFFu) << 8); v = ((v >> 4) & 0x0F0Fu) | ((v & 0x0F0Fu) << 4);
    v = ((v >> 2) & 0x3333u) | ((v & 0x3333u) << 2); return ((v >> 1) & 0x5555u) | ((v & 0x5555u) << 1);
}
// dword i of the mask "bytes >= t of a 16-byte group" (t <= 0: all of them, t >= 16: none)
__device__ __forceinline__ uint32_t e3_from(int t, int i) { const int k = t - 4 * i; return k <= 0 ? 0xFFFFFFFFu : (k >= 4 ? 0u : 0xFFFFFFFFu << (8 * k)); }
// v_alignbyte_b32
__device__ __forceinline__ uint32_t e3_align(uint32_t hi, uint32_t lo, int bytes) { return (uint32_t)((((unsigned long long)hi << 32) | lo) >> (8 * bytes)); }
template <bool IMPL, uint32_t N1CAP = ET_N1CAP> __global__ void __launch_bounds__(256, N1CAP == ET_N1CAP ? 5 : 4) k_dec_emit3(const uint8_t* __restrict__ img,
        const DChunk* __restrict__ CH, const DevHeader* __restrict__ D, DReadTab R,
                           uint64_t img_bytes, int split, uint8_t* __restrict__ out1, uint64_t cap1, uint8_t* __restrict__ out2, uint64_t cap2, DecStatus* st,
                           const uint32_t* __restrict__ plist, const unsigned long long* __restrict__ loff, const uint32_t* __restrict__ nent, const uint32_t* __restrict__ cellidx,
                           uint32_t ncell, uint32_t nstr, uint32_t kshift) {
    __shared__ uint4 t_q4[E3_QCAP / 16 + 6];                                // quality tile (16 bytes of slack in front, the rest behind)
    __shared__ uint4 t_pk4[E3_QCAP / 64 + 6];                               // the tile's packed bases
    __shared__ uint32_t t_nb[E3_QCAP / 32 + 8];                             // one bit per stored base of the tile: is N
    // (N1CAP: E3_N1BIG for files with long per-read names)
    __shared__ uint4 t_mid4_[64 * 40 / 16 + 5], t_n14_[N1CAP / 16 + 5], t_n24_[ET_N2CAP / 16 + 5], t_st4[ET_STCAP / 16 + 4];
    // (16 readable bytes in front of each: a 16-byte group of the name line may start before a piece)
    uint4* const t_mid4 = t_mid4_ + 1; uint4* const t_n14 = t_n14_ + 1; uint4* const t_n24 = t_n24_ + 1;
    __shared__ unsigned long long s_loff[NPOS_SLOT + 2]; __shared__ uint32_t s_nent[NPOS_SLOT + 2], s_val[NPOS_SLOT + 2];
    __shared__ uint32_t s_g[2][NPOS_SLOT + 2], s_kb[2][NPOS_SLOT + 2];
    const uint32_t c = blockIdx.y; const DChunk d = CH[c]; const uint8_t* cp = img + d.off;
    const uint32_t fl = d.flags, hf = D->flags, f = d.rbase; const bool il = (fl & C_PE_INTERLEAVED) != 0;
    constexpr bool implied_n = IMPL;                                        // (the host instantiates by the header: N positions implied by the quality, or listed)
    const uint32_t nq4 = (D->n_base_qual & 0xFFu) * 0x01010101u, dpos = D->name2_diff_pos, dch = D->name2_diff_char;
    const int l = lane_id(), w = (int)uni32((uint32_t)wave_id()); const uint32_t tid = threadIdx.x;
    const U4 pv0 = R.pv[f]; const uint32_t pq0 = R.pq[f];
    const uint32_t K = 1u << kshift, pshift = 8u - kshift, P = 1u << pshift;
    uint32_t per = (d.reads + gridDim.x - 1) / gridDim.x; per = (per + K - 1u) & ~(K - 1u);                 // whole tiles per workgroup (K is even: pairs stay together)
    const uint32_t rs = blockIdx.x * per; const uint32_t re = rs + per < d.reads ? rs + per : d.reads;
    if (rs >= re) return;
    const bool raw = (hf & H_DONT_QUAL) != 0, bycol = !raw && (hf & H_QUAL_BY_COL);
    const uint32_t nn = bycol ? (D->n_normal < NPOS_SLOT ? D->n_normal : NPOS_SLOT) : 0u; const bool hasn = (hf & H_N_POS) != 0;
    const uint32_t T = nn + (hasn ? 1u : 0u);                                // streams of the tile: t < nn quality value t, t == nn the N positions
    const uint32_t major4 = (D->major & 0xFFu) * 0x01010101u;
    const uint32_t qlen_c = R.pq[f + d.reads] - pq0, slen_c = R.pv[f + d.reads].d - pv0.d;      // qualities / stored bases of the chunk
    const bool same1 = (fl & C_NAME1_SAME) != 0, same2 = (fl & C_NAME2_SAME) != 0, same3 = (fl & C_STRAND_SAME) != 0;
    if (tid < T) { const uint32_t jj = tid < nn ? tid : D->n_normal; const size_t t_ = (size_t)c * nstr + jj; s_loff[tid] = loff[t_]; s_nent[tid] = nent[t_];
            s_val[tid] = tid < nn ? (uint32_t)D->normal[tid] : (uint32_t)'N'; }
    // exception records behind the streams (src/rfqcodec.cpp:1034-1043)
    uint32_t nrec = 0; const uint8_t* xrec = nullptr;
    if (bycol && 4ull * D->n_normal <= d.qual_size) {
        const uint8_t* qp = cp + d.o_qual; uint64_t off = 4ull * D->n_normal;
        for (uint32_t i = 0; i < D->n_normal; i++) off += ld_u32(qp + 4 * i);
        off = uni64(off);
        if (off <= d.qual_size) { nrec = (uint32_t)((d.qual_size - off) / 5); xrec = qp + off; }
    }
    auto cell_lookup = [&](uint32_t t, uint32_t qpos, uint32_t spos, bool bound) -> uint32_t {
        const uint32_t jj = t < nn ? t : D->n_normal; uint32_t cell = ((t < nn ? qpos : spos) + (bound ? E3_QCAP : 0u)) / POS2_CELL + (bound ? 1u : 0u);
        if (cell >= ncell) return bound ? 0xFFFFFFFFu : cellidx[((size_t)c * nstr + jj) * ncell + ncell - 1u];
        return cellidx[((size_t)c * nstr + jj) * ncell + cell];
    };
    // a tile's uniform parameters: quality / stored-base / name-piece prefixes at its first read and behind its last
    struct TileP { uint32_t q0, q1, s0, s1, a7, a8, a9, e7, e8, e9; };
    auto tile_params = [&](uint32_t r0, uint32_t r1) -> TileP {
        TileP t; const U4 a = R.pv[f + r0], b = R.pv[f + r1];
        t.q0 = uni32(R.pq[f + r0]) - pq0; t.q1 = uni32(R.pq[f + r1]) - pq0; t.s0 = uni32(a.d) - pv0.d; t.s1 = uni32(b.d) - pv0.d;
        t.a7 = uni32(a.a) - pv0.a; t.a8 = uni32(a.b) - pv0.b; t.a9 = uni32(a.c) - pv0.c; t.e7 = uni32(b.a) - pv0.a; t.e8 = uni32(b.b) - pv0.b; t.e9 = uni32(b.c) - pv0.c;
        return t;
    };
    uint32_t cur = rs, pb = 0;
    TileP tp_cur = tile_params(cur, cur + K < re ? cur + K : re);
    if (tid < T) { s_g[0][tid] = cell_lookup(tid, tp_cur.q0, tp_cur.s0, false); s_kb[0][tid] = cell_lookup(tid, tp_cur.q0, tp_cur.s0, true); }
    {   // pieces every read of the chunk shares: staged once
        const uint64_t ib = d.off;
        if (same1) span_dma<1>(make_span(t_n14, img, ib + d.o_n1, ib + d.o_n1 + d.n1_size, img_bytes, true));
        if (same2) span_dma<1>(make_span(t_n24, img, ib + d.o_n2, ib + d.o_n2 + d.n2_size, img_bytes, true));
        if (same3) span_dma<1>(make_span(t_st4, img, ib + d.o_st, ib + d.o_st + d.st_size, img_bytes, true));
    }
    // thread -> (read of the tile, part of it): the even reads first, then the odd ones - an interleaved chunk's mates are written back to front and
    // complemented, their R1 as stored, and a wave that holds both runs both paths
    const uint32_t jj = tid >> pshift, j = ((jj << 1) & (K - 1u)) | (jj >> (kshift - 1u)), part = tid & (P - 1u);
    __syncthreads();
    while (cur < re) {                                                       // block-uniform
        const uint32_t cnt = re - cur < K ? re - cur : K, g0 = f + cur, g1 = g0 + cnt;
        const TileP tp = tp_cur; const uint32_t q0 = tp.q0, q1 = tp.q1, s0 = tp.s0, s1 = tp.s1;
        const bool fits = q1 - q0 <= E3_QCAP && (same1 || tp.e7 - tp.a7 + 32u <= N1CAP) && (same2 || tp.e8 - tp.a8 + 32u <= ET_N2CAP) && (same3 || tp.e9 - tp.a9 + 32u <= ET_STCAP);
        // (the host sizes K by the longest read and by the chunks' average piece sizes; a tile whose pieces are longer than that allowed for: the host repeats the range
        // with k_dec_emit2)
        if (!fits) { if (tid == 0) atomicOr(&st->err, (uint32_t)DE_E3_RETRY); break; }
        // ---- stage: packed bases, middles, per-read name pieces (LDS-DMA); raw qualities for DONT_ENCODE_QUAL files
        const uint64_t ib = d.off;
        const uint64_t n1a = ib + d.o_n1 + (same1 ? 0u : tp.a7), n1e = same1 ? n1a + d.n1_size : ib + d.o_n1 + tp.e7;
        const uint64_t n2a = ib + d.o_n2 + (same2 ? 0u : tp.a8), n2e = same2 ? n2a + d.n2_size : ib + d.o_n2 + tp.e8;
        const uint64_t sta = ib + d.o_st + (same3 ? 0u : tp.a9), ste = same3 ? sta + d.st_size : ib + d.o_st + tp.e9;
        uint64_t pka = ib + d.o_seq + (s0 >> 2), pke = ib + d.o_seq + ((s1 + 3u) >> 2); { const uint64_t pend = ib + d.o_seq + d.seq_size; if (pke > pend) pke = pend;
                if (pka > pke) pka = pke; }
        uint64_t rqa = ib + d.o_qual + q0, rqe = ib + d.o_qual + q1; { const uint64_t qend = ib + d.o_qual + d.qual_size; if (rqe > qend) rqe = qend;
                if (rqa > rqe) rqa = rqe; }
        if (raw) span_dma<(int)((E3_QCAP / 16 + 4 + 255) / 256)>(make_span(t_q4 + 1, img, rqa, rqe, img_bytes, true));
        if (w == 0) span_dma_wave<(int)((E3_QCAP / 64 + 4 + 63) / 64)>(make_span(t_pk4, img, pka, pke, img_bytes, true), l);
        else if (w == 1) span_dma_wave<(int)((64 * 40 / 16 + 4 + 63) / 64)>(make_span(t_mid4, R.mid, (uint64_t)g0 * 40, (uint64_t)g1 * 40, ~0ull >> 1, true), l);
        else if (w == 2) { if (!same1) span_dma_wave<(int)((N1CAP / 16 + 4 + 63) / 64)>(make_span(t_n14, img, n1a, n1e, img_bytes, true), l); }
        else { if (!same2) span_dma_wave<(int)((ET_N2CAP / 16 + 4 + 63) / 64)>(make_span(t_n24, img, n2a, n2e, img_bytes, true), l);
               if (!same3) span_dma_wave<(int)((ET_STCAP / 16 + 4 + 63) / 64)>(make_span(t_st4, img, sta, ste, img_bytes, true), l); }
        // qualities start as the major value (src/rfqcodec.cpp:1089), the N bits as none
        if (bycol) { uint4* qt = t_q4 + 1; const uint32_t ng = (q1 - q0 + 15u) >> 4;
                for (uint32_t i = tid; i < ng; i += blockDim.x) qt[i] = make_uint4(major4, major4, major4, major4); }
        for (uint32_t i = tid; i < ((s1 - s0 + 31u) >> 5) + 1u; i += blockDim.x) t_nb[i] = 0;
        // ---- the tile's list entries, requested now and scattered after the barrier (wave w takes the lists w, w + 4, ...)
        uint32_t ls_t = (uint32_t)w, ls_base = 0, ls_k0 = 0, ls_ke = 0, ls_val = 0; const uint32_t* ls_p = plist;
        auto ls_open = [&](uint32_t t_) {
            const uint32_t g_ = uni32(s_g[pb][t_]), b_ = uni32(s_kb[pb][t_]), n_ = uni32(s_nent[t_]);
            ls_k0 = g_ == 0xFFFFFFFFu ? 0u : g_; ls_ke = g_ == 0xFFFFFFFFu ? 0u : (b_ < n_ ? b_ : n_); ls_base = 0;
            ls_val = uni32(s_val[t_]); ls_p = plist + uni64(s_loff[t_]);
        };
        auto ls_round = [&](uint32_t& e_, uint32_t& v_) {
            while (ls_t < nn && ls_k0 + ls_base >= ls_ke) { ls_t += 4u; if (ls_t < nn) ls_open(ls_t); }
            if (ls_t < nn) { const uint32_t kk = ls_k0 + ls_base + (uint32_t)l; if (kk < ls_ke) e_ = ls_p[kk]; v_ = ls_val; ls_base += 64u; }
        };
        if (ls_t < nn) ls_open(ls_t); else ls_t = nn;
        uint32_t fe[8], fv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { fe[i] = 0xFFFFFFFFu; fv[i] = 0; ls_round(fe[i], fv[i]); }
        uint32_t pn[2] = { 0xFFFFFFFFu, 0xFFFFFFFFu }; uint32_t nk0 = 0xFFFFFFFFu, nke = 0;
        if (hasn) {
            nk0 = s_g[pb][nn]; nke = s_kb[pb][nn]; if (nke > s_nent[nn]) nke = s_nent[nn];
            const uint32_t* lp = plist + s_loff[nn];
#pragma unroll
            for (int i = 0; i < 2; i++) { const uint32_t kk = nk0 + tid + 256u * (uint32_t)i; if (nk0 != 0xFFFFFFFFu && kk < nke) pn[i] = lp[kk]; }
        }
        // ---- my read (P lanes share one)
        const uint32_t r = cur + j; const bool on = j < cnt; const bool odd = (r & 1u) != 0, rc = il && odd, to2 = split && odd;
        uint32_t len = 0, n1 = 0, n2 = 0, sl = 0, md = 0, prevlen = 0, sp = 0, qp_ = 0, o7 = 0, o8 = 0, o9 = 0; int ov = 0; uint32_t toff = 0;
        if (on) {
            const uint32_t g_ = f + r; const U4 t4 = R.tp[g_], p4 = R.pv[g_];
            toff = to2 ? t4.b : t4.a; sp = p4.d - pv0.d - s0; qp_ = R.pq[g_] - pq0 - q0; o7 = p4.a - pv0.a - tp.a7; o8 = p4.b - pv0.b - tp.a8; o9 = p4.c - pv0.c - tp.a9;
            len = R.len[g_]; ov = R.ov[g_]; prevlen = odd ? R.len[g_ - 1] : 0u;
            n1 = cp[d.o_n1lens + ((fl & C_NAME1_LEN_SAME) ? 0u : r)]; n2 = (hf & H_NAME2) ? cp[d.o_n2lens + ((fl & C_NAME2_LEN_SAME) ? 0u : r)] : 0u;
            sl = cp[d.o_stlens + ((fl & C_STRAND_LEN_SAME) ? 0u : r)]; md = R.mid[(size_t)g_ * 40 + 39];
        }
        // ---- the next tile's parameters (consumed a tile from now)
        const uint32_t nxt = cur + cnt; TileP tp_n = tp;
        if (nxt < re) tp_n = tile_params(nxt, nxt + K < re ? nxt + K : re);
        __syncthreads();
        uint8_t* const q_t = (uint8_t*)(t_q4 + 1) + (raw ? (uint32_t)(rqa & 15ull) : 0u);              // quality of chunk position q0 + i at q_t[i]
        const uint8_t* const pk = (const uint8_t*)t_pk4 + (uint32_t)(pka & 15ull);                     // packed byte (s0 >> 2) + i at pk[i]
        const uint32_t have = (uint32_t)(pke - pka), sbit0 = 2u * (s0 & 3u);                              // staged packed bytes; bit offset of stored base s0 in pk
        // ---- quality lists, exception records, N list into the tiles
        {
#pragma unroll
            for (int i = 0; i < 8; i++) { const uint32_t p = fe[i]; if (p >= q0 && p < q1) q_t[p - q0] = (uint8_t)fv[i]; }
            while (ls_t < nn) {
                uint32_t e_[4], v_[4];
#pragma unroll
                for (int i = 0; i < 4; i++) { e_[i] = 0xFFFFFFFFu; v_[i] = 0; ls_round(e_[i], v_[i]); }
#pragma unroll
                for (int i = 0; i < 4; i++) { const uint32_t p = e_[i]; if (p >= q0 && p < q1) q_t[p - q0] = (uint8_t)v_[i]; }
            }
            if (nrec) for (uint32_t i = tid; i < nrec; i += blockDim.x) { const uint8_t* rr = xrec + 5ull * i; const uint32_t pos = ld_u32(rr + 1);
                    if (pos >= q0 && pos < q1 && pos < qlen_c) q_t[pos - q0] = rr[0]; }
            if (hasn) {
                const uint32_t send = s1 < slen_c ? s1 : slen_c;
#pragma unroll
                for (int i = 0; i < 2; i++) { const uint32_t p = pn[i]; if (p >= s0 && p < send) atomicOr(&t_nb[(p - s0) >> 5], 1u << ((p - s0) & 31u)); }
                if (nk0 != 0xFFFFFFFFu && nk0 + 512u < nke) { const uint32_t* lp = plist + s_loff[nn];
                        for (uint32_t kk = nk0 + 512u + tid; kk < nke; kk += 256u) { const uint32_t p = lp[kk];
                        if (p >= s0 && p < send) atomicOr(&t_nb[(p - s0) >> 5], 1u << ((p - s0) & 31u)); } }
            }
        }
        // the next tile's list cells (its parameters have come back by now)
        if (nxt < re && tid < T) { s_g[pb ^ 1u][tid] = cell_lookup(tid, tp_n.q0, tp_n.s0, false); s_kb[pb ^ 1u][tid] = cell_lookup(tid, tp_n.q0, tp_n.s0, true); }
        __syncthreads();
        // ---- compose: my share of my read's four lines, straight to the output (src/rfqcodec.cpp:1141-1254, Read::toString src/read.cpp:170)
        if (on) {
            uint8_t* const rec = (to2 ? out2 : out1) + toff; const uint64_t capo = to2 ? cap2 : cap1;
            const uint32_t e0 = n1 + md + n2, oseq = e0 + 1u, ost = oseq + len + 1u, oq = ost + sl + 1u, total = oq + len + 1u;
            if ((uint64_t)toff + total > capo) { if (part == 0) atomicOr(&st->err, 1u << 31); }
            else {
                // The name line = name1 + middle + name2 + '\n' (three LDS pieces at arbitrary byte offsets): a lane builds a whole 16-byte group of the line
                // in registers - one unaligned 16-byte LDS read per piece the group touches, later pieces laid over the earlier ones from their first byte
                // on - and stores it once.  (Piece by piece this was ~10 partial stores per read: 1.4 ms of the kernel's 5.1 on 2 x 4 GB.)
                const uint32_t L = e0 + 1u; const bool nfast = L >= 16u, jfast = sl == 1u && len >= 16u;
                const uint8_t* const src1 = (const uint8_t*)t_n14 + (uint32_t)(n1a & 15ull) + (same1 ? 0u : o7);
                const uint8_t* const src2 = (const uint8_t*)t_mid4 + (uint32_t)(((uint64_t)g0 * 40) & 15ull) + 40u * j;
                const uint8_t* const src3 = (const uint8_t*)t_n24 + (uint32_t)(n2a & 15ull) + (same2 ? 0u : o8);
                const uint8_t* const src4 = (const uint8_t*)t_st4 + (uint32_t)(sta & 15ull) + (same3 ? 0u : o9);
                const int pat2 = (same2 && rc && dch != 0 && dpos < n2) ? (int)dpos : -1;
                if (nfast) {
                    const uint32_t ngl = (L + 15u) >> 4;
                    for (uint32_t gi = part; gi < ngl; gi += P) {
                        uint32_t p0 = 16u * gi; if (p0 + 16u > L) p0 = L - 16u;
                        const int t1 = (int)n1 - (int)p0, t2 = t1 + (int)md, t3 = t2 + (int)n2;          // where the middle, name2 and the '\n' start in this group
                        uint32_t w[4] = { 0, 0, 0, 0 }, x[4];
                        if (t1 > 0) lds_get16(src1 + p0, 0, w);
                        if (t1 < 16 && t2 > 0 && md) { lds_get16(src2 - t1, 0, x);
#pragma unroll
                            for (int i = 0; i < 4; i++) { const uint32_t m = e3_from(t1, i); w[i] = (w[i] & ~m) | (x[i] & m); } }
                        if (t2 < 16 && t3 > 0 && n2) { lds_get16(src3 - t2, 0, x);
                            if (pat2 >= 0) { const int b = t2 + pat2; if (b >= 0 && b < 16) { const uint32_t sh = 8u * (uint32_t)(b & 3); uint32_t& y = x[b >> 2];
                                    y = (y & ~(0xFFu << sh)) | ((dch & 0xFFu) << sh); } }
#pragma unroll
                            for (int i = 0; i < 4; i++) { const uint32_t m = e3_from(t2, i); w[i] = (w[i] & ~m) | (x[i] & m); } }
                        if (t3 == 15) w[3] = (w[3] & 0x00FFFFFFu) | 0x0A000000u;                           // (the line's last byte, in its last group only)
                        GU16d v; v.a = w[0]; v.b = w[1]; v.c = w[2]; v.d = w[3]; *(GU16d*)(rec + p0) = v;
                    }
                } else {
                    if (part == 0) e3_copy(rec, src1, n1, 0, 1, -1, 0);
                    else if (part == 1 % P) e3_copy(rec + n1, src2, md, 0, 1, -1, 0);
                    if (part == 2 % P) e3_copy(rec + n1 + md, src3, n2, 0, 1, pat2, dch);
                    if (part == 3 % P) rec[e0] = '\n';
                }
                // "\n" + strand + "\n" behind the bases and the '\n' behind the qualities ride on the last 16-byte stores of those lines when the strand line is
                // one character (below); otherwise they are written here
                if (!jfast && part == 3 % P) { e3_copy(rec + ost, src4, sl, 0, 1, -1, 0); rec[ost - 1u] = '\n'; rec[oq - 1u] = '\n'; rec[total - 1u] = '\n'; }
                // bases and qualities, 16 positions per step.  I = the read in interleaved orientation: I[p] = stored[A + p] for p < xa, stored[Bs + p - xa] behind
                // (the part of a mate that overlaps R1 is R1's: src/rfqcodec.cpp:865-897); the output is I, or its reverse complement for an interleaved chunk's mate
                const uint32_t xa = ov < 0 ? len - (uint32_t)(-ov) : len; const uint32_t A = ov > 0 ? sp - (uint32_t)ov : sp, Bs = sp - prevlen;
                auto fetch = [&](uint32_t si, uint32_t& cw, uint32_t& nw) {    // 16 codes / N bits from tile-relative stored index si on
                    const uint32_t bit = sbit0 + 2u * si, byte = bit >> 3; const unsigned long long v = lds_get8(pk, byte);
                    cw = (uint32_t)(v >> (bit & 7u));
                    const uint32_t nb_ = lds_get4((const uint8_t*)t_nb, (si >> 3)) >> (si & 7u); nw = nb_ & 0xFFFFu;
                    // bases past the packed buffer read as N (the reference's 'N' prefill)
                    if (byte + 5u > have) { uint32_t lim_ = 4u * have > sbit0 / 2u + si ? 4u * have - sbit0 / 2u - si : 0u;
                            if (lim_ < 16u) nw |= (0xFFFFu << lim_) & 0xFFFFu; }
                };
                auto group = [&](uint32_t k0, uint32_t (&qw)[4], uint32_t (&sw)[4]) {   // output positions [k0, k0 + 16) of both lines (k0 + 16 <= len)
                    const uint32_t pa = rc ? len - k0 - 16u : k0;
                    lds_get16(q_t, qp_ + pa, qw);
                    uint32_t cw, nw;
                    if (pa + 16u <= xa) fetch(A + pa, cw, nw);
                    else if (pa >= xa) fetch(Bs + (pa - xa), cw, nw);
                    else { uint32_t c2, n2_; const uint32_t t1 = xa - pa; fetch(A + pa, cw, nw); fetch(Bs, c2, n2_);
                            cw = (cw & ((1u << (2u * t1)) - 1u)) | (c2 << (2u * t1)); nw = (nw & ((1u << t1) - 1u)) | ((n2_ << t1) & 0xFFFFu); }
                    if (rc) { const uint32_t x0 = bswap32(qw[3]), x1 = bswap32(qw[2]), x2 = bswap32(qw[1]), x3 = bswap32(qw[0]); qw[0] = x0; qw[1] = x1; qw[2] = x2;
                            qw[3] = x3;
                              cw = ~e3_rev2x16(cw); if (nw) nw = e3_rev1x16(nw); }
#pragma unroll
                    for (int i = 0; i < 4; i++) { const uint32_t b = (cw >> (8 * i)) & 0xFFu, y = (b | (b << 12)) & 0x000F000Fu, idx = (y | (y << 6)) & 0x03030303u;
                            sw[i] = __builtin_amdgcn_perm(0u, 0x43544147u, idx); }
                    if (nw) {                                                   // (rare: an N among the 16)
#pragma unroll
                        for (int i = 0; i < 4; i++) { const uint32_t mk = ((((nw >> (4 * i)) & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu;
                                sw[i] = (sw[i] & ~mk) | (0x4E4E4E4Eu & mk); }
                    }
                    if (implied_n) {
#pragma unroll
                        for (int i = 0; i < 4; i++) { const uint32_t mk = eq_bytes_full(qw[i], nq4); sw[i] = (sw[i] & ~mk) | (0x4E4E4E4Eu & mk); }
                    }
                };
                auto put = [&](uint32_t at_q, uint32_t at_s, const uint32_t (&qw)[4], const uint32_t (&sw)[4], bool both) {
                    GU16d v;
                    if (both) { v.a = qw[0]; v.b = qw[1]; v.c = qw[2]; v.d = qw[3]; *(GU16d*)(rec + at_q) = v; }
                    v.a = sw[0]; v.b = sw[1]; v.c = sw[2]; v.d = sw[3]; *(GU16d*)(rec + at_s) = v;
                };
                if (len >= 16u) {
                    const uint32_t nfull = len >> 4, rem = len & 15u; uint32_t qw[4], sw[4];
                    for (uint32_t gi = part; gi < nfull; gi += P) { group(16u * gi, qw, sw); put(oq + 16u * gi, oseq + 16u * gi, qw, sw, true); }
                    if ((nfull & (P - 1u)) == part && (rem || jfast)) {       // the lines' tails: positions [len - 16, len)
                        group(len - 16u, qw, sw);
                        if (!jfast) put(oq + len - 16u, oseq + len - 16u, qw, sw, true);
                        else {
                            if (rem > 13u) put(0u, oseq + len - 16u, qw, sw, false);                       // (the shifted store below starts behind position 16 * nfull)
                            const uint32_t jd = 0x000A000Au | ((uint32_t)src4[0] << 8);                    // '\n', the strand character, '\n'
                            uint32_t qs[4], ss[4];
                            ss[0] = e3_align(sw[1], sw[0], 3); ss[1] = e3_align(sw[2], sw[1], 3); ss[2] = e3_align(sw[3], sw[2], 3); ss[3] = e3_align(jd, sw[3], 3);
                            qs[0] = e3_align(qw[1], qw[0], 1); qs[1] = e3_align(qw[2], qw[1], 1); qs[2] = e3_align(qw[3], qw[2], 1); qs[3] = e3_align(0x0Au, qw[3], 1);
                            put(oq + len - 15u, oseq + len - 13u, qs, ss, true);
                        }
                    }
                } else if (part == 0) {
                    for (uint32_t k = 0; k < len; k++) {                      // a read of < 16 bases: byte by byte
                        const uint32_t p = rc ? len - 1u - k : k, si = p < xa ? A + p : Bs + (p - xa);
                        const uint32_t bit = sbit0 + 2u * si, byte = bit >> 3; const uint32_t code = byte < have ? (pk[byte] >> (bit & 7u)) & 3u : 0u;
                        const bool isn = byte >= have || ((t_nb[si >> 5] >> (si & 31u)) & 1u);
                        const uint8_t q = q_t[qp_ + p]; uint8_t b = isn ? (uint8_t)'N' : (uint8_t)("GATC"[rc ? 3u - code : code]);
                        if (implied_n && q == (uint8_t)(nq4 & 0xFFu)) b = 'N';
                        rec[oseq + k] = b; rec[oq + k] = q;
                    }
                }
            }
        }
        __syncthreads();                                                    // (the tiles are rewritten by the next round)
        cur += cnt; pb ^= 1u; tp_cur = tp_n;
    }
}
