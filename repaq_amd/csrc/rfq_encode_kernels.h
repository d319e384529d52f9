// rfq_encode_kernels.h — gfx950 kernels of the FASTQ -> RFQ encode path (included by rfq_encode.hip only).
//
// Pipeline (one batch = whole FASTQ stream(s) resident in HBM; reference call sites in brackets):
//   index    k_nl_bitmap -> scan -> k_line_offsets            newline bitmap, global line ranks, line-start table
//            [FastqReader::getLine/read, src/fastqreader.cpp:94-196, for '\n'-terminated text]
//   table    k_read_table                                     per-read lengths + FastqMeta::parse [src/fastqmeta.cpp:22-80]
//   cut      scan + k_partition                               chunk boundaries [Repaq::compress, src/repaq.cpp:546-553]
//   header   k_hdr_*                                          RfqCodec::makeHeader + makeQualityTable on chunk 0
//   chunk    k_chunk_flags, k_overlap, scans, k_gather, k_stream_plan, k_pos_coder, k_coords, k_chunk_layout,
//            k_assemble*                                      RfqCodec::encodeChunk + RfqChunk::write
#pragma once
#include "rfq_common.h"

struct Text {                    // the FASTQ streams of a batch and their line tables
    const uint8_t* fq[2];
    uint32_t n[2];
    const uint32_t* lo[2];       // lo[s][i] = start of line i; lo[s][i+1]-1 = its terminator (virtual at n for an unterminated tail)
    const uint32_t* ot[2];       // normalised text only (else null): offset of line i's terminator in the caller's text
    int paired;                  // RFQ_SE / RFQ_PE_TWO_FILES / RFQ_PE_INTERLEAVED
    uint32_t n_reads;            // reads in interleaved order (PE: 2 * pairs)
    uint32_t upr;                // reads per partition unit (1 SE, 2 PE)
};
// element s of the two-entry arrays above for a stream index that is only known per lane: a select between two kernel arguments (indexing
// the argument struct dynamically makes the compiler fetch the pointer from memory - a dependent load in front of every access)
__device__ __forceinline__ const uint8_t* t_fq(const Text& T, int s) { return s ? T.fq[1] : T.fq[0]; }
__device__ __forceinline__ const uint32_t* t_lo(const Text& T, int s) { return s ? T.lo[1] : T.lo[0]; }
__device__ __forceinline__ const uint32_t* t_ot(const Text& T, int s) { return s ? T.ot[1] : T.ot[0]; }
__device__ __forceinline__ uint32_t t_n(const Text& T, int s) { return s ? T.n[1] : T.n[0]; }
__device__ __forceinline__ void read_loc(const Text& T, uint32_t g, int& s, uint32_t& r) {
    if (T.paired == 1) { s = (int)(g & 1u); r = g >> 1; } else { s = 0; r = g; }
}
__device__ __forceinline__ uint32_t line_beg(const Text& T, uint32_t g, int k) { int s; uint32_t r; read_loc(T, g, s, r); return t_lo(T, s)[4 * (size_t)r + k]; }
__device__ __forceinline__ uint32_t line_len(const Text& T, uint32_t g, int k) { int s; uint32_t r; read_loc(T, g, s, r);
        const uint32_t* p = t_lo(T, s) + 4 * (size_t)r + k; return p[1] - 1 - p[0]; }
__device__ __forceinline__ const uint8_t* line_ptr(const Text& T, uint32_t g, int k) { int s; uint32_t r; read_loc(T, g, s, r);
        return t_fq(T, s) + t_lo(T, s)[4 * (size_t)r + k]; }

struct ReadTab {                 // per-read arrays, indexed by g (interleaved order)
    uint32_t* len;               // sequence length
    uint32_t* name1_len;
    uint32_t* name2_off;         // name2 = name[name2_off, name_len)
    uint32_t* x; uint32_t* y;
    uint16_t* tile; uint8_t* lane; uint8_t* ok;
    uint32_t* chunk;             // chunk id
    uint32_t* stored;            // bases kept in the sequence stream (after overlap trimming)
    uint8_t*  eq2;               // name2 == name2 of the chunk's read 0
    uint32_t* pq;                // exclusive prefix of len          (n_reads + 1 entries)
    U4*       pv;                // exclusive prefix of (name1_len, name2_len, strand_len, stored): only differences inside one chunk are ever used (pv[g] - pv[first[c]];
                                 // the chunk's totals: ChunkTab::ptot) - the tile path restarts it at 0 in every chunk (k_chunk_prefix), the byte-wise path scans the
                                 // whole batch
};

struct ChunkTab {                // per-chunk arrays
    uint32_t* first;             // first read of chunk c; first[n_chunks] = end
    uint32_t* flags;             // RfqChunk::mFlags (without line-break bits)
    uint32_t* il;                // final canBePeInterleaved
    uint32_t* ncount;            // 'N' bases in the stored sequence
    uint32_t* nmap;              // [c][NMAP_WORDS] bit b set: the chunk's stored bases contain an 'N' in 4096-base steps [b << shift, (b+1) << shift)
    uint32_t* scap;              // [c][MAX_STREAMS] scratch capacity of each stream
    uint64_t* soff;              // [c][MAX_STREAMS] scratch offset of each stream
    uint32_t* ssize;             // [c][MAX_STREAMS] bytes written by the stream coder
    uint32_t* xsize; uint32_t* ysize;
    uint64_t* qbase; uint64_t* sbase;   // 64-byte aligned bases of the chunk in qcat / scat
    uint64_t* img_size;          // bytes of the chunk image
    uint64_t* img_off;           // exclusive prefix (n_chunks + 1)
    U4*       ptot;              // the chunk's totals of (name1_len, name2_len, strand_len, stored): ReadTab::pv[g] - pv[first] is read g's offset inside the chunk
};

// =============================================================== index
// 64 bytes per lane -> one u64 newline mask; 256 lanes = 16 KiB per workgroup.
__device__ __forceinline__ uint32_t eq_mask4(uint32_t w, uint32_t pat) {   // bit k set iff byte k of w == pat byte
    uint32_t v = w ^ pat;
    uint32_t t = ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);   // 0x80 in every zero byte, exact
    return (((t >> 7) * 0x00204081u) >> 21) & 0xFu;
}
// 0x80 in every byte of w that equals the pattern byte, exact (no borrow between bytes)
__device__ __forceinline__ uint32_t eq_flags4(uint32_t w, uint32_t pat) { const uint32_t v = w ^ pat; return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu); }
// bits 0-7 = bytes of (a, b) that equal the pattern byte: the flags of the two words share one shift-or cascade (a's in the low nibble of every
// byte, b's in the high one), no multiply (v_mul_lo_u32 runs at a quarter of the rate)
__device__ __forceinline__ uint32_t eq_mask8(uint32_t a, uint32_t b, uint32_t pat) {
    uint32_t x = (eq_flags4(a, pat) >> 7) | (eq_flags4(b, pat) >> 3);
    x |= x >> 7; x |= x >> 14;
    return x & 0xFFu;
}
__device__ __forceinline__ uint32_t eq_mask16c(const uint4& q, uint32_t pat) { return eq_mask8(q.x, q.y, pat) | (eq_mask8(q.z, q.w, pat) << 8); }
// non-zero iff some byte of q equals the pattern byte (which one is not told: a borrow may flag the byte above a match as well)
__device__ __forceinline__ uint32_t has_byte16(const uint4& q, uint32_t pat) {
    const uint32_t a = q.x ^ pat, b = q.y ^ pat, c = q.z ^ pat, d = q.w ^ pat;
    return (((a - 0x01010101u) & ~a) | ((b - 0x01010101u) & ~b) | ((c - 0x01010101u) & ~c) | ((d - 0x01010101u) & ~d)) & 0x80808080u;
}
__device__ __forceinline__ uint32_t eq_mask16(const uint4& q, uint32_t pat) {
    return eq_mask4(q.x, pat) | (eq_mask4(q.y, pat) << 4) | (eq_mask4(q.z, pat) << 8) | (eq_mask4(q.w, pat) << 12);
}
// skip (< 16): leading bytes of the stream that do not belong to it (the stream starts at an unaligned address inside a larger text: the
// pointer was rounded down to 16 bytes); they hold no line end and line 0 starts behind them.
__global__ void k_nl_bitmap(const uint8_t* __restrict__ fq, uint32_t n, uint32_t skip, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ blkcnt, DevStatus* st) {
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t base = w * 64;
    uint64_t m = 0, crm = 0; uint32_t cr = 0;
    if (base + 64 <= n) {
        const uint4* p = (const uint4*)(fq + base);
#pragma unroll
        for (int k = 0; k < 4; k++) { uint4 q = p[k]; m |= (uint64_t)eq_mask16(q, 0x0A0A0A0Au) << (16 * k); crm |= (uint64_t)eq_mask16(q, 0x0D0D0D0Du) << (16 * k); }
    } else if (base < n) {
        for (uint32_t i = 0; i < 64 && base + i < n; i++) { uint8_t c = fq[base + i]; if (c == '\n') m |= 1ull << i; if (c == '\r') crm |= 1ull << i; }
    }
    if (w == 0 && skip) { const uint64_t keep = ~((1ull << skip) - 1ull); m &= keep; crm &= keep; }
    cr = crm != 0;
    if (base < n) bitmap[w] = m;
    uint32_t tot; (void)block_excl_sum<uint32_t>((uint32_t)__popcll(m), &tot);
    if (threadIdx.x == 0) blkcnt[blockIdx.x] = tot;
    if (__any(cr != 0) && lane_id() == 0) atomicOr(&st->err, (uint32_t)DE_HAS_CR);
}
// lo[rank+1] = position after the rank-th newline; lo[0] = 0.
__global__ void k_line_offsets(const uint64_t* __restrict__ bitmap, const uint32_t* __restrict__ blkbase, uint32_t n, uint32_t skip, uint32_t* __restrict__ lo) {
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t base = w * 64;
    uint64_t m = base < n ? bitmap[w] : 0ull;
    uint32_t ex = block_excl_sum<uint32_t>((uint32_t)__popcll(m), (uint32_t*)nullptr);
    uint32_t rank = blkbase[blockIdx.x] + ex;
    while (m) { int b = __ffsll((long long)m) - 1; m &= m - 1; lo[rank + 1] = (uint32_t)(base + (uint32_t)b + 1); rank++; }
    if (w == 0) lo[0] = skip;
}
// The two kernels above in ONE pass over the text: a workgroup turns its 16 KiB into newline masks, learns how many line ends lie in front of it
// from its predecessors (decoupled look-back: every workgroup publishes its own count at once and its inclusive prefix as soon as it knows it;
// a workgroup's first wave sums the counts behind it, 64 at a time, back to the nearest published prefix) and writes its line starts straight away -
// no bitmap in HBM, no second pass over it, no scan launches in between.  Workgroups take their place in the text from a ticket counter (order of
// arrival, not blockIdx): a workgroup only ever waits for workgroups that were started before it.  state[b]: bits 62-63 = 1 count / 2 inclusive
// prefix, low 32 bits = the value; the word IS the message (one aligned 8-byte agent-scope store / load: the XCDs' L2s are not coherent), zeroed by
// the host before every launch.  lo_cap: entries lo can hold; more lines than that (lines of a few bytes), or a wait that does not end, set
// st->err bit 30 and the host takes the two-pass path.
#define NLF_TILES 16                                 // 4 KiB tiles per wave: a workgroup indexes NLF_TILES x 16 KiB of contiguous text (RFQ_IDX_TILES = 4 / 8: the other instantiations; 2 x 4 GB: 2.10 / 1.94 / 1.83 ms with 4 / 8 / 16)
#define NLF_SPINS (1u << 18)
#ifdef RFQ_SIMT_EMULATION
__device__ __forceinline__ void nlf_store(unsigned long long* p, unsigned long long v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
__device__ __forceinline__ unsigned long long nlf_load(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
__device__ __forceinline__ void nlf_pause() {}
#else
typedef __attribute__((address_space(1))) unsigned long long nlf_gu64;
__device__ __forceinline__ void nlf_store(unsigned long long* p, unsigned long long v) { __hip_atomic_store((nlf_gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
__device__ __forceinline__ unsigned long long nlf_load(const unsigned long long* p) { return __hip_atomic_load((nlf_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
__device__ __forceinline__ void nlf_pause() { __builtin_amdgcn_s_sleep(2); }
#endif
// Why NLF_TILES x 16 KiB and not 16 KiB per workgroup: a look-back is a few dependent round trips to memory (the state words bypass the L2s); with
// 16 KiB of text per workgroup that is as long as the work itself, nobody's prefix is ever ready when its successors look, and every look-back walks
// far (measured 5.9 ms against the two passes' 2.1 ms on 2 x 4 GB).  With 256 KiB the wait is a small part of a workgroup's life.
template <int NLF_T> __global__ void __launch_bounds__(256) k_line_index(const uint8_t* __restrict__ fq, uint32_t n, uint32_t skip, uint32_t* __restrict__ lo,
        uint32_t lo_cap, unsigned long long* state,
                                                    uint32_t* ticket, uint32_t* total, DevStatus* st) {
    constexpr uint32_t NLF_BYTES = NLF_T * 16384u;
    __shared__ uint32_t s_blk, s_base, s_wave[4];
    if (threadIdx.x == 0) s_blk = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t blk = s_blk;
    const int lane = lane_id(), wv = wave_id();
    const uint64_t wbase = (uint64_t)blk * NLF_BYTES + (uint64_t)wv * (NLF_T * 4096u) + (uint32_t)lane * 64u;      // this lane's 64 bytes of tile 0
    uint64_t m[NLF_T]; uint32_t incl[NLF_T], tsum[NLF_T]; bool cr = false;
#pragma unroll
    for (int k = 0; k < NLF_T; k++) {
        const uint64_t base = wbase + (uint32_t)k * 4096u;
        uint64_t mk = 0, crm = 0;
        if (base + 64 <= n) {
            const uint4* p = (const uint4*)(fq + base);
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) { uint4 q = p[q4]; mk |= (uint64_t)eq_mask16c(q, 0x0A0A0A0Au) << (16 * q4); crm |= has_byte16(q, 0x0D0D0D0Du); }
            // (bytes in front of the stream do not count)
            if (base == 0 && skip && crm) { crm = 0; for (uint32_t i = skip; i < 64; i++) if (fq[i] == '\r') crm = 1ull << 63; }
        } else if (base < n) {
            for (uint32_t i = 0; i < 64 && base + i < n; i++) { uint8_t c = fq[base + i]; if (c == '\n') mk |= 1ull << i; if (c == '\r') crm |= 1ull << i; }
        }
        if (base == 0 && skip) { const uint64_t keep = ~((1ull << skip) - 1ull); mk &= keep; crm &= keep; }
        m[k] = mk; cr |= crm != 0;
    }
    if (__any(cr) && lane == 0) atomicOr(&st->err, (uint32_t)DE_HAS_CR);
    uint32_t wtot = 0;
#pragma unroll
    for (int k = 0; k < NLF_T; k++) { incl[k] = wave_incl_sum<uint32_t>((uint32_t)__popcll(m[k])); tsum[k] = wave_last(incl[k]); wtot += tsum[k]; }
    if (lane == 0) s_wave[wv] = wtot;
    __syncthreads();
    const uint32_t tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    uint32_t wavebase = 0;
    for (int k = 0; k < wv; k++) wavebase += s_wave[k];
    if (threadIdx.x < 64) {
        if (lane == 0) nlf_store(&state[blk], ((blk == 0 ? 2ull : 1ull) << 62) | tot);       // my count (or, as the first workgroup, my prefix)
        uint32_t before = 0;
        if (blk) {
            int64_t newest = (int64_t)blk - 1;                                                // nearest predecessor not summed yet
            for (uint32_t spins = 0;;) {
                const int64_t j = newest - lane;
                const unsigned long long v = j >= 0 ? nlf_load(&state[j]) : (2ull << 62);     // (in front of the text: prefix 0)
                const unsigned long long pm = __ballot((v >> 62) == 2), zm = __ballot((v >> 62) == 0);
                const int fp = pm ? __ffsll((long long)pm) - 1 : 63;                          // the nearest prefix among these 64, if any
                const unsigned long long need = (2ull << fp) - 1ull;                          // lanes 0 .. fp
                if (zm & need) {                                                              // somebody in that range has not published yet
                    if (++spins > NLF_SPINS) { if (lane == 0) atomicOr(&st->err, (uint32_t)DE_INDEX_RETRY); break; }
                    nlf_pause(); continue;
                }
                before += wave_sum<uint32_t>(lane <= fp ? (uint32_t)v : 0u);
                if (pm) break;
                newest -= 64;
            }
            if (lane == 0) nlf_store(&state[blk], (2ull << 62) | (unsigned long long)(uint32_t)(before + tot));
        }
        if (lane == 0) {
            s_base = before;
            if ((uint64_t)(blk + 1) * NLF_BYTES >= n) *total = before + tot;                  // the last workgroup of the text: the number of line ends
        }
    }
    __syncthreads();
    uint32_t tbase = s_base + wavebase;                                                       // line ends in front of this wave's tile k
    if ((uint64_t)tbase + wtot + 1u > lo_cap) { if (wtot && lane == 0) atomicOr(&st->err, (uint32_t)DE_INDEX_RETRY); }
    else {
#pragma unroll
        for (int k = 0; k < NLF_T; k++) {
            uint64_t mk = m[k]; uint32_t rank = tbase + incl[k] - (uint32_t)__popcll(mk);
            const uint32_t base = (uint32_t)(wbase + (uint32_t)k * 4096u);
            while (mk) { const int b = __ffsll((long long)mk) - 1; mk &= mk - 1; lo[rank + 1] = base + (uint32_t)b + 1u; rank++; }
            tbase += tsum[k];
        }
    }
    if (blk == 0 && threadIdx.x == 0) lo[0] = skip;
}
__global__ void k_line_tail(uint32_t* lo, uint32_t n_newlines, uint32_t n, int unterminated) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && unterminated) lo[n_newlines + 1] = n + 1;
}

// =============================================================== text normalisation (slow path: '\r' or blank lines present)
// FastqReader::getLine (src/fastqreader.cpp:94-156): a line ends at '\r' or '\n'; ONE '\n' directly after a terminator is
// swallowed ("\r\n", but also a single blank line) unless that terminator sits in the last two bytes of the reader's 1 MiB
// block (`end < mBufDataLen - 1`).  Every byte is K (kept), T (terminator) or S (swallowed); the normalised stream keeps K,
// writes '\n' for T and drops S, so the '\n'-only indexer above applies unchanged.  ot / onx map normalised line i back to
// the original text: offset of its terminator, offset of the line after it.
// Exactness: the class of a '\n' depends on its predecessors through the run of terminator characters before it; the walk
// below looks back over at most 4 of them.  The third terminator of any such run already is an empty line, where the reader
// stops for good (src/fastqreader.cpp:180-191), so classes beyond that point never reach the output.
#define FQ_BLOCK_BYTES (1u << 20)
struct NormIn { const uint8_t* fq; uint32_t n; uint64_t file_off, file_end; };
__device__ __forceinline__ bool norm_exc(const NormIn& c, uint32_t j) {            // '\n' at j (j >= 1) cannot be swallowed
    const uint64_t e = c.file_off + j - 1;                                           // absolute offset of the terminator
    uint64_t bend = (e | (uint64_t)(FQ_BLOCK_BYTES - 1)) + 1; if (bend > c.file_end) bend = c.file_end;
    return !(e + 1 < bend - 1);
}
__device__ __forceinline__ bool norm_state_at(const NormIn& c, uint32_t j) {       // is byte j-1 a terminator that may swallow byte j?
    if (j == 0) return false;                                                        // a batch starts at a line start
    uint32_t k = 0; while (k < 4 && k < j && c.fq[j - 1 - k] == '\n') k++;
    bool st = (k < 4 && j - k > 0) ? c.fq[j - k - 1] == '\r' : false;
    for (uint32_t i = j - k; i < j; i++) { if (st && !norm_exc(c, i)) st = false; else st = true; }
    return st;
}
// 64 bytes per thread: T and S bitmaps + per-block counts of kept bytes and of terminators
__global__ void k_norm_classify(NormIn c, uint64_t* __restrict__ tbits, uint64_t* __restrict__ sbits, uint32_t* __restrict__ blk_keep, uint32_t* __restrict__ blk_term) {
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; const uint64_t base = w * 64;
    uint64_t tm = 0, sm = 0; uint32_t valid = 0;
    if (base < c.n) {
        valid = (uint32_t)(c.n - base < 64 ? c.n - base : 64);
        bool st = norm_state_at(c, (uint32_t)base);
        for (uint32_t i = 0; i < valid; i++) {
            const uint8_t ch = c.fq[base + i];
            if (ch == '\r') { tm |= 1ull << i; st = true; }
            else if (ch == '\n') { if (st && !norm_exc(c, (uint32_t)base + i)) { sm |= 1ull << i; st = false; } else { tm |= 1ull << i; st = true; } }
            else st = false;
        }
        tbits[w] = tm; sbits[w] = sm;
    }
    uint32_t tk, tt; (void)block_excl_sum<uint32_t>(valid - (uint32_t)__popcll(sm), &tk); (void)block_excl_sum<uint32_t>((uint32_t)__popcll(tm), &tt);
    if (threadIdx.x == 0) { blk_keep[blockIdx.x] = tk; blk_term[blockIdx.x] = tt; }
}
__global__ void k_norm_emit(NormIn c, const uint64_t* __restrict__ tbits, const uint64_t* __restrict__ sbits, const uint32_t* __restrict__ keep_base,
        const uint32_t* __restrict__ term_base,
                            uint8_t* __restrict__ out, uint32_t* __restrict__ ot, uint32_t* __restrict__ onx) {
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; const uint64_t base = w * 64;
    uint64_t tm = 0, sm = 0; uint32_t valid = 0;
    if (base < c.n) { valid = (uint32_t)(c.n - base < 64 ? c.n - base : 64); tm = tbits[w]; sm = sbits[w]; }
    uint32_t kp = keep_base[blockIdx.x] + block_excl_sum<uint32_t>(valid - (uint32_t)__popcll(sm), (uint32_t*)nullptr);
    uint32_t tr = term_base[blockIdx.x] + block_excl_sum<uint32_t>((uint32_t)__popcll(tm), (uint32_t*)nullptr);
    for (uint32_t i = 0; i < valid; i++) {
        if ((sm >> i) & 1ull) continue;
        const bool t = ((tm >> i) & 1ull) != 0;
        out[kp++] = t ? (uint8_t)'\n' : c.fq[base + i];
        if (t) {
            const uint32_t pos = (uint32_t)base + i;
            bool sw = false;
            if (pos + 1 < c.n) sw = i + 1 < 64 ? ((sm >> (i + 1)) & 1ull) != 0 : (sbits[w + 1] & 1ull) != 0;
            ot[tr] = pos; onx[tr] = pos + 1 + (sw ? 1u : 0u); tr++;
        }
    }
}
__global__ void k_norm_tail(uint32_t* ot, uint32_t* onx, uint32_t n_terms, uint32_t n) {   // the virtual terminator of an unterminated last line
    if (threadIdx.x == 0 && blockIdx.x == 0) { ot[n_terms] = n; onx[n_terms] = n; }
}

// =============================================================== read table + name parse
// glibc atoi: (int)strtol — leading isspace, sign, digits, saturating at LONG_MIN/LONG_MAX.
__device__ __forceinline__ int dev_atoi(const uint8_t* s, uint32_t n) {
    uint32_t i = 0;
    while (i < n && (s[i] == ' ' || (s[i] >= 9 && s[i] <= 13))) i++;
    bool neg = false;
    if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
    const unsigned long long lim = neg ? 0x8000000000000000ull : 0x7FFFFFFFFFFFFFFFull;
    unsigned long long acc = 0; bool sat = false;
    for (; i < n && s[i] >= '0' && s[i] <= '9'; i++) {
        unsigned d = (unsigned)(s[i] - '0');
        if (sat || acc > (lim - d) / 10) { sat = true; acc = lim; } else acc = acc * 10 + d;
    }
    unsigned long long v = neg ? (0ull - acc) : acc;
    return (int)(uint32_t)v;
}
struct Meta { uint32_t ok, name1_len, name2_off, x, y; uint16_t tile; uint8_t lane; };
// FastqMeta::parse, src/fastqmeta.cpp:22-80
// scan / done: only the first `scan` bytes are looked at; *done says whether that settled the result (the loop met its stop, or scan covers the name)
__device__ __forceinline__ Meta dev_parse_name(const uint8_t* str, uint32_t len, uint32_t scan = 0xFFFFFFFFu, bool* done = nullptr) {
    int colon = 0, last_colon = 0, cstart = 0, cend = 0;
    uint8_t lane = 0; uint16_t tile = 0; uint32_t x = 0, y = 0;
    const uint32_t lim = len < scan ? len : scan; bool stopped = false;
    for (uint32_t i = 0; i < lim; i++) {
        const uint8_t c = str[i];
        if (c == ':') colon++;
        if ((c == ':' || c == ' ') && colon >= 4 && colon <= 7) {
            const int val = dev_atoi(str + last_colon + 1, i - (uint32_t)last_colon - 1);
            if (colon == 4) { lane = (uint8_t)val; cstart = last_colon + 1; }
            else if (colon == 5) tile = (uint16_t)val;
            else if (colon == 6) { if (c == ':') x = (uint32_t)val; }
            else y = (uint32_t)val;
            if (c == ' ' && colon == 6) y = (uint32_t)val;
        }
        if (c == ':') last_colon = (int)i;
        if (c == ' ' || (c == ':' && colon == 7)) { cend = (int)i; stopped = true; break; }
    }
    if (done) *done = stopped || lim == len;
    Meta m;
    if (cstart > 0 && cend > 0) { m.ok = 1; m.lane = lane; m.tile = tile; m.x = x; m.y = y; m.name1_len = (uint32_t)(cstart - 1); m.name2_off = (uint32_t)cend; }
    else { m.ok = 0; m.lane = 0; m.tile = 0; m.x = 0; m.y = 0; m.name1_len = len; m.name2_off = len; }
    return m;
}
// One thread per read.  Names are staged through LDS first: per-lane byte walks over 64 different cache lines thrash
// the 32 KiB L1 (each byte load re-fetches a line), so the wave copies the 64 names row by row with coalesced loads
// (lane i takes byte i of read j's name) and every lane then parses its own row from LDS (row stride 132 B = 33 banks).
#define NAME_CAP 128
#define NAME_STRIDE 132
// Stage the names of a wave's 64 reads into LDS rows: lane i copies byte i of read j's name.  Eight rows' loads are issued
// before the first LDS write (an in-order wave otherwise pays one full memory latency per row).  nb / nl / s: per-lane name
// start, name length and stream of the lane's own read.
__device__ __forceinline__ void stage_name_rows(const Text& T, uint8_t* rows, uint32_t nb, uint32_t nl, int s, int l) {
    for (int j0 = 0; j0 < 64; j0 += 8) {
        uint32_t take[8]; const uint8_t* src[8]; uint8_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t jb = __shfl(nb, j0 + u), jl = __shfl(nl, j0 + u); const int js = __shfl(s, j0 + u);
            take[u] = jl < NAME_CAP ? jl : NAME_CAP; src[u] = t_fq(T, js) + jb;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (uint32_t)l < take[u] ? src[u][l] : (uint8_t)0;
#pragma unroll
        for (int u = 0; u < 8; u++) if ((uint32_t)l < take[u]) rows[(j0 + u) * NAME_STRIDE + l] = v[u];
#pragma unroll
        for (int u = 0; u < 8; u++) for (uint32_t i = 64u + (uint32_t)l; i < take[u]; i += 64) rows[(j0 + u) * NAME_STRIDE + i] = src[u][i];   // names > 64 bytes
    }
}
__device__ __forceinline__ bool bytes_eq(const uint8_t* a, uint32_t alen, const uint8_t* b, uint32_t blen);
// k_read_table's staging: EIGHT lanes per name, each one aligned 16-byte group of the 128-byte window that starts at the name's
// 16-byte-aligned address - a wave stages its 64 names with 8 global_load_dwordx4 (all in flight together) instead of 64 rounds
// of byte loads.  A group is stored with one byte-granular ds_write_b128 at (row + 16 + 16 * part - (name start & 15)), so the
// name itself begins at row + 16 whatever its alignment was (the bytes in front of it land in the row's own 16-byte pad).
// Row stride 116 B = 29 banks: lanes walking their own rows byte by byte do not collide; 29.7 KB per block = five blocks per CU.
#define RT_NAME_CAP 80            // 15 + 80 < 96: a name this long sits inside the first six 16-byte groups of its window
#define RT_ROW 116
__device__ __forceinline__ void stage_name_rows_wide(const Text& T, uint8_t* rows, uint32_t nb, uint32_t nl, int s, int l) {
    const uint32_t part = (uint32_t)l & 7u;
    uint4 v[8]; bool ok[8]; uint32_t da[8];
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int j = it * 8 + (l >> 3);
        const uint32_t jb = __shfl(nb, j), jl = __shfl(nl, j); const int js = __shfl(s, j);
        const uint32_t a = (jb & ~15u) + 16u * part;                         // the group's offset in its stream
        // it holds bytes of the name (of its first RT_NAME_CAP bytes: a longer name is parsed and compared from that prefix, and from global memory only where the prefix
        // does not settle it)
        ok[it] = jl != 0 && a < jb + (jl < RT_NAME_CAP ? jl : RT_NAME_CAP);
        da[it] = (uint32_t)j * RT_ROW + 16u + 16u * part - (jb & 15u);
        v[it] = make_uint4(0, 0, 0, 0);
        if (ok[it]) {
            const uint8_t* g = t_fq(T, js) + a;
            if ((uint64_t)a + 16ull <= (uint64_t)t_n(T, js)) { const LdsU16 t = *(const LdsU16*)g; v[it] = make_uint4(t.a, t.b, t.c, t.d); }
            else { uint32_t w[4] = { 0, 0, 0, 0 }; for (uint32_t b = 0; b < 16 && a + b < t_n(T, js); b++) w[b >> 2] |= (uint32_t)g[b] << (8 * (b & 3));
                    v[it] = make_uint4(w[0], w[1], w[2], w[3]); }
        }
    }
#pragma unroll
    for (int it = 0; it < 8; it++) if (ok[it]) { LdsU16 t; t.a = v[it].x; t.b = v[it].y; t.c = v[it].z; t.d = v[it].w; *(LdsU16*)(rows + da[it]) = t; }
}
// equality of rows[a .. a+n) and rows[b .. b+n) (LDS, any alignment), 4 bytes per step
__device__ __forceinline__ bool lds_bytes_eq(const uint8_t* rows, uint32_t a, uint32_t b, uint32_t n) {
    for (uint32_t i = 0; i < n; i += 4) {
        uint32_t x = lds_get4(rows, a + i) ^ lds_get4(rows, b + i);
        if (n - i < 4) x &= (1u << (8 * (n - i))) - 1u;
        if (x) return false;
    }
    return true;
}
// FastqMeta::parse for reads [0, n_reads) by themselves - a lane per read, names staged in LDS rows - for the two callers that need the parsed
// names BEFORE the gather: the file header of a first batch (RfqCodec::makeHeader looks at chunk 0 only, src/rfqcodec.cpp:20-145: the host passes
// chunk 0's reads) and the byte-wise gather path (every read).  The tile gather k_gather2 parses the names of the tile it has staged anyway
// (g2_parse) - the text is not read a third time for them.
__global__ void k_read_table(Text T, ReadTab R, uint32_t n_reads) {
    __shared__ __attribute__((aligned(16))) uint8_t s_names[4 * 64 * RT_ROW + 16];
    const int l = lane_id(), w = wave_id();
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = g < n_reads;
    uint32_t nb = 0, nl = 0; int s = 0;
    if (valid) { uint32_t r; read_loc(T, g, s, r); const uint32_t* p = t_lo(T, s) + 4 * (size_t)r; nb = p[0]; nl = p[1] - 1 - nb; }
    uint8_t* rows = s_names + (size_t)w * 64 * RT_ROW + 16;              // (+16: a row's name begins 16 bytes into the row)
    stage_name_rows_wide(T, rows - 16, nb, nl, s, l);
    __syncthreads();
    if (valid) {
        bool settled = true;
        Meta m = dev_parse_name(rows + l * RT_ROW, nl, RT_NAME_CAP, &settled);
        if (!settled) m = dev_parse_name(t_fq(T, s) + nb, nl);
        R.name1_len[g] = m.name1_len; R.name2_off[g] = m.name2_off; R.x[g] = m.x; R.y[g] = m.y; R.tile[g] = m.tile; R.lane[g] = m.lane; R.ok[g] = (uint8_t)m.ok;
    }
}
// rfq_scan_batch: offset just past the last record of every chunk, per input stream, in the caller's coordinates (onx: normalised text)
__global__ void k_chunk_ends(Text T, const uint32_t* __restrict__ first, uint32_t n_chunks, const uint32_t* __restrict__ onx0, const uint32_t* __restrict__ onx1,
                             uint64_t* __restrict__ end1, uint64_t* __restrict__ end2) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; if (c >= n_chunks) return;
    const uint32_t g = first[c + 1];                                        // reads (interleaved order) before the end of chunk c
    const uint32_t rec = T.paired == 1 ? g >> 1 : g;                        // records consumed in each stream
    const size_t li = 4 * (size_t)rec;
    end1[c] = onx0 ? (rec ? (uint64_t)onx0[li - 1] : 0ull) : (uint64_t)T.lo[0][li];
    if (T.paired == 1) end2[c] = onx1 ? (rec ? (uint64_t)onx1[li - 1] : 0ull) : (uint64_t)T.lo[1][li];
}
// Sequence lengths from the line table alone (no text is read): len / stored per read, the line checks of FastqReader::read (an empty line ends
// the input there, src/fastqreader.cpp:180-191; a quality line shorter than its sequence is refused), bases per partition unit (a read, or a
// pair) + per-block min / max for the partitioner's uniform-length fast path and the longest record (k_gather2 sizes its tiles by it).
// (no atomics for the min / max: 44k waves hitting two words serialise at ~11 ns each, and a "skip if no change" test reads stale L1 lines)
__global__ void k_read_lens(Text T, uint32_t* __restrict__ len, uint32_t* __restrict__ stored, uint64_t* __restrict__ ulen, uint32_t n_units, uint32_t upr,
        uint32_t* __restrict__ blk_minmax, DevStatus* st) {
    __shared__ uint32_t s_mn[4], s_mx[4], s_rc[4], s_ml[4];
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t tot = 0; uint32_t rec = 0, ml = 0, err = 0, fe = 0xFFFFFFFFu;   // rec: bytes of the unit's longest record, ml: bases of its longest read
    if (u < n_units) {
        for (uint32_t j = 0; j < upr; j++) {
            const uint32_t g = u * upr + j; int s; uint32_t r; read_loc(T, g, s, r); const uint32_t* p = t_lo(T, s) + 4 * (size_t)r;
            const uint32_t p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4];
            const uint32_t nl = p1 - 1 - p0, sl = p2 - 1 - p1, tl = p3 - 1 - p2, ql = p4 - 1 - p3;
            if (nl == 0 || sl == 0 || tl == 0 || ql == 0) { err |= DE_EMPTY_LINE; if (g < fe) fe = g; }
            if (ql < sl) err |= DE_QUAL_SHORT;
            len[g] = sl; stored[g] = sl; tot += sl; if (p4 - p0 > rec) rec = p4 - p0; if (sl > ml) ml = sl;
        }
        ulen[u] = tot;
    }
    uint32_t mn = u < n_units ? (uint32_t)(tot > 0xFFFFFFFFull ? 0xFFFFFFFFu : tot) : 0xFFFFFFFFu, mx = u < n_units ? mn : 0u;
    mn = wave_min(mn); mx = wave_max(mx); rec = wave_max(rec); ml = wave_max(ml); fe = wave_min(fe); err = wave_or(err);
    if (lane_id() == 0) { s_mn[wave_id()] = mn; s_mx[wave_id()] = mx; s_rc[wave_id()] = rec; s_ml[wave_id()] = ml; if (err) { atomicOr(&st->err, err);
            if (fe != 0xFFFFFFFFu) atomicMin(&st->first_empty, fe); } }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t i = 1; i < (blockDim.x >> 6); i++) { if (s_mn[i] < mn) mn = s_mn[i]; if (s_mx[i] > mx) mx = s_mx[i]; if (s_rc[i] > rec) rec = s_rc[i];
                if (s_ml[i] > ml) ml = s_ml[i]; }
        blk_minmax[4 * blockIdx.x] = mn; blk_minmax[4 * blockIdx.x + 1] = mx; blk_minmax[4 * blockIdx.x + 2] = rec; blk_minmax[4 * blockIdx.x + 3] = ml;
    }
}

// =============================================================== chunk partition (one wave)
// P = inclusive prefix of unit lengths.  Chunk = minimal run of units whose bases reach chunk_bases (src/repaq.cpp:552-553).
__device__ __forceinline__ uint32_t wave_lower_bound(const uint64_t* __restrict__ P, uint32_t lo, uint32_t hi, uint64_t target) {
    // smallest e in [lo, hi) with P[e] >= target, or hi.  64-ary search, wave-uniform.
    const int l = lane_id();
    while (hi - lo > 64) {
        const uint32_t span = hi - lo, stride = (span + 63) / 64;
        uint64_t idx = (uint64_t)lo + (uint64_t)(l + 1) * stride - 1; if (idx >= hi) idx = hi - 1;
        const unsigned long long b = __ballot(P[idx] >= target);
        if (!b) return hi;
        const int j = __ffsll((long long)b) - 1;
        uint64_t nhi = (uint64_t)lo + (uint64_t)(j + 1) * stride; if (nhi > hi) nhi = hi;
        lo = lo + (uint32_t)j * stride; hi = (uint32_t)nhi;
    }
    const uint32_t i = lo + (uint32_t)l;
    const unsigned long long b = __ballot(i < hi && P[i] >= target);
    if (!b) return hi;
    return lo + (uint32_t)(__ffsll((long long)b) - 1);
}
// carry: bases the chunk that is open at unit 0 has taken from the text in front of this batch (plan pass of a share, rfq_encode_args.carry_bases)
__global__ void k_partition(const uint64_t* __restrict__ P, uint32_t n_units, uint32_t upr, uint32_t chunk_bases, uint32_t carry, int final_batch,
                            const uint32_t* __restrict__ blk_minmax, uint32_t n_blk, uint32_t* __restrict__ first, uint32_t cap_chunks, DevStatus* st) {
    const int l = lane_id();
    // shortest / longest unit: every thread of the workgroup (1024: a single wave walked 44 k block entries in 158 us), then wave 0 goes on alone
    __shared__ uint32_t s_mn[16], s_mx[16], s_rc[16], s_ml[16];
    uint32_t len_minmax[2], max_rec, max_len;
    { uint32_t mn = 0xFFFFFFFFu, mx = 0, rc = 0, ml = 0;
      for (uint32_t i = threadIdx.x; i < n_blk; i += blockDim.x) { const uint32_t a = blk_minmax[4 * i], b = blk_minmax[4 * i + 1], r = blk_minmax[4 * i + 2], m = blk_minmax[4 * i + 3]; if (a < mn) mn = a; if (b > mx) mx = b; if (r > rc) rc = r; if (m > ml) ml = m; }
      mn = wave_min(mn); mx = wave_max(mx); rc = wave_max(rc); ml = wave_max(ml);
      if (l == 0) { s_mn[wave_id()] = mn; s_mx[wave_id()] = mx; s_rc[wave_id()] = rc; s_ml[wave_id()] = ml; }
      __syncthreads();
      if (wave_id() != 0) return;
      const uint32_t nw = blockDim.x >> 6; mn = (uint32_t)l < nw ? s_mn[l] : 0xFFFFFFFFu; mx = (uint32_t)l < nw ? s_mx[l] : 0u; rc = (uint32_t)l < nw ? s_rc[l] : 0u;
              ml = (uint32_t)l < nw ? s_ml[l] : 0u;
      len_minmax[0] = wave_min(mn); len_minmax[1] = wave_max(mx); max_rec = wave_max(rc); max_len = wave_max(ml); }
    uint32_t c = 0, start = 0, max_units = 0; uint64_t prevP = 0, max_bases = 0;
    if (n_units > 0 && len_minmax[0] == len_minmax[1] && len_minmax[0] > 0) {
        // every unit has the same length L: a chunk is K = ceil(chunk_bases / L) units
        const uint32_t L = len_minmax[0]; const uint32_t K = (uint32_t)(((uint64_t)chunk_bases + L - 1) / L);
        // (the chunk open at unit 0 already holds `carry` bases: it closes after K0 units, the others after K each)
        const uint32_t K0 = carry ? (uint32_t)(((uint64_t)(chunk_bases - carry) + L - 1) / L) : K;
        const uint32_t head = n_units >= K0 ? K0 : 0u, full = head ? 1u + (n_units - K0) / K : 0u, used = head ? K0 + (full - 1u) * K : 0u, rem = n_units - used;
        const uint32_t nch = full + ((rem && final_batch) ? 1u : 0u);
        for (uint32_t i = (uint32_t)l; i <= nch && i < cap_chunks; i += 64) { uint64_t f = i == 0 ? 0ull : (uint64_t)K0 + (uint64_t)(i - 1u) * K;
                if (f > n_units) f = n_units; first[i] = (uint32_t)f * upr; }
        c = nch; start = (rem && !final_batch) ? used : n_units;
        max_units = head ? K0 : 0u; if (full > 1u && K > max_units) max_units = K; if (rem && final_batch && rem > max_units) max_units = rem;
                max_bases = (uint64_t)max_units * L;
    } else {
        uint32_t guess = 0;
        while (start < n_units) {
            const uint64_t target = prevP + chunk_bases - (c == 0 ? carry : 0u);
            uint32_t e = n_units; bool found = false;
            if (guess > 32 && start + guess - 32 < n_units) {            // probe a 64-wide window around the previous chunk's size
                const uint32_t w0 = start + guess - 32; const uint32_t i = w0 + (uint32_t)l;
                const unsigned long long b = __ballot(i < n_units && P[i] >= target);
                if (b && !(b & 1ull)) { e = w0 + (uint32_t)(__ffsll((long long)b) - 1); found = true; }
            }
            if (!found) e = wave_lower_bound(P, start, n_units, target);
            if (e >= n_units) { if (!final_batch) break; e = n_units - 1; }
            if (c < cap_chunks && l == 0) first[c] = start * upr;
            const uint64_t pe = P[e];
            if (pe - prevP > max_bases) max_bases = pe - prevP;
            if (e + 1 - start > max_units) max_units = e + 1 - start;
            guess = e + 1 - start; prevP = pe; start = e + 1; c++;
        }
        if (c < cap_chunks && l == 0) first[c] = start * upr;
    }
    if (l == 0) {
        st->n_chunks = c; st->n_units_used = start; st->max_chunk_reads = max_units * upr;
        st->max_chunk_bases = (uint32_t)(max_bases > 0xFFFFFFFFull ? 0xFFFFFFFFu : max_bases);
        st->total_bases = start ? P[start - 1] : 0; st->max_rec = max_rec; st->max_len = max_len;
                st->unit_bases = (n_units > 0 && len_minmax[0] == len_minmax[1]) ? len_minmax[0] : 0u;
    }
}
__global__ void k_chunk_ids(ChunkTab C, ReadTab R) {
    const uint32_t c = blockIdx.y; const uint32_t f = C.first[c], e = C.first[c + 1];
    const uint32_t g = f + blockIdx.x * blockDim.x + threadIdx.x;
    if (g < e) R.chunk[g] = c;
}

// =============================================================== header from chunk 0
// RfqCodec::makeHeader (src/rfqcodec.cpp:20-145) + RfqHeader::makeQualityTable (src/rfqheader.cpp:130-237).
struct HdrStats {
    uint32_t hist[128];
    uint32_t n_count;           // N bases in chunk 0
    uint32_t all_ok;            // AND of hasLaneTileXY (stored as "any not ok" = 0 -> ok)
    uint32_t any_not_ok;
    uint32_t max_len;
    uint64_t first_n_key;       // (read << 32 | offset) of the first N base, ~0 if none
    uint64_t first_err_key;     // first position with a bad quality / bad base, ~0 if none
    uint32_t q0;                // quality of the first N
    uint32_t need_npos;         // N with another quality, or a non-N base carrying q0 after the first N
    uint32_t pe_support;        // PE: starts 1, cleared by any failing pair
    uint32_t dpos, dch;         // name2 diff of pair 0
};
__global__ void k_hdr_init(HdrStats* H) {
    for (int i = threadIdx.x; i < 128; i += blockDim.x) H->hist[i] = 0;
    if (threadIdx.x == 0) { H->n_count = 0; H->all_ok = 1; H->any_not_ok = 0; H->max_len = 0; H->first_n_key = ~0ull; H->first_err_key = ~0ull; H->q0 = 0;
            H->need_npos = 0; H->pe_support = 1; H->dpos = 0; H->dch = 0; }
}
// pass 1: one wave per read of chunk 0 (grid-stride)
__global__ void k_hdr_stats(Text T, ReadTab R, const uint32_t* __restrict__ first, HdrStats* H) {
    __shared__ uint32_t sh[128];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    const uint32_t nreads = first[1];
    const int l = lane_id(); const uint32_t wpb = blockDim.x >> 6;
    uint32_t ncnt = 0, notok = 0, mxl = 0; uint64_t fn = ~0ull, fe = ~0ull;
    for (uint32_t g = blockIdx.x * wpb + (uint32_t)wave_id(); g < nreads; g += gridDim.x * wpb) {
        const uint32_t len = R.len[g]; const uint8_t* sq = line_ptr(T, g, 1); const uint8_t* ql = line_ptr(T, g, 3);
        if (!R.ok[g]) notok = 1;
        if (len > mxl) mxl = len;
        for (uint32_t i = (uint32_t)l; i < len; i += 64) {
            const uint8_t q = ql[i], b = sq[i]; const uint64_t key = ((uint64_t)g << 32) | i;
            if (q >= 128) { if (key < fe) fe = key; }
            else atomicAdd(&sh[q], 1u);
            if (b == 'N') { ncnt++; if (key < fn) fn = key; }
            else if (b != 'A' && b != 'C' && b != 'G' && b != 'T') { if (key < fe) fe = key; }
        }
    }
    ncnt = wave_sum(ncnt); notok = wave_or(notok); mxl = wave_max(mxl); fn = wave_min(fn); fe = wave_min(fe);
    if (l == 0) {
        if (ncnt) atomicAdd(&H->n_count, ncnt);
        if (notok) atomicOr(&H->any_not_ok, 1u);
        atomicMax(&H->max_len, mxl);
        if (fn != ~0ull) atomicMin((unsigned long long*)&H->first_n_key, (unsigned long long)fn);
        if (fe != ~0ull) atomicMin((unsigned long long*)&H->first_err_key, (unsigned long long)fe);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 128; i += blockDim.x) if (sh[i]) atomicAdd(&H->hist[i], sh[i]);
}
__global__ void k_hdr_q0(Text T, HdrStats* H) {
    if (threadIdx.x || blockIdx.x) return;
    if (H->first_n_key != ~0ull) { const uint32_t g = (uint32_t)(H->first_n_key >> 32), i = (uint32_t)H->first_n_key; H->q0 = line_ptr(T, g, 3)[i]; }
}
// pass 2: (a) an N whose quality differs from q0, (b) a non-N base with quality q0 located after the first N
__global__ void k_hdr_pass2(Text T, ReadTab R, const uint32_t* __restrict__ first, HdrStats* H) {
    const uint64_t fnk = H->first_n_key;
    if (fnk == ~0ull) return;                                   // uniform: no N at all
    const uint32_t q0 = H->q0; const uint32_t nreads = first[1];
    const int l = lane_id(); const uint32_t wpb = blockDim.x >> 6; uint32_t need = 0;
    for (uint32_t g = blockIdx.x * wpb + (uint32_t)wave_id(); g < nreads; g += gridDim.x * wpb) {
        const uint32_t len = R.len[g]; const uint8_t* sq = line_ptr(T, g, 1); const uint8_t* ql = line_ptr(T, g, 3);
        for (uint32_t i = (uint32_t)l; i < len; i += 64) {
            const uint8_t q = ql[i], b = sq[i]; const uint64_t key = ((uint64_t)g << 32) | i;
            if (b == 'N') { if (q != q0) need = 1; }
            else if (q == q0 && key > fnk) need = 1;
        }
    }
    need = wave_or(need);
    if (l == 0 && need) atomicOr(&H->need_npos, 1u);
}
// (a with a[pos] = ch when ch != 0) == b   — the name2 mate rule of src/rfqcodec.cpp:105-113 and :237-245
__device__ __forceinline__ bool name2_eq_replaced(const uint8_t* a, uint32_t alen, const uint8_t* b, uint32_t blen, uint32_t pos, uint32_t ch) {
    if (alen != blen) return false;
    for (uint32_t i = 0; i < alen; i++) { uint8_t c = a[i]; if (ch != 0 && i == pos) c = (uint8_t)ch; if (c != b[i]) return false; }
    return true;
}
__device__ __forceinline__ bool bytes_eq(const uint8_t* a, uint32_t alen, const uint8_t* b, uint32_t blen) {
    if (alen != blen) return false;
    for (uint32_t i = 0; i < alen; i++) if (a[i] != b[i]) return false;
    return true;
}
// PE: one thread per pair of chunk 0 (src/rfqcodec.cpp:89-114)
__global__ void k_hdr_pe(Text T, ReadTab R, const uint32_t* __restrict__ first, HdrStats* H) {
    const uint32_t npairs = first[1] / 2; const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    // pair 0 fixes (dpos, dch); every thread derives it (a name2 is a handful of bytes)
    const uint8_t* a0 = line_ptr(T, 0, 0) + R.name2_off[0]; const uint32_t al0 = line_len(T, 0, 0) - R.name2_off[0];
    const uint8_t* b0 = line_ptr(T, 1, 0) + R.name2_off[1]; const uint32_t bl0 = line_len(T, 1, 0) - R.name2_off[1];
    uint32_t dpos = 0, dch = 0;
    for (uint32_t i = 0; i < al0; i++) { const uint8_t c2 = i < bl0 ? b0[i] : 0; if (a0[i] != c2) { dpos = i; dch = c2; break; } }
    bool bad = false;
    if (p < npairs) {
        const uint32_t g = 2 * p;
        const uint8_t* a = line_ptr(T, g, 0) + R.name2_off[g]; const uint32_t al = line_len(T, g, 0) - R.name2_off[g];
        const uint8_t* b = line_ptr(T, g + 1, 0) + R.name2_off[g + 1]; const uint32_t bl = line_len(T, g + 1, 0) - R.name2_off[g + 1];
        if (p == 0 && al != bl) bad = true;
        if (al < dpos) bad = true;
        else if (!name2_eq_replaced(a, al, b, bl, dpos, dch)) bad = true;
    }
    if (__any(bad) && lane_id() == 0) atomicAnd(&H->pe_support, 0u);
    if (p == 0) { H->dpos = dpos; H->dch = dch; }
}
// derived tables shared by "made" and "set" headers: majorQual / normalQualBins / normalQualBuf (src/rfqheader.cpp:263,308-328)
__device__ __forceinline__ void hdr_derive(DevHeader* D) {
    const uint8_t* b = D->bytes;
    D->read_len_bytes = b[9]; D->flags = (uint32_t)b[10] | ((uint32_t)b[11] << 8);
    D->name2_diff_pos = b[12]; D->name2_diff_char = b[13]; D->n_base_qual = b[14]; D->overlap_shift = (int32_t)(int8_t)b[15];
    const uint32_t bins = b[16]; D->len = 17 + bins;
    D->support_interleaved = (D->flags & H_PE_OVERLAP) ? 1u : 0u;
    const uint8_t* qb = b + 17;
    D->major = bins ? qb[0] : 0;
    const int mq = (int)(int8_t)D->major, nq = (int)(int8_t)D->n_base_qual;
    const uint32_t nb = (mq == nq) ? bins : (bins ? bins - 1 : 0);
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < 256; i++) { D->stream_of[i] = 0xFF; D->is_exception[i] = 1; D->normal[i] = 0; }
    for (uint32_t i = 0; i < bins; i++) {
        const int v = qb[i];
        if (v != mq || v == nq) { if (cnt < nb) { D->normal[cnt] = (uint8_t)v; } cnt++; if (cnt > nb) break; }
    }
    D->n_normal = nb;
    // a byte equal to several normal entries is claimed by the FIRST stream only for the mask; later equal entries would
    // re-emit the same positions (the reference loops per entry).  Entries are distinct by construction (histogram bins).
    for (uint32_t i = 0; i < nb; i++) { const uint8_t v = D->normal[i]; if (D->stream_of[v] == 0xFF) D->stream_of[v] = (uint8_t)i; D->is_exception[v] = 0; }
    D->is_exception[D->major & 0xFF] = 0;
    D->valid = 1;
}
__global__ void k_hdr_from_bytes(DevHeader* D) { if (threadIdx.x == 0 && blockIdx.x == 0) hdr_derive(D); }
__global__ void k_hdr_finalize(Text T, HdrStats* H, DevHeader* D, int is_pe, DevStatus* st) {
    if (threadIdx.x || blockIdx.x) return;
    if (H->first_err_key != ~0ull) {
        const uint32_t g = (uint32_t)(H->first_err_key >> 32), i = (uint32_t)H->first_err_key;
        const uint8_t q = line_ptr(T, g, 3)[i];
        st->err |= (q >= 128) ? DE_BAD_QUAL : DE_BAD_BASE; st->err_read = g; st->err_key = H->first_err_key;
        return;
    }
    uint8_t* b = D->bytes;
    b[0] = 'R'; b[1] = 'F'; b[2] = 'Q'; b[3] = '0'; b[4] = '.'; b[5] = '5'; b[6] = '.'; b[7] = '1'; b[8] = 2;
    uint32_t flags = 0; int nbq = '#';
    const bool ltxy = H->any_not_ok == 0;
    if (ltxy) flags |= H_LANE | H_TILE | H_X | H_Y | H_NAME2;
    uint32_t dpos = 0, dch = 0;
    if (is_pe) { flags |= H_PAIRED; if (ltxy && H->pe_support) { flags |= H_PE_OVERLAP; dpos = H->dpos; dch = H->dch; } }
    // N-quality inference (src/rfqheader.cpp:145-184)
    if (H->n_count > 0) nbq = (int)H->q0;
    if (H->need_npos) { flags |= H_N_POS; nbq = -1; }
    if (H->n_count < 100) { flags |= H_N_POS; nbq = -1; }
    uint32_t bins = 0, maxnum = 0; int major = 0; bool has_n = false;
    for (int i = 0; i < 128; i++) { if (H->hist[i] > 0) { bins++; if (i == nbq) has_n = true; } if (H->hist[i] > maxnum) { maxnum = H->hist[i]; major = i; } }
    if (bins == 0) { st->err |= DE_NO_QUAL_BINS; return; }
    if (bins >= 64) flags |= H_DONT_QUAL;
    if (!has_n) bins += 1;
    b[17] = (uint8_t)major; uint32_t cur = 1;
    for (int i = 0; i < 128; i++) { if (i == major) continue; if (H->hist[i] > 0) b[17 + cur++] = (uint8_t)i; }
    if (!has_n) b[17 + bins - 1] = (uint8_t)nbq;
    if (bins <= 64) flags |= H_QUAL_BY_COL;
    b[9] = H->max_len > 255 ? 2 : 1;                       // never 4: src/rfqcodec.cpp:48-53 (second `if` is not `else if`)
    b[10] = (uint8_t)flags; b[11] = (uint8_t)(flags >> 8); b[12] = (uint8_t)dpos; b[13] = (uint8_t)dch; b[14] = (uint8_t)nbq; b[15] = (uint8_t)(-24);
            b[16] = (uint8_t)bins;
    hdr_derive(D);
}

// match-mask mode of k_gather2: which coded values get a plane built in LDS - the most frequent ones of chunk 0 (a NovaSeq-binned file codes ':' and ','
// a few percent of the time each, '#' only under N bases, and the table's 0xFF entry never)
__global__ void k_dense_order(const HdrStats* __restrict__ H, DevHeader* D) {
    if (threadIdx.x || blockIdx.x) return;
    const uint32_t nn = D->n_normal < 4u ? D->n_normal : 4u; uint32_t fr[4], ix[4];
    for (uint32_t j = 0; j < 4; j++) { ix[j] = j; const uint32_t v = D->normal[j]; fr[j] = (j < nn && v < 128u) ? H->hist[v] : 0u; }
    // (stable: ties keep the table's order)
    for (uint32_t a = 1; a < 4; a++) for (uint32_t b = a; b > 0 && fr[ix[b]] > fr[ix[b - 1]]; b--) { const uint32_t t = ix[b]; ix[b] = ix[b - 1]; ix[b - 1] = t; }
    for (uint32_t j = 0; j < 4; j++) D->dense[j] = (uint8_t)ix[j];
    D->dense_valid = 1;
}

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
#define OV2_CAP 256u              // bases per read held in a row
#define OV2_CROW 68u              // code row: 64 bytes + 4 of slack for the last unaligned word; 17 dwords, so that lanes reading their own rows at one offset hit 64 different banks
#define OV2_NROW 36u              // N-bit row: 32 bytes + 4; 9 dwords
#define OV2_WAVE_BYTES (128u * (OV2_CROW + OV2_NROW))
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
template <bool LOOSE> __global__ void __launch_bounds__(256) k_overlap(Text T, OvLoose Z, int16_t* __restrict__ ovraw, uint32_t n_pairs) {
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
            uint32_t W[17], Dm[16];
#pragma unroll
            for (int d = 0; d < 17; d++) W[d] = (uint32_t)d <= nd ? ((const uint32_t*)wc)[d] : 0u;
#pragma unroll
            for (int d = 0; d < 16; d++) Dm[d] = 0u;
#pragma unroll
            for (int k = 0; k < OV2_FILTER; k++) {
                const uint32_t rep = ((head >> (2 * k)) & 3u) * 0x55555555u;
#pragma unroll
                for (int d = 0; d < 16; d++) if ((uint32_t)d < nd) {
                    const uint32_t sk = k ? (uint32_t)(((((unsigned long long)W[d + 1]) << 32) | W[d]) >> (2 * k)) : W[d];
                    Dm[d] |= sk ^ rep;
                }
            }
            // the starts that pass are verified by their own lane, in ascending o = descending start: the window row from base wl - o on
            // against the head of the other row, 32 bases (64 code bits) or 64 N bits per step - byte-granular 8-byte LDS reads + a
            // sub-byte funnel shift.  (The whole wave used to verify ONE candidate at a time: ~27 rounds of ~45 instructions for 64 pairs.)
            const int i_lo = wl - minlen, i_hi = wl - 12;       // starts that exist for this pair (12 <= o <= minlen)
#pragma unroll
            for (int d = 0; d < 16; d++) {
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
                for (int d = 15; d >= 0; d--) if ((uint32_t)d < nd && cand < 0 && Dm[d]) cand = 16 * d + ((31 - __clz((int)Dm[d])) >> 1);
                if (!__any(cand >= 0)) break;                    // wave-uniform
                const uint32_t pa = cand >= 0 ? (uint32_t)cand : 0u, o = cand >= 0 ? (uint32_t)(wl - cand) : 0u;
                unsigned long long diff = 0;
                for (uint32_t c = 0; c < nch; c++) {
                    if (32u * c >= o) continue;
                    const uint32_t bit = 2u * (pa + 32u * c), off = bit >> 3, sh = bit & 7u;
                    unsigned long long x = lds_get8(wc, off) >> sh; if (sh) x |= (unsigned long long)wc[off + 8u] << (64u - sh);
                    const uint32_t nb = o - 32u * c;
                    diff |= (x ^ lds_get8(oc, 8u * c)) & (nb >= 32u ? ~0ull : (1ull << (2u * nb)) - 1ull);
                }
                for (uint32_t c = 0; c < nch2; c++) {
                    if (64u * c >= o) continue;
                    const uint32_t bit = pa + 64u * c, off = bit >> 3, sh = bit & 7u;
                    unsigned long long x = lds_get8(wn, off) >> sh; if (sh) x |= (unsigned long long)wn[off + 8u] << (64u - sh);
                    const uint32_t nb = o - 64u * c;
                    diff |= (x ^ lds_get8(on_, 8u * c)) & (nb >= 64u ? ~0ull : (1ull << nb) - 1ull);
                }
                if (cand >= 0) {
                    if (diff == 0) { done = true; ov = dir ? -(int)o : (int)o;
#pragma unroll
                        for (int d = 0; d < 16; d++) Dm[d] = 0u; }
                    else {
#pragma unroll
                        for (int d = 0; d < 16; d++) if (d == (cand >> 4)) Dm[d] &= ~(1u << (2 * (cand & 15)));
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
    U4 carry; carry.a = carry.b = carry.c = carry.d = 0;
    for (uint32_t base = f; base < e; base += 1024u) {                     // block-uniform
        U4 v[4]; U4 acc; acc.a = acc.b = acc.c = acc.d = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t g = base + 4u * tid + (uint32_t)i; v[i].a = v[i].b = v[i].c = v[i].d = 0;
            if (g < e) {
                int s_; uint32_t r_; read_loc(T, g, s_, r_); const uint4 lo4 = *(const uint4*)(t_lo(T, s_) + 4 * (size_t)r_);
                uint32_t st = R.len[g];
                if (enc && ((g - f) & 1u)) {
                    int ov = ovraw[g >> 1];
                    if (ov + shift > 127) ov = 0;
                    if (ov + shift < -127) ov = 0;
                    ovb[g >> 1] = (int8_t)(ov + shift); st -= (uint32_t)(ov < 0 ? -ov : ov); R.stored[g] = st;
                }
                v[i].a = R.name1_len[g]; v[i].b = (lo4.y - 1u - lo4.x) - R.name2_off[g]; v[i].c = lo4.w - 1u - lo4.z; v[i].d = st;
                acc = acc + v[i];
            }
        }
        U4 tot; U4 run = carry + block_excl_sum<U4>(acc, &tot);
#pragma unroll
        for (int i = 0; i < 4; i++) { const uint32_t g = base + 4u * tid + (uint32_t)i; if (g < e) { R.pv[g] = run; run = run + v[i]; } }
        carry = carry + tot;
    }
    if (tid == 0) { C.ptot[c] = carry; C.sbase[c] = C.qbase[c]; }          // (the tight streams are laid out like the qualities: stored <= len)
}
// byte-wise path: the totals from the batch-wide prefix
__global__ void k_chunk_ptot(ReadTab R, ChunkTab C, uint32_t n_chunks) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_chunks) C.ptot[c] = R.pv[C.first[c + 1]] - R.pv[C.first[c]];
}

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
        uint8_t* __restrict__ rflag) {
    const uint32_t ng = (m.len + 15u) >> 4;
    for (uint32_t gi = part; gi < ng; gi += P) {
        uint32_t w[4], code = 0, nbits = 0;
        const uint32_t nv0 = m.len - 16u * gi;                             // valid bases of this step
        // (the bytes of a last, partial step that lie outside the line would fail the all-ACGT test in some lane of nearly every wave - and a wave
        // runs the exact path if any of its lanes does: they are made 'A' first; their codes are masked off below)
        auto blank = [&](uint32_t from, uint32_t to) { for (uint32_t k = from; k < to; k++) { uint32_t& x = w[k >> 2]; const uint32_t sh = 8u * (k & 3u);
                x = (x & ~(0xFFu << sh)) | (0x41u << sh); } };
        if (!m.rc) {
            lds_get16(s_text, m.ssrc + 16u * gi, w);
            if (nv0 < 16u) blank(nv0, 16u);
            if (!pack16_fast(w, code)) {                                    // (an N, a lower-case or any other byte among the 16)
                uint32_t bad = 0; code = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) { uint32_t c4, n4, b4; pack4_codes(w[i], c4, n4, b4); code |= c4 << (8 * i); nbits |= n4 << (4 * i); bad |= b4 << (4 * i); }
                // a byte outside A/C/G/T/N: it equals nothing in RfqCodec::overlap (k_overlap's byte-wise path)
                if (bad) rflag[m.gi] = 1;
            }
        } else {
            lds_get16(s_text, m.ssrc + m.len - 16u * gi - 16u, w);           // the 16 file bases that END at len - 16 gi (the last step reaches in front of the line)
            if (nv0 < 16u) blank(0u, 16u - nv0);
            if (pack16_fast(w, code)) code = ~g2_rev2x16(code);             // reverse complement in 2-bit space: the fields back to front, G 0 <-> C 3, A 1 <-> T 2
            else {
                const uint32_t x0 = bswap32(w[3]), x1 = bswap32(w[2]), x2 = bswap32(w[1]), x3 = bswap32(w[0]); w[0] = x0; w[1] = x1; w[2] = x2; w[3] = x3; code = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) { uint32_t c4, n4; pack4_codes_rc(w[i], c4, n4); code |= c4 << (8 * i); nbits |= n4 << (4 * i); }
            }
        }
        if (nv0 < 16u) { code &= (1u << (2u * nv0)) - 1u; nbits &= (1u << nv0) - 1u; }   // (what lies outside the line is not the read's)
        lpk[m.ld + gi] = code; lnb[m.ld + gi] = (uint16_t)nbits;
    }
}
// my share (groups part, part + P, ...) of my read's two lines: qualities -> qcat, bases -> the loose slot
__device__ __forceinline__ void g2_compose(const uint8_t* s_text, const G2Read& m, uint32_t part, uint32_t P, uint8_t* qd, uint32_t* __restrict__ lpk,
        uint16_t* __restrict__ lnb, uint8_t* __restrict__ rflag, QualCount& qc) {
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
    g2_bases(s_text, m, part, P, lpk, lnb, rflag);
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
// ---- MASKS mode (files with at most three coded quality values - a NovaSeq-binned file has three): no quality bytes leave the kernel.  A lane turns its 16
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
template <bool MASKS> __global__ void __launch_bounds__(256, 6) k_gather2(Text T, ReadTab R, const uint32_t* __restrict__ first, const uint64_t* __restrict__ qbase,
                                                 const DevHeader* __restrict__ D, uint8_t* __restrict__ qcat, uint32_t* __restrict__ lpk, uint16_t* __restrict__ lnb, uint8_t* __restrict__ rflag,
                                                 uint32_t* __restrict__ segm, int* __restrict__ segc, uint32_t n_seg, uint32_t kshift,
                                                 uint32_t* __restrict__ cbits, uint32_t* __restrict__ cfail, const uint32_t* __restrict__ only, uint32_t text4, G2Planes M) {
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
    const bool two = T.paired == 1, can0 = T.paired != 0 && D->support_interleaved != 0, il = can0 && !redo;
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
            g2_bases(tx, m, part, P, lpk, lnb, rflag);
        } else g2_compose(tx, m, part, P, qd, lpk, lnb, rflag, qc);
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
__global__ void __launch_bounds__(256) k_seqpack(const uint32_t* __restrict__ pq, const U4* __restrict__ pv, const U4* __restrict__ ptot,
        const uint32_t* __restrict__ first, const uint32_t* __restrict__ ilv, const int8_t* __restrict__ ovb,
                                                 const DevHeader* __restrict__ D, const uint64_t* __restrict__ sbase, const uint32_t* __restrict__ lpk, const uint16_t* __restrict__ lnb,
                                                 uint32_t* __restrict__ spk, uint16_t* __restrict__ snm, uint32_t* __restrict__ ncount, uint32_t* __restrict__ nmap, uint32_t* __restrict__ segm, int* __restrict__ segc, uint32_t n_seg,
                                                 uint32_t rshift) {
    __shared__ uint32_t s_sd[256 + SP_EXTRA + 1], s_ld[256 + SP_EXTRA], s_sk[256 + SP_EXTRA]; __shared__ uint8_t s_own[SP_OWN];
    const uint32_t c = blockIdx.y, f = first[c], e = first[c + 1], tid = threadIdx.x;
    const uint32_t ps0 = pv[f].d, S = ptot[c].d;
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
            const uint32_t g = r0 + t; s_sd[t] = g < e ? pv[g].d - ps0 : S;  // (pv[e] belongs to the next chunk)
            if (t < nx) { s_ld[t] = (pq[g] >> 4) + g; s_sk[t] = skip_of(g); }
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
            struct Dw { uint32_t k, need, t1, jj, sh, sh2, take2, n, n2; unsigned long long c, c2; } q[SP_U];
#pragma unroll
            for (int u = 0; u < SP_U; u++) {
                Dw& x = q[u]; x.need = 0; x.take2 = 0; x.k = x.t1 = x.jj = x.sh = x.sh2 = x.n = x.n2 = 0; x.c = x.c2 = 0;
                const uint32_t i = i0 + (uint32_t)u * blockDim.x + tid;
                if (i < ndw) {
                    const uint32_t k = kbase + i, j = s_own[i];
                    const uint32_t B = 16u * k, si = B - s_sd[j], need = S - B < 16u ? S - B : 16u, av = s_sd[j + 1] - B, t1 = need < av ? need : av;
                    const uint32_t b0 = s_sk[j] + si, d = s_ld[j] + (b0 >> 4);
                    x.k = k; x.need = need; x.t1 = t1; x.sh = b0 & 15u; x.jj = j + 1u;
                    { const SpU8 v = *(const SpU8*)(lpk + d); x.c = (((unsigned long long)v.b) << 32) | v.a; x.n = ((const SpU4*)(lnb + d))->a; }
                    if (t1 < need && j + 1u < nx) {                          // the read behind: its LDS entries are there
                        const uint32_t avail = s_sd[j + 2] - s_sd[j + 1]; x.jj = j + 2u;
                        if (avail) {
                            const uint32_t s2 = s_sk[j + 1], d2 = s_ld[j + 1] + (s2 >> 4); x.sh2 = s2 & 15u; x.take2 = avail < need - t1 ? avail : need - t1;
                            const SpU8 v = *(const SpU8*)(lpk + d2); x.c2 = (((unsigned long long)v.b) << 32) | v.a; x.n2 = ((const SpU4*)(lnb + d2))->a;
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
                    else { const uint32_t gg = r0 + jj; a = pv[gg].d - ps0; b = gg + 1u < e ? pv[gg + 1].d - ps0 : S; l2 = (pq[gg] >> 4) + gg; s2 = skip_of(gg); }
                    const uint32_t avail = b - a;
                    if (avail) {
                        const uint32_t take = avail < need - filled ? avail : need - filled;
                        acc |= (unsigned long long)(loose_codes(lpk, l2, s2) & (take >= 16u ? 0xFFFFFFFFu : (1u << (2u * take)) - 1u)) << (2u * filled);
                        nacc |= (loose_nbits(lnb, l2, s2) & ((1u << take) - 1u)) << filled; filled += take;
                    }
                    jj++;
                }
                ok[k] = (uint32_t)acc; on[k] = (uint16_t)nacc;
                if (nacc) { const uint32_t n = (uint32_t)__popc(nacc); nsum += n; nmap_mark(nm, nshift, B); atomicAdd(&segm[nsi + B / PC_SEG_POS], n);
                        atomicMax(&segc[nsi + B / PC_SEG_POS], (int)(B + 31u - (uint32_t)__clz((int)nacc))); }
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
__global__ void k_pos_coder(ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, const uint8_t* __restrict__ qcat, const uint16_t* __restrict__ snm,
                            uint8_t* __restrict__ scratch, const uint64_t* __restrict__ cbase, uint8_t* __restrict__ scratch_n, const uint64_t* __restrict__ cbase_n,
                            uint32_t* __restrict__ segb, const int* __restrict__ segc, const uint32_t* __restrict__ segm, uint32_t n_seg, uint32_t n_chunks,
                            uint32_t n_qgroups, uint32_t g0, uint32_t gn, DevStatus* st, const uint32_t* __restrict__ planes, uint64_t pstride) {
    // XCD-aware mapping: workgroup b runs on XCD b % 8 (observed dispatch order; a different placement only costs speed).  All
    // (group, segment) workgroups of chunk c are given ids congruent to c mod 8, so a chunk's data stays in ONE private L2.
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
__global__ void __launch_bounds__(64, 3) k_pos_coder_list(ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, const uint8_t* __restrict__ qcat,
        uint8_t* __restrict__ scratch, const uint64_t* __restrict__ cbase,
                                                       uint32_t* __restrict__ segb, const int* __restrict__ segc, const uint32_t* __restrict__ segm, uint32_t n_seg, uint32_t n_chunks, DevStatus* st) {
    __shared__ PlLds S;
    // [stream][lane]: matches among the lane's 64 positions, then the lane's next free entry in the stream's part (n_normal x 64 u16: dynamic)
    RFQ_DYN_SHARED(uint16_t, pl_base);
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

// =============================================================== coordinate coder (encodeCoords, src/rfqcodec.cpp:1262-1330)
// One wave per (axis, chunk).  `last` always equals the previous element, so every token is local: a repeat element
// closes a 0xC0|k token when it is the 32nd of its group or the next element is not a repeat.
__global__ void k_coords(ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, uint8_t* __restrict__ xs, uint8_t* __restrict__ ys, DevStatus* st) {
    const uint32_t axis = blockIdx.x, c = blockIdx.y;
    if (!(D->flags & (axis ? H_Y : H_X))) return;
    const uint32_t f = C.first[c], e = C.first[c + 1]; const bool il = C.il[c] != 0;
    const uint32_t stride = il ? 2u : 1u, num = (e - f) / stride;
    const uint32_t* V = (axis ? R.y : R.x) + f;
    uint8_t* out = (axis ? ys : xs) + 3ull * f;
    const int l = lane_id();
    uint32_t outpos = 0, carry_prev = 1000u, carry_rep = 0; long long carry_start = -1;
    for (uint32_t base = 0; base < num; base += 64) {
        const uint32_t i = base + (uint32_t)l; const bool valid = i < num;
        const uint32_t v = valid ? V[(size_t)i * stride] : 0u;
        const uint32_t p = wave_shr1(v, carry_prev);
        const uint32_t rep = (valid && v == p) ? 1u : 0u;
        const bool rep_next = (i + 1 < num) && V[(size_t)(i + 1) * stride] == v;
        const uint32_t rep_prev = wave_shr1(rep, carry_rep);
        long long sidx = (rep && !rep_prev) ? (long long)i : -1;
        sidx = wave_incl_max(sidx); if (carry_start > sidx) sidx = carry_start;
        uint32_t bytes = 0, kind = 0;                                       // kind 1: repeat close, 2: +diff, 3: 15-bit, 4: 21-bit
        if (valid) {
            if (rep) { const uint32_t k = (uint32_t)((long long)i - sidx); if (((k + 1) & 31u) == 0 || !rep_next) { bytes = 1; kind = 1; } }
            else {
                const int diff = (int)(v - p);
                if (diff > 0 && diff <= 64) { bytes = 1; kind = 2; }
                else if (v <= 32767u) { bytes = 2; kind = 3; }
                else if (v < (1u << 21)) { bytes = 3; kind = 4; }
                else { atomicOr(&st->err, (uint32_t)DE_COORD_RANGE);
                        atomicMin((unsigned long long*)&st->coord_key, ((unsigned long long)c << 34) | ((unsigned long long)axis << 33) | (unsigned long long)i); }
            }
        }
        const uint32_t incl = wave_incl_sum(bytes); uint32_t o = outpos + incl - bytes;
        if (kind == 1) out[o] = (uint8_t)(0xC0u | (((uint32_t)((long long)i - sidx)) & 31u));
        else if (kind == 2) out[o] = (uint8_t)(0x80u | (uint32_t)((int)(v - p) - 1));
        else if (kind == 3) { out[o] = (uint8_t)(v >> 8); out[o + 1] = (uint8_t)v; }
        else if (kind == 4) { out[o] = (uint8_t)((v >> 16) | 0xE0u); out[o + 1] = (uint8_t)(v >> 8); out[o + 2] = (uint8_t)v; }
        outpos += wave_last(incl);
        carry_prev = wave_last(v); carry_rep = wave_last(rep); carry_start = wave_last(sidx);
    }
    if (l == 0) { if (axis) C.ysize[c] = outpos; else C.xsize[c] = outpos; }
}

// =============================================================== chunk image (RfqChunk::calcTotalBufSize + write, src/rfqchunk.cpp:141-159,230-311)
// mode 0: upper bound of the image size from stream capacities (before coding); mode 1: exact layout (after coding).
__global__ void k_chunk_layout(Text T, ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, Layout* __restrict__ L, uint32_t n_chunks, int exact, DevStatus* st) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const uint32_t f = C.first[c], e = C.first[c + 1], s = e - f, fl = C.flags[c], hf = D->flags, rlb = D->read_len_bytes, nn = D->n_normal;
    const bool il = (fl & C_PE_INTERLEAVED) != 0; const uint32_t h = il ? s / 2 : s;
    const U4 b = C.ptot[c];
    const uint32_t len = R.pq[e] - R.pq[f], seqCopied = b.d;
    Layout o;
    o.n_reads = s; o.flags = fl;
    const uint32_t readLenBuf = (fl & C_READ_LEN_SAME) ? rlb : rlb * s;
    const uint32_t n1Len = (fl & C_NAME1_LEN_SAME) ? 1 : s, n2Len = (fl & C_NAME2_LEN_SAME) ? 1 : s, stLen = (fl & C_STRAND_LEN_SAME) ? 1 : s;
    o.n1_size = (fl & C_NAME1_SAME) ? R.name1_len[f] : b.a;
    o.n2_size = (fl & C_NAME2_SAME) ? name2_len_of(T, R, f) : b.b;
    o.st_size = (fl & C_STRAND_SAME) ? line_len(T, f, 2) : b.c;
    o.seq_size = (seqCopied + 3) / 4;
    const size_t k0 = (size_t)c * MAX_STREAMS;
    uint32_t qsz = 0;
    if (hf & H_DONT_QUAL) qsz = len;
    else if (hf & H_QUAL_BY_COL) { qsz = 4 * nn; for (uint32_t j = 0; j < nn && j < NPOS_SLOT; j++) qsz += exact ? C.ssize[k0 + j] : C.scap[k0 + j];
            qsz += exact ? C.ssize[k0 + EXC_SLOT] : C.scap[k0 + EXC_SLOT]; }
    o.qual_size = qsz;
    o.npos_size = (hf & H_N_POS) ? (exact ? C.ssize[k0 + NPOS_SLOT] : C.scap[k0 + NPOS_SLOT]) : 0;
    o.x_size = (hf & H_X) ? (exact ? C.xsize[c] : 3 * h) : 0; o.y_size = (hf & H_Y) ? (exact ? C.ysize[c] : 3 * h) : 0;
    uint32_t k = 18 + ((hf & H_N_POS) ? 4 : 0);
    o.off_readlens = k; k += readLenBuf;
    o.off_n1lens = k; k += n1Len;
    o.off_n2lens = k; if (hf & H_NAME2) k += n2Len;
    o.off_stlens = k; k += stLen;
    o.off_lanes = k; if (hf & H_LANE) k += (fl & C_LANE_SAME) ? 1 : h;
    o.off_tiles = k; if (hf & H_TILE) k += 2 * ((fl & C_TILE_SAME) ? 1 : h);
    o.off_x = k; if (hf & H_X) k += 4 + o.x_size;
    o.off_y = k; if (hf & H_Y) k += 4 + o.y_size;
    o.off_n1 = k; k += o.n1_size;
    o.off_n2 = k; if (hf & H_NAME2) k += o.n2_size;
    o.off_st = k; k += o.st_size;
    o.off_seq = k; k += o.seq_size;
    o.off_qual = k; k += o.qual_size;
    o.off_ov = k; if (il && (hf & H_PE_OVERLAP)) k += s / 2;
    o.off_npos = k; if (hf & H_N_POS) k += o.npos_size;
    o.total = k;
    // mSize with the reference's accounting bug (Q1): tile bytes land in mLaneBufSize, mTileBufSize stays 0; the
    // name2-length / name2 / "tile" bytes are counted even when the header lacks NAME2 / TILE; lane bytes never are.
    const uint32_t laneBug = (fl & C_TILE_SAME) ? 2u : (il ? (2u * s) / 2u : 2u * s);
    uint32_t ms = 18 + readLenBuf + n1Len + n2Len + stLen + laneBug + o.n1_size + o.n2_size + o.st_size + o.seq_size + o.qual_size;
    if (il && (hf & H_PE_OVERLAP)) ms += s / 2;
    if (hf & H_N_POS) ms += 4 + o.npos_size;
    if (hf & H_X) ms += 4 + o.x_size;
    if (hf & H_Y) ms += 4 + o.y_size;
    o.msize = ms;
    L[c] = o; C.img_size[c] = k;
    if (exact && (hf & H_QUAL_BY_COL) && !(hf & H_DONT_QUAL)) {
        // reference scratch is int(totalReadLen * 1.5) bytes (src/rfqcodec.cpp:413): a larger payload overflows its heap
        const uint32_t lim = (uint32_t)((double)len * 1.5);
        if (qsz > lim) atomicOr(&st->err, (uint32_t)DE_QUAL_OVERFLOW);
    }
}

// dst[0..n) = src[0..n) by the threads t, t+NT, ... : bytes up to dst's 4-byte boundary, then ALIGNED dword stores fed by aligned dword
// loads + a funnel shift (src may sit at any phase), four of them in flight per thread, then the tail bytes.  src must be readable
// up to the next multiple of 4 past n (all callers copy out of 16-byte padded scratch buffers or out of the text itself).
__device__ __forceinline__ void copy_to_image(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n, uint32_t t, uint32_t NT) {
    uint32_t head = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u); if (head > n) head = n;
    if (t < head) dst[t] = src[t];
    const uint32_t body = (n - head) / 4;
    const uint8_t* sp = src + head; const uint32_t sh = (uint32_t)((uintptr_t)sp & 3u) * 8u;
    const uint32_t* sw = (const uint32_t*)(sp - ((uintptr_t)sp & 3u)); uint32_t* dw = (uint32_t*)(dst + head);
    for (uint32_t k0 = t; k0 < body; k0 += 4 * NT) {
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t k = k0 + (uint32_t)u * NT; lo[u] = hi[u] = 0; if (k < body) { lo[u] = sw[k]; if (sh) hi[u] = sw[k + 1]; } }
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t k = k0 + (uint32_t)u * NT; if (k < body) dw[k] = sh ? (uint32_t)((((uint64_t)hi[u] << 32) | lo[u]) >> sh) : lo[u]; }
    }
    const uint32_t done = head + 4 * body;
    if (t < n - done) dst[done + t] = src[done + t];
}
// grid (blocks_per_chunk, n_chunks): fixed fields, per-read arrays, coordinate streams, "same" names, packed bases,
// quality payload, overlap bytes, N positions.
__global__ void k_assemble(Text T, ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, const Layout* __restrict__ L,
                           const uint8_t* __restrict__ qcat, const uint32_t* __restrict__ spk, const uint8_t* __restrict__ scratch, const uint64_t* __restrict__ cbase,
                           const uint8_t* __restrict__ scratch_n, const uint64_t* __restrict__ cbase_n, const uint8_t* __restrict__ xs, const uint8_t* __restrict__ ys, const int8_t* __restrict__ ovb,
                           uint8_t* __restrict__ img, uint64_t img_cap, uint64_t img_base, uint64_t off1, uint64_t off2, uint64_t nolb1, uint64_t nolb2,
                           const uint32_t* __restrict__ segb, const uint32_t* __restrict__ segd, const uint32_t* __restrict__ segs, uint32_t n_seg, DevStatus* st,
                           uint32_t tail_bases, uint32_t tail_units, uint32_t tail_nl1, uint32_t tail_nl2, uint64_t tail_n1, uint64_t tail_n2) {
    const uint32_t c = blockIdx.y; const Layout o = L[c];
    const uint64_t at = img_base + C.img_off[c];
    if (at + o.total > img_cap) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(&st->err, 1u << 31); return; }
    uint8_t* out = img + at;
    const uint32_t f = C.first[c], s = o.n_reads, fl = o.flags, hf = D->flags, rlb = D->read_len_bytes, nn = D->n_normal;
    const bool il = (fl & C_PE_INTERLEAVED) != 0; const uint32_t h = il ? s / 2 : s, hs = il ? 2u : 1u;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, NT = gridDim.x * blockDim.x;
    const size_t k0 = (size_t)c * MAX_STREAMS;
    if (t == 0) {
        // line-break bits: set once the reference's reader has loaded the final (short) 1 MiB block (Q10)
        // A chunk is written right after its last record was read, so what counts is where that record ends.  The TAIL chunk of the input -
        // fewer than chunk_bases bases, written by the final flush (src/repaq.cpp:590-624, 715-761) - is written only after the reader(s) went
        // on and FAILED: what counts there is how far that last attempt got.  FastqReader::read (src/fastqreader.cpp:166-196) takes three lines,
        // gives up if one of them is empty, else takes the fourth; it reads to the end of the file when the lines run out (a truncated last
        // record included).  FastqReaderPair::read (:287-299) asks both files - or the one file twice - before it looks at either answer.
        // tail_bases = chunk_bases when this call ends the input (its end, or an empty line), else 0; tail_units = units encoded.
        uint32_t flags = fl;
        const uint32_t last = f + s - 1;
        const bool tail = tail_bases && c + 1 == gridDim.y && R.pq[f + s] - R.pq[f] < tail_bases;
        auto line_end = [&](int st_, size_t q) -> uint64_t { return T.ot[st_] ? (uint64_t)T.ot[st_][q] : (uint64_t)T.lo[st_][q + 1] - 1; };
        // one read() from line l0 on: where it stops (relative to the stream)
        auto attempt = [&](int st_, uint32_t l0, uint32_t nl, uint64_t n_, uint32_t& next) -> uint64_t {
            if (l0 + 2u >= nl) { next = nl; return n_; }                                       // fewer than three lines left: read to the end
            const uint32_t* lo_ = T.lo[st_]; bool e3 = false;
            for (uint32_t k = 0; k < 3; k++) if (lo_[l0 + k + 1] - 1u - lo_[l0 + k] == 0u) e3 = true;
            // (on text that was not normalised - only the encoded records were looked at - an empty line here may be a blank line the reader
            // swallows, src/fastqreader.cpp:112-114: the host repeats the call on the normalised text)
            if (!T.ot[st_] && (e3 || (l0 + 3u < nl && lo_[l0 + 4u] - 1u - lo_[l0 + 3u] == 0u))) atomicOr(&st->err, (uint32_t)DE_TAIL_BLANK);
            if (!e3 && l0 + 3u >= nl) { next = nl; return n_; }                                // the quality line is asked for at the end of the file
            const uint32_t lastl = e3 ? l0 + 2u : l0 + 3u; next = lastl + 1u;
            return line_end(st_, lastl);
        };
        if (T.paired == 1) {
            const size_t q = 4 * (size_t)(last >> 1) + 3;                  // the pair's quality lines
            uint64_t e1 = off1 + line_end(0, q), e2 = off2 + line_end(1, q);
            if (tail) { uint32_t nx; e1 = off1 + attempt(0, 4u * tail_units, tail_nl1, tail_n1, nx); e2 = off2 + attempt(1, 4u * tail_units, tail_nl2, tail_n2, nx); }
            if (e1 >= nolb1) flags |= C_NO_LB;
            if (e2 >= nolb2) flags |= C_NO_LB_R2;
        } else {
            const size_t q = 4 * (size_t)last + 3;
            uint64_t e1 = off1 + line_end(0, q);
            if (tail) {
                uint32_t nx; e1 = off1 + attempt(0, 4u * tail_units * T.upr, tail_nl1, tail_n1, nx);
                if (T.paired == 2) e1 = off1 + attempt(0, nx, tail_nl1, tail_n1, nx);          // the second mate is asked for whatever the first answered
            }
            if (e1 >= nolb1) { flags |= C_NO_LB; if (T.paired == 2) flags |= C_NO_LB_R2; }
        }
        st_u32(out, o.msize); st_u32(out + 4, s); st_u16(out + 8, flags); st_u32(out + 10, o.seq_size); st_u32(out + 14, o.qual_size);
        if (hf & H_N_POS) st_u32(out + 18, o.npos_size);
        if (fl & C_READ_LEN_SAME) { const uint32_t l0 = R.len[f]; for (uint32_t b = 0; b < rlb; b++) out[o.off_readlens + b] = (uint8_t)(l0 >> (8 * b)); }
        if (fl & C_NAME1_LEN_SAME) out[o.off_n1lens] = (uint8_t)R.name1_len[f];
        if ((hf & H_NAME2) && (fl & C_NAME2_LEN_SAME)) out[o.off_n2lens] = (uint8_t)name2_len_of(T, R, f);
        if (fl & C_STRAND_LEN_SAME) out[o.off_stlens] = (uint8_t)line_len(T, f, 2);
        if ((hf & H_LANE) && (fl & C_LANE_SAME)) out[o.off_lanes] = R.lane[f];
        if ((hf & H_TILE) && (fl & C_TILE_SAME)) st_u16(out + o.off_tiles, R.tile[f]);
        if (hf & H_X) st_u32(out + o.off_x, o.x_size);
        if (hf & H_Y) st_u32(out + o.off_y, o.y_size);
        if ((hf & H_QUAL_BY_COL) && !(hf & H_DONT_QUAL)) for (uint32_t j = 0; j < nn; j++) st_u32(out + o.off_qual + 4 * j, C.ssize[k0 + j]);
    }
    // per-read arrays
    if (!(fl & C_READ_LEN_SAME)) for (uint32_t i = t; i < s; i += NT) { const uint32_t v = R.len[f + i]; uint8_t* p = out + o.off_readlens + (size_t)i * rlb;
            for (uint32_t b = 0; b < rlb; b++) p[b] = (uint8_t)(v >> (8 * b)); }
    if (!(fl & C_NAME1_LEN_SAME)) for (uint32_t i = t; i < s; i += NT) out[o.off_n1lens + i] = (uint8_t)R.name1_len[f + i];
    if ((hf & H_NAME2) && !(fl & C_NAME2_LEN_SAME)) for (uint32_t i = t; i < s; i += NT) out[o.off_n2lens + i] = (uint8_t)name2_len_of(T, R, f + i);
    if (!(fl & C_STRAND_LEN_SAME)) for (uint32_t i = t; i < s; i += NT) out[o.off_stlens + i] = (uint8_t)line_len(T, f + i, 2);
    if ((hf & H_LANE) && !(fl & C_LANE_SAME)) for (uint32_t i = t; i < h; i += NT) out[o.off_lanes + i] = R.lane[f + (size_t)i * hs];
    if ((hf & H_TILE) && !(fl & C_TILE_SAME)) for (uint32_t i = t; i < h; i += NT) st_u16(out + o.off_tiles + 2 * (size_t)i, R.tile[f + (size_t)i * hs]);
    if (hf & H_X) copy_to_image(out + o.off_x + 4, xs + 3ull * f, o.x_size, t, NT);
    if (hf & H_Y) copy_to_image(out + o.off_y + 4, ys + 3ull * f, o.y_size, t, NT);
    // names / strand that are stored once
    if (fl & C_NAME1_SAME) { const uint8_t* src = line_ptr(T, f, 0); for (uint32_t i = t; i < o.n1_size; i += NT) out[o.off_n1 + i] = src[i]; }
    if ((hf & H_NAME2) && (fl & C_NAME2_SAME)) { const uint8_t* src = line_ptr(T, f, 0) + R.name2_off[f];
            for (uint32_t i = t; i < o.n2_size; i += NT) out[o.off_n2 + i] = src[i]; }
    if (fl & C_STRAND_SAME) { const uint8_t* src = line_ptr(T, f, 2); for (uint32_t i = t; i < o.st_size; i += NT) out[o.off_st + i] = src[i]; }
    // 2-bit bases (src/rfqcodec.cpp:590-604): k_seqpack / k_packbytes left the section's bytes in spk
    copy_to_image(out + o.off_seq, (const uint8_t*)(spk + (size_t)(C.sbase[c] >> 4)), o.seq_size, t, NT);
    // quality payload
    if (hf & H_DONT_QUAL) copy_to_image(out + o.off_qual, qcat + C.qbase[c], o.qual_size, t, NT);
    else if (hf & H_QUAL_BY_COL) {
        // a stream sits in its scratch area as one slot per coder segment (pc_seg_cap); the image wants the slots' bytes back to back,
        // normal streams in header order, then the exception records.  One wave per (stream, segment) piece.
        const uint8_t* sc = scratch + cbase[c]; const uint32_t qlen = R.pq[f + s] - R.pq[f];
        const uint32_t nw = NT >> 6, wv = t >> 6; const uint32_t l = t & 63u;
        for (uint32_t pc = wv; pc < (nn + 1) * n_seg; pc += nw) {
            const uint32_t jj = pc / n_seg, seg = pc - jj * n_seg, js = jj < nn ? jj : (uint32_t)EXC_SLOT; const size_t si0 = (k0 + js) * n_seg;
            const uint32_t sz = C.scap[k0 + js] ? segb[si0 + seg] : 0u;
            if (!sz) continue;                                               // wave-uniform
            (void)qlen;
            copy_to_image(out + o.off_qual + 4 * nn + segd[si0 + seg], sc + C.soff[k0 + js] + segs[si0 + seg], sz, l, 64u);
        }
    }
    if (il && (hf & H_PE_OVERLAP)) for (uint32_t i = t; i < s / 2; i += NT) out[o.off_ov + i] = (uint8_t)ovb[(f >> 1) + i];
    if ((hf & H_N_POS) && C.scap[k0 + NPOS_SLOT]) {
        const size_t si0 = (k0 + NPOS_SLOT) * n_seg; const uint32_t nw = NT >> 6, wv = t >> 6; const uint32_t l = t & 63u;
        for (uint32_t seg = wv; seg < n_seg; seg += nw) {
            const uint32_t sz = segb[si0 + seg]; if (!sz) continue;
            copy_to_image(out + o.off_npos + segd[si0 + seg], scratch_n + cbase_n[c] + C.soff[k0 + NPOS_SLOT] + segs[si0 + seg], sz, l, 64u);
        }
    }
}
// names / strands that differ inside the chunk: EIGHT lanes per read (a wave takes eight consecutive reads) copy its pieces to their prefix-sum
// offsets in 16-byte groups, byte-granular on both sides (consecutive reads' pieces are neighbours in the image, so a wave's stores still cover
// one contiguous span); the last group of a piece is moved back to end exactly at its end, pieces < 16 bytes go byte by byte
struct __attribute__((packed, aligned(1))) GU16 { uint32_t a, b, c, d; };
__device__ __forceinline__ void copy_piece8(uint8_t* __restrict__ d, const uint8_t* __restrict__ src, uint32_t n, uint32_t part) {
    if (n < 16u) { for (uint32_t i = part; i < n; i += 8u) d[i] = src[i]; return; }
    const uint32_t ng = (n + 15u) >> 4;
    for (uint32_t g = part; g < ng; g += 8u) { uint32_t p0 = 16u * g; if (p0 + 16u > n) p0 = n - 16u; *(GU16*)(d + p0) = *(const GU16*)(src + p0); }
}
__global__ void k_assemble_names(Text T, ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, const Layout* __restrict__ L, uint8_t* __restrict__ img,
        uint64_t img_cap, uint64_t img_base) {
    const uint32_t c = blockIdx.y; const uint32_t fl = C.flags[c];
    const bool need1 = !(fl & C_NAME1_SAME), need2 = (D->flags & H_NAME2) && !(fl & C_NAME2_SAME), need3 = !(fl & C_STRAND_SAME);
    if (!need1 && !need2 && !need3) return;
    const Layout o = L[c]; const uint64_t at = img_base + C.img_off[c];
    if (at + o.total > img_cap) return;
    uint8_t* out = img + at;
    const uint32_t f = C.first[c], e = f + o.n_reads; const uint32_t wpb = blockDim.x >> 6; const int l = lane_id();
    const U4 a = R.pv[f]; const uint32_t part = (uint32_t)l & 7u, sub = (uint32_t)l >> 3;
    for (uint32_t g0 = f + 8u * (blockIdx.x * wpb + (uint32_t)wave_id()); g0 < e; g0 += 8u * gridDim.x * wpb) {
        const uint32_t g = g0 + sub; if (g >= e) continue;
        const U4 p = R.pv[g]; const uint8_t* nm = line_ptr(T, g, 0);
        if (need1) copy_piece8(out + o.off_n1 + (p.a - a.a), nm, R.name1_len[g], part);
        if (need2) copy_piece8(out + o.off_n2 + (p.b - a.b), nm + R.name2_off[g], name2_len_of(T, R, g), part);
        if (need3) copy_piece8(out + o.off_st + (p.c - a.c), line_ptr(T, g, 2), line_len(T, g, 2), part);
    }
}
__global__ void k_enc_totals(ChunkTab C, const uint64_t* __restrict__ ctotal_prefix, uint32_t n_chunks, int which, DevStatus* st) {
    if (threadIdx.x || blockIdx.x) return;
    if (which == 0) { st->total_scratch = ctotal_prefix[n_chunks]; }                     // ctotal_prefix: the quality arena's
    else if (which == 2) { st->total_scratch_n = ctotal_prefix[n_chunks]; st->image_bound = C.img_off[n_chunks]; }   // ... the N arena's
    else st->total_image = C.img_off[n_chunks];
}
