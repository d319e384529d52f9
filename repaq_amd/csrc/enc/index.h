// enc/index.h - line index (one pass with a decoupled look-back; two-pass fallback) and the text normaliser of the slow path
// Part of rfq_encode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
// =============================================================== index
// 64 bytes per lane -> one u64 newline mask; 256 lanes = 16 KiB per workgroup.
__device__ __forceinline__ uint32_t eq_mask4(uint32_t w, uint32_t pat) {   // bit k set iff byte k of w == pat byte
    uint32_t v = w ^ pat;
    uint32_t t = ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);   // 0x80 in every zero byte, exact
    return (((t >> 7) * 0x00204081u) >> 21) & 0xFu;
}
// 0x80 in every byte of w that equals the pattern byte, exact (no borrow between bytes)
__device__ __forceinline__ uint32_t eq_flags4(uint32_t w, uint32_t pat) { const uint32_t v = w ^ pat; return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu); }
// bits 0-7 = bytes of (a, b) that equal the pattern byte: the 0x80 flags of the two words become mask bits (<< 7) by two chained dot products with weights 1, 2, 4, ... 128
// (v_dot4_u32_u8, full rate; the shift-or cascade this replaces took 8 instructions, a multiply runs at a quarter of the rate)
__device__ __forceinline__ uint32_t eq_mask8(uint32_t a, uint32_t b, uint32_t pat) {
    return udot4(eq_flags4(b, pat), 0x80402010u, udot4(eq_flags4(a, pat), 0x08040201u, 0u)) >> 7;
}
// bits 0-15 = bytes of q that equal the pattern byte (19 instructions for the 16 bytes; 32 with the cascade: k_gather2, bound by its instructions, 3.76 -> 3.59 ms)
__device__ __forceinline__ uint32_t eq_mask16c(const uint4& q, uint32_t pat) {
    const uint32_t lo = udot4(eq_flags4(q.y, pat), 0x80402010u, udot4(eq_flags4(q.x, pat), 0x08040201u, 0u));
    const uint32_t hi = udot4(eq_flags4(q.w, pat), 0x80402010u, udot4(eq_flags4(q.z, pat), 0x08040201u, 0u));
    return ((hi << 8) | lo) >> 7;
}
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
// What the host used to read back behind the index (line totals, last bytes) and do with it (k_line_tail, records -> units), on the device: the kernels up to the
// partition take their unit count from st->idx_units, the host learns everything with the partition's results - one round trip less per batch.  guess: the units the
// host sized the per-read tables for (from the bytes per record of the context's earlier batches); more than that: DE_UNITS_GUESS, the batch is repeated the slow way.
__global__ void k_index_totals(const uint32_t* __restrict__ tot0, const uint32_t* __restrict__ tot1, const uint8_t* __restrict__ fq0, uint32_t n0, const uint8_t* __restrict__ fq1, uint32_t n1,
                               uint32_t* lo0, uint32_t* lo1, int final_batch, int paired, uint32_t unit_cap, uint32_t guess, DevStatus* st) {
    if (threadIdx.x || blockIdx.x) return;
    uint32_t rec[2] = { 0, 0 };
    for (int s = 0; s < (paired == 1 ? 2 : 1); s++) {
        const uint32_t nl = s ? *tot1 : *tot0, n = s ? n1 : n0; const uint8_t* fq = s ? fq1 : fq0; uint32_t* lo = s ? lo1 : lo0;
        // (an unterminated tail is the file's last line only in the final batch)
        const uint32_t unterm = (final_batch && n > 0 && fq[n - 1] != '\n') ? 1u : 0u;
        if (unterm) lo[nl + 1] = n + 1;
        st->idx_lines[s] = n ? nl + unterm : 0u; rec[s] = n ? (nl + unterm) / 4 : 0u;
    }
    uint32_t units = paired == 0 ? rec[0] : (paired == 1 ? (rec[0] < rec[1] ? rec[0] : rec[1]) : rec[0] / 2);
    if (units > unit_cap) units = unit_cap;
    st->idx_units_true = units; st->idx_units = units < guess ? units : guess;
    if (units > guess) atomicOr(&st->err, (uint32_t)DE_UNITS_GUESS);
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
