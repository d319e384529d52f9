// enc/reads_cut.h - name parse, read lengths, chunk partition
// Part of rfq_encode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
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
#define LENS_BLK 8u               // words per block of k_read_lens' summary: shortest / longest unit (bases), longest record (bytes), longest / shortest read (bases)
// Every read of the batch has the same number of bases L - sequencer output, as a rule - : the base prefix of read g is g x L, no scan needed.  One workgroup reduces
// k_read_lens' block summaries to uni[0] = 1 / 0, uni[1] = L; the two prefix scans behind it (units, reads: six launches over 11 M / 22 M elements on the headline
// workload, 0.25 ms) look at uni[0] and leave at once, k_fill_pq writes the closed form instead, and k_partition's uniform branch never needs the unit prefix.
// (nu, here and in the kernels below: null, or the device word that holds the unit count - k_index_totals' st->idx_units - when the host launched without it)
__global__ void k_lens_uniform(const uint32_t* __restrict__ blk, uint32_t n_blk, uint32_t n_units, uint32_t* __restrict__ uni, const uint32_t* __restrict__ nu) {
    if (nu) n_units = *nu;
    __shared__ uint32_t s_a[16], s_b[16];
    uint32_t ml = 0, msl = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < n_blk; i += blockDim.x) { const uint32_t a = blk[LENS_BLK * i + 3], b = blk[LENS_BLK * i + 4]; if (a > ml) ml = a; if (b < msl) msl = b; }
    ml = wave_max(ml); msl = wave_min(msl);
    if (lane_id() == 0) { s_a[wave_id()] = ml; s_b[wave_id()] = msl; }
    __syncthreads();
    if (threadIdx.x == 0) { for (uint32_t i = 1; i < (blockDim.x >> 6); i++) { if (s_a[i] > ml) ml = s_a[i]; if (s_b[i] < msl) msl = s_b[i]; }
            uni[0] = (n_units > 0 && ml == msl && ml > 0) ? 1u : 0u; uni[1] = ml; }
}
__global__ void k_fill_pq(uint32_t* __restrict__ pq, uint32_t n_reads, const uint32_t* __restrict__ uni, const uint32_t* __restrict__ nu, uint32_t upr) {
    if (nu) n_reads = *nu * upr;
    if (!uni[0]) return;                                                     // (uniform: mixed lengths took the scan)
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, L = uni[1];
    if (g <= n_reads) pq[g] = g * L;                                         // (entry n_reads: the total; < 2^32: a call's stream is < 4 GiB of text)
}
__global__ void k_read_lens(Text T, uint32_t* __restrict__ len, uint32_t* __restrict__ stored, uint64_t* __restrict__ ulen, uint32_t n_units, uint32_t upr,
        uint32_t* __restrict__ blk_minmax, DevStatus* st, const uint32_t* __restrict__ nu) {
    if (nu) n_units = *nu;
    __shared__ uint32_t s_mn[4], s_mx[4], s_rc[4], s_ml[4], s_sl[4];
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t tot = 0; uint32_t rec = 0, ml = 0, msl = 0xFFFFFFFFu, err = 0, fe = 0xFFFFFFFFu;   // rec: bytes of the unit's longest record, ml: bases of its longest read
    if (u < n_units) {
        for (uint32_t j = 0; j < upr; j++) {
            const uint32_t g = u * upr + j; int s; uint32_t r; read_loc(T, g, s, r); const uint32_t* p = t_lo(T, s) + 4 * (size_t)r;
            const uint32_t p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4];
            const uint32_t nl = p1 - 1 - p0, sl = p2 - 1 - p1, tl = p3 - 1 - p2, ql = p4 - 1 - p3;
            if (nl == 0 || sl == 0 || tl == 0 || ql == 0) { err |= DE_EMPTY_LINE; if (g < fe) fe = g; }
            if (ql < sl) err |= DE_QUAL_SHORT;
            len[g] = sl; stored[g] = sl; tot += sl; if (p4 - p0 > rec) rec = p4 - p0; if (sl > ml) ml = sl; if (sl < msl) msl = sl;
        }
        ulen[u] = tot;
    }
    uint32_t mn = u < n_units ? (uint32_t)(tot > 0xFFFFFFFFull ? 0xFFFFFFFFu : tot) : 0xFFFFFFFFu, mx = u < n_units ? mn : 0u;
    mn = wave_min(mn); mx = wave_max(mx); rec = wave_max(rec); ml = wave_max(ml); msl = wave_min(msl); fe = wave_min(fe); err = wave_or(err);
    if (lane_id() == 0) { s_mn[wave_id()] = mn; s_mx[wave_id()] = mx; s_rc[wave_id()] = rec; s_ml[wave_id()] = ml; s_sl[wave_id()] = msl; if (err) { atomicOr(&st->err, err);
            if (fe != 0xFFFFFFFFu) atomicMin(&st->first_empty, fe); } }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t i = 1; i < (blockDim.x >> 6); i++) { if (s_mn[i] < mn) mn = s_mn[i]; if (s_mx[i] > mx) mx = s_mx[i]; if (s_rc[i] > rec) rec = s_rc[i];
                if (s_ml[i] > ml) ml = s_ml[i]; if (s_sl[i] < msl) msl = s_sl[i]; }
        uint32_t* const o = blk_minmax + LENS_BLK * blockIdx.x; o[0] = mn; o[1] = mx; o[2] = rec; o[3] = ml; o[4] = msl;
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
                            const uint32_t* __restrict__ blk_minmax, uint32_t n_blk, uint32_t* __restrict__ first, uint32_t cap_chunks, DevStatus* st, const uint32_t* __restrict__ uni,
                            int have_scans, const uint32_t* __restrict__ nu) {
    if (nu) n_units = *nu;
    const int l = lane_id();
    // have_scans = 0: the host launched neither prefix scan (it expects reads of one length - what a sequencer writes - and saves their six launches); reads of
    // several lengths say so and leave: the host runs the scans and this kernel once more
    if (!have_scans && !uni[0]) { if (threadIdx.x == 0) atomicOr(&st->err, (uint32_t)DE_NEED_SCAN); return; }
    // shortest / longest unit: every thread of the workgroup (1024: a single wave walked 44 k block entries in 158 us), then wave 0 goes on alone
    __shared__ uint32_t s_mn[16], s_mx[16], s_rc[16], s_ml[16];
    uint32_t len_minmax[2], max_rec, max_len;
    { uint32_t mn = 0xFFFFFFFFu, mx = 0, rc = 0, ml = 0;
      for (uint32_t i = threadIdx.x; i < n_blk; i += blockDim.x) { const uint32_t a = blk_minmax[LENS_BLK * i], b = blk_minmax[LENS_BLK * i + 1], r = blk_minmax[LENS_BLK * i + 2], m = blk_minmax[LENS_BLK * i + 3]; if (a < mn) mn = a; if (b > mx) mx = b; if (r > rc) rc = r; if (m > ml) ml = m; }
      mn = wave_min(mn); mx = wave_max(mx); rc = wave_max(rc); ml = wave_max(ml);
      if (l == 0) { s_mn[wave_id()] = mn; s_mx[wave_id()] = mx; s_rc[wave_id()] = rc; s_ml[wave_id()] = ml; }
      __syncthreads();
      if (wave_id() != 0) return;
      const uint32_t nw = blockDim.x >> 6; mn = (uint32_t)l < nw ? s_mn[l] : 0xFFFFFFFFu; mx = (uint32_t)l < nw ? s_mx[l] : 0u; rc = (uint32_t)l < nw ? s_rc[l] : 0u;
              ml = (uint32_t)l < nw ? s_ml[l] : 0u;
      len_minmax[0] = wave_min(mn); len_minmax[1] = wave_max(mx); max_rec = wave_max(rc); max_len = wave_max(ml); }
    uint32_t c = 0, start = 0, max_units = 0; uint64_t prevP = 0, max_bases = 0; bool closed = false;
    if (n_units > 0 && len_minmax[0] == len_minmax[1] && len_minmax[0] > 0) {
        closed = true;                                                     // (the unit prefix P may not exist: k_lens_uniform lets its scan return at once when every READ is L bases)
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
        // (the unit prefix P exists only when the scans ran: k_lens_uniform's "every read has L bases" lets them return at once, and implies the closed branch
        // above - the coupling is stated here instead of being left to two predicates that happen to agree)
        if (uni[0]) { if (l == 0) atomicOr(&st->err, (uint32_t)DE_INTERNAL); return; }
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
        st->total_bases = start ? (closed ? (uint64_t)start * len_minmax[0] : P[start - 1]) : 0; st->max_rec = max_rec; st->max_len = max_len;
                st->unit_bases = (n_units > 0 && len_minmax[0] == len_minmax[1]) ? len_minmax[0] : 0u;
    }
}
__global__ void k_chunk_ids(ChunkTab C, ReadTab R) {
    const uint32_t c = blockIdx.y; const uint32_t f = C.first[c], e = C.first[c + 1];
    const uint32_t g = f + blockIdx.x * blockDim.x + threadIdx.x;
    if (g < e) R.chunk[g] = c;
}
