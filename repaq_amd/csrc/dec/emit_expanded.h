// dec/emit_expanded.h - text emission of the expanded path (tile emitter k_dec_emit)
// Part of rfq_decode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
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
// 16 global bytes per lane -> 64 consecutive 16-byte groups at a wave-uniform LDS address, as the builtin issues it - but out of the compiler's sight (HIDDEN): it orders every
// later LDS access of the wave behind an LDS-DMA it knows of with s_waitcnt vmcnt(0), which is right for a tile's own data and defeats a request made a tile AHEAD, into
// buffers nothing reads until the next barrier but one.  The caller waits (any later load's wait covers it: the counter is in order) and synchronises.
template <bool HIDDEN> __device__ __forceinline__ void dma16(const uint8_t* g, uint4* lds) {
#ifndef RFQ_SIMT_EMULATION
    if (HIDDEN) {
        const uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)lds);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(la) : "memory", "m0");
        return;
    }
#endif
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
// The same by ONE wave (lane l of it): R rounds of 64 groups.  A tile's small spans are dealt out one per wave - a wave then runs the
// address arithmetic and the issue of its own span only.
template <int R, bool HIDDEN = false> __device__ __forceinline__ void span_dma_wave(const StageSpan& sp, int l) {
    const bool inside = sp.a0 + 16ull * sp.ng <= sp.lim;                     // wave-uniform
    const uint8_t* const gp = sp.g + sp.a0;
    uint32_t done = 0;
    if (inside) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t i = (uint32_t)l + 64u * (uint32_t)r;
            if (i < sp.ng) dma16<HIDDEN>(gp + 16u * i, sp.l + 64u * (uint32_t)r);
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
