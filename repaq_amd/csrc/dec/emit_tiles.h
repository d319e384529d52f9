// dec/emit_tiles.h - text emission, fused path: k_dec_emit3 (fixed tiles, no output tile)
// Part of rfq_decode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
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
#define E3_SHARED_OK 1             // 0: never take the SHARED instantiation (A/B on the box: tools/build_variant.sh)
#define E3_ALIGNED 1               // 0: the two long lines in line-relative 16-byte groups (A/B on the box: tools/build_variant.sh)
#define E3_PROBE_WRAP 0            // measurement aid (tools/build_variant.sh): a mask, e.g. 0xFFFFFu = every record is written into the first MiB of its output - WRONG text,
                                   // the emitter's time with its stores kept inside the L2 (profiles/r06_zn_emit_wrap_probe.txt: 2.53 instead of 3.67 ms)
#define E3_PREFETCH 0              // 1: the shared-pieces instantiation requests a tile's loads a tile AHEAD (below).  Built, bit-exact, measured (profiles/r06_zv*): the registers
                                   // it keeps across the compose take the kernel from six waves per SIMD to four (114 VGPRs; at five it spills, and a spill's reload is a wait), and
                                   // there it only wins back what the occupancy lost - 3.91 against 3.86 ms.  Off.
#define E3_L2PF 1                  // 0: no warming of the L2 for the next tile (A/B on the box: tools/build_variant.sh)
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
// SHARED: every chunk of the range shares its three name pieces among its reads (C_NAME1_SAME, C_NAME2_SAME, C_STRAND_SAME: what FastqMeta::parse leaves of a sequencer's
// names; the walk's summary knows - DecStatus::per_read_pieces == 0).  That instantiation carries no per-read piece prefixes: a tile's uniform parameters are four words
// instead of ten, three of the four staging waves' branches, the piece offsets and the fit tests of the pieces are gone - the tile loop kept ~110 wave-uniform values
// alive, more than a wave has SGPRs.
template <bool IMPL, uint32_t N1CAP = ET_N1CAP, bool SHARED = false> __global__ void __launch_bounds__(256, (N1CAP == ET_N1CAP && !(SHARED && E3_PREFETCH)) ? 5 : 4) k_dec_emit3(const uint8_t* __restrict__ img,
        const DChunk* __restrict__ CH, const DevHeader* __restrict__ D, DFused F,
                           uint64_t img_bytes, int split, uint8_t* __restrict__ out1, uint64_t cap1, uint8_t* __restrict__ out2, uint64_t cap2, DecStatus* st,
                           const plist_t* __restrict__ plist, const unsigned long long* __restrict__ loff, const uint32_t* __restrict__ nent, const uint32_t* __restrict__ cellidx,
                           uint32_t ncell, uint32_t nstr, uint32_t kshift, unsigned long long list_cap) {
    __shared__ uint4 t_q4[E3_QCAP / 16 + 6];                                // quality tile (16 bytes of slack in front, the rest behind)
    constexpr uint32_t NB = (SHARED && E3_PREFETCH) ? 2u : 1u;               // staging buffers of the packed bases and the middles: two where the next tile's are requested a tile ahead
    __shared__ uint4 t_pk4[NB][E3_QCAP / 64 + 6];                           // the tile's packed bases
    __shared__ uint32_t t_nb[E3_QCAP / 32 + 8];                             // one bit per stored base of the tile: is N
    // (N1CAP: E3_N1BIG for files with long per-read names)
    // (the shared-pieces instantiation stages ONE name1 / name2 / strand piece of at most 255 bytes each)
    __shared__ uint4 t_mid4_[NB][64 * E3_MIDROW / 16 + 5], t_n14_[(SHARED ? 256u : N1CAP) / 16 + 5], t_n24_[(SHARED ? 256u : ET_N2CAP) / 16 + 5], t_st4[(SHARED ? 256u : ET_STCAP) / 16 + 4];
    // (16 readable bytes in front of each: a 16-byte group of the name line may start before a piece)
    uint4* const t_n14 = t_n14_ + 1; uint4* const t_n24 = t_n24_ + 1;
    __shared__ unsigned long long s_loff[NPOS_SLOT + 2]; __shared__ uint32_t s_nent[NPOS_SLOT + 2], s_val[NPOS_SLOT + 2];
    __shared__ uint32_t s_g[2][NPOS_SLOT + 2], s_kb[2][NPOS_SLOT + 2];
    const uint32_t c = blockIdx.y; const DChunk d = CH[c]; const uint8_t* cp = img + d.off;
    const uint32_t fl = d.flags, hf = D->flags, f = d.rbase; const bool il = (fl & C_PE_INTERLEAVED) != 0;
    constexpr bool implied_n = IMPL;                                        // (the host instantiates by the header: N positions implied by the quality, or listed)
    const uint32_t nq4 = (D->n_base_qual & 0xFFu) * 0x01010101u, dpos = D->name2_diff_pos, dch = D->name2_diff_char;
    const int l = lane_id(), w = (int)uni32((uint32_t)wave_id()); const uint32_t tid = threadIdx.x;
    const size_t fp = (size_t)f + c;                                       // the chunk's entries of the chunk-local prefixes (one more than it has reads: see DFused)
    const U4 tb = F.tbase[c];
    const uint32_t K = 1u << kshift, pshift = 8u - kshift, P = 1u << pshift;
    uint32_t per = (d.reads + gridDim.x - 1) / gridDim.x; per = (per + K - 1u) & ~(K - 1u);                 // whole tiles per workgroup (K is even: pairs stay together)
    const uint32_t rs = blockIdx.x * per; const uint32_t re = rs + per < d.reads ? rs + per : d.reads;
    if (rs >= re) return;
    // (list_cap != 0: launched WITHOUT the host having looked at the status behind the list chain and the read table - rfq_decode.hip, "speculative" - : whatever the host
    // would have stopped at stops the kernel; the host looks afterwards, once, and takes the path it would have taken)
    if (list_cap && ((st->err & (uint32_t)(DE_CORRUPT | DE_CORRUPT_OV | DE_E3_RETRY)) || st->list_need > list_cap)) return;
    // (SHARED is only taken for files with coded qualities: raw quality bytes - more than 64 values, rare - keep the general instantiation)
    const bool raw = !SHARED && (hf & H_DONT_QUAL) != 0, bycol = SHARED || (!raw && (hf & H_QUAL_BY_COL));
    const uint32_t nn = bycol ? (D->n_normal < NPOS_SLOT ? D->n_normal : NPOS_SLOT) : 0u; const bool hasn = (hf & H_N_POS) != 0;
    const uint32_t T = nn + (hasn ? 1u : 0u);                                // streams of the tile: t < nn quality value t, t == nn the N positions
    const uint32_t major4 = (D->major & 0xFFu) * 0x01010101u;
    const uint32_t qlen_c = F.pql[fp + d.reads], slen_c = F.sdl[fp + d.reads];                // qualities / stored bases of the chunk
    const bool same1 = SHARED || (fl & C_NAME1_SAME) != 0, same2 = SHARED || (fl & C_NAME2_SAME) != 0, same3 = SHARED || (fl & C_STRAND_SAME) != 0;
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
        TileP t; t.q0 = uni32(F.pql[fp + r0]); t.q1 = uni32(F.pql[fp + r1]);
        t.s0 = uni32(F.sdl[fp + r0]); t.s1 = uni32(F.sdl[fp + r1]);
        if constexpr (SHARED) { t.a7 = t.a8 = t.a9 = t.e7 = t.e8 = t.e9 = 0u; }
        else { const U4 a = F.pvl[fp + r0], b = F.pvl[fp + r1];
               t.a7 = uni32(a.a); t.a8 = uni32(a.b); t.a9 = uni32(a.c); t.e7 = uni32(b.a); t.e8 = uni32(b.b); t.e9 = uni32(b.c); }
        return t;
    };
    // ---- PREFETCH (the shared-pieces instantiation): everything a tile reads from global memory - its packed bases and middles (LDS-DMA into the other pair of staging
    // buffers), its reads' table entries, the first rounds of its list entries - is requested BEFORE the tile in front of it is composed, and has landed when its turn comes.
    // Requested at the top of its own tile, the wait for it was 37 - 41 % of the kernel: one round trip to memory per tile - 5 us under the emitter's own store traffic - with
    // every wave of the workgroup in it at the same time (profiles/r06_zt_emit_phase_probe.txt, r06_zt2_emit_drain_time.txt: the drain of a tile's stores is 5 %, the rest is
    // the loads).  The other instantiations stage per-read name pieces too and keep the order they had.
    constexpr bool PF = SHARED && E3_PREFETCH;
    uint32_t cur = rs, pb = 0; uint32_t l2pf = 0; (void)l2pf;
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
    // a tile's spans in the image
    struct TileD { uint64_t n1a, n1e, n2a, n2e, sta, ste, pka, pke, rqa, rqe; };
    auto tile_spans = [&](const TileP& tp) -> TileD {
        TileD t; const uint64_t ib = d.off;
        t.n1a = ib + d.o_n1 + (same1 ? 0u : tp.a7); t.n1e = same1 ? t.n1a + d.n1_size : ib + d.o_n1 + tp.e7;
        t.n2a = ib + d.o_n2 + (same2 ? 0u : tp.a8); t.n2e = same2 ? t.n2a + d.n2_size : ib + d.o_n2 + tp.e8;
        t.sta = ib + d.o_st + (same3 ? 0u : tp.a9); t.ste = same3 ? t.sta + d.st_size : ib + d.o_st + tp.e9;
        t.pka = ib + d.o_seq + (tp.s0 >> 2); t.pke = ib + d.o_seq + ((tp.s1 + 3u) >> 2); { const uint64_t pend = ib + d.o_seq + d.seq_size; if (t.pke > pend) t.pke = pend;
                if (t.pka > t.pke) t.pka = t.pke; }
        t.rqa = ib + d.o_qual + tp.q0; t.rqe = ib + d.o_qual + tp.q1; { const uint64_t qend = ib + d.o_qual + d.qual_size; if (t.rqe > qend) t.rqe = qend;
                if (t.rqa > t.rqe) t.rqa = t.rqe; }
        return t;
    };
    auto tile_fits = [&](const TileP& tp) -> bool {
        return tp.q1 - tp.q0 <= E3_QCAP && (same1 || tp.e7 - tp.a7 + 32u <= N1CAP) && (same2 || tp.e8 - tp.a8 + 32u <= ET_N2CAP) && (same3 || tp.e9 - tp.a9 + 32u <= ET_STCAP);
    };
    // ---- stage: packed bases, middles, per-read name pieces (LDS-DMA); raw qualities for DONT_ENCODE_QUAL files
    auto stage = [&](const TileD& t, uint32_t g0_, uint32_t g1_, uint32_t bsel) {
        if (raw) span_dma<(int)((E3_QCAP / 16 + 4 + 255) / 256)>(make_span(t_q4 + 1, img, t.rqa, t.rqe, img_bytes, true));
        if (w == 0) span_dma_wave<(int)((E3_QCAP / 64 + 4 + 63) / 64), PF>(make_span(t_pk4[bsel], img, t.pka, t.pke, img_bytes, true), l);
        else if (w == 1) span_dma_wave<(int)((64 * E3_MIDROW / 16 + 4 + 63) / 64), PF>(make_span(t_mid4_[bsel] + 1, F.mid, (uint64_t)g0_ * E3_MIDROW, (uint64_t)g1_ * E3_MIDROW, ~0ull >> 1, true), l);
        else if (w == 2) { if (!same1) span_dma_wave<(int)((N1CAP / 16 + 4 + 63) / 64)>(make_span(t_n14, img, t.n1a, t.n1e, img_bytes, true), l); }
        else { if (!same2) span_dma_wave<(int)((ET_N2CAP / 16 + 4 + 63) / 64)>(make_span(t_n24, img, t.n2a, t.n2e, img_bytes, true), l);
               if (!same3) span_dma_wave<(int)((ET_STCAP / 16 + 4 + 63) / 64)>(make_span(t_st4, img, t.sta, t.ste, img_bytes, true), l); }
    };
    // ---- my read of a tile (P lanes share one): its table entries as they are stored (the tile's bases are taken off when the tile is composed)
    struct ReadM { uint32_t len, md, prevlen, sdl, pql, tx; int ov; U4 p4; uint8_t n1, n2, sl; };       // (the bytes stay bytes: widening them where they are loaded is a use - a wait)
    auto load_read = [&](uint32_t cur_, uint32_t cnt_) -> ReadM {
        ReadM m; m.len = m.n1 = m.n2 = m.sl = m.md = m.prevlen = m.sdl = m.pql = m.tx = 0; m.ov = 0; m.p4.a = m.p4.b = m.p4.c = m.p4.d = 0;
        const uint32_t r = cur_ + j;
        if (j < cnt_) {
            const uint32_t g_ = f + r; const uint2 t2 = F.tpl[g_];
            m.sdl = F.sdl[fp + r];
            if constexpr (!SHARED) m.p4 = F.pvl[fp + r];
            m.tx = t2.x; m.pql = F.pql[fp + r];
            m.len = F.len[g_]; m.ov = F.ov[g_]; m.prevlen = (r & 1u) ? F.len[g_ - 1] : 0u;
            m.n1 = cp[d.o_n1lens + ((fl & C_NAME1_LEN_SAME) ? 0u : r)]; m.n2 = (hf & H_NAME2) ? cp[d.o_n2lens + ((fl & C_NAME2_LEN_SAME) ? 0u : r)] : 0u;
            m.sl = cp[d.o_stlens + ((fl & C_STRAND_LEN_SAME) ? 0u : r)]; m.md = t2.y;
        }
        return m;
    };
    // ---- a tile's list entries: the first eight rounds of the quality lists (wave w takes the lists w, w + 4, ...) and two of the N list are requested here and scattered
    // behind the tile's first barrier; what is left of longer lists is fetched there
    struct ListS { uint32_t fe[8], fv[8], pn[2], nk0, nke, t, base, k0, ke, val; const plist_t* p; };
    auto ls_open = [&](ListS& L, uint32_t pb_, uint32_t t_) {
        const uint32_t g_ = uni32(s_g[pb_][t_]), b_ = uni32(s_kb[pb_][t_]), n_ = uni32(s_nent[t_]);
        L.k0 = g_ == 0xFFFFFFFFu ? 0u : g_; L.ke = g_ == 0xFFFFFFFFu ? 0u : (b_ < n_ ? b_ : n_); L.base = 0;
        L.val = uni32(s_val[t_]); L.p = plist + uni64(s_loff[t_]);
    };
    auto ls_round = [&](ListS& L, uint32_t pb_, uint32_t& e_, uint32_t& v_) {
        while (L.t < nn && L.k0 + L.base >= L.ke) { L.t += 4u; if (L.t < nn) ls_open(L, pb_, L.t); }
        if (L.t < nn) { const uint32_t kk = L.k0 + L.base + (uint32_t)l; if (kk < L.ke) e_ = L.p[kk]; v_ = L.val; L.base += 64u; }
    };
    auto load_lists = [&](uint32_t pb_) -> ListS {
        ListS L; L.t = (uint32_t)w; L.base = L.k0 = L.ke = L.val = 0; L.p = plist;
        if (L.t < nn) ls_open(L, pb_, L.t); else L.t = nn;
#pragma unroll
        for (int i = 0; i < 8; i++) { L.fe[i] = 0xFFFFFFFFu; L.fv[i] = 0; ls_round(L, pb_, L.fe[i], L.fv[i]); }
        L.pn[0] = L.pn[1] = 0xFFFFFFFFu; L.nk0 = 0xFFFFFFFFu; L.nke = 0;
        if (hasn) {
            L.nk0 = s_g[pb_][nn]; L.nke = s_kb[pb_][nn]; if (L.nke > s_nent[nn]) L.nke = s_nent[nn];
            const plist_t* lp = plist + s_loff[nn];
#pragma unroll
            for (int i = 0; i < 2; i++) { const uint32_t kk = L.nk0 + tid + 256u * (uint32_t)i; if (L.nk0 != 0xFFFFFFFFu && kk < L.nke) L.pn[i] = lp[kk]; }
        }
        return L;
    };
    __syncthreads();
    ReadM mN; ListS lN;
    if constexpr (PF) {
        if (tile_fits(tp_cur)) { const uint32_t cnt0 = re - cur < K ? re - cur : K; stage(tile_spans(tp_cur), f + cur, f + cur + cnt0, 0u); mN = load_read(cur, cnt0); lN = load_lists(0u); }
        else { mN = load_read(cur, 0u); lN = load_lists(0u); }
    }
    while (cur < re) {                                                       // block-uniform
        const uint32_t cnt = re - cur < K ? re - cur : K, g0 = f + cur, g1 = g0 + cnt;
        const TileP tp = tp_cur; const uint32_t q0 = tp.q0, q1 = tp.q1, s0 = tp.s0, s1 = tp.s1;
        // (the host sizes K by the longest read and by the chunks' average piece sizes; a tile whose pieces are longer than that allowed for: the host repeats the range
        // on the expanded path, k_dec_emit)
        if (!tile_fits(tp)) { if (tid == 0) atomicOr(&st->err, (uint32_t)DE_E3_RETRY); break; }
        const TileD td = tile_spans(tp); const uint32_t bs = PF ? pb : 0u;       // this tile's staging buffers
        const uint64_t n1a = td.n1a, n2a = td.n2a, sta = td.sta, pka = td.pka, pke = td.pke, rqa = td.rqa;
        if constexpr (!PF) { stage(td, g0, g1, 0u); mN = load_read(cur, cnt); lN = load_lists(pb); }
        const ReadM m = mN; ListS ls = lN;
        // qualities start as the major value (src/rfqcodec.cpp:1089), the N bits as none
        if (bycol) { uint4* qt = t_q4 + 1; const uint32_t ng = (q1 - q0 + 15u) >> 4;
                for (uint32_t i = tid; i < ng; i += blockDim.x) qt[i] = make_uint4(major4, major4, major4, major4); }
        for (uint32_t i = tid; i < ((s1 - s0 + 31u) >> 5) + 1u; i += blockDim.x) t_nb[i] = 0;
        // ---- my read
        const uint32_t r = cur + j; const bool on = j < cnt; const bool odd = (r & 1u) != 0, rc = il && odd, to2 = split && odd;
        const uint32_t len = m.len, n1 = m.n1, n2 = m.n2, sl = m.sl, md = m.md, prevlen = m.prevlen; const int ov = m.ov;
        uint32_t sp = 0, qp_ = 0, o7 = 0, o8 = 0, o9 = 0, toff = 0;
        if (on) { sp = m.sdl - s0; qp_ = m.pql - q0; toff = (to2 ? tb.b : tb.a) + m.tx; if constexpr (!SHARED) { o7 = m.p4.a - tp.a7; o8 = m.p4.b - tp.a8; o9 = m.p4.c - tp.a9; } }
        // ---- the next tile's parameters (consumed a tile from now)
        const uint32_t nxt = cur + cnt; TileP tp_n = tp;
        if (nxt < re) tp_n = tile_params(nxt, nxt + K < re ? nxt + K : re);
#ifndef RFQ_SIMT_EMULATION
        if constexpr (PF) __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the hidden LDS-DMA of this tile's staging buffers (requested a tile ago) has landed
#endif
        __syncthreads();
#if E3_L2PF && !defined(RFQ_SIMT_EMULATION)
        if (SHARED) asm volatile("" :: "v"(l2pf));                                      // (the warming load of a tile ago has returned: every load of this tile behind it has)
#endif
        uint8_t* const q_t = (uint8_t*)(t_q4 + 1) + (raw ? (uint32_t)(rqa & 15ull) : 0u);              // quality of chunk position q0 + i at q_t[i]
        const uint8_t* const pk = (const uint8_t*)t_pk4[bs] + (uint32_t)(pka & 15ull);                 // packed byte (s0 >> 2) + i at pk[i]
        const uint4* const t_mid4 = t_mid4_[bs] + 1;
        const uint32_t have = (uint32_t)(pke - pka), sbit0 = 2u * (s0 & 3u);                              // staged packed bytes; bit offset of stored base s0 in pk
        // ---- quality lists, exception records, N list into the tiles
        {
#pragma unroll
            // (an entry holds the low 16 bits of its position: see plist_t; 0xFFFFFFFF = no entry)
            for (int i = 0; i < 8; i++) { const uint32_t p = plist_pos(ls.fe[i], q0); if (p < q1) q_t[p - q0] = (uint8_t)ls.fv[i]; }
            while (ls.t < nn) {
                uint32_t e_[4], v_[4];
#pragma unroll
                for (int i = 0; i < 4; i++) { e_[i] = 0xFFFFFFFFu; v_[i] = 0; ls_round(ls, pb, e_[i], v_[i]); }
#pragma unroll
                for (int i = 0; i < 4; i++) { const uint32_t p = plist_pos(e_[i], q0); if (p < q1) q_t[p - q0] = (uint8_t)v_[i]; }
            }
            if (nrec) for (uint32_t i = tid; i < nrec; i += blockDim.x) { const uint8_t* rr = xrec + 5ull * i; const uint32_t pos = ld_u32(rr + 1);
                    if (pos >= q0 && pos < q1 && pos < qlen_c) q_t[pos - q0] = rr[0]; }
            if (hasn) {
                const uint32_t send = s1 < slen_c ? s1 : slen_c;
#pragma unroll
                for (int i = 0; i < 2; i++) { const uint32_t p = plist_pos(ls.pn[i], s0); if (p < send) atomicOr(&t_nb[(p - s0) >> 5], 1u << ((p - s0) & 31u)); }
                if (ls.nk0 != 0xFFFFFFFFu && ls.nk0 + 512u < ls.nke) { const plist_t* lp = plist + s_loff[nn];
                        for (uint32_t kk = ls.nk0 + 512u + tid; kk < ls.nke; kk += 256u) { const uint32_t p = plist_pos((uint32_t)lp[kk], s0);
                        if (p < send) atomicOr(&t_nb[(p - s0) >> 5], 1u << ((p - s0) & 31u)); } }
            }
        }
        // the next tile's list cells (its parameters have come back by now)
        if (nxt < re && tid < T) { s_g[pb ^ 1u][tid] = cell_lookup(tid, tp_n.q0, tp_n.s0, false); s_kb[pb ^ 1u][tid] = cell_lookup(tid, tp_n.q0, tp_n.s0, true); }
        __syncthreads();
        // the next tile's loads: on their way while this one is composed (the staging buffers this tile reads from are the other pair)
        if constexpr (PF) {
            if (nxt < re) {
                const uint32_t cntn = re - nxt < K ? re - nxt : K;
                if (tile_fits(tp_n)) { stage(tile_spans(tp_n), f + nxt, f + nxt + cntn, pb ^ 1u); mN = load_read(nxt, cntn); lN = load_lists(pb ^ 1u); }
            }
        }
        // ---- the next tile's lines into the L2, on their way while this tile is composed: one 4-byte load per thread and 128-byte line of what the next tile will ask for at its
        // top - packed bases, middles, its reads' table entries, its lists' ranges (the first four streams and the N list) - into a register nothing reads.  The round trip at the
        // top of a tile - 37 - 41 % of the kernel, see above - then ends in the L2 instead of behind the emitter's own stores in the queue to memory; unlike E3_PREFETCH this keeps
        // ONE register across the compose (the load's target must stay allocated until the load has returned: it is "used" behind the next tile's first wait).
#if E3_L2PF && !defined(RFQ_SIMT_EMULATION)
        if (SHARED && nxt < re) {                                           // (the shared-pieces instantiation has the register to spare at six waves per SIMD)
            const uint32_t cntn = re - nxt < K ? re - nxt : K; const TileD tn = tile_spans(tp_n);
            const uint8_t* pb_ = nullptr; uint32_t nbytes = 0;              // my region: nbytes from pb_ on; my line of it
            const uint32_t grp = tid >> 5, li = tid & 31u; uint32_t line = li;  // 32 threads = 32 lines = 4 KiB per region
            if (grp == 0) { pb_ = img + tn.pka; nbytes = (uint32_t)(tn.pke - tn.pka); }
            else if (grp == 1) { pb_ = F.mid + (size_t)(f + nxt) * E3_MIDROW; nbytes = cntn * E3_MIDROW; }
            else if (grp == 2) {                                             // the reads' table entries: five arrays, four lines of each
                const uint32_t a = li >> 2; line = li & 3u;
                if (a == 0) { pb_ = (const uint8_t*)(F.tpl + f + nxt); nbytes = cntn * 8u; }
                else if (a == 1) { pb_ = (const uint8_t*)(F.sdl + fp + nxt); nbytes = cntn * 4u; }
                else if (a == 2) { pb_ = (const uint8_t*)(F.pql + fp + nxt); nbytes = cntn * 4u; }
                else if (a == 3) { pb_ = (const uint8_t*)(F.len + f + nxt); nbytes = cntn * 4u; }
                else if (a == 4) { pb_ = (const uint8_t*)(F.ov + f + nxt); nbytes = cntn * 4u; }
            }
            else {                                                           // groups 3 .. 7: the lists of streams 0 .. 3 and the N list
                const uint32_t t_ = grp - 3u, st_ = t_ < 4u ? t_ : nn;       // (index among the tile's streams; group 7 = the N list)
                if ((t_ < 4u && t_ < nn) || (t_ == 4u && hasn)) {
                    const uint32_t g_ = s_g[pb ^ 1u][st_], b_ = s_kb[pb ^ 1u][st_], n_ = s_nent[st_]; const uint32_t ke = b_ < n_ ? b_ : n_;
                    if (g_ != 0xFFFFFFFFu && g_ < ke) { pb_ = (const uint8_t*)(plist + s_loff[st_] + g_); nbytes = (ke - g_) * (uint32_t)sizeof(plist_t); }
                }
            }
            if (nbytes) {
                const uint8_t* const a0 = (const uint8_t*)((uintptr_t)pb_ & ~(uintptr_t)127); const uint32_t span = (uint32_t)(pb_ - a0) + nbytes;
                if (128u * line < span) { const uint8_t* q = a0 + 128u * line; asm volatile("global_load_dword %0, %1, off" : "=v"(l2pf) : "v"(q)); }
            }
        }
#endif
        // ---- compose: my share of my read's four lines, straight to the output (src/rfqcodec.cpp:1141-1254, Read::toString src/read.cpp:170)
        if (on) {
            uint8_t* const rec = (to2 ? out2 : out1) + (E3_PROBE_WRAP ? (toff & (uint32_t)E3_PROBE_WRAP) : toff); const uint64_t capo = to2 ? cap2 : cap1;
            const uint32_t e0 = n1 + md + n2, oseq = e0 + 1u, ost = oseq + len + 1u, oq = ost + sl + 1u, total = oq + len + 1u;
            if ((uint64_t)toff + total > capo) { if (part == 0) atomicOr(&st->err, 1u << 31); }
            else {
                // The name line = name1 + middle + name2 + '\n' (three LDS pieces at arbitrary byte offsets): a lane builds a whole 16-byte group of the line
                // in registers - one unaligned 16-byte LDS read per piece the group touches, later pieces laid over the earlier ones from their first byte
                // on - and stores it once.  (Piece by piece this was ~10 partial stores per read: 1.4 ms of the kernel's 5.1 on 2 x 4 GB.)
                const uint32_t L = e0 + 1u; const bool nfast = L >= 16u, jfast = sl == 1u && len >= 16u;
                const uint8_t* const src1 = (const uint8_t*)t_n14 + (uint32_t)(n1a & 15ull) + (same1 ? 0u : o7);
                const uint8_t* const src2 = (const uint8_t*)t_mid4 + (uint32_t)(((uint64_t)g0 * E3_MIDROW) & 15ull) + E3_MIDROW * j;
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
                auto group_q = [&](uint32_t k0, uint32_t (&qw)[4]) {                  // output positions [k0, k0 + 16) of the quality line (k0 + 16 <= len)
                    const uint32_t pa = rc ? len - k0 - 16u : k0;
                    lds_get16(q_t, qp_ + pa, qw);
                    if (rc) { const uint32_t x0 = bswap32(qw[3]), x1 = bswap32(qw[2]), x2 = bswap32(qw[1]), x3 = bswap32(qw[0]); qw[0] = x0; qw[1] = x1; qw[2] = x2; qw[3] = x3; }
                };
                // ... of the bases' line; qw: the qualities of the same positions (group_q's), looked at by files whose N bases are implied by their quality
                auto group_s = [&](uint32_t k0, const uint32_t (&qw)[4], uint32_t (&sw)[4]) {
                    const uint32_t pa = rc ? len - k0 - 16u : k0;
                    uint32_t cw, nw;
                    if (pa + 16u <= xa) fetch(A + pa, cw, nw);
                    else if (pa >= xa) fetch(Bs + (pa - xa), cw, nw);
                    else { uint32_t c2, n2_; const uint32_t t1 = xa - pa; fetch(A + pa, cw, nw); fetch(Bs, c2, n2_);
                            cw = (cw & ((1u << (2u * t1)) - 1u)) | (c2 << (2u * t1)); nw = (nw & ((1u << t1) - 1u)) | ((n2_ << t1) & 0xFFFFu); }
                    if (rc) { cw = ~e3_rev2x16(cw); if (nw) nw = e3_rev1x16(nw); }
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
                auto group = [&](uint32_t k0, uint32_t (&qw)[4], uint32_t (&sw)[4]) { group_q(k0, qw); group_s(k0, qw, sw); };     // both lines' positions [k0, k0 + 16)
                auto st16 = [&](uint32_t at, const uint32_t (&x)[4]) { GU16d v; v.a = x[0]; v.b = x[1]; v.c = x[2]; v.d = x[3]; *(GU16d*)(rec + at) = v; };
                auto put = [&](uint32_t at_q, uint32_t at_s, const uint32_t (&qw)[4], const uint32_t (&sw)[4], bool both) {
                    GU16d v;
                    if (both) { v.a = qw[0]; v.b = qw[1]; v.c = qw[2]; v.d = qw[3]; *(GU16d*)(rec + at_q) = v; }
                    v.a = sw[0]; v.b = sw[1]; v.c = sw[2]; v.d = sw[3]; *(GU16d*)(rec + at_s) = v;
                };
                if (len >= 16u) {
                    uint32_t qw[4], sw[4];
                    // Where the lines' 16-byte groups are cut.  By the OUTPUT address (E3_ALIGNED, lines of 32 bases and more): one group at the line's start, then groups at the
                    // 16-aligned addresses inside it, then the tail that ends with the line - the body's stores are aligned, which is what the memory side wants: byte-granular
                    // 16-byte stores in this record shape reach 2.7 TB/s, with the two long lines' bodies aligned 3.8 (tools/micro/store_align.hip, profiles/r06_zo_store_align.txt),
                    // and the emitter with its writes kept inside the L2 runs 2.5 instead of 3.7 ms (profiles/r06_zn_emit_wrap_probe.txt): a third of it is the store path.  The two
                    // lines start at different alignments, so each has its own groups (bases first, then qualities: one path per wave at a time).
                    const bool al = E3_ALIGNED && len >= 32u;
                    uint32_t rs, tail_part;                                  // bases between the last whole group and the line's end; the lane that writes the tails
                    if (al) {
                        const uint32_t hs = (16u - (uint32_t)((uintptr_t)(rec + oseq) & 15u)) & 15u, hq = (16u - (uint32_t)((uintptr_t)(rec + oq) & 15u)) & 15u;
                        const uint32_t nts = ((len - hs) >> 4) + (hs ? 1u : 0u), ntq = ((len - hq) >> 4) + (hq ? 1u : 0u);       // head (if the line does not start aligned) + body
                        for (uint32_t t = part; t < nts; t += P) { const uint32_t k0 = hs ? (t ? hs + 16u * (t - 1u) : 0u) : 16u * t;
                                if (implied_n) group_q(k0, qw);
                                group_s(k0, qw, sw); st16(oseq + k0, sw); }
                        const uint32_t p2 = (part + P - (nts & (P - 1u))) & (P - 1u);                            // (the lane behind the bases' last group takes the qualities' first)
                        for (uint32_t t = p2; t < ntq; t += P) { const uint32_t k0 = hq ? (t ? hq + 16u * (t - 1u) : 0u) : 16u * t; group_q(k0, qw); st16(oq + k0, qw); }
                        rs = (len - hs) & 15u; tail_part = (nts + ntq) & (P - 1u);
                        if (!jfast && tail_part == part) {                    // (a strand line of its own: plain tails where the body left bytes)
                            const uint32_t rq = (len - hq) & 15u;
                            if (rs || rq) { group(len - 16u, qw, sw); if (rq) st16(oq + len - 16u, qw); if (rs) st16(oseq + len - 16u, sw); }
                        }
                    } else {
                        const uint32_t nfull = len >> 4;
                        for (uint32_t gi = part; gi < nfull; gi += P) { group(16u * gi, qw, sw); put(oq + 16u * gi, oseq + 16u * gi, qw, sw, true); }
                        rs = len & 15u; tail_part = nfull & (P - 1u);
                        if (!jfast && tail_part == part && rs) { group(len - 16u, qw, sw); put(oq + len - 16u, oseq + len - 16u, qw, sw, true); }
                    }
                    if (jfast && tail_part == part) {                         // the lines' tails, positions [len - 16, len), with what follows the lines riding on them
                        group(len - 16u, qw, sw);
                        {
                            if (rs > 13u) put(0u, oseq + len - 16u, qw, sw, false);                        // (the shifted store below starts behind the last whole group)
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
