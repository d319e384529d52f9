// rfq_decode.hip — host orchestration of the gfx950 RFQ -> FASTQ path (rfq_decode_batch of include/rfq_hip.h).
// Replaces, per image: RfqHeader::read (src/rfqheader.cpp:19-43), RfqChunk::read (src/rfqchunk.cpp:161-228),
// RfqCodec::decodeChunk (src/rfqcodec.cpp:826-1260) and the text emission of Repaq::decompress / decompressPE
// (src/repaq.cpp:262-417; final-newline rule :301-328,375-413).
#include "rfq_ctx.h"
#include "rfq_decode_kernels.h"
#include <algorithm>
#include <cstring>
#include <cstdlib>

int rfq_upload_header(rfq_ctx* c, const uint8_t* h, size_t n);   // rfq_encode.hip

enum DecBuf {   // indices into rfq_ctx::b (disjoint from the encoder's, so one context can alternate encode / decode)
    DB_CHUNKS = 80, DB_STATUS, DB_LEN, DB_CHUNKID, DB_OV, DB_PVIN, DB_PV, DB_PQ, DB_TIN, DB_TP, DB_QBASE, DB_SBASE, DB_QDEC, DB_SDEC, DB_XV, DB_YV, DB_SCAN, DB_MID, DB_SEGF, DB_SEGA, DB_SEGN, DB_SEGS, DB_SEGP, DB_OFFT, DB_CELL, DB_SEGK, DB_NENT, DB_LOFF, DB_PLIST, DB_GWCAND, DB_GWLIST, DB_GWCNT, DB_GWLAND, DB_PQL, DB_PVL, DB_TPL, DB_CTEXT, DB_TBASE, DB_SDL, DB_END
};
static_assert(DB_END <= 120, "rfq_ctx::b too small");

// fused path: k_dec_pos_list for the quality streams and the N-position stream (the arena is whatever B[DB_PLIST] holds)
static void launch_pos_list(rfq_ctx* ctx, const rfq_decode_args* a, const DChunk* CH, uint32_t n_chunks, uint32_t maxseg, uint32_t ncell, uint32_t nstr,
                            uint32_t mq, uint32_t mn, uint32_t nn, bool hasn, uint32_t segb, hipStream_t LS) {
    DBuf* B = ctx->b; const DevHeader* D = ctx->d_hdr.as<DevHeader>(); const DecStatus* dst = B[DB_STATUS].as<DecStatus>();
    const unsigned long long cap = B[DB_PLIST].cap / sizeof(plist_t);
#define RFQ_LIST_ARGS a->d_rfq, CH, D, (const uint8_t*)B[DB_SEGS].as<uint8_t>(), (const int*)B[DB_SEGP].as<int>(), (const uint32_t*)B[DB_SEGK].as<uint32_t>(), \
                      (const unsigned long long*)B[DB_LOFF].as<unsigned long long>(), B[DB_PLIST].as<plist_t>(), cap, B[DB_CELL].as<uint32_t>(), maxseg, ncell, (uint64_t)a->n
    if (nn) hipLaunchKernelGGL(k_dec_pos_list, dim3((mq + 3) / 4, nn, n_chunks), dim3(256), 0, LS, RFQ_LIST_ARGS, 0u, nstr, dst, segb);
    if (hasn) hipLaunchKernelGGL(k_dec_pos_list, dim3((mn + 3) / 4, 1, n_chunks), dim3(256), 0, LS, RFQ_LIST_ARGS, ctx->h_hdr.n_normal, nstr, dst, segb);
#undef RFQ_LIST_ARGS
}
#define RFQ_RANGE_TOO_BIG 2          // internal: the range's text would not fit the 32-bit text offsets of one pass
// bases: sum of the range's read lengths (the walk's 64-bit total)
struct DecRange { const DChunk* CH; uint32_t n_chunks, n_reads, max_reads, max_stream, max_npos, max_len, max_bases, max_nrec, max_one, pieces, piece_avg, piece_n1;
        uint64_t bases; };
// RfqCodec::decodeChunk + Read::toString for the chunks of one range (reads, bases and text of a range are placed by 32-bit prefix sums).
// out1 / out2: caller buffers (16-byte aligned) or null = the context's own result buffers; *p1 / *p2 = where the text went.
// force_expanded: the retry of a range whose fused emit raised DE_E3_RETRY - straight to the expanded path, whatever made the tile test fail (the retry
// must terminate by construction, not because a flag happens to be honoured: ADVICE r4)
static int decode_range(rfq_ctx* ctx, const rfq_decode_args* a, const DecRange& g, uint8_t* out1, uint64_t ocap1, uint8_t* out2, uint64_t ocap2,
                        uint8_t** p1, uint8_t** p2, size_t* n1, size_t* n2, uint64_t* nbases, bool force_expanded = false) {
    hipStream_t S = ctx->stream; DBuf* B = ctx->b;
    // (RFQ_MATERIALISE: the expanding decode of a streaming caller's non-final slices, on every call)
    const int tune = ctx->opt.materialise ? 2048 : 0;
    const DevHeader& HH = ctx->h_hdr; const DevHeader* D = ctx->d_hdr.as<DevHeader>();
    DecStatus* dst = B[DB_STATUS].as<DecStatus>(); DecStatus hs; memset(&hs, 0, sizeof hs);
    hs.max_stream = g.max_stream; hs.max_npos = g.max_npos;
    const uint32_t n_chunks = g.n_chunks, n_reads = g.n_reads, max_reads = std::max(g.max_reads, 1u);
    const DChunk* CH = g.CH;
    HIPCHK(ctx, hipMemsetAsync(dst, 0, sizeof(DecStatus), S));
    // ---- the position-list chain of the fused path (rfq_decode_kernels.h "fused path"): the emitter builds qualities and bases tile by tile in LDS
    // from the packed bytes and lists of coded positions; the lists need nothing but the chunk table, so their chain (stream summaries, link,
    // offsets, lists + cell index) starts NOW on the second stream, beside read table, prefixes, coordinates and text lengths, and is joined in
    // front of the last status read-back before the emitter.  Taken for files with few quality streams whose reads and exception lists fit a
    // tile; RFQ_MATERIALISE=1 forces the materialising path.
    const bool bycol_h = (HH.flags & H_QUAL_BY_COL) && !(HH.flags & H_DONT_QUAL);
    const bool rle_h = !(HH.flags & (H_DONT_QUAL | H_QUAL_BY_COL));          // legacy run-length quality coding: k_dec_rle on the materialising path
    // The fused path ends in k_dec_emit3 (no output tile, K reads per tile: the largest power of two whose qualities fit its tile).  Names FastqMeta::parse does not
    // take apart and strand lines with text are stored per read: K then also goes by the chunks' average piece size, a fifth of the tile left for reads above the
    // average, full tiles of 64 reads only (name1 has a second instantiation with a 13 KB tile: names of up to ~165 bytes on average, e.g. the configs[4] shape).
    // What does not fit - longer names, reads of more than 2000 bases, chunks of more than 4096 exception records, legacy run-length images - takes the ONE
    // fallback: qualities and bases expanded in HBM, k_dec_emit.  A tile whose pieces turn out longer than the averages allowed for raises DE_E3_RETRY: the range
    // is decoded again on the fallback (force_expanded: the retry cannot come back here), and the context remembers it for the ranges that follow until
    // the next header is set or cleared.
    uint32_t e3k = 6; bool n1big = false;
    while (e3k >= 1 && ((uint64_t)g.max_len << e3k) > E3_QCAP) e3k--;
    if (g.pieces) {
        while (e3k >= 1 && ((uint64_t)g.piece_avg << e3k) > 204u) e3k--;
        const uint64_t need1 = ((uint64_t)g.piece_n1 << e3k) * 5u / 4u + 32u;
        if (need1 > ET_N1CAP) { if (e3k == 6 && need1 <= E3_N1BIG) n1big = true; else e3k = 0; }
    }
    const bool e3_ok = e3k >= (g.pieces ? 6u : 1u) && !(g.pieces && ctx->e3_pieces_failed);
    // (chunks of 2^30 bases and more: the list chain's saturating position sums - POS_ADV_MAX, dec/pos_lists.h - would no longer be exact)
    const bool fused = !force_expanded && !(tune & 2048) && !rle_h && g.max_len <= 2000u && g.max_nrec <= 4096u && e3_ok && g.max_bases < (1u << 30) - 65536u;
    uint32_t f_maxseg = 1, f_ncell = 1, f_mq = 0, f_mn = 0, f_nn = 0, f_segb = POS2_SEG; bool f_lists = false, f_hasn = false, f_join = false; hipStream_t f_aux = S;
            const uint32_t f_nstr = HH.n_normal + 1;
    // (an early return must not leave the chain running over buffers the next call reuses)
    struct AuxJoin { rfq_ctx* c; bool armed; ~AuxJoin() { if (armed) (void)hipStreamSynchronize(c->aux); } } aux_guard = { ctx, false };
    if (fused) {
        const bool hasn = (HH.flags & H_N_POS) != 0; const uint32_t nn = bycol_h ? std::min<uint32_t>(HH.n_normal, NPOS_SLOT) : 0u;
        const bool forked = ctx->aux_ready(); hipStream_t A = forked ? ctx->aux : S;
        if (nn || hasn) {
            if (forked) { HIPCHK(ctx, hipEventRecord(ctx->ev_fork, S)); HIPCHK(ctx, hipStreamWaitEvent(A, ctx->ev_fork, 0)); aux_guard.armed = true; }
            f_segb = ctx->opt.pos_seg ? (uint32_t)ctx->opt.pos_seg : (std::max(g.max_one, g.max_npos) >= 32768u ? POS2_SEG_BIG : POS2_SEG);
            const uint32_t mq = nn ? g.max_one / f_segb + 1 : 0u, mn = hasn ? g.max_npos / f_segb + 1 : 0u; f_maxseg = std::max(1u, std::max(mq, mn));
            f_ncell = g.max_bases / POS2_CELL + 2;
            const size_t nst = (size_t)n_chunks * f_nstr, nseg = nst * f_maxseg, ncl = nst * f_ncell;
            HIPCHK(ctx, B[DB_SEGF].ensure(nseg + 16)); HIPCHK(ctx, B[DB_SEGA].ensure(nseg * 32 + 16)); HIPCHK(ctx, B[DB_SEGN].ensure(nst * 4 + 16));
            HIPCHK(ctx, B[DB_SEGS].ensure(nseg + 16)); HIPCHK(ctx, B[DB_SEGP].ensure(nseg * 4 + 16)); HIPCHK(ctx, B[DB_SEGK].ensure(nseg * 4 + 16));
                    HIPCHK(ctx, B[DB_CELL].ensure(ncl * 4 + 16));
            HIPCHK(ctx, B[DB_NENT].ensure(nst * 4 + 16)); HIPCHK(ctx, B[DB_LOFF].ensure(nst * 8 + 16));
            // the arena of the position lists: a position belongs to at most one stream, the coded ones are a few percent of the bases; if a
            // file needs more than the arena holds, k_dec_pos_list leaves it alone and the pass is repeated below with the right size
            // (an entry per eight bases to begin with - half a byte per base; a NovaSeq-binned file codes one position in twelve.  A file that codes more
            // - forty quality values code most positions - runs the list pass a second time, once per context: the arena keeps its size.  Round 3 asked for
            // 2 bytes per base up front: 6.7 GB on 2 x 4 GB of text)
            if (B[DB_PLIST].ensure((size_t)(g.bases / 8 + 1024) * sizeof(plist_t)) != hipSuccess) { (void)hipGetLastError();
                    HIPCHK(ctx, B[DB_PLIST].ensure((size_t)(g.bases / 32 + 1024) * sizeof(plist_t))); }
            { ClearList z; memset(&z, 0, sizeof z); z.add(B[DB_SEGN].p, nst * 4, 0u); z.add(B[DB_NENT].p, nst * 4, 0u); z.add(B[DB_CELL].p, ncl * 4, 0xFFFFFFFFu); clear_list(A, z); }
#define RFQ_SUM2_ARGS a->d_rfq, CH, D, B[DB_SEGF].as<uint8_t>(), B[DB_SEGA].as<int>(), B[DB_SEGN].as<uint32_t>(), f_maxseg, dst, (uint64_t)a->n
            if (nn) hipLaunchKernelGGL(k_dec_pos_sum2, dim3((mq + 3) / 4, nn, n_chunks), dim3(256), 0, A, RFQ_SUM2_ARGS, 0u, f_nstr, f_segb);
            if (hasn) hipLaunchKernelGGL(k_dec_pos_sum2, dim3((mn + 3) / 4, 1, n_chunks), dim3(256), 0, A, RFQ_SUM2_ARGS, HH.n_normal, f_nstr, f_segb);
#undef RFQ_SUM2_ARGS
            hipLaunchKernelGGL(k_dec_pos_link2, dim3((uint32_t)((nst + 3) / 4)), dim3(256), 0, A, (const uint8_t*)B[DB_SEGF].as<uint8_t>(),
                    (const int*)B[DB_SEGA].as<int>(),
                               (const uint32_t*)B[DB_SEGN].as<uint32_t>(), B[DB_SEGS].as<uint8_t>(), B[DB_SEGP].as<int>(), B[DB_SEGK].as<uint32_t>(), B[DB_NENT].as<uint32_t>(), f_maxseg, (uint32_t)nst,
                               CH, f_nstr, dst);
            hipLaunchKernelGGL(k_dec_pos_off, dim3(1), dim3(256), 0, A, (const uint32_t*)B[DB_NENT].as<uint32_t>(), B[DB_LOFF].as<unsigned long long>(), (uint32_t)nst,
                    dst);
            launch_pos_list(ctx, a, CH, n_chunks, f_maxseg, f_ncell, f_nstr, mq, mn, nn, hasn, f_segb, A);
            f_lists = true; f_mq = mq; f_mn = mn; f_nn = nn; f_hasn = hasn;
            f_join = forked; f_aux = A;
            KCHK(ctx, "k_dec_pos_*");
        }
    }
    // ---- read table + prefixes
    ctx->timer.begin("read_table", S);
    const size_t nr = (size_t)n_reads + 2, nc = (size_t)n_chunks + 2;
    HIPCHK(ctx, B[DB_LEN].ensure(nr * 4)); HIPCHK(ctx, B[DB_OV].ensure(nr * 4));
    HIPCHK(ctx, B[DB_XV].ensure(nr * 4)); HIPCHK(ctx, B[DB_YV].ensure(nr * 4)); HIPCHK(ctx, B[DB_MID].ensure(nr * 40));
    DReadTab R; memset(&R, 0, sizeof R); DFused F; memset(&F, 0, sizeof F);
    R.len = B[DB_LEN].as<uint32_t>(); R.ov = B[DB_OV].as<int32_t>(); R.mid = B[DB_MID].as<uint8_t>();
    uint32_t total_bases = 0; U4 pv_tot; memset(&pv_tot, 0, sizeof pv_tot);
    if (fused) {
        // chunk-local prefixes made where the per-read values are made (k_dec_readtab2): no per-read scan inputs, no batch-wide scans, no read-back here - the
        // status word is looked at with the next one, the range's bases are the walk's 64-bit total
        HIPCHK(ctx, B[DB_PQL].ensure((nr + nc) * 4)); HIPCHK(ctx, B[DB_TPL].ensure(nr * 8));
        if (g.pieces || !E3_SHARED_OK || (HH.flags & H_DONT_QUAL)) HIPCHK(ctx, B[DB_PVL].ensure((nr + nc) * 16));
        HIPCHK(ctx, B[DB_CTEXT].ensure(nc * 16)); HIPCHK(ctx, B[DB_TBASE].ensure(nc * 16)); HIPCHK(ctx, B[DB_SCAN].ensure((nc / SCAN_TILE + 2) * 16 + 1024));
        // (the per-read piece prefixes only where some chunk stores pieces per read: sequencer names leave none, and the stored-base prefix alone is 4 bytes per read instead of 16)
        const bool shared_pieces = E3_SHARED_OK && !g.pieces && !(HH.flags & H_DONT_QUAL);
        HIPCHK(ctx, B[DB_SDL].ensure((nr + nc) * 4));
        F.len = R.len; F.ov = R.ov; F.pql = B[DB_PQL].as<uint32_t>(); F.pvl = shared_pieces ? (const U4*)nullptr : B[DB_PVL].as<U4>(); F.sdl = B[DB_SDL].as<uint32_t>();
        F.tpl = B[DB_TPL].as<uint2>(); F.tbase = B[DB_TBASE].as<U4>(); F.mid = R.mid;
        hipLaunchKernelGGL(k_dec_readtab2, dim3(n_chunks), dim3(256), 0, S, a->d_rfq, CH, D, R.len, R.ov, B[DB_PQL].as<uint32_t>(), shared_pieces ? (U4*)nullptr : B[DB_PVL].as<U4>(),
                           B[DB_SDL].as<uint32_t>(), dst);
        KCHK(ctx, "k_dec_readtab2");
        ctx->timer.end(S);
        *nbases = g.bases;
    } else {
        HIPCHK(ctx, B[DB_CHUNKID].ensure(nr * 4));
        HIPCHK(ctx, B[DB_PVIN].ensure(nr * 16)); HIPCHK(ctx, B[DB_PV].ensure(nr * 16)); HIPCHK(ctx, B[DB_PQ].ensure(nr * 4));
        HIPCHK(ctx, B[DB_TIN].ensure(nr * 16)); HIPCHK(ctx, B[DB_TP].ensure(nr * 16));
        HIPCHK(ctx, B[DB_QBASE].ensure(nc * 8)); HIPCHK(ctx, B[DB_SBASE].ensure(nc * 8));
        HIPCHK(ctx, B[DB_SCAN].ensure((nr / SCAN_TILE + 2) * 16 + 1024));
        R.chunk = B[DB_CHUNKID].as<uint32_t>();
        R.pvin = B[DB_PVIN].as<U4>(); R.pv = B[DB_PV].as<U4>(); R.pq = B[DB_PQ].as<uint32_t>(); R.tin = B[DB_TIN].as<U4>(); R.tp = B[DB_TP].as<U4>();
        hipLaunchKernelGGL(k_dec_readtab, dim3((max_reads + 255) / 256, n_chunks), dim3(256), 0, S, a->d_rfq, CH, D, R, dst);
        KCHK(ctx, "k_dec_readtab");
        scan_exclusive<uint32_t>(S, R.len, R.pq, n_reads, B[DB_SCAN].as<uint32_t>(), 1);
        scan_exclusive<U4>(S, R.pvin, R.pv, n_reads, B[DB_SCAN].as<U4>(), 1);
        HIPCHK(ctx, ctx->fetch(&total_bases, R.pq + n_reads, 4, S));
        HIPCHK(ctx, ctx->fetch(&pv_tot, R.pv + n_reads, 16, S));
        { DecStatus h2; HIPCHK(ctx, ctx->fetch(&h2, dst, sizeof h2, S)); HIPCHK(ctx, ctx->fetch_sync(S)); hs.err = h2.err; }
        ctx->timer.end(S);
        if (hs.err & DE_CORRUPT) return rfq_fail(ctx, RFQ_E_FORMAT, "corrupt overlap buffer");
        *nbases = total_bases;
    }

    // ---- streams (fused path: the coordinate decoder here, the list chain has been running since the top)
    if (fused) {
        ctx->timer.begin("streams", S);
        hipLaunchKernelGGL(k_dec_coords, dim3(2, n_chunks), dim3(64), 0, S, a->d_rfq, CH, D, B[DB_XV].as<uint32_t>(), B[DB_YV].as<uint32_t>());
        KCHK(ctx, "k_dec_coords");
    }
    uint64_t* qbase = nullptr; uint64_t* sbase = nullptr; uint8_t* qdec = nullptr; uint8_t* sdec = nullptr; size_t qbytes = 0, sbytes = 0;
    if (!fused) {
    ctx->timer.begin("streams", S);
    qbytes = (size_t)total_bases + 64 * nc + 256; sbytes = (size_t)pv_tot.d + 64 * nc + 256;
    HIPCHK(ctx, B[DB_QDEC].ensure(qbytes)); HIPCHK(ctx, B[DB_SDEC].ensure(sbytes));
    qbase = B[DB_QBASE].as<uint64_t>(); sbase = B[DB_SBASE].as<uint64_t>();
    qdec = B[DB_QDEC].as<uint8_t>(); sdec = B[DB_SDEC].as<uint8_t>();
    hipLaunchKernelGGL(k_dec_bases, dim3((n_chunks + 255) / 256), dim3(256), 0, S, CH, R, qbase, sbase, n_chunks);
    // Two chains run side by side (rfq_ctx::aux, fork / join by events):
    //   main  prefill of qdec, coordinate decoder, [summaries linked] quality position streams scattered into qdec, exception records
    //   aux   position-stream summaries + link, 2-bit unpack into sdec, N positions scattered into sdec
    const bool forked = ctx->aux_ready();
    hipStream_t A = forked ? ctx->aux : S;
    if (forked) { HIPCHK(ctx, hipEventRecord(ctx->ev_fork, S)); HIPCHK(ctx, hipStreamWaitEvent(A, ctx->ev_fork, 0)); }
    const uint32_t bpc = std::max(1u, std::min(64u, 4096u / n_chunks));
    // the prefill is pure bandwidth, the coordinate decoder and the stream summaries are pure latency: third chain
    hipStream_t F = forked ? ctx->aux2 : S;
    if (forked) HIPCHK(ctx, hipStreamWaitEvent(F, ctx->ev_fork, 0));
    // (many short blocks: slots keep turning over for the two other chains)
    hipLaunchKernelGGL(k_dec_fill, dim3((uint32_t)std::min<size_t>(8192, (qbytes / 16 + 255) / 256 + 1)), dim3(256), 0, F, qdec, (uint64_t)qbytes, D);
    if (forked) HIPCHK(ctx, hipEventRecord(ctx->ev_f, F));
    hipLaunchKernelGGL(k_dec_coords, dim3(2, n_chunks), dim3(64), 0, S, a->d_rfq, CH, D, B[DB_XV].as<uint32_t>(), B[DB_YV].as<uint32_t>());
    if ((HH.flags & H_N_POS) || ((HH.flags & H_QUAL_BY_COL) && !(HH.flags & H_DONT_QUAL))) {
        // position streams in POS_SEG-byte segments: summary -> link -> emit (see rfq_decode_kernels.h)
        // (grid sizes from the largest quality / N-position section: a file with raw qualities has no quality streams to walk)
        const bool bycol = (HH.flags & H_QUAL_BY_COL) && !(HH.flags & H_DONT_QUAL), hasn = (HH.flags & H_N_POS) != 0;
        const uint32_t nn = bycol ? std::min<uint32_t>(HH.n_normal, NPOS_SLOT) : 0u, nstr = HH.n_normal + 1;
        const uint32_t mq = nn ? hs.max_stream / POS_SEG + 1 : 0u, mn = hasn ? hs.max_npos / POS_SEG + 1 : 0u, maxseg = std::max(1u, std::max(mq, mn));
        const size_t nseg = (size_t)n_chunks * nstr * maxseg;
        HIPCHK(ctx, B[DB_SEGF].ensure(nseg + 16)); HIPCHK(ctx, B[DB_SEGA].ensure(nseg * 16 + 16)); HIPCHK(ctx, B[DB_SEGN].ensure((size_t)n_chunks * nstr * 4 + 16));
        HIPCHK(ctx, B[DB_SEGS].ensure(nseg + 16)); HIPCHK(ctx, B[DB_SEGP].ensure(nseg * 4 + 16));
        HIPCHK(ctx, hipMemsetAsync(B[DB_SEGN].p, 0, (size_t)n_chunks * nstr * 4, A));
#define RFQ_SUM_ARGS a->d_rfq, CH, D, R, (const uint64_t*)qbase, (const uint64_t*)sbase, qdec, sdec, B[DB_SEGF].as<uint8_t>(), B[DB_SEGA].as<int>(), B[DB_SEGN].as<uint32_t>(), maxseg, dst, (uint64_t)a->n
        if (nn) hipLaunchKernelGGL(k_dec_pos_sum, dim3(mq, nn, n_chunks), dim3(64), 0, A, RFQ_SUM_ARGS, 0u, nstr);
        if (hasn) hipLaunchKernelGGL(k_dec_pos_sum, dim3(mn, 1, n_chunks), dim3(64), 0, A, RFQ_SUM_ARGS, HH.n_normal, nstr);
#undef RFQ_SUM_ARGS
        hipLaunchKernelGGL(k_dec_pos_link, dim3((n_chunks * nstr + 255) / 256), dim3(256), 0, A, (const uint8_t*)B[DB_SEGF].as<uint8_t>(),
                (const int*)B[DB_SEGA].as<int>(),
                           (const uint32_t*)B[DB_SEGN].as<uint32_t>(), B[DB_SEGS].as<uint8_t>(), B[DB_SEGP].as<int>(), maxseg, n_chunks * nstr);
        if (forked) { HIPCHK(ctx, hipEventRecord(ctx->ev_mid, A)); HIPCHK(ctx, hipStreamWaitEvent(S, ctx->ev_mid, 0)); HIPCHK(ctx, hipStreamWaitEvent(S, ctx->ev_f, 0)); }
        hipLaunchKernelGGL(k_dec_unpack, dim3(bpc, n_chunks), dim3(256), 0, A, a->d_rfq, CH, R, (const uint64_t*)sbase, sdec, (uint64_t)a->n);
        if (nn) hipLaunchKernelGGL(k_dec_pos_emit, dim3(mq, nn, n_chunks), dim3(64), 0, S, a->d_rfq, CH, D, R, (const uint64_t*)qbase, (const uint64_t*)sbase, qdec, sdec,
                                   (const uint8_t*)B[DB_SEGS].as<uint8_t>(), (const int*)B[DB_SEGP].as<int>(), maxseg, (uint64_t)a->n, 0u, nstr);
        if (hasn) hipLaunchKernelGGL(k_dec_pos_emit, dim3(mn, 1, n_chunks), dim3(64), 0, A, a->d_rfq, CH, D, R, (const uint64_t*)qbase, (const uint64_t*)sbase, qdec,
                sdec,
                                     (const uint8_t*)B[DB_SEGS].as<uint8_t>(), (const int*)B[DB_SEGP].as<int>(), maxseg, (uint64_t)a->n, HH.n_normal, nstr);
    } else hipLaunchKernelGGL(k_dec_unpack, dim3(bpc, n_chunks), dim3(256), 0, A, a->d_rfq, CH, R, (const uint64_t*)sbase, sdec, (uint64_t)a->n);
    if (forked) { HIPCHK(ctx, hipEventRecord(ctx->ev_join, A)); HIPCHK(ctx, hipStreamWaitEvent(S, ctx->ev_join, 0)); HIPCHK(ctx, hipStreamWaitEvent(S, ctx->ev_f, 0)); }
    if (rle_h) hipLaunchKernelGGL(k_dec_rle, dim3(1, n_chunks), dim3(256), 0, S, a->d_rfq, CH, D, R, (const uint64_t*)qbase, qdec);
    hipLaunchKernelGGL(k_dec_except, dim3(bpc, n_chunks), dim3(256), 0, S, a->d_rfq, CH, D, R, (const uint64_t*)qbase, qdec);
    KCHK(ctx, "k_dec_streams");
    ctx->timer.end(S);

    }

    // ---- text
    if (!fused) ctx->timer.begin("textlen", S);                          // (fused path: still inside "streams", beside the list chain)
    const int split = a->split_pe ? 1 : 0;
    U4 tt;
    if (fused) {
        hipLaunchKernelGGL(k_dec_textlen2, dim3(n_chunks), dim3(256), 0, S, a->d_rfq, CH, D, (const uint32_t*)R.len, R.mid, (const uint32_t*)B[DB_XV].as<uint32_t>(),
                (const uint32_t*)B[DB_YV].as<uint32_t>(), split, B[DB_TPL].as<uint2>(), B[DB_CTEXT].as<U4>(), dst);
        KCHK(ctx, "k_dec_textlen2");
        scan_exclusive<U4>(S, B[DB_CTEXT].as<U4>(), B[DB_TBASE].as<U4>(), n_chunks, B[DB_SCAN].as<U4>(), 1);      // where every chunk's text starts; entry n_chunks: the totals
    } else {
        hipLaunchKernelGGL(k_dec_textlen, dim3((max_reads + 255) / 256, n_chunks), dim3(256), 0, S, a->d_rfq, CH, D, R, (const uint32_t*)B[DB_XV].as<uint32_t>(),
                (const uint32_t*)B[DB_YV].as<uint32_t>(), split, dst);
        scan_exclusive<U4>(S, R.tin, R.tp, n_reads, B[DB_SCAN].as<U4>(), 1);
    }
    if (f_join) { HIPCHK(ctx, hipEventRecord(ctx->ev_join, f_aux)); HIPCHK(ctx, hipStreamWaitEvent(S, ctx->ev_join, 0)); }
    aux_guard.armed = false;       // (everything below is ordered behind the chain on the main stream)
    // Speculative emit (fused path, the caller's output buffers, lists that were built): the emitter is launched WITHOUT this read-back - it stops itself on what the checks
    // below would have stopped at (k_dec_emit3: list_cap) -, and the host looks once, behind it: one round trip less per range (~35 us: 3 % of a 1 GB decode).
    const bool spec = fused && f_lists && out1 && (!a->split_pe || out2) && !((uintptr_t)out1 & 15u) && !(out2 && ((uintptr_t)out2 & 15u)) && !ctx->opt.no_spec;
    bool late = false;
    auto pre_emit_sync = [&]() -> int {
        HIPCHK(ctx, ctx->fetch(&tt, fused ? B[DB_TBASE].as<U4>() + n_chunks : R.tp + n_reads, 16, S));
        HIPCHK(ctx, ctx->fetch(&hs, dst, sizeof hs, S));
        HIPCHK(ctx, ctx->fetch_sync(S));
        return RFQ_OK;
    };
    if (!spec) { const int rc = pre_emit_sync(); if (rc) return rc; }
    else {
        ctx->timer.end(S);
        ctx->timer.begin("emit", S);
        const uint32_t K = 1u << e3k, b3 = grid_x_for(n_chunks, (max_reads + K - 1) / K, 6u * ctx->n_cu);
        const unsigned long long lcap = B[DB_PLIST].cap / sizeof(plist_t);
#define RFQ_EMIT3_SPEC a->d_rfq, CH, D, F, (uint64_t)a->n, a->split_pe ? 1 : 0, out1, ocap1, out2, ocap2, dst, (const plist_t*)B[DB_PLIST].as<plist_t>(), (const unsigned long long*)B[DB_LOFF].as<unsigned long long>(), \
                       (const uint32_t*)B[DB_NENT].as<uint32_t>(), (const uint32_t*)B[DB_CELL].as<uint32_t>(), f_ncell, f_nstr, e3k, lcap
        if (n1big) {
            const uint32_t b4 = grid_x_for(n_chunks, (max_reads + K - 1) / K, 4u * ctx->n_cu);
            if (HH.flags & H_N_POS) hipLaunchKernelGGL((k_dec_emit3<false, E3_N1BIG>), dim3(b4, n_chunks), dim3(256), 0, S, RFQ_EMIT3_SPEC);
            else hipLaunchKernelGGL((k_dec_emit3<true, E3_N1BIG>), dim3(b4, n_chunks), dim3(256), 0, S, RFQ_EMIT3_SPEC);
        }
        else if (F.pvl == nullptr) {
            if (HH.flags & H_N_POS) hipLaunchKernelGGL((k_dec_emit3<false, ET_N1CAP, true>), dim3(b3, n_chunks), dim3(256), 0, S, RFQ_EMIT3_SPEC);
            else hipLaunchKernelGGL((k_dec_emit3<true, ET_N1CAP, true>), dim3(b3, n_chunks), dim3(256), 0, S, RFQ_EMIT3_SPEC);
        }
        else if (HH.flags & H_N_POS) hipLaunchKernelGGL(k_dec_emit3<false>, dim3(b3, n_chunks), dim3(256), 0, S, RFQ_EMIT3_SPEC);
        else hipLaunchKernelGGL(k_dec_emit3<true>, dim3(b3, n_chunks), dim3(256), 0, S, RFQ_EMIT3_SPEC);
#undef RFQ_EMIT3_SPEC
        KCHK(ctx, "k_dec_emit");
        ctx->timer.end(S);
        const int rc = pre_emit_sync(); if (rc) return rc;
        late = true;
    }
    if (hs.err & DE_CORRUPT_OV) return rfq_fail(ctx, RFQ_E_FORMAT, "corrupt overlap buffer");   // (k_dec_readtab2's verdict, a bit of its own: nothing reads the status in between, and the list chain's DE_CORRUPT means the quality buffer - the same two messages as on the expanded path)
    // a stream codes positions far beyond its chunk's length table (k_dec_pos_link2): 16-bit list entries would alias - the range goes to the expanded path
    if (fused && (hs.err & DE_E3_RETRY)) return decode_range(ctx, a, g, out1, ocap1, out2, ocap2, p1, p2, n1, n2, nbases, true);
    if (hs.err & DE_CORRUPT) return rfq_fail(ctx, RFQ_E_FORMAT, "corrupt quality buffer");
    // text prefix sums are 32-bit: one decode call emits < 4 GiB per output stream
    if (fused && f_lists && hs.list_need > B[DB_PLIST].cap / sizeof(plist_t)) {            // the lists did not fit the arena: now that their size is known, build them
        if (late) { late = false; ctx->timer.begin("streams", S); }                            // (the speculative emitter left at once: the ordinary order from here on)
        HIPCHK(ctx, B[DB_PLIST].ensure((size_t)(hs.list_need + 1024) * sizeof(plist_t)));
        launch_pos_list(ctx, a, CH, n_chunks, f_maxseg, f_ncell, f_nstr, f_mq, f_mn, f_nn, f_hasn, f_segb, S);
        KCHK(ctx, "k_dec_pos_list");
    }
    hs.text1 = hs.text2 = 0; for (int i = 0; i < 64; i++) { hs.text1 += hs.text_slots[0][i]; hs.text2 += hs.text_slots[1][i]; }
    if (hs.text1 >= 0xFFFFFFF0ull || hs.text2 >= 0xFFFFFFF0ull) return RFQ_RANGE_TOO_BIG;       // (the caller decodes the range in two halves)
    uint8_t *o1, *o2; uint64_t cap1, cap2;
    if ((out1 && ((uintptr_t)out1 & 15u)) || (out2 && ((uintptr_t)out2 & 15u))) return rfq_fail(ctx, RFQ_E_ARG, "output device pointers must be 16-byte aligned");
    if (out1) { o1 = out1; cap1 = ocap1; } else { HIPCHK(ctx, ctx->out_fq1.ensure((size_t)tt.a + 64)); o1 = ctx->out_fq1.as<uint8_t>(); cap1 = ctx->out_fq1.cap; }
    if (out2) { o2 = out2; cap2 = ocap2; } else { HIPCHK(ctx, ctx->out_fq2.ensure((size_t)tt.b + 64)); o2 = ctx->out_fq2.as<uint8_t>(); cap2 = ctx->out_fq2.cap; }
    if (late) {                                                        // the emitter has run: what is left of the ordinary path below is its own verdict
        if (hs.err & DE_E3_RETRY) { if (g.pieces) ctx->e3_pieces_failed = true; return decode_range(ctx, a, g, out1, ocap1, out2, ocap2, p1, p2, n1, n2, nbases, true); }
        ctx->timer.collect();
        if (hs.err & (1u << 31)) return rfq_fail(ctx, RFQ_E_NOSPACE, "output buffer too small: need %u / %u bytes", tt.a, tt.b);
        *n1 = tt.a; *n2 = split ? tt.b : 0; *p1 = o1; *p2 = o2;
        return RFQ_OK;
    }
    ctx->timer.end(S);
    ctx->timer.begin(fused ? "emit" : "emit_expanded", S);             // the emitter alone: the path's largest kernel (bench.py roofline)
    {
        if (fused) {
            const uint32_t K = 1u << e3k, b3 = grid_x_for(n_chunks, (max_reads + K - 1) / K, 6u * ctx->n_cu);       // (25 KB of LDS: six workgroups per CU)
#define RFQ_EMIT3_ARGS a->d_rfq, CH, D, F, (uint64_t)a->n, split, o1, cap1, o2, cap2, dst, (const plist_t*)B[DB_PLIST].as<plist_t>(), (const unsigned long long*)B[DB_LOFF].as<unsigned long long>(), \
                       (const uint32_t*)B[DB_NENT].as<uint32_t>(), (const uint32_t*)B[DB_CELL].as<uint32_t>(), f_ncell, f_nstr, e3k, 0ull
            if (n1big) {
                const uint32_t b4 = grid_x_for(n_chunks, (max_reads + K - 1) / K, 4u * ctx->n_cu);
                if (HH.flags & H_N_POS) hipLaunchKernelGGL((k_dec_emit3<false, E3_N1BIG>), dim3(b4, n_chunks), dim3(256), 0, S, RFQ_EMIT3_ARGS);
                else hipLaunchKernelGGL((k_dec_emit3<true, E3_N1BIG>), dim3(b4, n_chunks), dim3(256), 0, S, RFQ_EMIT3_ARGS);
            }
            else if (F.pvl == nullptr) {                                          // (every chunk shares its name pieces among its reads: the instantiation without per-read piece prefixes)
                if (HH.flags & H_N_POS) hipLaunchKernelGGL((k_dec_emit3<false, ET_N1CAP, true>), dim3(b3, n_chunks), dim3(256), 0, S, RFQ_EMIT3_ARGS);
                else hipLaunchKernelGGL((k_dec_emit3<true, ET_N1CAP, true>), dim3(b3, n_chunks), dim3(256), 0, S, RFQ_EMIT3_ARGS);
            }
            else if (HH.flags & H_N_POS) hipLaunchKernelGGL(k_dec_emit3<false>, dim3(b3, n_chunks), dim3(256), 0, S, RFQ_EMIT3_ARGS);
            else hipLaunchKernelGGL(k_dec_emit3<true>, dim3(b3, n_chunks), dim3(256), 0, S, RFQ_EMIT3_ARGS);
#undef RFQ_EMIT3_ARGS
        } else {
            const uint32_t bx = grid_x_for(n_chunks, (max_reads + ET_READS - 1) / ET_READS, 4u * ctx->n_cu);   // (39 KB of LDS: four workgroups per CU)
            hipLaunchKernelGGL(k_dec_emit, dim3(bx, n_chunks), dim3(256), 0, S, a->d_rfq, CH, D, R, (const uint64_t*)qbase, (const uint64_t*)sbase,
                               (const uint8_t*)qdec, (const uint8_t*)sdec, (uint64_t)qbytes, (uint64_t)sbytes, (uint64_t)a->n, split, o1, cap1, o2, cap2, dst);
        }
        KCHK(ctx, "k_dec_emit");
        ctx->timer.end(S);
        HIPCHK(ctx, ctx->fetch(&hs, dst, sizeof hs, S));
        HIPCHK(ctx, ctx->fetch_sync(S));
        if (fused && (hs.err & DE_E3_RETRY)) {                             // per-read name pieces that did not fit a tile: once more, expanded
            if (g.pieces) ctx->e3_pieces_failed = true;                    // (files like this one: the ranges that follow go straight to the expanded path)
            return decode_range(ctx, a, g, out1, ocap1, out2, ocap2, p1, p2, n1, n2, nbases, true);
        }
    }
    ctx->timer.collect();
    if (hs.err & (1u << 31)) return rfq_fail(ctx, RFQ_E_NOSPACE, "output buffer too small: need %u / %u bytes", tt.a, tt.b);
    *n1 = tt.a; *n2 = split ? tt.b : 0; *p1 = o1; *p2 = o2;
    return RFQ_OK;
}

extern "C" int rfq_decode_batch(rfq_ctx* ctx, const rfq_decode_args* a, rfq_decode_result* res) {
    if (!ctx || !a || !res) return RFQ_E_ARG;
    memset(res, 0, sizeof *res);
    ctx->err.clear();
    if (a->n && !a->d_rfq) return rfq_fail(ctx, RFQ_E_ARG, "null rfq pointer");
    if (a->n == 0 && a->has_header) {
        // an empty .rfq: every read of RfqHeader::read fails and the constructor's defaults stand (valid magic, flags 0;
        // src/rfqheader.cpp:7-17,19-43), no chunk follows: the reference writes an empty FASTQ
        if (a->split_pe) return rfq_fail(ctx, RFQ_E_DATA, "The input RFQ file was encoded by single-end FASTQ, you should not specify <out2>");
        return RFQ_OK;
    }
    hipStream_t S = ctx->stream; DBuf* B = ctx->b;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->timer.reset();
    ctx->pend.clear(); ctx->pin_used = 0;                                   // (read-backs an earlier call left behind on an error path)
    uint64_t start = 0;
    if (a->has_header) {
        uint8_t hb[RFQ_HEADER_MAX]; const size_t take = std::min<size_t>(a->n, sizeof hb);
        if (take) HIPCHK(ctx, hipMemcpy(hb, a->d_rfq, take, hipMemcpyDeviceToHost));
        int rc = rfq_upload_header(ctx, hb, take);
        if (rc) return rc;
        start = ctx->h_hdr.len;
    } else if (!ctx->have_hdr) return rfq_fail(ctx, RFQ_E_STATE, "decode without a header: pass has_header=1 or call rfq_set_header first");
    const DevHeader& HH = ctx->h_hdr;
    if (a->split_pe && !(HH.flags & H_PAIRED)) return rfq_fail(ctx, RFQ_E_DATA, "The input RFQ file was encoded by single-end FASTQ, you should not specify <out2>");
    if (HH.read_len_bytes != 1 && HH.read_len_bytes != 2 && HH.read_len_bytes != 4) return rfq_fail(ctx, RFQ_E_DATA,
            "header incorrect: read length bytes should be 1/2/4");
    const DevHeader* D = ctx->d_hdr.as<DevHeader>();

    // ---- walk the chunk chain
    ctx->timer.begin("walk", S);
    HIPCHK(ctx, B[DB_STATUS].ensure(sizeof(DecStatus)));
    DecStatus* dst = B[DB_STATUS].as<DecStatus>(); DecStatus hs;
    uint32_t cap = (uint32_t)std::max<size_t>(B[DB_CHUNKS].cap / sizeof(DChunk), 4096);
    // the chunk starts: the caller's chunk index (verified below); else guess-and-verify (k_dec_gw_*: an index made on the device, verified the same
    // way); else the exact serial walk (one wave, a full parse per chunk: foreign writers, corrupt images)
    bool use_table = a->h_chunk_off && a->n_chunk_off && a->h_chunk_off[0] == start && a->h_chunk_off[a->n_chunk_off] <= a->n;
    bool guess = !use_table && !ctx->opt.walk_exact;                         // (RFQ_WALK=exact: straight to the serial walk; tests run both)
    for (;;) {
        if (use_table && a->n_chunk_off + 1u > cap) cap = a->n_chunk_off + 1u;
        HIPCHK(ctx, B[DB_CHUNKS].ensure((size_t)cap * sizeof(DChunk)));
        ClearList wz; memset(&wz, 0, sizeof wz); wz.add(dst, sizeof(DecStatus), 0u);       // (the walk's fills in one launch: status, candidates, the "no candidate" count)
        const bool indexed = use_table || guess;                             // chunk starts that have to verify
        if (use_table) {
            const size_t tb = ((size_t)a->n_chunk_off + 1) * 8;
            clear_list(S, wz);
            HIPCHK(ctx, B[DB_OFFT].ensure(tb));
            HIPCHK(ctx, hipMemcpyAsync(B[DB_OFFT].p, a->h_chunk_off, tb, hipMemcpyHostToDevice, S));
            hipLaunchKernelGGL(k_dec_table, dim3(1), dim3(256), 0, S, a->d_rfq, (uint64_t)a->n, (const uint64_t*)B[DB_OFFT].as<uint64_t>(), a->n_chunk_off,
                    B[DB_CHUNKS].as<DChunk>(), dst);
        } else if (guess) {
            HIPCHK(ctx, B[DB_OFFT].ensure(((size_t)cap + 2) * 8));
            HIPCHK(ctx, B[DB_GWCAND].ensure(GW_SEGS * 8 + 64)); HIPCHK(ctx, B[DB_GWLIST].ensure((size_t)GW_SEGS * GW_LCAP * 8));
                    HIPCHK(ctx, B[DB_GWCNT].ensure(GW_SEGS * 4 + 64)); HIPCHK(ctx, B[DB_GWLAND].ensure(GW_SEGS * 8 + 64));
            unsigned long long* cand = B[DB_GWCAND].as<unsigned long long>(); uint32_t* gbad = B[DB_GWCNT].as<uint32_t>() + GW_SEGS;
            wz.add(cand, GW_SEGS * 8, 0xFFFFFFFFu); wz.add(gbad, 4, 0u); clear_list(S, wz);
            // (a segment is 16 chunks or more: no use in more segments than 64 KB pieces; RFQ_GW_SHIFT: test aid)
            const uint32_t mseg = (uint32_t)std::min<uint64_t>(GW_SEGS, std::max<uint64_t>(1, (a->n - start) >> ctx->opt.gw_shift));
            if (mseg > 1) hipLaunchKernelGGL(k_dec_gw_find, dim3(16, mseg - 1), dim3(256), 0, S, a->d_rfq, (uint64_t)a->n, start, D, cand, mseg);
            hipLaunchKernelGGL(k_dec_gw_walk, dim3(mseg), dim3(64), 0, S, a->d_rfq, (uint64_t)a->n, start, D, (const unsigned long long*)cand,
                    B[DB_GWLIST].as<unsigned long long>(), B[DB_GWCNT].as<uint32_t>(), B[DB_GWLAND].as<unsigned long long>(), gbad, mseg, a->final ? 1 : 0);
            hipLaunchKernelGGL(k_dec_gw_stitch, dim3(1), dim3(1024), 0, S, a->d_rfq, (uint64_t)a->n, start, D, (const unsigned long long*)cand,
                    (const unsigned long long*)B[DB_GWLIST].as<unsigned long long>(),
                               (const uint32_t*)B[DB_GWCNT].as<uint32_t>(), (const unsigned long long*)B[DB_GWLAND].as<unsigned long long>(), (const uint32_t*)gbad, B[DB_OFFT].as<uint64_t>(), cap, dst, mseg);
            hipLaunchKernelGGL(k_dec_table, dim3(1), dim3(256), 0, S, a->d_rfq, (uint64_t)a->n, (const uint64_t*)B[DB_OFFT].as<uint64_t>(), 0xFFFFFFFFu,
                    B[DB_CHUNKS].as<DChunk>(), dst);
        }
        else { clear_list(S, wz); hipLaunchKernelGGL(k_dec_walk, dim3(1), dim3(64), 0, S, a->d_rfq, (uint64_t)a->n, start, D, B[DB_CHUNKS].as<DChunk>(), cap, dst, a->final ? 1 : 0); }
        KCHK(ctx, "k_dec_walk");
        // the verifying parse rides behind the index without a host round trip in between (its grid covers the first
        // PARSE_AHEAD chunks; the walk's own verdict is in the status words it reads)
        const uint32_t PARSE_AHEAD = use_table ? std::max(a->n_chunk_off, 1u) : 4096u;
        if (indexed) {
            hipLaunchKernelGGL(k_dec_parse, dim3(std::min(cap, PARSE_AHEAD)), dim3(64), 0, S, a->d_rfq, (uint64_t)a->n, D, B[DB_CHUNKS].as<DChunk>(), dst, 0u);
            hipLaunchKernelGGL(k_dec_summary, dim3(1), dim3(256), 0, S, (const DChunk*)B[DB_CHUNKS].as<DChunk>(), dst, 0u, std::min(cap, PARSE_AHEAD));
            KCHK(ctx, "k_dec_parse");
        }
        HIPCHK(ctx, ctx->fetch(&hs, dst, sizeof hs, S));
        HIPCHK(ctx, ctx->fetch_sync(S));
        if (use_table) {
            // the table must cover whole chunks up to the end of the image (or up to a tail too short to be a chunk): anything else walks
            // (a range that does not end the image may end inside a chunk: whole chunks in front of it are all a table has to cover there)
            if (hs.pad || (hs.consumed != a->n && a->n - hs.consumed >= 18 && (a->final || hs.n_chunks == 0))) { use_table = false; guess = !ctx->opt.walk_exact;
                    continue; }
            break;
        }
        if (hs.overflow) { cap = hs.n_chunks + 1024; continue; }
        // (the exact walk does not look into the quality payloads: such images take the materialising path)
        if (!indexed) { hs.max_nrec = 0xFFFFFFFFu; hs.max_one = hs.max_stream; hs.per_read_pieces = 1; hs.piece_avg = 0xFFFFu; hs.piece_n1 = 0xFFFFu; break; }
        if (hs.pad) { guess = false; continue; }                           // the guessed index did not verify (foreign writer / corrupt image): walk it properly
        if (hs.n_chunks > PARSE_AHEAD) {
            hipLaunchKernelGGL(k_dec_parse, dim3(hs.n_chunks - PARSE_AHEAD), dim3(64), 0, S, a->d_rfq, (uint64_t)a->n, D, B[DB_CHUNKS].as<DChunk>(), dst, PARSE_AHEAD);
            hipLaunchKernelGGL(k_dec_summary, dim3(1), dim3(256), 0, S, (const DChunk*)B[DB_CHUNKS].as<DChunk>(), dst, PARSE_AHEAD, hs.n_chunks - PARSE_AHEAD);
            KCHK(ctx, "k_dec_parse");
            DecStatus h2; HIPCHK(ctx, ctx->fetch(&h2, dst, sizeof h2, S));
            HIPCHK(ctx, ctx->fetch_sync(S));
            if (h2.pad) { guess = false; continue; }
            hs.max_stream = h2.max_stream; hs.max_npos = h2.max_npos; hs.max_len = h2.max_len; hs.max_bases = h2.max_bases; hs.max_nrec = h2.max_nrec;
                    hs.max_one = h2.max_one; hs.per_read_pieces = h2.per_read_pieces; hs.piece_avg = h2.piece_avg; hs.piece_n1 = h2.piece_n1;
                    memcpy(hs.base_slots, h2.base_slots, sizeof hs.base_slots);
        }
        // let the exact walk decide about a trailing partial chunk (a range that does not end the image may end inside one)
        if (hs.consumed != a->n && a->n - hs.consumed >= 18 && (a->final || hs.n_chunks == 0)) { guess = false; continue; }
        break;
    }
    if (ctx->opt.trace) fprintf(stderr, "[rfq] chunk starts: %s, %u chunks\n", use_table ? "caller's index" : (guess ? "guess-and-verify" : "exact walk"), hs.n_chunks);
    ctx->timer.end(S);
    if (hs.err & DE_CORRUPT) return rfq_fail(ctx, RFQ_E_FORMAT, "truncated or corrupt rfq chunk at byte %llu", (unsigned long long)hs.consumed);
    // 64-bit total of the read lengths: bases, qualities and text are placed by 32-bit prefix sums below (a corrupt length table
    // must be refused before any buffer is sized from a wrapped sum)
    uint64_t tb = 0; for (int i = 0; i < 16; i++) tb += hs.base_slots[i];
    const uint32_t n_chunks = hs.n_chunks; const uint64_t n_reads64 = hs.total_reads;
    res->consumed = (size_t)hs.consumed; res->n_chunks = n_chunks; res->n_reads = n_reads64;
    if (n_chunks == 0) return RFQ_OK;
    const uint32_t last_flags = hs.last_flags; const int split = a->split_pe ? 1 : 0;
    DChunk* CHm = B[DB_CHUNKS].as<DChunk>();
    // (RFQ_SLICE_BASES: test aid - ranges of that many bases, so that the slicing logic runs on small images)
    const uint64_t slice_env = ctx->opt.slice_bases;
    // (a pass whose text turns out to be >= 4 GiB is redone in ranges)
    const uint64_t slice_bases = slice_env ? slice_env : 1500000000ull, one_pass = slice_env ? slice_env : 0xFFFFFFF0ull;
    size_t n1 = 0, n2 = 0; uint64_t nb = 0; uint8_t *o1 = nullptr, *o2 = nullptr;
    int rc = RFQ_RANGE_TOO_BIG;
    // bug_compat: Repaq::decompressPE as it stands (src/repaq.cpp:330-417).  A chunk with a NO_LINE_BREAK bit makes the loop read the chunk behind it
    // to see whether it was the last (:376-387); when it was not, `continue` (:389-392, :400-403) goes on with a fresh read - decompressPE declares
    // its chunk inside the loop - and the chunk it peeked at is never decoded; it also leaves the body before the flagged chunk's R2 text is written
    // when it is the R1 bit that is set.  The text is decoded piece by piece: runs of chunks that survive, R1 only where the reference drops R2.
    // Repaq::decompress (one output, :262-328) peeks the same way but keeps the peeked chunk in a variable that outlives the iteration and decodes it
    // next: nothing is lost there, and the only effect of the bit - the last chunk's final '\n' dropped - is the default behaviour.
    struct Rng { uint32_t c0, c1; bool r1_only; };
    std::vector<Rng> pieces; std::vector<DChunk> hc; bool strip1 = (last_flags & C_NO_LB) != 0, strip2 = (last_flags & C_NO_LB_R2) != 0;
    const bool compat = a->bug_compat != 0 && split;
    if (compat) {
        hc.resize(n_chunks);
        HIPCHK(ctx, hipMemcpy(hc.data(), CHm, (size_t)n_chunks * sizeof(DChunk), hipMemcpyDeviceToHost));
        uint32_t run = 0, end = n_chunks; strip1 = strip2 = false;
        for (uint32_t i = 0; i < n_chunks;) {
            const bool f1 = (hc[i].flags & C_NO_LB) != 0, f2 = split && (hc[i].flags & C_NO_LB_R2) != 0;
            if (!f1 && !f2) { i++; continue; }
            if (i + 1 == n_chunks) {                                            // nothing behind it in this image
                if (!a->final) { end = i; res->consumed = (size_t)hc[i].off; }  // the caller has to show what follows: the chunk stays unconsumed
                else { strip1 = f1; strip2 = f2; }
                break;
            }
            if (split && f1) { if (i > run) pieces.push_back({ run, i, false }); pieces.push_back({ i, i + 1, true }); }
            else pieces.push_back({ run, i + 1, false });
            run = i + 2; i += 2;                                                // (the chunk behind a flagged one that is not the last: lost)
        }
        if (run < end) pieces.push_back({ run, end, false });
        res->n_chunks = 0; for (auto& q : pieces) res->n_chunks += q.c1 - q.c0;
    }
    if (!compat && tb < one_pass && n_reads64 < 0x7FFFFFF0ull) {
        DecRange g; g.CH = CHm; g.n_chunks = n_chunks; g.n_reads = (uint32_t)n_reads64; g.max_reads = hs.max_reads; g.max_stream = hs.max_stream;
                g.max_npos = hs.max_npos; g.max_len = hs.max_len; g.max_bases = hs.max_bases; g.max_nrec = hs.max_nrec; g.max_one = hs.max_one;
                g.pieces = hs.per_read_pieces; g.piece_avg = hs.piece_avg; g.piece_n1 = hs.piece_n1; g.bases = tb;
        rc = decode_range(ctx, a, g, a->d_out1, a->cap1, a->d_out2, a->cap2, &o1, &o2, &n1, &n2, &nb);
        if (rc != RFQ_OK && rc != RFQ_RANGE_TOO_BIG) return rc;
    }
    if (rc == RFQ_RANGE_TOO_BIG) {
        // Reads, bases and text of one pass are placed by 32-bit prefix sums: a larger image is decoded range by range (contiguous chunks of
        // about slice_bases bases; a range whose text still does not fit is halved), every range into the context's own buffers and from
        // there to its place in the result.
        if (!compat) { hc.resize(n_chunks); HIPCHK(ctx, hipMemcpy(hc.data(), CHm, (size_t)n_chunks * sizeof(DChunk), hipMemcpyDeviceToHost));
                pieces.push_back({ 0, n_chunks, false }); }
        std::vector<Rng> todo;                                               // stack of [c0, c1), first range on top
        { std::vector<Rng> fw;
          for (auto& pc : pieces) {
              uint32_t c0 = pc.c0; uint64_t acc = 0, rd = 0;
              for (uint32_t c = pc.c0; c < pc.c1; c++) {
                  if (c > c0 && (acc + hc[c].bases > slice_bases || rd + hc[c].reads > 0x7FFFFFF0ull)) { fw.push_back({ c0, c, pc.r1_only }); c0 = c; acc = 0; rd = 0; }
                  acc += hc[c].bases; rd += hc[c].reads;
              }
              fw.push_back({ c0, pc.c1, pc.r1_only });
          }
          for (size_t i = fw.size(); i-- > 0;) todo.push_back(fw[i]); }
        size_t w1 = 0, w2 = 0; n1 = n2 = 0; nb = 0;
        std::vector<std::pair<const char*, float>> acc_ms;
        ctx->timer.collect();                                                // (the walk, and whatever a failed one-pass attempt got through)
        for (size_t i = 0; i < ctx->timer.names.size() && i < 1; i++) acc_ms.emplace_back(ctx->timer.names[i], ctx->timer.ms[i]);
        while (!todo.empty()) {
            const Rng r = todo.back(); todo.pop_back();
            const uint32_t c0 = r.c0, c1 = r.c1; uint64_t reads = 0, rbases = 0; for (uint32_t c = c0; c < c1; c++) { reads += hc[c].reads; rbases += hc[c].bases; }
            hipLaunchKernelGGL(k_dec_rebase, dim3((c1 - c0 + 255) / 256), dim3(256), 0, S, CHm + c0, c1 - c0, hc[c0].rbase_abs);
            KCHK(ctx, "k_dec_rebase");
            DecRange g; g.CH = CHm + c0; g.n_chunks = c1 - c0; g.n_reads = (uint32_t)reads; g.max_reads = hs.max_reads; g.max_stream = hs.max_stream;
                    g.max_npos = hs.max_npos; g.max_len = hs.max_len; g.max_bases = hs.max_bases; g.max_nrec = hs.max_nrec; g.max_one = hs.max_one;
                    g.pieces = hs.per_read_pieces; g.piece_avg = hs.piece_avg; g.piece_n1 = hs.piece_n1; g.bases = rbases;
            uint8_t *q1 = nullptr, *q2 = nullptr; size_t m1 = 0, m2 = 0; uint64_t mb = 0;
            ctx->timer.reset();
            rc = reads > 0x7FFFFFF0ull ? RFQ_RANGE_TOO_BIG : decode_range(ctx, a, g, nullptr, 0, nullptr, 0, &q1, &q2, &m1, &m2, &mb);
            if (rc == RFQ_RANGE_TOO_BIG) {
                if (c1 - c0 < 2) return rfq_fail(ctx, RFQ_E_ARG, "a single chunk decodes to 4 GiB of text or more");
                const uint32_t mid = c0 + (c1 - c0) / 2; todo.push_back({ mid, c1, r.r1_only }); todo.push_back({ c0, mid, r.r1_only }); continue;
            }
            if (rc != RFQ_OK) return rc;
            for (size_t i = 0; i < ctx->timer.names.size(); i++) {
                bool hit = false;
                for (auto& q : acc_ms) if (q.first == ctx->timer.names[i]) { q.second += ctx->timer.ms[i]; hit = true; break; }
                if (!hit) acc_ms.emplace_back(ctx->timer.names[i], ctx->timer.ms[i]);
            }
            for (int k = 0; k < (split ? 2 : 1); k++) {
                uint8_t* src = k ? q2 : q1; const size_t m = k ? m2 : m1; size_t& w = k ? w2 : w1; uint8_t* dcall = k ? a->d_out2 : a->d_out1;
                        const size_t dcap = k ? a->cap2 : a->cap1;
                if (!m || (k && r.r1_only)) continue;
                uint8_t* to;
                if (dcall) { if (w + m > dcap) return rfq_fail(ctx, RFQ_E_NOSPACE, "output buffer too small"); to = dcall + w; }
                else { DBuf& ab = k ? ctx->out_acc2 : ctx->out_acc1; HIPCHK(ctx, ab.ensure_keep(w + m + 64, w, S)); to = ab.as<uint8_t>() + w; }
                HIPCHK(ctx, hipMemcpyAsync(to, src, m, hipMemcpyDeviceToDevice, S));
                w += m;
            }
            HIPCHK(ctx, hipStreamSynchronize(S));
            nb += mb;
        }
        n1 = w1; n2 = w2;
        o1 = a->d_out1 ? a->d_out1 : ctx->out_acc1.as<uint8_t>(); o2 = a->d_out2 ? a->d_out2 : ctx->out_acc2.as<uint8_t>();
        ctx->timer.names.clear(); ctx->timer.ms.clear();
        for (auto& q : acc_ms) { ctx->timer.names.push_back(q.first); ctx->timer.ms.push_back(q.second); }
    }
    res->n_bases = nb;
    if (a->final) {   // Repaq::decompress*: drop the final '\n' when the last chunk carries the NO_LINE_BREAK bit
        if (strip1 && n1) n1--;
        if (split && strip2 && n2) n2--;
    }
    res->d_fq1 = o1; res->n1 = n1; res->d_fq2 = split ? o2 : nullptr; res->n2 = split ? n2 : 0;
    return RFQ_OK;
}
