// rfq_encode.hip — host orchestration of the gfx950 FASTQ -> RFQ path (rfq_encode_batch of include/rfq_hip.h).
// Replaces, per batch: Repaq::compress / compressPE chunking (src/repaq.cpp:530-762), RfqCodec::makeHeader
// (src/rfqcodec.cpp:20-145), RfqCodec::encodeChunk (:147-586) and RfqChunk::write (src/rfqchunk.cpp:230-311).
#include "rfq_ctx.h"
#include "rfq_encode_kernels.h"
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <cstddef>

enum EncBuf {   // indices into rfq_ctx::b
    B_BITMAP0 = 0, B_BITMAP1, B_BLK0, B_BLK1, B_LO0, B_LO1, B_SCANTMP,
    B_LEN, B_N1LEN, B_N2OFF, B_X, B_Y, B_TILE, B_LANE, B_OK, B_CHUNK, B_STORED, B_EQ2, B_PQ, B_PV, B_PVIN,
    B_ULEN, B_P, B_MINMAX, B_FIRST, B_CFLAGS, B_IL, B_HIST, B_NCOUNT, B_SCAP, B_SOFF, B_SSIZE, B_XSIZE, B_YSIZE, B_QBASE, B_SBASE,
    B_IMGSIZE, B_IMGOFF, B_CTOTAL, B_CBASE, B_LAYOUT, B_HSTATS, B_OVB, B_OVRAW, B_QCAT, B_SCAT, B_SCRATCH, B_XS, B_YS, B_SEGB, B_SEGC,
    B_NORM0, B_NORM1, B_OT0, B_OT1, B_ONX0, B_ONX1, B_TBITS, B_SBITS, B_NKEEP, B_NTERM, B_NMAP, B_ADJ, B_PINFO, B_SEGM, B_LPK, B_LNB, B_SPK, B_SNM, B_RFLAG, B_SCANTMP2, B_CTOTALN, B_CBASEN, B_SCRATCHN, B_PTOT, B_QPLANE, B_SEGD, B_SEGS, B_SD, B_RN, B_ENC_END
};

static_assert(B_ENC_END <= 80, "encode buffers must stay below the decode buffer indices of rfq_ctx::b");

static int fetch_bytes(rfq_ctx* ctx, const uint8_t* d, size_t n, std::string& out) {
    out.resize(n);
    if (n) HIPCHK(ctx, hipMemcpy(&out[0], d, n, hipMemcpyDeviceToHost));
    return RFQ_OK;
}
// text of line k of read g (host copy), for the reference's error messages
static int fetch_line(rfq_ctx* ctx, const Text& T, uint32_t g, int k, std::string& out) {
    int s = 0; uint32_t r = g; if (T.paired == 1) { s = (int)(g & 1u); r = g >> 1; }
    uint32_t lo2[2];
    HIPCHK(ctx, hipMemcpy(lo2, T.lo[s] + 4 * (size_t)r + k, 8, hipMemcpyDeviceToHost));
    return fetch_bytes(ctx, T.fq[s] + lo2[0], lo2[1] - 1 - lo2[0], out);
}

// RfqHeader::read (src/rfqheader.cpp:19-43) + the derived tables, on device and mirrored on the host
int rfq_upload_header(rfq_ctx* c, const uint8_t* h, size_t n) {
    if (n < 17) return rfq_fail(c, RFQ_E_FORMAT, "Not a valid repaq file!");
    if (h[8] != 2) return rfq_fail(c, RFQ_E_FORMAT, "The data is encoded by different version of repaq, please try repaq v%.5s. \nSee: https://github.com/OpenGene/repaq/releases", (const char*)h + 3);
    const size_t len = 17u + h[16];
    if (n < len) return rfq_fail(c, RFQ_E_FORMAT, "Not a valid repaq file!");
    if (h[0] != 'R' || h[1] != 'F' || h[2] != 'Q') return rfq_fail(c, RFQ_E_FORMAT, "Not a valid repaq file!");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->d_hdr.ensure(sizeof(DevHeader)));
    // The derived tables (hdr_derive: the same function the header kernels call) are made HERE, on the host, and the whole DevHeader goes to the device in one
    // stream-ordered copy from a page-locked block: no kernel, no read-back, no wait (round 5: upload, k_hdr_from_bytes, fetch, synchronise - two round trips in
    // front of every decode that starts with a header).  Bytes equal to the header the device already holds: nothing is sent.
    DevHeader tmp; memset(&tmp, 0, sizeof tmp); memcpy(tmp.bytes, h, len); hdr_derive(&tmp);
    const bool same = c->have_hdr && c->hdr_on_device && c->h_hdr.len == tmp.len && !memcmp(c->h_hdr.bytes, tmp.bytes, tmp.len);
    if (!same) {
        if (!c->pin_up) { if (hipHostMalloc((void**)&c->pin_up, sizeof(DevHeader), 0) != hipSuccess) { c->pin_up = nullptr; (void)hipGetLastError(); } }
        if (c->pin_up) {
            if (c->ev_up_pending) { HIPCHK(c, hipEventSynchronize(c->ev_up)); c->ev_up_pending = false; }       // (the block's previous copy)
            if (!c->ev_up) HIPCHK(c, hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
            memcpy(c->pin_up, &tmp, sizeof tmp);
            HIPCHK(c, hipMemcpyAsync(c->d_hdr.p, c->pin_up, sizeof tmp, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipEventRecord(c->ev_up, c->stream)); c->ev_up_pending = true;
        } else { HIPCHK(c, hipMemcpyAsync(c->d_hdr.p, &tmp, sizeof tmp, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
        c->h_hdr = tmp; c->hdr_on_device = true;
    }
    c->have_hdr = true; c->dense_ok = false; c->e3_pieces_failed = false; c->mixed_lengths = false;   // (another file: what its name pieces fit / whether its reads have one length is not known yet)
    return RFQ_OK;
}


// Mapping of a normalised stream (see k_norm_classify) back to the caller's text
struct NormMap { const uint32_t* ot[2]; const uint32_t* onx[2]; size_t orig_n[2]; };
#define RFQ_NEED_NORM 1            // internal: the '\n'-only indexer met '\r' or an empty line; redo on normalised text
#define RFQ_RETRY_ROOM 3           // internal: an arena sized in advance was too small (it has been grown): encode the batch again

// FastqReader::getLine semantics for text with '\r' / blank lines: rewrite stream s as '\n'-terminated text + the line maps
static int normalize_stream(rfq_ctx* ctx, const uint8_t* fq, size_t n, uint64_t file_off, bool final, int s, NormMap& nm, const uint8_t** out, size_t* out_n) {
    hipStream_t S = ctx->stream; DBuf* B = ctx->b;
    nm.orig_n[s] = n; nm.ot[s] = nm.onx[s] = nullptr; *out = nullptr; *out_n = 0;
    if (n == 0) return RFQ_OK;
    const uint64_t nwords = (n + 63) / 64; const uint32_t nblk = (uint32_t)((nwords + 255) / 256);
    HIPCHK(ctx, B[B_TBITS].ensure(nwords * 8 + 64)); HIPCHK(ctx, B[B_SBITS].ensure(nwords * 8 + 64));
    HIPCHK(ctx, B[B_NKEEP].ensure(((size_t)nblk + 2) * 4)); HIPCHK(ctx, B[B_NTERM].ensure(((size_t)nblk + 2) * 4));
    HIPCHK(ctx, B[B_SCANTMP].ensure(std::max<size_t>(1024, ((size_t)nblk / SCAN_TILE + 2) * 16)));
    NormIn in; in.fq = fq; in.n = (uint32_t)n; in.file_off = file_off; in.file_end = final ? file_off + n : ~0ull;
    hipLaunchKernelGGL(k_norm_classify, dim3(nblk), dim3(256), 0, S, in, B[B_TBITS].as<uint64_t>(), B[B_SBITS].as<uint64_t>(), B[B_NKEEP].as<uint32_t>(),
            B[B_NTERM].as<uint32_t>());
    KCHK(ctx, "k_norm_classify");
    scan_exclusive<uint32_t>(S, B[B_NKEEP].as<uint32_t>(), B[B_NKEEP].as<uint32_t>(), nblk, B[B_SCANTMP].as<uint32_t>(), 1);
    scan_exclusive<uint32_t>(S, B[B_NTERM].as<uint32_t>(), B[B_NTERM].as<uint32_t>(), nblk, B[B_SCANTMP].as<uint32_t>(), 1);
    uint32_t keep = 0, terms = 0;
    HIPCHK(ctx, ctx->fetch(&keep, B[B_NKEEP].as<uint32_t>() + nblk, 4, S));
    HIPCHK(ctx, ctx->fetch(&terms, B[B_NTERM].as<uint32_t>() + nblk, 4, S));
    HIPCHK(ctx, ctx->fetch_sync(S));
    HIPCHK(ctx, B[B_NORM0 + s].ensure((size_t)keep + 64)); HIPCHK(ctx, B[B_OT0 + s].ensure(((size_t)terms + 4) * 4));
            HIPCHK(ctx, B[B_ONX0 + s].ensure(((size_t)terms + 4) * 4));
    hipLaunchKernelGGL(k_norm_emit, dim3(nblk), dim3(256), 0, S, in, (const uint64_t*)B[B_TBITS].as<uint64_t>(), (const uint64_t*)B[B_SBITS].as<uint64_t>(),
                       (const uint32_t*)B[B_NKEEP].as<uint32_t>(), (const uint32_t*)B[B_NTERM].as<uint32_t>(), B[B_NORM0 + s].as<uint8_t>(), B[B_OT0 + s].as<uint32_t>(), B[B_ONX0 + s].as<uint32_t>());
    hipLaunchKernelGGL(k_norm_tail, dim3(1), dim3(64), 0, S, B[B_OT0 + s].as<uint32_t>(), B[B_ONX0 + s].as<uint32_t>(), terms, (uint32_t)n);
    KCHK(ctx, "k_norm_emit");
    nm.ot[s] = B[B_OT0 + s].as<uint32_t>(); nm.onx[s] = B[B_ONX0 + s].as<uint32_t>();
    *out = B[B_NORM0 + s].as<uint8_t>(); *out_n = keep;
    return RFQ_OK;
}

static int encode_impl(rfq_ctx* ctx, const rfq_encode_args* a, rfq_encode_result* res, const NormMap* nm, uint32_t unit_cap, bool ended, bool scan_only,
        const uint32_t* skip);

// One call's worth of text (< 4 GiB per stream; the stream pointers may sit at any byte address: they are rounded down to 16 bytes and the
// bytes in front are skipped by the indexer)
static int encode_one(rfq_ctx* ctx, const rfq_encode_args* a, rfq_encode_result* res, bool scan_only) {
    memset(res, 0, sizeof *res);
    rfq_encode_args al = *a; uint32_t skip[2] = { 0, 0 };
    if (a->n1 && a->d_fq1) { skip[0] = (uint32_t)((uintptr_t)a->d_fq1 & 15u); al.d_fq1 = a->d_fq1 - skip[0]; al.n1 = a->n1 + skip[0];
            al.file_off1 = a->file_off1 - skip[0]; }
    if (a->paired == RFQ_PE_TWO_FILES && a->n2 && a->d_fq2) { skip[1] = (uint32_t)((uintptr_t)a->d_fq2 & 15u); al.d_fq2 = a->d_fq2 - skip[1]; al.n2 = a->n2 + skip[1];
            al.file_off2 = a->file_off2 - skip[1]; }
    int rc = RFQ_RETRY_ROOM;
    for (int attempt = 0; attempt < 3 && rc == RFQ_RETRY_ROOM; attempt++) rc = encode_impl(ctx, &al, res, nullptr, ~0u, false, scan_only, skip);
    if (rc == RFQ_RETRY_ROOM) return rfq_fail(ctx, RFQ_E_HIP, "internal: the stream arenas did not settle");
    if (rc != RFQ_NEED_NORM) return rc;
    // slow path: '\r' line ends or blank lines (src/fastqreader.cpp:94-196)
    NormMap nm; memset(&nm, 0, sizeof nm);
    rfq_encode_args a2 = *a;
    const uint8_t* p; size_t pn; const uint32_t noskip[2] = { 0, 0 };
    if ((rc = normalize_stream(ctx, a->d_fq1, a->n1, a->file_off1, a->final != 0, 0, nm, &p, &pn)) != RFQ_OK) return rc;
    a2.d_fq1 = p; a2.n1 = pn;
    if (a->paired == RFQ_PE_TWO_FILES) {
        if ((rc = normalize_stream(ctx, a->d_fq2, a->n2, a->file_off2, a->final != 0, 1, nm, &p, &pn)) != RFQ_OK) return rc;
        a2.d_fq2 = p; a2.n2 = pn;
    }
    memset(res, 0, sizeof *res);
    rc = RFQ_RETRY_ROOM;
    for (int attempt = 0; attempt < 3 && rc == RFQ_RETRY_ROOM; attempt++) rc = encode_impl(ctx, &a2, res, &nm, ~0u, false, scan_only, noskip);
    if (rc == RFQ_RETRY_ROOM) return rfq_fail(ctx, RFQ_E_HIP, "internal: the stream arenas did not settle");
    if (rc == RFQ_NEED_NORM) return rfq_fail(ctx, RFQ_E_HIP, "internal: normalised text still needs normalisation");
    return rc;
}

// Texts of 4 GiB and more per stream (offsets inside one call are 32-bit): the call is cut into slices of RFQ_SLICE bytes per stream.  A slice
// that is not the last one stops at its last full chunk (final = 0) and the next slice starts right behind the bytes it consumed - in
// place, nothing is copied.  Chunk images are appended in order, so the result is the one-shot image.
#define RFQ_SLICE ((size_t)3 << 30)
static int encode_or_scan(rfq_ctx* ctx, const rfq_encode_args* a, rfq_encode_result* res, bool scan_only) {
    if (!ctx || !a || !res) return RFQ_E_ARG;
    memset(res, 0, sizeof *res);
    ctx->err.clear();
    if (a->paired < 0 || a->paired > 2) return rfq_fail(ctx, RFQ_E_ARG, "paired must be RFQ_SE, RFQ_PE_TWO_FILES or RFQ_PE_INTERLEAVED");
    const bool two = a->paired == RFQ_PE_TWO_FILES;
    // (RFQ_SLICE_BYTES: test aid - slices of that many bytes, so that the slicing logic runs on small inputs)
    const size_t slice_env = ctx->opt.slice_bytes;
    const size_t slice = slice_env ? slice_env : RFQ_SLICE, lim = slice_env ? slice_env : 0xFFFFFFF0ull - 16;
    if (a->n1 < lim && (!two || a->n2 < lim)) return encode_one(ctx, a, res, scan_only);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    size_t pos1 = 0, pos2 = 0, written = 0; bool first = true;
    std::vector<uint64_t> offs(1, 0), e1, e2; std::vector<std::pair<const char*, float>> acc;
    uint32_t chunks = 0; uint64_t reads = 0, bases = 0; int ended = 0; int32_t ub = 0;
    for (;;) {
        const size_t r1 = a->n1 - pos1, r2 = two ? a->n2 - pos2 : 0;
        size_t t1 = std::min(r1, slice), t2 = std::min(r2, slice);
        // Two files of different length: once one file's slice holds all that is left of it, no pair lies beyond it - the other file's slice grows
        // to what is left of ITS file (as far as one call can address), and the call ends the input there like the reference does (it truncates
        // to the shorter file).  (With the other slice left at its size the slice ran as a non-final one, the short file's last records - less
        // than a chunk - were never flushed, and the call failed with "no whole chunk".)
        if (two && (t1 == r1) != (t2 == r2)) { const size_t grow = slice_env ? 64 * slice_env : lim; if (t1 == r1) t2 = std::min(r2, grow); else t1 = std::min(r1, grow);
                }
        const bool last = t1 == r1 && t2 == r2;
        rfq_encode_args s = *a;
        s.d_fq1 = a->d_fq1 + pos1; s.n1 = t1; s.file_off1 = a->file_off1 + pos1;
        if (two) { s.d_fq2 = a->d_fq2 + pos2; s.n2 = t2; s.file_off2 = a->file_off2 + pos2; }
        s.final = last ? a->final : 0; s.flush_all = last ? a->flush_all : 0; s.emit_header = first ? a->emit_header : 0; s.carry_bases = first ? a->carry_bases : 0u;
        if (a->d_out) { s.d_out = a->d_out + written; s.out_cap = a->out_cap > written ? a->out_cap - written : 0; }
        rfq_encode_result r;
        const int rc = encode_one(ctx, &s, &r, scan_only);
        if (rc != RFQ_OK) return rc;
        for (size_t i = 0; i < ctx->timer.names.size(); i++) {
            bool hit = false;
            for (auto& q : acc) if (q.first == ctx->timer.names[i]) { q.second += ctx->timer.ms[i]; hit = true; break; }
            if (!hit) acc.emplace_back(ctx->timer.names[i], ctx->timer.ms[i]);
        }
        if (scan_only) {
            for (uint32_t c = 0; c < r.n_chunks; c++) { e1.push_back(ctx->scan_end[0][c] + pos1); if (two) e2.push_back(ctx->scan_end[1][c] + pos2); }
        } else if (r.rfq_len) {
            if (!a->d_out) {            // the slice's image sits in the context's result buffer: append it to the call's
                HIPCHK(ctx, ctx->out_acc.ensure_keep(written + r.rfq_len + 64, written, ctx->stream));
                HIPCHK(ctx, hipMemcpyAsync(ctx->out_acc.as<uint8_t>() + written, r.d_rfq, r.rfq_len, hipMemcpyDeviceToDevice, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            }
            for (uint32_t c = 1; c <= r.n_chunks; c++) offs.push_back(written + r.h_chunk_off[c]);
            if (first && r.n_chunks) offs[0] = r.h_chunk_off[0];
            written += r.rfq_len;
        }
        if (first) ub = r.reserved; else if (r.reserved != ub) ub = 0;      // (scan: the units' common length, if the slices agree on one)
        chunks += r.n_chunks; reads += r.n_reads; bases += r.n_bases; pos1 += r.consumed1; pos2 += r.consumed2; first = false;
        if (r.input_ended) { ended = 1; break; }
        if (last) break;
        if (r.consumed1 == 0) return rfq_fail(ctx, RFQ_E_ARG, "no whole chunk inside %zu bytes of text: chunk_bases is too large for a sliced call", slice);
    }
    ctx->timer.names.clear(); ctx->timer.ms.clear();
    for (auto& q : acc) { ctx->timer.names.push_back(q.first); ctx->timer.ms.push_back(q.second); }
    res->n_chunks = chunks; res->n_reads = reads; res->n_bases = bases; res->consumed1 = pos1; res->consumed2 = pos2; res->input_ended = ended;
            res->reserved = scan_only ? ub : 0;
    if (scan_only) { ctx->scan_end[0] = e1; ctx->scan_end[1] = e2; return RFQ_OK; }
    ctx->chunk_off = offs; res->h_chunk_off = ctx->chunk_off.data();
    res->rfq_len = written; res->d_rfq = written ? (a->d_out ? a->d_out : ctx->out_acc.as<uint8_t>()) : nullptr;
    return RFQ_OK;
}
extern "C" int rfq_encode_batch(rfq_ctx* ctx, const rfq_encode_args* a, rfq_encode_result* res) { return encode_or_scan(ctx, a, res, false); }
extern "C" int rfq_scan_batch(rfq_ctx* ctx, const rfq_encode_args* a, rfq_scan_result* out) {
    if (!out) return RFQ_E_ARG;
    memset(out, 0, sizeof *out);
    rfq_encode_result r;
    const int rc = encode_or_scan(ctx, a, &r, true);
    if (rc != RFQ_OK) return rc;
    out->n_chunks = r.n_chunks; out->n_reads = r.n_reads; out->consumed1 = r.consumed1; out->consumed2 = r.consumed2; out->input_ended = r.input_ended;
            out->unit_bases = (uint32_t)r.reserved;
    out->h_end1 = r.n_chunks ? ctx->scan_end[0].data() : nullptr;
    out->h_end2 = (r.n_chunks && a->paired == RFQ_PE_TWO_FILES) ? ctx->scan_end[1].data() : nullptr;
    return RFQ_OK;
}

static int encode_impl(rfq_ctx* ctx, const rfq_encode_args* a, rfq_encode_result* res, const NormMap* nm, uint32_t unit_cap, bool ended, bool scan_only,
        const uint32_t* skip) {
    memset(res, 0, sizeof *res);
    res->input_ended = ended ? 1 : 0;
    const bool fin = a->final || ended || a->flush_all;
    if (a->paired < 0 || a->paired > 2) return rfq_fail(ctx, RFQ_E_ARG, "paired must be RFQ_SE, RFQ_PE_TWO_FILES or RFQ_PE_INTERLEAVED");
    if (a->chunk_bases == 0) return rfq_fail(ctx, RFQ_E_ARG, "chunk_bases must be >= 1");
    const int nstreams = a->paired == RFQ_PE_TWO_FILES ? 2 : 1;
    const uint8_t* fq[2] = { a->d_fq1, nstreams == 2 ? a->d_fq2 : nullptr };
    // skip[s] (< 16): leading bytes of stream s that are not part of it (see k_nl_bitmap); a stream that holds nothing else is empty
    const size_t nbytes[2] = { a->n1 > skip[0] ? a->n1 : 0, nstreams == 2 ? (a->n2 > skip[1] ? a->n2 : 0) : 0 };
    for (int s = 0; s < nstreams; s++) {
        if (nbytes[s] >= 0xFFFFFFF0ull) return rfq_fail(ctx, RFQ_E_ARG, "a FASTQ stream of one batch must be < 4 GiB (got %zu bytes); split at record boundaries",
                nbytes[s]);
        if (nbytes[s] && !fq[s]) return rfq_fail(ctx, RFQ_E_ARG, "null FASTQ pointer");
        if (((uintptr_t)fq[s]) & 15u) return rfq_fail(ctx, RFQ_E_ARG, "FASTQ device pointers must be 16-byte aligned");
    }
    hipStream_t S = ctx->stream;
    DBuf* B = ctx->b;
    ctx->timer.reset();
    ctx->pend.clear(); ctx->pin_used = 0;                                   // (read-backs an earlier call left behind on an error path)
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // (a marker, not a phase: this is the repeat of a batch whose arenas were too small - tests look for it)
    if (ctx->retried_room) { ctx->timer.begin("retry_room", S); ctx->timer.end(S); ctx->retried_room = false; }

    // ---- status block
    DevStatus hs; memset(&hs, 0, sizeof hs); hs.err_key = ~0ull; hs.coord_key = ~0ull; hs.first_empty = ~0u;
    HIPCHK(ctx, ctx->d_status.ensure(sizeof(DevStatus)));
    DevStatus* dst = ctx->d_status.as<DevStatus>();
    HIPCHK(ctx, hipMemcpyAsync(dst, &hs, sizeof hs, hipMemcpyHostToDevice, S));

    // ---- phase 1: the line index lo[] of every stream.  One pass (k_line_index) where the lines are long enough for a table sized in advance
    // (RFQ_INDEX=2pass, or more than one line per 16 bytes: newline bitmap -> scan -> k_line_offsets, the table sized exactly)
    ctx->timer.begin("index", S);
    uint32_t nblk[2] = { 0, 0 }; uint64_t nwords[2] = { 0, 0 };
    for (int s = 0; s < nstreams; s++) { nwords[s] = (nbytes[s] + 63) / 64; nblk[s] = (uint32_t)((nwords[s] + 255) / 256); }
    int idx_tiles = ctx->opt.idx_tiles ? ctx->opt.idx_tiles : NLF_TILES;
    if (idx_tiles != 4 && idx_tiles != 8 && idx_tiles != 16) idx_tiles = NLF_TILES;
    uint32_t nidx[2] = { 0, 0 };                                                           // workgroups of the one-pass index
    for (int s = 0; s < nstreams; s++) nidx[s] = (uint32_t)((nbytes[s] + idx_tiles * 16384u - 1) / (idx_tiles * 16384u));
    uint32_t n_newlines[2] = { 0, 0 }; uint8_t lastbyte[2] = { '\n', '\n' };
    bool one_pass = !ctx->opt.index_2pass;
    // LAZY: no read-back behind the index.  The per-read tables are sized for a unit count guessed from the records per byte of the context's earlier batches; the
    // index's totals stay on the device (k_index_totals: lines, units, the unterminated tail) and reach the host with the partition's results.  A batch that holds
    // more units than guessed, an index that has to fall back to two passes: once more with the read-back (ctx->lazy_block).
    const bool lazy_allowed = !ctx->lazy_block; ctx->lazy_block = false;
    bool lazy = one_pass && lazy_allowed && ctx->rec_per_byte > 0.0 && !ctx->mixed_lengths && unit_cap == ~0u && !ended;
    uint32_t guess_units = 0;
    if (lazy) {
        double g = 1e300;
        for (int s = 0; s < nstreams; s++) g = std::min(g, (double)nbytes[s] * ctx->rec_per_byte * 1.03 + 64.0);
        if (a->paired == RFQ_PE_INTERLEAVED) g *= 0.5;
        if (g > 2.0e9 || g < 1.0) lazy = false; else guess_units = (uint32_t)g;
        for (int s = 0; s < nstreams; s++) if (!nblk[s]) lazy = false;
    }
    if (one_pass) {
        size_t cap[2] = { 0, 0 };
        for (int s = 0; s < nstreams; s++) {
            if (!nblk[s]) continue;
            cap[s] = std::max(B[B_LO0 + s].cap / 4, nbytes[s] / 16 + 4096);
            HIPCHK(ctx, B[B_LO0 + s].ensure(cap[s] * 4));
            HIPCHK(ctx, B[B_BLK0 + s].ensure((size_t)nidx[s] * 8 + 64));                  // state words, then the ticket and the total
            HIPCHK(ctx, hipMemsetAsync(B[B_BLK0 + s].p, 0, (size_t)nidx[s] * 8 + 64, S));
            unsigned long long* state = B[B_BLK0 + s].as<unsigned long long>();
            uint32_t* tt = (uint32_t*)(state + nidx[s]);
            const uint32_t lo_cap = (uint32_t)std::min<size_t>(cap[s] - 4, 0xFFFFFFF0u);
            auto kern = idx_tiles == 16 ? k_line_index<16> : (idx_tiles == 4 ? k_line_index<4> : k_line_index<8>);
            hipLaunchKernelGGL(kern, dim3(nidx[s]), dim3(256), 0, S, fq[s], (uint32_t)nbytes[s], skip[s], B[B_LO0 + s].as<uint32_t>(), lo_cap, state, tt, tt + 1, dst);
            KCHK(ctx, "k_line_index");
            if (lazy) continue;
            HIPCHK(ctx, ctx->fetch(&n_newlines[s], tt + 1, 4, S));
            HIPCHK(ctx, ctx->fetch(&lastbyte[s], fq[s] + nbytes[s] - 1, 1, S));
        }
        if (lazy) {
            const uint32_t* t0 = (const uint32_t*)(B[B_BLK0].as<unsigned long long>() + nidx[0]) + 1;
            const uint32_t* t1 = nstreams == 2 ? (const uint32_t*)(B[B_BLK1].as<unsigned long long>() + nidx[1]) + 1 : t0;
            hipLaunchKernelGGL(k_index_totals, dim3(1), dim3(64), 0, S, t0, t1, fq[0], (uint32_t)nbytes[0], fq[1], (uint32_t)nbytes[1], B[B_LO0].as<uint32_t>(),
                               nstreams == 2 ? B[B_LO1].as<uint32_t>() : (uint32_t*)nullptr, a->final ? 1 : 0, (int)a->paired, unit_cap, guess_units, dst);
            KCHK(ctx, "k_index_totals");
        } else {
        HIPCHK(ctx, ctx->fetch(&hs, dst, sizeof hs, S));
        HIPCHK(ctx, ctx->fetch_sync(S));
        }
        if (!lazy && (hs.err & DE_INDEX_RETRY)) {                                                    // start over with a clean status block
            one_pass = false;
            memset(&hs, 0, sizeof hs); hs.err_key = ~0ull; hs.coord_key = ~0ull; hs.first_empty = ~0u;
            HIPCHK(ctx, hipMemcpyAsync(dst, &hs, sizeof hs, hipMemcpyHostToDevice, S));
            HIPCHK(ctx, hipStreamSynchronize(S));                                         // (hs is a stack object the copy reads)
            n_newlines[0] = n_newlines[1] = 0;
        }
    }
    if (!one_pass) {
        ctx->timer.end(S); ctx->timer.begin("index_2pass", S);
        size_t scantmp = 1024;
        for (int s = 0; s < nstreams; s++) {
            HIPCHK(ctx, B[B_BITMAP0 + s].ensure(nwords[s] * 8 + 64));
            HIPCHK(ctx, B[B_BLK0 + s].ensure(((size_t)nblk[s] + 2) * 4));
            scantmp = std::max(scantmp, ((size_t)nblk[s] / SCAN_TILE + 2) * 16);
        }
        HIPCHK(ctx, B[B_SCANTMP].ensure(scantmp));
        for (int s = 0; s < nstreams; s++) {
            if (!nblk[s]) continue;
            hipLaunchKernelGGL(k_nl_bitmap, dim3(nblk[s]), dim3(256), 0, S, fq[s], (uint32_t)nbytes[s], skip[s], B[B_BITMAP0 + s].as<uint64_t>(),
                    B[B_BLK0 + s].as<uint32_t>(), dst);
            KCHK(ctx, "k_nl_bitmap");
            scan_exclusive<uint32_t>(S, B[B_BLK0 + s].as<uint32_t>(), B[B_BLK0 + s].as<uint32_t>(), nblk[s], B[B_SCANTMP].as<uint32_t>(), 1);
        }
        for (int s = 0; s < nstreams; s++) {
            if (!nblk[s]) continue;
            HIPCHK(ctx, ctx->fetch(&n_newlines[s], B[B_BLK0 + s].as<uint32_t>() + nblk[s], 4, S));
            HIPCHK(ctx, ctx->fetch(&lastbyte[s], fq[s] + nbytes[s] - 1, 1, S));
        }
        HIPCHK(ctx, ctx->fetch(&hs, dst, sizeof hs, S));
        HIPCHK(ctx, ctx->fetch_sync(S));
    }
    if (!lazy && (hs.err & DE_HAS_CR)) return nm ? rfq_fail(ctx, RFQ_E_HIP, "internal: '\\r' in normalised text") : RFQ_NEED_NORM;
    uint32_t nlines[2] = { 0, 0 }, nrec[2] = { 0, 0 };
    for (int s = 0; s < nstreams && !lazy; s++) {
        // an unterminated tail is the file's last line only in the final batch; in a non-final batch it is a line cut by the
        // batch boundary and belongs to the next batch
        const int unterm = a->final && nbytes[s] > 0 && lastbyte[s] != '\n';
        nlines[s] = n_newlines[s] + (unterm ? 1u : 0u); nrec[s] = nlines[s] / 4;
        if (!one_pass || !nblk[s]) HIPCHK(ctx, B[B_LO0 + s].ensure(((size_t)nlines[s] + 4) * 4));
        if (nblk[s]) {
            if (!one_pass) hipLaunchKernelGGL(k_line_offsets, dim3(nblk[s]), dim3(256), 0, S, B[B_BITMAP0 + s].as<uint64_t>(), B[B_BLK0 + s].as<uint32_t>(),
                    (uint32_t)nbytes[s], skip[s], B[B_LO0 + s].as<uint32_t>());
            hipLaunchKernelGGL(k_line_tail, dim3(1), dim3(64), 0, S, B[B_LO0 + s].as<uint32_t>(), n_newlines[s], (uint32_t)nbytes[s], unterm);
            KCHK(ctx, "k_line_offsets");
        }
    }
    ctx->timer.end(S);
    // (a marker, not a phase: the index's totals stay on the device until the partition's read-back - tests look for it)
    if (lazy) { ctx->timer.begin("lazy_index", S); ctx->timer.end(S); }

    // ---- phase 2: read table, chunk cuts
    Text T; memset(&T, 0, sizeof T);
    for (int s = 0; s < 2; s++) { T.fq[s] = fq[s]; T.n[s] = (uint32_t)nbytes[s]; T.lo[s] = s < nstreams ? B[B_LO0 + s].as<uint32_t>() : nullptr;
            T.ot[s] = nm && s < nstreams ? nm->ot[s] : nullptr; }
    T.paired = a->paired; T.upr = a->paired == RFQ_SE ? 1u : 2u;
    uint32_t n_units = a->paired == RFQ_SE ? nrec[0] : (a->paired == RFQ_PE_TWO_FILES ? std::min(nrec[0], nrec[1]) : nrec[0] / 2);
    if (n_units > unit_cap) n_units = unit_cap;                       // the reader stopped at an empty line (src/fastqreader.cpp:180-191)
    if (lazy) n_units = guess_units;                                  // (what the tables are sized for; the true count comes back with the partition)
    uint32_t n_reads = n_units * T.upr; T.n_reads = n_reads;
    const uint32_t* const nu = lazy ? &dst->idx_units : (const uint32_t*)nullptr;      // where the kernels up to the partition find the unit count
    res->d_rfq = nullptr;
    if (n_units == 0) { ctx->chunk_off.assign(1, 0); res->h_chunk_off = ctx->chunk_off.data(); return RFQ_OK; }
    const bool is_pe = a->paired != RFQ_SE;

    // (an early return must not leave the second stream running over buffers that are about to be reused)
    struct AuxGuard { rfq_ctx* c; bool armed; void sync() { if (armed) { (void)hipStreamSynchronize(c->aux); armed = false; } } ~AuxGuard() { sync();
            } } ovl_guard = { ctx, false };
    ctx->timer.begin("lens+cut", S);
    const size_t nr = (size_t)n_reads + 2;
    HIPCHK(ctx, B[B_LEN].ensure(nr * 4)); HIPCHK(ctx, B[B_N1LEN].ensure(nr * 4)); HIPCHK(ctx, B[B_N2OFF].ensure(nr * 4));
    HIPCHK(ctx, B[B_X].ensure(nr * 4)); HIPCHK(ctx, B[B_Y].ensure(nr * 4)); HIPCHK(ctx, B[B_TILE].ensure(nr * 2)); HIPCHK(ctx, B[B_LANE].ensure(nr));
            HIPCHK(ctx, B[B_OK].ensure(nr));
    HIPCHK(ctx, B[B_CHUNK].ensure(nr * 4)); HIPCHK(ctx, B[B_STORED].ensure(nr * 4)); HIPCHK(ctx, B[B_EQ2].ensure(nr)); HIPCHK(ctx, B[B_PQ].ensure(nr * 4));
    HIPCHK(ctx, B[B_PV].ensure(nr * 16)); HIPCHK(ctx, B[B_PVIN].ensure(nr * 16));
    HIPCHK(ctx, B[B_ULEN].ensure(((size_t)n_units + 2) * 8)); HIPCHK(ctx, B[B_P].ensure(((size_t)n_units + 2) * 8));
    HIPCHK(ctx, B[B_SCANTMP].ensure(std::max<size_t>(1024, (nr / SCAN_TILE + 2) * 16)));
    ReadTab R;
    R.len = B[B_LEN].as<uint32_t>(); R.name1_len = B[B_N1LEN].as<uint32_t>(); R.name2_off = B[B_N2OFF].as<uint32_t>(); R.x = B[B_X].as<uint32_t>();
            R.y = B[B_Y].as<uint32_t>();
    R.tile = B[B_TILE].as<uint16_t>(); R.lane = B[B_LANE].as<uint8_t>(); R.ok = B[B_OK].as<uint8_t>(); R.chunk = B[B_CHUNK].as<uint32_t>();
            R.stored = B[B_STORED].as<uint32_t>();
    R.eq2 = B[B_EQ2].as<uint8_t>(); R.pq = B[B_PQ].as<uint32_t>(); R.pv = B[B_PV].as<U4>();
    HIPCHK(ctx, B[B_SD].ensure(nr * 4)); R.sd = B[B_SD].as<uint32_t>();
    // sequence lengths come from the line table alone; the names are parsed where the text is staged anyway (k_gather2), or by k_read_table for
    // the reads that need them earlier (chunk 0 of a first batch: the file header) / on the byte-wise gather path (all of them)
    const uint32_t ublocks = (n_units + 255) / 256;
    HIPCHK(ctx, B[B_MINMAX].ensure(((size_t)ublocks + 2) * LENS_BLK * 4));
    hipLaunchKernelGGL(k_read_lens, dim3(ublocks), dim3(256), 0, S, T, R.len, R.stored, B[B_ULEN].as<uint64_t>(), n_units, T.upr, B[B_MINMAX].as<uint32_t>(), dst, nu);
    KCHK(ctx, "k_read_lens");
    // every read the same length (sequencer output): both prefixes have a closed form - the scans see the flag and return, k_fill_pq writes g x L (no host round trip)
    uint32_t* const uni = B[B_MINMAX].as<uint32_t>() + (size_t)ublocks * LENS_BLK;
    hipLaunchKernelGGL(k_lens_uniform, dim3(1), dim3(1024), 0, S, (const uint32_t*)B[B_MINMAX].as<uint32_t>(), ublocks, n_units, uni, nu);
    // The two prefix scans (units for the cut, reads for the base prefix: six launches) see `uni` and return at once when every read has L bases.  A context that has
    // not met reads of several lengths in this file does not even launch them: k_partition says DE_NEED_SCAN if they were needed after all, and scans + partition run
    // then - one more round trip, once per file (ctx->mixed_lengths stays up until the header is cleared).
    auto prefix_scans = [&]() {
        scan_exclusive<uint64_t>(S, B[B_ULEN].as<uint64_t>(), B[B_P].as<uint64_t>(), n_units, B[B_SCANTMP].as<uint64_t>(), 1, uni);
        scan_exclusive<uint32_t>(S, R.len, R.pq, n_reads, B[B_SCANTMP].as<uint32_t>(), 1, uni);
    };
    bool have_scans = ctx->mixed_lengths;
    if (have_scans) prefix_scans();
    hipLaunchKernelGGL(k_fill_pq, dim3(n_reads / 256 + 1), dim3(256), 0, S, R.pq, n_reads, (const uint32_t*)uni, nu, T.upr);
    const uint64_t cap64 = (uint64_t)(nbytes[0] + nbytes[1]) / (2ull * a->chunk_bases) + 3;
    const uint32_t cap_chunks = (uint32_t)std::min<uint64_t>(cap64, (uint64_t)n_units + 1);
    HIPCHK(ctx, B[B_FIRST].ensure(((size_t)cap_chunks + 2) * 4));
    ChunkTab C; memset(&C, 0, sizeof C);
    C.first = B[B_FIRST].as<uint32_t>();
    if (a->carry_bases && !scan_only) return rfq_fail(ctx, RFQ_E_ARG, "carry_bases is for the plan pass (rfq_scan_batch): an encode starts on a chunk boundary");
    if (a->carry_bases >= a->chunk_bases) return rfq_fail(ctx, RFQ_E_ARG, "carry_bases must be < chunk_bases");
    for (;;) {
        hipLaunchKernelGGL(k_partition, dim3(1), dim3(1024), 0, S, (const uint64_t*)(B[B_P].as<uint64_t>() + 1), n_units, T.upr, a->chunk_bases, a->carry_bases, fin ? 1 : 0,
                           (const uint32_t*)B[B_MINMAX].as<uint32_t>(), ublocks, C.first, cap_chunks + 1, dst, (const uint32_t*)uni, have_scans ? 1 : 0, nu);
        KCHK(ctx, "k_partition");
        HIPCHK(ctx, ctx->fetch(&hs, dst, sizeof hs, S));
        HIPCHK(ctx, ctx->fetch_sync(S));
        if (have_scans || !(hs.err & DE_NEED_SCAN)) break;
        ctx->mixed_lengths = true; have_scans = true; prefix_scans();       // (the bit stays in the device's status word: nobody else reads it)
    }
    hs.err &= ~(uint32_t)DE_NEED_SCAN;
    ctx->timer.end(S);
    if (lazy) {
        // the index's verdict, which the other form reads right behind it
        if (hs.err & (DE_INDEX_RETRY | DE_UNITS_GUESS | DE_NEED_SCAN)) { ctx->lazy_block = true; if (hs.err & DE_NEED_SCAN) ctx->mixed_lengths = true;
                return encode_impl(ctx, a, res, nm, unit_cap, ended, scan_only, skip); }
        if (hs.err & DE_HAS_CR) return nm ? rfq_fail(ctx, RFQ_E_HIP, "internal: '\\r' in normalised text") : RFQ_NEED_NORM;
        for (int s = 0; s < nstreams; s++) { nlines[s] = hs.idx_lines[s]; nrec[s] = nlines[s] / 4; }
        n_units = hs.idx_units_true; n_reads = n_units * T.upr; T.n_reads = n_reads;
        if (n_units == 0) { ctx->chunk_off.assign(1, 0); res->h_chunk_off = ctx->chunk_off.data(); return RFQ_OK; }
    }
    { double r = 0.0; for (int s = 0; s < nstreams; s++) if (nbytes[s]) r = std::max(r, (double)nrec[s] / (double)nbytes[s]); if (r > 0.0) ctx->rec_per_byte = r; }
    if (hs.err & DE_EMPTY_LINE) {
        // "\n\n" is a swallowed blank line, not an empty one: classify the text properly first.  On normalised text an empty line is
        // where FastqReader::read returns NULL (src/fastqreader.cpp:180-191): the record and everything after it are never read.
        if (!nm) return RFQ_NEED_NORM;
        ovl_guard.sync();                                                   // (the repeat rebuilds nothing, but starts its own search over the same buffers)
        return encode_impl(ctx, a, res, nm, hs.first_empty / T.upr, true, scan_only, skip);
    }
    if (hs.err & DE_INTERNAL) return rfq_fail(ctx, RFQ_E_HIP, "internal: reads of one length, units of several (k_lens_uniform / k_partition disagree)");
    if (hs.err & DE_QUAL_SHORT) return rfq_fail(ctx, RFQ_E_UNPINNED, "a quality line is shorter than its sequence line (the reference reads past the string: undefined)");
    const uint32_t n_chunks = hs.n_chunks;
    if (n_chunks > cap_chunks) return rfq_fail(ctx, RFQ_E_HIP, "internal: chunk table overflow (%u > %u)", n_chunks, cap_chunks);
    if (n_chunks == 0) { ctx->chunk_off.assign(1, 0); res->h_chunk_off = ctx->chunk_off.data(); return RFQ_OK; }
    const uint32_t units_used = hs.n_units_used, reads_used = units_used * T.upr;
    const uint64_t total_bases = hs.total_bases;
    if (scan_only) {
        // rfq_scan_batch stops here: where every chunk ends in the caller's stream(s)
        HIPCHK(ctx, B[B_P].ensure(((size_t)n_chunks + 2) * 16));              // (the unit prefix is no longer needed)
        uint64_t* e1 = B[B_P].as<uint64_t>(); uint64_t* e2 = e1 + n_chunks + 1;
        hipLaunchKernelGGL(k_chunk_ends, dim3((n_chunks + 255) / 256), dim3(256), 0, S, T, (const uint32_t*)C.first, n_chunks,
                           nm ? nm->onx[0] : nullptr, (nm && nstreams == 2) ? nm->onx[1] : nullptr, e1, e2);
        KCHK(ctx, "k_chunk_ends");
        ctx->scan_end[0].assign(n_chunks, 0); ctx->scan_end[1].assign(nstreams == 2 ? n_chunks : 0, 0);
        HIPCHK(ctx, ctx->fetch(ctx->scan_end[0].data(), e1, (size_t)n_chunks * 8, S));
        if (nstreams == 2) HIPCHK(ctx, ctx->fetch(ctx->scan_end[1].data(), e2, (size_t)n_chunks * 8, S));
        HIPCHK(ctx, ctx->fetch_sync(S));
        const size_t lim0 = nm ? nm->orig_n[0] : nbytes[0], lim1 = nm ? nm->orig_n[1] : nbytes[1];
        // (a virtual terminator past an unterminated last line; offsets count from the stream's own first byte)
        for (auto& v : ctx->scan_end[0]) { if (v > lim0) v = lim0; if (!nm) v -= skip[0]; }
        for (auto& v : ctx->scan_end[1]) { if (v > lim1) v = lim1; if (!nm) v -= skip[1]; }
        res->n_chunks = n_chunks; res->n_reads = reads_used; res->n_bases = total_bases; res->reserved = (int32_t)hs.unit_bases;
        res->consumed1 = (size_t)ctx->scan_end[0].back(); res->consumed2 = nstreams == 2 ? (size_t)ctx->scan_end[1].back() : 0;
        ctx->timer.collect();
        return RFQ_OK;
    }

    // ---- phase 3: header (first batch), chunk analysis, gather, plan
    const size_t nc = (size_t)n_chunks + 2;
    HIPCHK(ctx, B[B_CFLAGS].ensure(nc * 4)); HIPCHK(ctx, B[B_IL].ensure(nc * 4)); HIPCHK(ctx, B[B_NCOUNT].ensure(nc * 4));
    HIPCHK(ctx, B[B_SCAP].ensure(nc * MAX_STREAMS * 4)); HIPCHK(ctx, B[B_SOFF].ensure(nc * MAX_STREAMS * 8)); HIPCHK(ctx, B[B_SSIZE].ensure(nc * MAX_STREAMS * 4));
    HIPCHK(ctx, B[B_XSIZE].ensure(nc * 4)); HIPCHK(ctx, B[B_YSIZE].ensure(nc * 4)); HIPCHK(ctx, B[B_QBASE].ensure(nc * 8)); HIPCHK(ctx, B[B_SBASE].ensure(nc * 8));
    HIPCHK(ctx, B[B_IMGSIZE].ensure(nc * 8)); HIPCHK(ctx, B[B_IMGOFF].ensure(nc * 8)); HIPCHK(ctx, B[B_CTOTAL].ensure(nc * 8)); HIPCHK(ctx, B[B_CBASE].ensure(nc * 8));
    HIPCHK(ctx, B[B_LAYOUT].ensure(nc * sizeof(Layout))); HIPCHK(ctx, B[B_HSTATS].ensure(sizeof(HdrStats) + 8192)); HIPCHK(ctx, B[B_OVB].ensure(nr / 2 + 16));
    const size_t catbytes = (size_t)total_bases + 64 * nc + 256;
    HIPCHK(ctx, B[B_QCAT].ensure(catbytes));
    HIPCHK(ctx, ctx->d_hdr.ensure(sizeof(DevHeader)));
    C.flags = B[B_CFLAGS].as<uint32_t>(); C.il = B[B_IL].as<uint32_t>(); C.ncount = B[B_NCOUNT].as<uint32_t>();
    HIPCHK(ctx, B[B_NMAP].ensure(nc * NMAP_WORDS * 4)); C.nmap = B[B_NMAP].as<uint32_t>();
    HIPCHK(ctx, B[B_PTOT].ensure(nc * sizeof(U4))); C.ptot = B[B_PTOT].as<U4>();
    C.scap = B[B_SCAP].as<uint32_t>(); C.soff = B[B_SOFF].as<uint64_t>(); C.ssize = B[B_SSIZE].as<uint32_t>(); C.xsize = B[B_XSIZE].as<uint32_t>();
            C.ysize = B[B_YSIZE].as<uint32_t>();
    C.qbase = B[B_QBASE].as<uint64_t>(); C.sbase = B[B_SBASE].as<uint64_t>(); C.img_size = B[B_IMGSIZE].as<uint64_t>(); C.img_off = B[B_IMGOFF].as<uint64_t>();
    DevHeader* D = ctx->d_hdr.as<DevHeader>();
    Layout* L = B[B_LAYOUT].as<Layout>();
    int8_t* ovb = B[B_OVB].as<int8_t>();
    const uint32_t max_reads = std::max(hs.max_chunk_reads, 1u);

    // SE: nothing before the gather needs the file header, so the (small, serial) header kernels of a first batch run on the aux
    // stream beside chunk ids / flags / prefix scans; PE needs it for the interleave test right away.
    const bool make_header = !ctx->have_hdr;
    const bool hdr_aside = make_header && !is_pe && ctx->aux_ready();
    hipStream_t HS = hdr_aside ? ctx->aux : S;
    if (hdr_aside) { HIPCHK(ctx, hipEventRecord(ctx->ev_fork, S)); HIPCHK(ctx, hipStreamWaitEvent(HS, ctx->ev_fork, 0)); }
    // Which gather: the tile gather k_gather2 (+ k_seqpack) whenever two records fit its staged-text buffer - tiles of K reads, the largest power of two
    // that always fits - else the byte-wise k_gather (+ k_packbytes).  RFQ_GATHER=old forces the latter (tests run both).
    uint32_t kshift = 6;
    while (kshift >= 1 && ((uint64_t)hs.max_rec << kshift) + 64u > G2_CAP) kshift--;
    const bool fast = kshift >= 1 && !ctx->opt.gather_old;
    const uint32_t max_rec = hs.max_rec;
    const uint32_t np = reads_used / 2;
    ctx->timer.begin("header", S);
    if (!fast) hipLaunchKernelGGL(k_chunk_ids, dim3((max_reads + 255) / 256, n_chunks), dim3(256), 0, S, C, R);   // (a read's chunk: only the byte-wise path's k_overlap_apply asks)
    // parsed names ahead of the gather: chunk 0's for the file header of a first batch; every read's on the byte-wise path
    const uint32_t c0_reads = std::max(1u, std::min(max_reads, reads_used));
    if (!fast) hipLaunchKernelGGL(k_read_table, dim3((n_reads + 255) / 256), dim3(256), 0, S, T, R, n_reads);
    else if (make_header) hipLaunchKernelGGL(k_read_table, dim3((c0_reads + 255) / 256), dim3(256), 0, S, T, R, c0_reads);
    if (hdr_aside) { HIPCHK(ctx, hipEventRecord(ctx->ev_fork, S)); HIPCHK(ctx, hipStreamWaitEvent(HS, ctx->ev_fork, 0)); }
    if (make_header) {
        HdrStats* H = B[B_HSTATS].as<HdrStats>();
        const uint32_t hb = std::min<uint32_t>(1024, (c0_reads + 3) / 4);
        hipLaunchKernelGGL(k_hdr_init, dim3(1), dim3(128), 0, HS, H);
        hipLaunchKernelGGL(k_hdr_stats, dim3(hb), dim3(256), 0, HS, T, R, (const uint32_t*)C.first, H);
        hipLaunchKernelGGL(k_hdr_q0, dim3(1), dim3(64), 0, HS, T, H);
        hipLaunchKernelGGL(k_hdr_pass2, dim3(hb), dim3(256), 0, HS, T, R, (const uint32_t*)C.first, H);
        if (is_pe) hipLaunchKernelGGL(k_hdr_pe, dim3((c0_reads / 2 + 255) / 256), dim3(256), 0, HS, T, R, (const uint32_t*)C.first, H);
        hipLaunchKernelGGL(k_hdr_finalize, dim3(1), dim3(64), 0, HS, T, H, D, is_pe ? 1 : 0, dst);
        if (fast) { hipLaunchKernelGGL(k_dense_order, dim3(1), dim3(64), 0, HS, (const HdrStats*)H, D); ctx->dense_ok = true; }
        KCHK(ctx, "k_hdr_*");
        if (hdr_aside) HIPCHK(ctx, hipEventRecord(ctx->ev_mid, HS));
    } else if (fast && !ctx->dense_ok) {
        // a header that was set, not made (rfq_set_header: a worker of a multi-GPU queue, a later file): which coded values are frequent is taken from this batch's chunk
        // 0
        HdrStats* H = B[B_HSTATS].as<HdrStats>();
        const uint32_t hb = std::min<uint32_t>(1024, (c0_reads + 3) / 4);
        hipLaunchKernelGGL(k_hdr_init, dim3(1), dim3(128), 0, S, H);
        hipLaunchKernelGGL(k_hdr_stats, dim3(hb), dim3(256), 0, S, T, R, (const uint32_t*)C.first, H);
        hipLaunchKernelGGL(k_dense_order, dim3(1), dim3(64), 0, S, (const HdrStats*)H, D);
        ctx->dense_ok = true;
        HIPCHK(ctx, ctx->fetch(ctx->h_hdr.dense, (const uint8_t*)D + offsetof(DevHeader, dense), 4, S));
        HIPCHK(ctx, ctx->fetch_sync(S));
    }
    if (make_header && fast) {
        // the tile gather is instantiated by the header (match masks for <= 4 coded quality values, bytes otherwise): a first batch waits for it here
        if (hdr_aside) HIPCHK(ctx, hipStreamWaitEvent(S, ctx->ev_mid, 0));
        HIPCHK(ctx, ctx->fetch(&ctx->h_hdr, D, sizeof(DevHeader), S));
        { DevStatus h2; HIPCHK(ctx, ctx->fetch(&h2, dst, sizeof h2, S)); HIPCHK(ctx, ctx->fetch_sync(S)); hs.err |= h2.err; hs.err_read = h2.err_read; hs.err_key = h2.err_key; }
    }
    ctx->timer.end(S);
    // RfqHeader::makeQualityTable's refusals (src/rfqheader.cpp:140-166), as soon as the header kernels' verdict is on the host (tile path: here; byte-wise path: behind its gather)
    auto header_errors = [&]() -> int {
        if (hs.err & (DE_BAD_QUAL | DE_BAD_BASE)) {
            const uint32_t g = hs.err_read, i = (uint32_t)hs.err_key; std::string ln;
            if (hs.err & DE_BAD_QUAL) { if (fetch_line(ctx, T, g, 3, ln)) return RFQ_E_HIP; return rfq_fail(ctx, RFQ_E_DATA, "bad quality value: %d", (int)(int8_t)ln[i]); }
            if (fetch_line(ctx, T, g, 1, ln)) return RFQ_E_HIP;
            const char b = ln[i];
            if (b == 'a' || b == 't' || b == 'c') return rfq_fail(ctx, RFQ_E_DATA, "repaq doesn't support FASTQ with lowercase bases (a/t/c/g)\nbut we get:\n%s", ln.c_str());
            return rfq_fail(ctx, RFQ_E_DATA, "repaq only supports FASTQ with uppercase bases (A/T/C/G/N)\nbut we get:\n%s", ln.c_str());
        }
        if (hs.err & DE_NO_QUAL_BINS) return rfq_fail(ctx, RFQ_E_DATA, "bad quality string, is this a valid FASTQ file?");
        if (make_header) { if (!ctx->h_hdr.valid) return rfq_fail(ctx, RFQ_E_HIP, "internal: header was not finalised"); ctx->have_hdr = true; ctx->hdr_on_device = true; }
        return RFQ_OK;
    };
    if (fast && make_header) { ovl_guard.sync(); const int rc = header_errors(); if (rc) return rc; }
    // match masks for files with at most four coded quality values (a NovaSeq-binned file: ':' ',' '#' and the 0xFF entry the reference's table gets when the
    // N bases have no quality of their own); the most frequent two or three get planes built in LDS, the others are set bit by bit
    const bool masks = fast && !ctx->opt.qual_bytes && ctx->h_hdr.valid && (ctx->h_hdr.flags & H_QUAL_BY_COL) && !(ctx->h_hdr.flags & H_DONT_QUAL) && ctx->h_hdr.n_normal >= 1u && ctx->h_hdr.n_normal <= 4u;

    HIPCHK(ctx, B[B_OVRAW].ensure((size_t)(is_pe ? n_units : 0) * 2 + 64));
    HIPCHK(ctx, B[B_SCANTMP2].ensure(std::max<size_t>(4096, (nr / SCAN_TILE + 2) * 16 + (nc / SCAN_TILE + 2) * 8)));
    HIPCHK(ctx, B[B_CTOTALN].ensure(nc * 8)); HIPCHK(ctx, B[B_CBASEN].ensure(nc * 8));
    const OvLoose noz = { nullptr, nullptr, nullptr, nullptr };

    // per-chunk accumulators of the read-0 / mate comparisons (CF_ALL): all ones; k_chunk_flags_b makes the flag words from them.  The tile gather
    // fills them itself (and the flags follow it); the byte-wise path needs the flags first (overlap search on the text, stored prefix).
    HIPCHK(ctx, B[B_ADJ].ensure(3 * nc * 4));
    uint32_t* cbits = B[B_ADJ].as<uint32_t>(); uint32_t* cfail = cbits + nc; uint32_t* redo = cfail + nc;
    // the position coder's per-(chunk, stream, 32768-position segment) tables: match counts and last matches are left by the gather
    const uint32_t pc_max_steps = (hs.max_chunk_bases + 4095u) / 4096u; const uint32_t n_seg = std::max(1u, (pc_max_steps + PC_SEG_STEPS - 1) / PC_SEG_STEPS);
    const size_t nsb = nc * MAX_STREAMS * (size_t)n_seg;
    HIPCHK(ctx, B[B_SEGB].ensure(nsb * 4)); HIPCHK(ctx, B[B_SEGC].ensure(nsb * 4)); HIPCHK(ctx, B[B_SEGM].ensure(nsb * 4));
    if (fast) {
        // every table of the batch that starts all-zero / all-ones, in one launch (k_clear_list)
        HIPCHK(ctx, B[B_RFLAG].ensure((nr + 15) & ~(size_t)15)); HIPCHK(ctx, B[B_RN].ensure((nr + 15) & ~(size_t)15));
        ClearList z; memset(&z, 0, sizeof z);
        z.add(cbits, 2 * nc * 4, 0xFFFFFFFFu); z.add(C.ncount, nc * 4, 0u); z.add(C.nmap, nc * NMAP_WORDS * 4, 0u);
        z.add(B[B_SEGB].p, nsb * 4, 0u); z.add(B[B_SEGM].p, nsb * 4, 0u); z.add(B[B_SEGC].p, nsb * 4, 0xFFFFFFFFu); z.add(B[B_RFLAG].p, nr, 0u);
        z.add(B[B_RN].p, nr, 0u);
        clear_list(S, z);
    } else HIPCHK(ctx, hipMemsetAsync(cbits, 0xFF, 2 * nc * 4, S));
    ctx->timer.begin("chunk_flags", S);
    hipLaunchKernelGGL(k_chunk_bases, dim3((n_chunks + 255) / 256), dim3(256), 0, S, R, C, n_chunks, 1);
    // the stored-base prefix (it needs the mates' overlaps): k_overlap_apply, per-read prefix inputs, their scan, the chunks' bases in the tight streams
    auto stored_prefix = [&](hipStream_t Q, U4* tmp) {
        if (is_pe) hipLaunchKernelGGL(k_overlap_apply, dim3((np + 255) / 256), dim3(256), 0, Q, R, C, (const DevHeader*)D, (const int16_t*)B[B_OVRAW].as<int16_t>(), ovb,
                np);
        hipLaunchKernelGGL(k_pv_in, dim3((n_reads + 255) / 256), dim3(256), 0, Q, T, R, B[B_PVIN].as<U4>(), n_reads);
        scan_exclusive<U4>(Q, B[B_PVIN].as<U4>(), R.pv, n_reads, tmp, 1);
        hipLaunchKernelGGL(k_chunk_bases, dim3((n_chunks + 255) / 256), dim3(256), 0, Q, R, C, n_chunks, 2);
        hipLaunchKernelGGL(k_chunk_ptot, dim3((n_chunks + 255) / 256), dim3(256), 0, Q, R, C, n_chunks);
    };
    if (hdr_aside) HIPCHK(ctx, hipStreamWaitEvent(S, ctx->ev_mid, 0));      // from here on everything needs the header (major quality, flags, the mates' name2 rule)
    if (!fast) {
        // byte-wise gather: it writes the stored bases themselves, so chunk flags, the overlap search (on the text) and the stored prefix come first
        const uint32_t fbx = std::max(1u, std::min<uint32_t>((max_reads + 255) / 256, std::max(1u, 4096u / n_chunks)));
        hipLaunchKernelGGL(k_chunk_flags_a, dim3(fbx, n_chunks), dim3(256), 0, S, T, R, C, (const DevHeader*)D, is_pe ? 1 : 0, cbits, cfail);
        hipLaunchKernelGGL(k_chunk_flags_b, dim3(n_chunks), dim3(64), 0, S, R, C, (const DevHeader*)D, is_pe ? 1 : 0, (const uint32_t*)cbits, (const uint32_t*)cfail,
                (uint32_t*)nullptr);
        if (is_pe) {
            const uint32_t ob = std::min<uint32_t>((n_units + 255) / 256, 65535u * 16u);
            hipLaunchKernelGGL(k_overlap<false>, dim3(ob), dim3(256), 0, S, T, noz, B[B_OVRAW].as<int16_t>(), n_units);
        }
        stored_prefix(S, B[B_SCANTMP].as<U4>());
        KCHK(ctx, "k_chunk_flags");
    }
    ctx->timer.end(S);

    // (a marker, not a phase: k_gather2 leaves match masks instead of quality bytes - tests and the bench look for it)
    if (masks) { ctx->timer.begin("quality_masks", S); ctx->timer.end(S); }
    ctx->timer.begin(fast ? "gather" : "gather_bytes", S);                  // (which formulation ran: tests and the bench look at it)
    HIPCHK(ctx, B[B_SPK].ensure((catbytes >> 4) * 4 + 64)); HIPCHK(ctx, B[B_SNM].ensure((catbytes >> 4) * 2 + 64));
    if (!fast) {
        HIPCHK(ctx, hipMemsetAsync(C.ncount, 0, nc * 4, S)); HIPCHK(ctx, hipMemsetAsync(C.nmap, 0, nc * NMAP_WORDS * 4, S));
        HIPCHK(ctx, hipMemsetAsync(B[B_SEGB].p, 0, nsb * 4, S)); HIPCHK(ctx, hipMemsetAsync(B[B_SEGM].p, 0, nsb * 4, S));
                HIPCHK(ctx, hipMemsetAsync(B[B_SEGC].p, 0xFF, nsb * 4, S));
    }
    uint64_t* const ctot = B[B_CTOTAL].as<uint64_t>(); uint64_t* const cbase = B[B_CBASE].as<uint64_t>(); uint64_t* const ctot_n = B[B_CTOTALN].as<uint64_t>();
            uint64_t* const cbase_n = B[B_CBASEN].as<uint64_t>();
    bool aux_chain = false, coder_waits = false;
    if (fast) {
        const size_t nld = (size_t)(total_bases >> 4) + reads_used + 16;
        HIPCHK(ctx, B[B_LPK].ensure(nld * 4)); HIPCHK(ctx, B[B_LNB].ensure(nld * 2));
        const uint32_t K = 1u << kshift;
        const uint32_t bx = grid_x_for(n_chunks, (max_reads + K - 1) / K, 6u * ctx->n_cu);      // (26 KB of LDS: six workgroups per CU)
        // dynamic LDS of k_gather2: the staged text of K of the batch's longest records (+ slack), read 0's name / strand line, and - match-mask mode - three
        // bit planes of K of the longest reads.  Six workgroups per CU need <= 26.8 KB each (measured: with five the kernel is 10 % slower).
        const uint32_t text4 = (uint32_t)((((uint64_t)max_rec << kshift) + 64u + 15u) / 16u) + 8u;
        G2Planes M; M.planes = nullptr; M.rare = nullptr; M.pstride = 0; M.nd = 0; M.pw = (uint32_t)((((uint64_t)hs.max_len << kshift) + 31u) / 32u) + 2u;
        auto dyn_of = [&](uint32_t nd_) -> uint32_t { return text4 * 16u + (G2_REFN + G2_REFS + 32u) + 4u * nd_ * M.pw; };
        if (masks) {
            // dense planes: three if the workgroup still fits six to a CU (26.8 KB of LDS each: with five the kernel is 10 % slower), else two
            M.nd = std::min<uint32_t>(ctx->h_hdr.n_normal, 3u);
            if (M.nd == 3u && dyn_of(3u) + 64u > 26880u) M.nd = 2u;
            // five planes laid out by the buffer's capacity (so that the planes' places are fixed while the buffer is), + rare[n_chunks] behind them
            const size_t need_w = (catbytes >> 5) + 16, extra_w = nc * (1u + G2_RARE_LIST) / G2_PLANES + 16;
            if (B[B_QPLANE].cap / 4 / G2_PLANES < need_w + extra_w || ctx->qplane_stride < need_w) {
                HIPCHK(ctx, B[B_QPLANE].ensure((need_w + extra_w) * G2_PLANES * 4)); ctx->qplane_stride = B[B_QPLANE].cap / 4 / G2_PLANES - extra_w;
                        ctx->qplane_dirty = true;
            }
            if ((ctx->qplane_stride + extra_w) * G2_PLANES * 4 > B[B_QPLANE].cap) { ctx->qplane_stride = B[B_QPLANE].cap / 4 / G2_PLANES - extra_w;
                    ctx->qplane_dirty = true; }
            uint32_t dmask = 0; for (uint32_t d = 0; d < M.nd; d++) dmask |= 1u << (ctx->h_hdr.dense[d] & 7u);
            if (ctx->qplane_mask & ~dmask) ctx->qplane_dirty = true;        // (a plane that was stored whole is now set bit by bit: it has to start all-zero)
            ctx->qplane_mask = dmask;
            M.planes = B[B_QPLANE].as<uint32_t>(); M.pstride = ctx->qplane_stride; M.rare = M.planes + G2_PLANES * M.pstride;
            if (ctx->qplane_dirty) HIPCHK(ctx, hipMemsetAsync(M.planes, 0, ((size_t)M.pstride * G2_PLANES + nc * (1u + G2_RARE_LIST)) * 4, S));
            ctx->qplane_dirty = true; ctx->qplane_nd = M.nd;                // (dirty until this call's cleanup is queued)
        }
        const uint32_t dyn = dyn_of(M.nd) + ctx->opt.g2_pad; (void)dyn;   // (the interpreter's launch macro takes its dynamic LDS from a buffer of its own)
        // phase 1: every chunk, names parsed on the way, mates taken for interleaved wherever the header allows; then the flag words; then phase 2 for the
        // (rare) chunks whose interleave test failed somewhere: their workgroups are the only ones of that launch that do not return at once
        for (int phase = 1; phase <= (is_pe ? 2 : 1); phase++) {
            const uint32_t* only = phase == 2 ? (const uint32_t*)redo : (const uint32_t*)nullptr;
            if (phase == 2) hipLaunchKernelGGL(k_gather_redo_reset, dim3(n_chunks), dim3(64), 0, S, only, B[B_SEGM].as<uint32_t>(), B[B_SEGC].as<int>(), n_seg,
                                               masks ? M.planes : (uint32_t*)nullptr, M.pstride, (const DevHeader*)D, M.nd, (const uint32_t*)R.pq, (const uint32_t*)C.first, (const uint64_t*)C.qbase);
            if (masks) hipLaunchKernelGGL(k_mask_bounds, dim3(n_chunks), dim3(64), 0, S, (const uint32_t*)R.pq, (const uint32_t*)C.first, (const uint64_t*)C.qbase,
                    M.planes, M.pstride, (const DevHeader*)D, M.nd, bx, only);
#define RFQ_G2_ARGS T, R, (const uint32_t*)C.first, (const uint64_t*)C.qbase, (const DevHeader*)D, B[B_QCAT].as<uint8_t>(), B[B_LPK].as<uint32_t>(), B[B_LNB].as<uint16_t>(), B[B_RFLAG].as<uint8_t>(), \
                    B[B_RN].as<uint8_t>(), B[B_SEGM].as<uint32_t>(), B[B_SEGC].as<int>(), n_seg, kshift, cbits, cfail, only, text4, M
            // (single-end input with match masks: the instantiation without mates - 132 spilled SGPRs instead of 182, no VGPR in scratch; the byte-stream form of it
            // spills 64 VGPRs instead and is not used)
            if (masks && !is_pe && G2_SE_OK) hipLaunchKernelGGL((k_gather2<true, 0>), dim3(bx, n_chunks), dim3(256), dyn, S, RFQ_G2_ARGS);
            else if (masks && a->paired == RFQ_PE_TWO_FILES && G2_SE_OK) hipLaunchKernelGGL((k_gather2<true, 1>), dim3(bx, n_chunks), dim3(256), dyn, S, RFQ_G2_ARGS);
            else if (masks) hipLaunchKernelGGL(k_gather2<true>, dim3(bx, n_chunks), dim3(256), dyn, S, RFQ_G2_ARGS);
            else hipLaunchKernelGGL(k_gather2<false>, dim3(bx, n_chunks), dim3(256), dyn, S, RFQ_G2_ARGS);
#undef RFQ_G2_ARGS
            if (phase == 1) hipLaunchKernelGGL(k_chunk_flags_b, dim3(n_chunks), dim3(64), 0, S, R, C, (const DevHeader*)D, is_pe ? 1 : 0, (const uint32_t*)cbits,
                    (const uint32_t*)cfail, redo);
        }
        // the quality streams' scratch plan needs nothing else: the position coder can start as soon as the host has sized its arena
        // The arenas of the coded streams and the image are sized BEFORE their sizes exist (what the context holds from earlier batches, or a guess from the bases): no
        // read-back between the gather and the coders, none behind the second chain.  A total beyond its arena raises DE_SCRATCH(N)_SMALL on the device - the coders and
        // the assembler leave at once - and the batch is repeated with room (RFQ_RETRY_ROOM: once per context as a rule, the arenas keep their size).
        // (the header tells the two shapes apart: a file with at most four coded quality values - match masks - codes a few percent of its positions; one with
        // forty codes most of them, a byte or so each)
        HIPCHK(ctx, B[B_SCRATCH].ensure(std::max<size_t>(B[B_SCRATCH].cap, (size_t)(masks ? total_bases / 8 : total_bases + total_bases / 4) + nc * 4096 + 256)));
        HIPCHK(ctx, B[B_SCRATCHN].ensure(std::max<size_t>(B[B_SCRATCHN].cap, (size_t)(total_bases / 64) + nc * 1024 + 256)));
        hipLaunchKernelGGL(k_stream_plan, dim3(n_chunks), dim3(64), 0, S, R, C, (const DevHeader*)D, ctot, ctot_n, n_chunks, (const uint32_t*)B[B_SEGM].as<uint32_t>(),
                n_seg, 1);
        scan_exclusive<uint64_t>(S, ctot, cbase, n_chunks, B[B_SCANTMP].as<uint64_t>(), 1);
        hipLaunchKernelGGL(k_enc_totals, dim3(1), dim3(64), 0, S, C, (const uint64_t*)cbase, n_chunks, 0, dst, (uint64_t)B[B_SCRATCH].cap);
        // Second chain (aux stream), beside the position coder: overlap search on the loose slots the gather has just left, stored prefix, sequence packer
        // (tight 2-bit stream + N mask + N counts), the N streams' plan, the image's upper bound.  These are chains of small latency-bound kernels
        // and a search that is VALU-bound; the coder hides them.
        aux_chain = ctx->aux_ready() && !ctx->opt.one_stream; hipStream_t A = aux_chain ? ctx->aux : S;
        if (aux_chain) { HIPCHK(ctx, hipEventRecord(ctx->ev_fork, S)); HIPCHK(ctx, hipStreamWaitEvent(A, ctx->ev_fork, 0)); ovl_guard.armed = true; }
        if (is_pe) {
            const OvLoose Z = { (const uint32_t*)R.pq, (const uint32_t*)B[B_LPK].as<uint32_t>(), (const uint16_t*)B[B_LNB].as<uint16_t>(), (const uint8_t*)B[B_RFLAG].as<uint8_t>() };
            const uint32_t ob = std::min<uint32_t>((np + 255) / 256, 65535u * 16u);
            // (rows of 160 bases where no read is longer: sixteen resident waves per CU instead of twelve)
            if (hs.max_len <= 160u) hipLaunchKernelGGL((k_overlap<true, 160u>), dim3(ob), dim3(256), 0, A, T, Z, B[B_OVRAW].as<int16_t>(), np);
            else hipLaunchKernelGGL(k_overlap<true>, dim3(ob), dim3(256), 0, A, T, Z, B[B_OVRAW].as<int16_t>(), np);
            // The search and the position coder are both VALU-bound: side by side they only share the issue slots, and the latency-bound chain behind the search
            // (stored prefix -> sequence packer -> N plan -> N coder) then runs alone, with nothing to hide its round trips (round 4's timeline: coder 1.6 ms and
            // search 2.5 ms together, then 1.9 ms of that chain on an empty device).  The coder waits for the search instead and runs beside the chain
            // (4.4 -> 4.1 ms for the stage.  The packer beside the coder still takes twice its time alone - the coder's single-wave workgroups take the slots
            // that free up - and on a stream of the highest priority it is the other way round, 3.1 ms for the coder: the two kernels take turns, in either order).
        }
        // (the coder waits for the search only: the stored prefix behind it is bound by memory and shares the device well - 3.55 -> 3.48 ms for the phase)
        if (aux_chain) { HIPCHK(ctx, hipEventRecord(ctx->ev_ovl, A)); coder_waits = true; }
        hipLaunchKernelGGL(k_chunk_prefix, dim3(n_chunks), dim3(256), 0, A, T, R, C, (const DevHeader*)D, (const int16_t*)B[B_OVRAW].as<int16_t>(), ovb);
        {
            const uint32_t max_len = max_rec / 2u;                             // (a record holds its sequence twice over: bases and qualities)
            // reads per step of k_seqpack: as many as keep the step's tight dwords inside its owner table (a read of L bases owns at most L / 16 + 1)
            uint32_t rshift = 8; while (rshift && ((uint64_t)(max_len / 16u + 1u) << rshift) > SP_OWN) rshift--;
            uint32_t sx = grid_x_for(n_chunks, (max_reads >> rshift) + 1u, 8u * ctx->n_cu);
            hipLaunchKernelGGL(k_seqpack, dim3(sx, n_chunks), dim3(256), aux_chain ? ctx->opt.sp_pad : 0u, A, (const uint32_t*)R.pq, (const uint32_t*)R.sd, (const U4*)C.ptot,
                    (const uint32_t*)C.first, (const uint32_t*)C.il, (const int8_t*)ovb, (const DevHeader*)D,
                               (const uint64_t*)C.sbase, (const uint32_t*)B[B_LPK].as<uint32_t>(), (const uint16_t*)B[B_LNB].as<uint16_t>(), (const uint8_t*)B[B_RN].as<uint8_t>(), B[B_SPK].as<uint32_t>(), B[B_SNM].as<uint16_t>(),
                               C.ncount, C.nmap, B[B_SEGM].as<uint32_t>(), B[B_SEGC].as<int>(), n_seg, rshift);
        }
        uint64_t* tmp2 = B[B_SCANTMP2].as<uint64_t>() + (nr / SCAN_TILE + 2) * 2;   // (behind the U4 scan's part of the buffer)
        hipLaunchKernelGGL(k_stream_plan, dim3(n_chunks), dim3(64), 0, A, R, C, (const DevHeader*)D, ctot, ctot_n, n_chunks, (const uint32_t*)B[B_SEGM].as<uint32_t>(),
                n_seg, 2);
        scan_exclusive<uint64_t>(A, ctot_n, cbase_n, n_chunks, tmp2, 1);
        hipLaunchKernelGGL(k_chunk_layout, dim3((n_chunks + 63) / 64), dim3(64), 0, A, T, R, C, (const DevHeader*)D, L, n_chunks, 0, dst);
        scan_exclusive<uint64_t>(A, C.img_size, C.img_off, n_chunks, tmp2, 1);
        hipLaunchKernelGGL(k_enc_totals, dim3(1), dim3(64), 0, A, C, (const uint64_t*)cbase_n, n_chunks, 2, dst, (uint64_t)B[B_SCRATCHN].cap);
        KCHK(ctx, "k_gather2");
    } else {
        HIPCHK(ctx, B[B_SCAT].ensure(catbytes));
        // workgroups per chunk: each takes a contiguous run of reads in tiles of <= 32
        const uint32_t bx = grid_x_for(n_chunks, (max_reads + GT_READS - 1) / GT_READS, 5u * ctx->n_cu);   // (30 KB of LDS: five workgroups per CU)
        hipLaunchKernelGGL(k_gather, dim3(bx, n_chunks), dim3(256), 0, S, T, R, C, (const int8_t*)ovb, (const DevHeader*)D, B[B_QCAT].as<uint8_t>(),
                B[B_SCAT].as<uint8_t>(), B[B_SEGM].as<uint32_t>(), B[B_SEGC].as<int>(), n_seg);
        const uint32_t px = grid_x_for(n_chunks, (hs.max_chunk_bases / 16u + 255u) / 256u + 1u, 8u * ctx->n_cu);
        hipLaunchKernelGGL(k_packbytes, dim3(px, n_chunks), dim3(256), 0, S, (const U4*)R.pv, (const uint32_t*)C.first, (const uint64_t*)C.sbase,
                (const uint8_t*)B[B_SCAT].as<uint8_t>(),
                           B[B_SPK].as<uint32_t>(), B[B_SNM].as<uint16_t>());
        hipLaunchKernelGGL(k_stream_plan, dim3(n_chunks), dim3(64), 0, S, R, C, (const DevHeader*)D, ctot, ctot_n, n_chunks, (const uint32_t*)B[B_SEGM].as<uint32_t>(),
                n_seg, 3);
        scan_exclusive<uint64_t>(S, ctot, cbase, n_chunks, B[B_SCANTMP].as<uint64_t>(), 1);
        scan_exclusive<uint64_t>(S, ctot_n, cbase_n, n_chunks, B[B_SCANTMP].as<uint64_t>(), 1);
        hipLaunchKernelGGL(k_chunk_layout, dim3((n_chunks + 63) / 64), dim3(64), 0, S, T, R, C, (const DevHeader*)D, L, n_chunks, 0, dst);
        scan_exclusive<uint64_t>(S, C.img_size, C.img_off, n_chunks, B[B_SCANTMP].as<uint64_t>(), 1);
        hipLaunchKernelGGL(k_enc_totals, dim3(1), dim3(64), 0, S, C, (const uint64_t*)cbase, n_chunks, 0, dst, ~0ull);
        hipLaunchKernelGGL(k_enc_totals, dim3(1), dim3(64), 0, S, C, (const uint64_t*)cbase_n, n_chunks, 2, dst, ~0ull);
        KCHK(ctx, "k_gather");
    }
    if (!fast) {
        // byte-wise path: arenas by their exact sizes (a read-back here), the header's verdict with them
        HIPCHK(ctx, ctx->fetch(&hs, dst, sizeof hs, S));
        if (make_header) HIPCHK(ctx, ctx->fetch(&ctx->h_hdr, D, sizeof(DevHeader), S));
        HIPCHK(ctx, ctx->fetch_sync(S));
    }
    ctx->timer.end(S);
    if (!fast) { const int rc = header_errors(); if (rc) return rc; }
    const DevHeader& HH = ctx->h_hdr;

    // ---- phase 4: code streams, exact layout, assemble
    if (!fast) HIPCHK(ctx, B[B_SCRATCH].ensure((size_t)hs.total_scratch + 256));
    HIPCHK(ctx, B[B_XS].ensure(3 * nr + 64)); HIPCHK(ctx, B[B_YS].ensure(3 * nr + 64));
    const uint32_t nqg = (std::min<uint32_t>(HH.n_normal, NPOS_SLOT) + PC_G - 1) / PC_G;              // quality-value streams, PC_G per wave
    // the value streams of a file with many coded quality values (no match masks): the list coder - one wave per (chunk, segment) for all of them, work
    // proportional to the coded positions - instead of a wave per four streams testing every position (RFQ_CODER=list / mask force one or the other)
    const bool coder_list = !masks && !(HH.flags & H_DONT_QUAL) && (HH.flags & H_QUAL_BY_COL) && HH.n_normal >= 1 && (ctx->opt.coder == 1 || (ctx->opt.coder == 0 && HH.n_normal >= 5));
    auto launch_coder = [&](hipStream_t Q, uint32_t g0, uint32_t gn) -> int {
        if (coder_list && g0 == 0 && gn >= nqg) {                          // the value streams; what is left of the request (exception group, N group) below
            const uint64_t mb = (uint64_t)((n_chunks + 7) / 8) * 8ull * n_seg;
            if (mb > 0x7FFFFFFFull) return rfq_fail(ctx, RFQ_E_ARG, "batch too large for the position-coder grid");
            hipLaunchKernelGGL(k_pos_coder_list, dim3((uint32_t)mb), dim3(64), std::min<uint32_t>(HH.n_normal, NPOS_SLOT) * 128u, Q, R, C, (const DevHeader*)D,
                    (const uint8_t*)B[B_QCAT].as<uint8_t>(), B[B_SCRATCH].as<uint8_t>(), (const uint64_t*)cbase,
                               B[B_SEGB].as<uint32_t>(), (const int*)B[B_SEGC].as<int>(), (const uint32_t*)B[B_SEGM].as<uint32_t>(), n_seg, n_chunks, dst);
            g0 = nqg; gn -= nqg;
            if (gn == 0) return RFQ_OK;
        }
        const uint64_t pc_blocks = (uint64_t)((n_chunks + 7) / 8) * 8ull * gn * n_seg;
        if (pc_blocks > 0x7FFFFFFFull) return rfq_fail(ctx, RFQ_E_ARG, "batch too large for the position-coder grid");
        hipLaunchKernelGGL(k_pos_coder, dim3((uint32_t)pc_blocks), dim3(64), 0, Q, R, C, (const DevHeader*)D, (const uint8_t*)B[B_QCAT].as<uint8_t>(),
                (const uint16_t*)B[B_SNM].as<uint16_t>(),
                           B[B_SCRATCH].as<uint8_t>(), (const uint64_t*)cbase, B[B_SCRATCHN].as<uint8_t>(), (const uint64_t*)cbase_n,
                           B[B_SEGB].as<uint32_t>(), (const int*)B[B_SEGC].as<int>(), (const uint32_t*)B[B_SEGM].as<uint32_t>(), n_seg, n_chunks, nqg, g0, gn, dst,
                           masks ? (const uint32_t*)B[B_QPLANE].as<uint32_t>() : (const uint32_t*)nullptr, (uint64_t)ctx->qplane_stride);
        return RFQ_OK;
    };
    ctx->timer.begin("pos_coder", S);
    const bool fork_coords = ctx->aux_ready();
    if (!fast && fork_coords) { HIPCHK(ctx, hipEventRecord(ctx->ev_fork, S)); HIPCHK(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0)); }
    if (fast) {
        // the quality / exception streams now; the N streams when the second chain has planned them (its totals come back while the coder runs)
        if (coder_waits) HIPCHK(ctx, hipStreamWaitEvent(S, ctx->ev_ovl, 0));
        { const int rc = launch_coder(S, 0, nqg + 1); if (rc) return rc; }
        // the coordinate coder (one dependent chain of ~100 steps per (axis, chunk)) needs nothing of either chain: behind the coder on the main stream
        hipLaunchKernelGGL(k_coords, dim3(2, n_chunks), dim3(64), 0, S, R, C, (const DevHeader*)D, B[B_XS].as<uint8_t>(), B[B_YS].as<uint8_t>(), dst);
    }
    if (!fast) HIPCHK(ctx, B[B_SCRATCHN].ensure((size_t)hs.total_scratch_n + 256));
    const uint64_t hdr_bytes = a->emit_header ? HH.len : 0;
    uint8_t* img; uint64_t img_cap;
    if (a->d_out) { img = a->d_out; img_cap = a->out_cap; }
    else {
        // (tile path: the image's bound is not on the host - what the context holds, or a third of the text to begin with (7/8 of it for a file with many coded quality
        // values); k_assemble checks every chunk against the room)
        const size_t nb_all = nbytes[0] + nbytes[1];
        const size_t want = fast ? std::max<size_t>(ctx->out_img.cap, (masks ? nb_all / 3 : nb_all - nb_all / 8) + (1u << 20)) : (size_t)(hs.image_bound + hdr_bytes + 64);
        HIPCHK(ctx, ctx->out_img.ensure(want)); img = ctx->out_img.as<uint8_t>(); img_cap = ctx->out_img.cap;
    }
    if (hdr_bytes) {
        if (img_cap < hdr_bytes) return rfq_fail(ctx, RFQ_E_NOSPACE, "output buffer too small for the header");
        HIPCHK(ctx, hipMemcpyAsync(img, HH.bytes, hdr_bytes, hipMemcpyHostToDevice, S));
    }
    // byte-wise path: the coordinate coder runs beside the quality streams on the aux stream; tile path: the N streams behind the second chain (which has
    // planned them)
    {
        hipStream_t A2 = (fast ? aux_chain : fork_coords) ? ctx->aux : S;
        if (!fast) hipLaunchKernelGGL(k_coords, dim3(2, n_chunks), dim3(64), 0, A2, R, C, (const DevHeader*)D, B[B_XS].as<uint8_t>(), B[B_YS].as<uint8_t>(), dst);
        if (fast) { const int rc = launch_coder(A2, nqg + 1, 1); if (rc) return rc; }
        else { const int rc = launch_coder(S, 0, nqg + 2); if (rc) return rc; }
        if (A2 != S) { HIPCHK(ctx, hipEventRecord(ctx->ev_join, A2)); HIPCHK(ctx, hipStreamWaitEvent(S, ctx->ev_join, 0)); }
    }
    KCHK(ctx, "k_pos_coder");
    if (masks) {                                                            // the rare planes back to all-zero (stream-ordered behind the coder that read them)
        hipLaunchKernelGGL(k_rare_cleanup, dim3(n_chunks), dim3(256), 0, S, B[B_QPLANE].as<uint32_t>() + G2_PLANES * ctx->qplane_stride, B[B_QPLANE].as<uint32_t>(),
                (uint64_t)ctx->qplane_stride,
                           (const DevHeader*)D, ctx->qplane_nd, (const uint32_t*)R.pq, (const uint32_t*)C.first, (const uint64_t*)C.qbase);
        ctx->qplane_dirty = false;
    }
    ctx->timer.end(S);
    ctx->timer.begin("coords+layout", S);
    HIPCHK(ctx, B[B_SEGD].ensure(nsb * 4)); HIPCHK(ctx, B[B_SEGS].ensure(nsb * 4));
    hipLaunchKernelGGL(k_pos_sizes, dim3(n_chunks), dim3(64), 0, S, C, (const DevHeader*)D, (const uint32_t*)B[B_SEGB].as<uint32_t>(),
            (const uint32_t*)B[B_SEGM].as<uint32_t>(), n_seg, B[B_SEGD].as<uint32_t>(), B[B_SEGS].as<uint32_t>());
    hipLaunchKernelGGL(k_chunk_layout, dim3((n_chunks + 63) / 64), dim3(64), 0, S, T, R, C, (const DevHeader*)D, L, n_chunks, 1, dst);
    scan_exclusive<uint64_t>(S, C.img_size, C.img_off, n_chunks, B[B_SCANTMP].as<uint64_t>(), 1);
    hipLaunchKernelGGL(k_enc_totals, dim3(1), dim3(64), 0, S, C, (const uint64_t*)cbase, n_chunks, 1, dst, ~0ull);
    KCHK(ctx, "k_coords");
    ctx->timer.end(S);
    ctx->timer.begin("assemble", S);
    {
        // the line-break bit of the input's tail chunk looks at how far the readers got on their last, failed attempt (see k_assemble): when this
        // call ends the input - at its end or at an empty line - and not at a worker's chunk boundary
        const uint32_t tail_bases = ((a->final && !a->flush_all) || ended) ? a->chunk_bases : 0u;
        const uint32_t bpc = grid_x_for(n_chunks, 64u, 8u * ctx->n_cu);       // (no LDS, 28 VGPRs: eight workgroups per CU)
        hipLaunchKernelGGL(k_assemble, dim3(bpc, n_chunks), dim3(256), 0, S, T, R, C, (const DevHeader*)D, (const Layout*)L,
                           (const uint8_t*)B[B_QCAT].as<uint8_t>(), (const uint32_t*)B[B_SPK].as<uint32_t>(), (const uint8_t*)B[B_SCRATCH].as<uint8_t>(), (const uint64_t*)cbase,
                           (const uint8_t*)B[B_SCRATCHN].as<uint8_t>(), (const uint64_t*)cbase_n,
                           (const uint8_t*)B[B_XS].as<uint8_t>(), (const uint8_t*)B[B_YS].as<uint8_t>(), (const int8_t*)ovb, img, img_cap, hdr_bytes,
                           a->file_off1, a->file_off2, a->nolb_from1, a->nolb_from2,
                           (const uint32_t*)B[B_SEGB].as<uint32_t>(), (const uint32_t*)B[B_SEGD].as<uint32_t>(), (const uint32_t*)B[B_SEGS].as<uint32_t>(), n_seg, dst,
                           tail_bases, units_used, nlines[0], nlines[1], (uint64_t)(nm ? nm->orig_n[0] : nbytes[0]), (uint64_t)(nm ? nm->orig_n[1] : nbytes[1]));
        // (a wave per eight reads, three dependent loads each: as many waves as there are groups of eight, not a serial walk per wave)
        // (most files share their names' fixed parts: the workgroups of such chunks leave at once, so the grid stays small - a workgroup loops over its share)
        const uint32_t bx = std::max(1u, std::min<uint32_t>((max_reads + 31) / 32, std::max(1u, 16384u / n_chunks)));
        hipLaunchKernelGGL(k_assemble_names, dim3(bx, n_chunks), dim3(256), 0, S, T, R, C, (const DevHeader*)D, (const Layout*)L, img, img_cap, hdr_bytes);
        KCHK(ctx, "k_assemble");
    }
    ctx->timer.end(S);
    ctx->chunk_off.resize((size_t)n_chunks + 1);
    HIPCHK(ctx, ctx->fetch(ctx->chunk_off.data(), C.img_off, ((size_t)n_chunks + 1) * 8, S));
    uint32_t cons[2] = { 0, 0 };
    for (int s = 0; s < nstreams; s++) {
        const uint32_t recs = a->paired == RFQ_PE_INTERLEAVED ? reads_used : (a->paired == RFQ_PE_TWO_FILES ? units_used : reads_used);
        if (nm) { cons[s] = 0; if (recs) HIPCHK(ctx, ctx->fetch(&cons[s], nm->onx[s] + 4 * (size_t)recs - 1, 4, S)); }
        else HIPCHK(ctx, ctx->fetch(&cons[s], B[B_LO0 + s].as<uint32_t>() + 4 * (size_t)recs, 4, S));
    }
    HIPCHK(ctx, ctx->fetch(&hs, dst, sizeof hs, S));
    HIPCHK(ctx, ctx->fetch_sync(S));
    ovl_guard.armed = false;                                                // (the second chain was joined in front of the assembler)
    ctx->timer.collect();
    if (hs.err & DE_COORD_RANGE) {
        // RfqCodec::encodeCoords error_exit, src/rfqcodec.cpp:1315-1317: first offender in (chunk, x-before-y, index) order
        const uint32_t c = (uint32_t)(hs.coord_key >> 34), axis = (uint32_t)((hs.coord_key >> 33) & 1u), i = (uint32_t)(hs.coord_key & 0xFFFFFFFFu);
        uint32_t f = 0, ilv = 0, v = 0;
        HIPCHK(ctx, hipMemcpy(&f, C.first + c, 4, hipMemcpyDeviceToHost)); HIPCHK(ctx, hipMemcpy(&ilv, C.il + c, 4, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(&v, (axis ? R.y : R.x) + f + (size_t)i * (ilv ? 2 : 1), 4, hipMemcpyDeviceToHost));
        return rfq_fail(ctx, RFQ_E_DATA, "The X/Y coordinate cannot be larger than 2M, but we get: %u", v);
    }
    // (rare: the tail chunk's line-break bits need the normaliser's verdict on a blank line behind the records)
    if ((hs.err & DE_TAIL_BLANK) && !nm) return RFQ_NEED_NORM;
    if (fast && ((hs.err & (DE_SCRATCH_SMALL | DE_SCRATCHN_SMALL)) || ((hs.err & (1u << 31)) && !a->d_out))) {
        // an arena (or the context's own image buffer) sized in advance was too small: now that the sizes are known, make room and repeat the batch
        ovl_guard.sync();
        HIPCHK(ctx, B[B_SCRATCH].ensure((size_t)hs.total_scratch + 256)); HIPCHK(ctx, B[B_SCRATCHN].ensure((size_t)hs.total_scratch_n + 256));
        if (!a->d_out) HIPCHK(ctx, ctx->out_img.ensure((size_t)(hs.image_bound + hdr_bytes + 64)));
        ctx->retried_room = true;
        return RFQ_RETRY_ROOM;
    }
    if (hs.err & DE_QUAL_OVERFLOW) return rfq_fail(ctx, RFQ_E_UNPINNED, "quality payload exceeds the reference's 1.5x scratch buffer (reference heap overflow, SURVEY.md App. C Q6)");
    if (hs.err & DE_CORRUPT) return rfq_fail(ctx, RFQ_E_HIP, "internal: a stream exceeded its scratch capacity");
    if (hs.err & (1u << 31)) return rfq_fail(ctx, RFQ_E_NOSPACE, "output buffer too small: need %llu bytes", (unsigned long long)(hs.total_image + hdr_bytes));
    for (auto& o : ctx->chunk_off) o += hdr_bytes;
    res->d_rfq = img; res->rfq_len = (size_t)(hs.total_image + hdr_bytes); res->n_chunks = n_chunks; res->n_reads = reads_used; res->n_bases = total_bases;
    const size_t lim[2] = { nm ? nm->orig_n[0] : nbytes[0], nm ? nm->orig_n[1] : nbytes[1] };
    res->consumed1 = (cons[0] > lim[0] ? lim[0] : cons[0]) - (nm ? 0 : skip[0]);
    res->consumed2 = nstreams == 2 ? (cons[1] > lim[1] ? lim[1] : cons[1]) - (nm ? 0 : skip[1]) : 0;
    res->h_chunk_off = ctx->chunk_off.data();
    return RFQ_OK;
}
