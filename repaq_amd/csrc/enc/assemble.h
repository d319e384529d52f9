// enc/assemble.h - chunk layout (incl. the mSize bug) and image assembly
// Part of rfq_encode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
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

// dst[0..n) = src[0..n) by the threads t, t+NT, ... : bytes up to dst's 16-byte boundary, then ALIGNED 16-byte stores fed by 16-byte loads at whatever phase src sits
// (the hardware takes an unaligned dwordx4 load; a store it splits), four of them in flight per thread, then the tail bytes.  Nothing outside [src, src + n) is read.
// (Round 6: dword stores fed by aligned dword loads and a funnel shift before - four times the instructions for the same bytes, the assembler ran at 4.1 TB/s.)
struct __attribute__((packed, aligned(1))) AsmU16 { uint32_t a, b, c, d; };
__device__ __forceinline__ void copy_to_image(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n, uint32_t t, uint32_t NT) {
    uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u); if (head > n) head = n;
    if (t < head) dst[t] = src[t];
    const uint32_t body = (n - head) / 16u;
    const uint8_t* sp = src + head; uint4* dw = (uint4*)(dst + head);
    for (uint32_t k0 = t; k0 < body; k0 += 4 * NT) {
        AsmU16 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t k = k0 + (uint32_t)u * NT; v[u].a = v[u].b = v[u].c = v[u].d = 0; if (k < body) v[u] = *(const AsmU16*)(sp + 16u * (size_t)k); }
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t k = k0 + (uint32_t)u * NT; if (k < body) dw[k] = make_uint4(v[u].a, v[u].b, v[u].c, v[u].d); }
    }
    const uint32_t done = head + 16u * body;
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
    if (enc_arena_small(st)) return;                                       // (the streams were not coded: the host repeats the batch)
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
// cap: bytes the arena of `which` holds (the host sizes it BEFORE the counts exist - no read-back between the gather and the coders; ~0 = exact sizing, the byte-wise path):
// a total beyond it raises DE_SCRATCH(N)_SMALL, at which every kernel that would touch the arena leaves (enc_arena_small) and the host repeats the batch with room
__global__ void k_enc_totals(ChunkTab C, const uint64_t* __restrict__ ctotal_prefix, uint32_t n_chunks, int which, DevStatus* st, uint64_t cap) {
    if (threadIdx.x || blockIdx.x) return;
    if (which == 0) { st->total_scratch = ctotal_prefix[n_chunks]; if (ctotal_prefix[n_chunks] + 256ull > cap) atomicOr(&st->err, (uint32_t)DE_SCRATCH_SMALL); }   // ctotal_prefix: the quality arena's
    else if (which == 2) { st->total_scratch_n = ctotal_prefix[n_chunks]; st->image_bound = C.img_off[n_chunks];                                     // ... the N arena's
            if (ctotal_prefix[n_chunks] + 256ull > cap) atomicOr(&st->err, (uint32_t)DE_SCRATCHN_SMALL); }
    else st->total_image = C.img_off[n_chunks];
}
