// dec/read_table.h - per-read table, chunk bases, 2-bit unpack of the expanded path
// Part of rfq_decode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
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
// ---- fused path (k_dec_emit3): CHUNK-LOCAL prefixes, made where the per-read values are made.
// Round 4 wrote the per-read inputs (pvin, len) to HBM and ran two batch-wide scans over them (3 launches each: reduce, partials, apply) - 1.0 ms of kernels and
// 2.0 GB of traffic per 8 GB of text for prefixes of which the emitter only ever uses differences INSIDE one chunk.  Here a workgroup owns a chunk
// and walks its reads 1024 at a time (four consecutive reads per thread, one block scan per step, the step's total carried): the prefixes restart at 0 in every
// chunk and every chunk gets one entry more than it has reads - its totals - so chunk c's entries sit at [rbase + c, rbase + c + reads].
#define E3_MIDROW 32u
struct DFused {
    const uint32_t* len; const int32_t* ov;
    const uint32_t* pql;        // [rbase + c + r]: qualities (= bases) of the chunk in front of read r; entry `reads` = the chunk's total
    const U4* pvl;              // ... (name1, name2, strand piece bytes; stored bases) in front of read r; null where every chunk shares its name pieces (the three sums are all zero there)
    const uint32_t* sdl;        // ... the stored bases alone (pvl[].d): what the emitter's SHARED instantiation reads - 4 bytes per read instead of 16
    const uint2* tpl;           // [rbase + r]: x = text bytes of the chunk in front of read r in ITS output (out1, or out2 for a split decode's mates), y = bytes of its name middle
    // mid: [g][E3_MIDROW]: the formatted ":lane:tile:x:y" middle of the name, zero-padded (":255:65535:4294967295:4294967295" is 32 bytes: the row; its length rides in tpl.y)
    const U4* tbase;            // [c]: text bytes in front of chunk c (a: out1, b: out2); entry n_chunks = the range's totals
    const uint8_t* mid;
};
__global__ void __launch_bounds__(256) k_dec_readtab2(const uint8_t* __restrict__ img, const DChunk* __restrict__ CH, const DevHeader* __restrict__ D,
                                                      uint32_t* __restrict__ len_o, int32_t* __restrict__ ov_o, uint32_t* __restrict__ pql, U4* __restrict__ pvl, uint32_t* __restrict__ sdl, DecStatus* st) {
    const uint32_t c = blockIdx.x; const DChunk d = CH[c];
    const uint8_t* cp = img + d.off; const uint32_t fl = d.flags, hf = D->flags, rlb = D->read_len_bytes, shift = (uint32_t)D->overlap_shift;
    const bool ovl = (fl & C_PE_INTERLEAVED) && (hf & H_PE_OVERLAP);
    const size_t fp = (size_t)d.rbase + c;
    U4 carry; carry.a = carry.b = carry.c = carry.d = 0; uint32_t qcarry = 0, bad = 0;
    for (uint32_t r0 = 0; r0 < d.reads; r0 += 4u * blockDim.x) {             // block-uniform
        const uint32_t rb = r0 + 4u * threadIdx.x;
        U4 v[4]; uint32_t ln[4]; U4 sum; sum.a = sum.b = sum.c = sum.d = 0; uint32_t qsum = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t r = rb + (uint32_t)i; v[i].a = v[i].b = v[i].c = v[i].d = 0; ln[i] = 0;
            if (r < d.reads) {
                const uint32_t len = dec_read_len(cp, d, rlb, r);
                v[i].a = (fl & C_NAME1_SAME) ? 0u : cp[d.o_n1lens + ((fl & C_NAME1_LEN_SAME) ? 0u : r)];
                v[i].b = ((hf & H_NAME2) && !(fl & C_NAME2_SAME)) ? cp[d.o_n2lens + ((fl & C_NAME2_LEN_SAME) ? 0u : r)] : 0u;
                v[i].c = (fl & C_STRAND_SAME) ? 0u : cp[d.o_stlens + ((fl & C_STRAND_LEN_SAME) ? 0u : r)];
                int ov = 0; uint32_t stored = len;
                if (ovl && (r & 1u)) {
                    ov = (int)(int8_t)cp[d.o_ov + r / 2] - (int)shift;
                    const uint32_t a = (uint32_t)(ov < 0 ? -ov : ov);
                    const uint32_t prevlen = dec_read_len(cp, d, rlb, r - 1);
                    if (a > len || a > prevlen) { bad = 1; ov = 0; } else stored = len - a;
                }
                v[i].d = stored; ln[i] = len;
                len_o[(size_t)d.rbase + r] = len; ov_o[(size_t)d.rbase + r] = ov;
            }
            sum = sum + v[i]; qsum += ln[i];
        }
        U4 tot; uint32_t qtot;
        U4 run = carry + block_excl_sum<U4>(sum, &tot); uint32_t qrun = qcarry + block_excl_sum<uint32_t>(qsum, &qtot);
#pragma unroll
        for (int i = 0; i < 4; i++) { const uint32_t r = rb + (uint32_t)i; if (r < d.reads) { if (pvl) pvl[fp + r] = run; sdl[fp + r] = run.d; pql[fp + r] = qrun; } run = run + v[i]; qrun += ln[i]; }
        carry = carry + tot; qcarry += qtot;
    }
    if (threadIdx.x == 0) { if (pvl) pvl[fp + d.reads] = carry; sdl[fp + d.reads] = carry.d; pql[fp + d.reads] = qcarry; }
    if (__any(bad != 0) && lane_id() == 0) atomicOr(&st->err, (uint32_t)DE_CORRUPT_OV);
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
