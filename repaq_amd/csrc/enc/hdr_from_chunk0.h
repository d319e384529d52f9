// enc/hdr_from_chunk0.h - the file header from chunk 0 (makeHeader, makeQualityTable)
// Part of rfq_encode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
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
    // one atomic of each kind per WORKGROUP (every wave sent its own: four thousand atomics in a row on each of five words - most of the kernel's 53 us, and the header of a
    // file's first batch is a chain the gather waits for)
    __shared__ uint32_t s_n[4], s_ok[4], s_mx[4]; __shared__ unsigned long long s_fn[4], s_fe[4];
    if (l == 0) { const int w = wave_id(); s_n[w] = ncnt; s_ok[w] = notok; s_mx[w] = mxl; s_fn[w] = fn; s_fe[w] = fe; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < wpb; w++) { ncnt += s_n[w]; notok |= s_ok[w]; if (s_mx[w] > mxl) mxl = s_mx[w]; if (s_fn[w] < fn) fn = s_fn[w]; if (s_fe[w] < fe) fe = s_fe[w]; }
        if (ncnt) atomicAdd(&H->n_count, ncnt);
        if (notok) atomicOr(&H->any_not_ok, 1u);
        if (mxl) atomicMax(&H->max_len, mxl);
        if (fn != ~0ull) atomicMin((unsigned long long*)&H->first_n_key, (unsigned long long)fn);
        if (fe != ~0ull) atomicMin((unsigned long long*)&H->first_err_key, (unsigned long long)fe);
    }
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
__host__ __device__ __forceinline__ void hdr_derive(DevHeader* D) {
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
