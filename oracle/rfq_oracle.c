/*
 * rfq_oracle.c — CPU restatement (plain C) of the repaq v0.5.1 RfqCodec path.
 * TEST INFRASTRUCTURE ONLY — see rfq_oracle.h.  Single-threaded, straightforward, and
 * deliberately bug-compatible; each block cites the reference lines.
 */
#include "rfq_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define FQ_BLOCK ((size_t)1 << 20)            /* src/fastqreader.cpp:5 */

/* ------------------------------------------------------------------ byte buffer */
typedef struct { uint8_t* p; size_t n, cap; } bb_t;
static void bb_reserve(bb_t* b, size_t extra) {
    if (b->n + extra <= b->cap) return;
    size_t nc = b->cap ? b->cap * 2 : 4096;
    while (nc < b->n + extra) nc *= 2;
    b->p = (uint8_t*)realloc(b->p, nc); b->cap = nc;
}
static void bb_put(bb_t* b, const void* d, size_t n) { bb_reserve(b, n); if (n) memcpy(b->p + b->n, d, n); b->n += n; }
static void bb_u8(bb_t* b, uint8_t v) { bb_put(b, &v, 1); }
static void bb_u16(bb_t* b, uint16_t v) { uint8_t t[2] = { (uint8_t)v, (uint8_t)(v >> 8) }; bb_put(b, t, 2); }
static void bb_u32(bb_t* b, uint32_t v) { uint8_t t[4] = { (uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24) }; bb_put(b, t, 4); }
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
void rfqo_free(void* p) { free(p); }

/* ------------------------------------------------------------------ name parse */
/* glibc atoi == (int)strtol(s, NULL, 10): leading isspace, sign, digits, saturating at LONG range. */
static int atoi_span(const uint8_t* s, uint32_t n) {
    uint32_t i = 0;
    while (i < n && (s[i] == ' ' || (s[i] >= '\t' && s[i] <= '\r'))) i++;
    int neg = 0;
    if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
    unsigned long long acc = 0; int sat = 0;
    const unsigned long long lim = neg ? (unsigned long long)LONG_MAX + 1ull : (unsigned long long)LONG_MAX;
    for (; i < n && s[i] >= '0' && s[i] <= '9'; i++) {
        unsigned d = (unsigned)(s[i] - '0');
        if (sat || acc > (lim - d) / 10) { sat = 1; acc = lim; } else acc = acc * 10 + d;
    }
    long v = neg ? (long)(0ull - acc) : (long)acc;
    return (int)v;
}

/* src/fastqmeta.cpp:22-80 */
void rfqo_parse_name(const uint8_t* str, uint32_t len, rfqo_meta* m) {
    int colon = 0; int last_colon = 0, cstart = 0, cend = 0;
    uint8_t lane = 0; uint16_t tile = 0; uint32_t x = 0, y = 0;
    for (uint32_t i = 0; i < len; i++) {
        uint8_t c = str[i];
        if (c == ':') colon++;
        if (c == ':' || c == ' ') {
            if (colon >= 4 && colon <= 7) {
                int val = atoi_span(str + last_colon + 1, i - (uint32_t)last_colon - 1);
                switch (colon) {
                    case 4: lane = (uint8_t)val; cstart = last_colon + 1; break;
                    case 5: tile = (uint16_t)val; break;
                    case 6: if (c == ':') x = (uint32_t)val; break;
                    case 7: y = (uint32_t)val; break;
                }
                if (c == ' ' && colon == 6) y = (uint32_t)val;
            }
        }
        if (c == ':') last_colon = (int)i;
        if (c == ' ' || (c == ':' && colon == 7)) { cend = (int)i; break; }
    }
    memset(m, 0, sizeof *m);
    if (cstart > 0 && cend > 0) {
        m->ok = 1; m->lane = lane; m->tile = tile; m->x = x; m->y = y;
        m->name1_len = (uint32_t)(cstart - 1);
        m->name2_off = (uint32_t)cend; m->name2_len = len - (uint32_t)cend;
    } else {
        m->name1_len = len; m->name2_off = len; m->name2_len = 0;
    }
}

/* ------------------------------------------------------------------ small coders */
/* src/rfqcodec.cpp:1391-1438 */
int rfqo_overlap(const uint8_t* r1, int len1, const uint8_t* r2, int len2) {
    const int minlen = len1 < len2 ? len1 : len2;
    for (int o = 12; o <= minlen; o++) {
        if (memcmp(r1 + len1 - o, r2, (size_t)o) == 0) return o;
    }
    for (int o = 12; o <= minlen; o++) {
        if (memcmp(r2 + len2 - o, r1, (size_t)o) == 0) return -o;
    }
    return 0;
}

/* src/rfqcodec.cpp:1262-1330 */
int64_t rfqo_encode_coords(const uint32_t* data, uint32_t num, uint8_t* buf) {
    uint32_t last = 1000; uint8_t repeat = 0; int64_t n = 0;
    for (uint32_t i = 0; i < num; i++) {
        uint32_t val = data[i];
        if (repeat > 0 && (val != last || repeat == 32)) { buf[n++] = (uint8_t)((repeat - 1) | 0xC0); repeat = 0; }
        if (val == last) { repeat++; continue; }
        int diff = (int)(val - last);
        last = val;
        if (diff > 0 && diff <= 64) { buf[n++] = (uint8_t)((diff - 1) | 0x80); continue; }
        if (val <= 32767) { buf[n++] = (uint8_t)(val >> 8); buf[n++] = (uint8_t)(val & 0xFF); }
        else if (val < (1u << 21)) { buf[n++] = (uint8_t)((val >> 16) | 0xE0); buf[n++] = (uint8_t)((val >> 8) & 0xFF); buf[n++] = (uint8_t)(val & 0xFF); }
        else return -1 - (int64_t)i; /* error_exit("The X/Y coordinate cannot be larger than 2M...") */
    }
    if (repeat > 0) buf[n++] = (uint8_t)((repeat - 1) | 0xC0);
    return n;
}

/* src/rfqcodec.cpp:1332-1389 (bounds added: the reference trusts num) */
void rfqo_decode_coords(const uint8_t* buf, uint32_t len, uint32_t* data, uint32_t num) {
    uint32_t last = 1000, consumed = 0, decoded = 0;
    while (consumed < len) {
        uint32_t b0 = buf[consumed++];
        if ((b0 & 0x80) == 0) {
            uint32_t b1 = consumed < len ? buf[consumed] : 0; consumed++;
            uint32_t v = (b0 << 8) | b1; if (decoded < num) data[decoded] = v; decoded++; last = v;
        } else if ((b0 & 0x40) == 0) {
            uint32_t v = last + (b0 & 0x3F) + 1; if (decoded < num) data[decoded] = v; decoded++; last = v;
        } else if ((b0 & 0x20) == 0) {
            uint32_t rep = (b0 & 0x1F) + 1;
            for (uint32_t i = 0; i < rep; i++) { if (decoded < num) data[decoded] = last; decoded++; }
        } else {
            uint32_t b1 = consumed < len ? buf[consumed] : 0; consumed++;
            uint32_t b2 = consumed < len ? buf[consumed] : 0; consumed++;
            uint32_t v = ((b0 & 0x1F) << 16) | (b1 << 8) | b2; if (decoded < num) data[decoded] = v; decoded++; last = v;
        }
    }
}

/* src/rfqcodec.cpp:625-710.  mask may be NULL. */
uint32_t rfqo_pos_encode(const uint8_t* qual, uint32_t len, uint8_t q, uint8_t* enc, uint8_t* mask) {
    uint32_t n = 0; int64_t last = -1; int64_t cur = 0;
    while (cur < (int64_t)len) {
        while (qual[cur] != q) { cur++; if (cur >= (int64_t)len) return n; }
        if (mask) mask[cur] = 1;
        if (cur - last == 1 && cur > 1) {
            uint32_t run = 1;
            while (!(cur + run == (int64_t)len || run >= 32)) { if (qual[cur + run] == q) run++; else break; }
            if (mask) memset(mask + cur, 1, run);
            enc[n++] = (uint8_t)((run - 1) | 0xC0);
            cur += run; last = cur - 1;
            continue;
        }
        int64_t d = cur - last;
        if (d <= 128) { enc[n++] = (uint8_t)(d - 1); }
        else if (d <= (1 << 14)) { uint32_t v = (uint32_t)(d - 1); enc[n++] = (uint8_t)((v >> 8) | 0x80); enc[n++] = (uint8_t)(v & 0xFF); }
        else { uint32_t v = (uint32_t)(d - 1); enc[n++] = (uint8_t)((v >> 24) | 0xE0); enc[n++] = (uint8_t)(v >> 16); enc[n++] = (uint8_t)(v >> 8); enc[n++] = (uint8_t)v; }
        last = cur; cur++;
    }
    return n;
}

/* src/rfqcodec.cpp:957-1007 (writes bounded by out_len; the reference writes unchecked) */
void rfqo_pos_decode(const uint8_t* buf, uint32_t blen, uint8_t q, uint8_t* out, uint32_t out_len) {
    uint32_t consumed = 0; int64_t last = -1;
#define RFQO_SET(p) do { int64_t _p = (p); if (_p >= 0 && _p < (int64_t)out_len) out[_p] = q; } while (0)
    while (consumed < blen) {
        uint8_t b0 = buf[consumed];
        if ((b0 & 0x80) == 0) { int64_t d = (int64_t)b0 + 1; RFQO_SET(last + d); consumed += 1; last += d; }
        else if ((b0 & 0x40) == 0) {
            uint32_t b1 = consumed + 1 < blen ? buf[consumed + 1] : 0;
            int64_t d = (int64_t)((((uint32_t)b0 & 0x3F) << 8) | b1) + 1; RFQO_SET(last + d); consumed += 2; last += d;
        } else if ((b0 & 0x20) == 0) {
            int run = (b0 & 0x1F) + 1;
            for (int i = 1; i <= run; i++) RFQO_SET(last + i);
            consumed += 1; last += run;
        } else {
            uint32_t b1 = consumed + 1 < blen ? buf[consumed + 1] : 0, b2 = consumed + 2 < blen ? buf[consumed + 2] : 0, b3 = consumed + 3 < blen ? buf[consumed + 3] : 0;
            /* int arithmetic as in the reference: ((b0&0x1F)<<8|b1)<<8|b2)<<8|b3, then +1 */
            int32_t d32 = (int32_t)((((uint32_t)b0 & 0x1F) << 24) | (b1 << 16) | (b2 << 8) | b3);
            int64_t d = (int64_t)d32 + 1; RFQO_SET(last + d); consumed += 4; last += d;
        }
    }
#undef RFQO_SET
}

/* src/read.cpp:77-115 — reverses quality too; non-ACGT (either case) -> N */
void rfqo_revcomp(uint8_t* seq, uint8_t* qual, int len) {
    for (int i = 0; i < len / 2; i++) {
        uint8_t t = qual[i]; qual[i] = qual[len - 1 - i]; qual[len - 1 - i] = t;
        t = seq[i]; seq[i] = seq[len - 1 - i]; seq[len - 1 - i] = t;
    }
    for (int i = 0; i < len; i++) {
        switch (seq[i]) {
            case 'A': case 'a': seq[i] = 'T'; break;
            case 'T': case 't': seq[i] = 'A'; break;
            case 'C': case 'c': seq[i] = 'G'; break;
            case 'G': case 'g': seq[i] = 'C'; break;
            default: seq[i] = 'N';
        }
    }
}

/* ------------------------------------------------------------------ header */
static void header_init(rfqo_header* h) {            /* src/rfqheader.cpp:7-17 */
    memset(h, 0, sizeof *h);
    memcpy(h->version, "0.5.1", 5);
    h->algo = 2; h->read_len_bytes = 1; h->n_base_qual = '#'; h->overlap_shift = (uint8_t)(-24);
}
size_t rfqo_header_write(const rfqo_header* h, uint8_t* out) {   /* src/rfqheader.cpp:84-97 */
    size_t k = 0;
    out[k++] = 'R'; out[k++] = 'F'; out[k++] = 'Q';
    memcpy(out + k, h->version, 5); k += 5;
    out[k++] = h->algo; out[k++] = h->read_len_bytes;
    out[k++] = (uint8_t)h->flags; out[k++] = (uint8_t)(h->flags >> 8);
    out[k++] = h->name2_diff_pos; out[k++] = h->name2_diff_char; out[k++] = h->n_base_qual; out[k++] = h->overlap_shift;
    out[k++] = h->qual_bins;
    memcpy(out + k, h->qual_buf, h->qual_bins); k += h->qual_bins;
    return k;
}
int rfqo_header_read(const uint8_t* in, size_t n, rfqo_header* h, size_t* used, char* err) {   /* src/rfqheader.cpp:19-43 */
    memset(h, 0, sizeof *h);
    if (n < 17) { snprintf(err, 256, "Not a valid repaq file!"); return -1; }
    memcpy(h->version, in + 3, 5); h->algo = in[8];
    if (h->algo != 2) {
        snprintf(err, 256, "The data is encoded by different version of repaq, please try repaq v%.5s. \nSee: https://github.com/OpenGene/repaq/releases", (const char*)h->version);
        return -1;
    }
    h->read_len_bytes = in[9]; h->flags = rd16(in + 10);
    h->name2_diff_pos = in[12]; h->name2_diff_char = in[13]; h->n_base_qual = in[14]; h->overlap_shift = in[15];
    h->qual_bins = in[16];
    if (n < 17u + h->qual_bins) { snprintf(err, 256, "Not a valid repaq file!"); return -1; }
    memcpy(h->qual_buf, in + 17, h->qual_bins);
    if (in[0] != 'R' || in[1] != 'F' || in[2] != 'Q') { snprintf(err, 256, "Not a valid repaq file!"); return -1; }
    /* mSupportInterleaved is not stored; BIT_ENCODE_PE_BY_OVERLAP is set iff it was true (src/rfqcodec.cpp:117-122) */
    h->support_interleaved = (h->flags & RFQO_H_PE_OVERLAP) != 0;
    *used = 17u + h->qual_bins;
    return 0;
}
static uint8_t hdr_major(const rfqo_header* h) { return h->qual_buf[0]; }          /* mBit2QualTable[0], src/rfqheader.cpp:103-115,263 */
/* src/rfqheader.cpp:308-328; char-vs-uint8 comparison semantics preserved: 0xFF never "equals" -1 */
static int hdr_normal(const rfqo_header* h, uint8_t* out) {
    int8_t nq = (int8_t)h->n_base_qual; int8_t mq = (int8_t)hdr_major(h);
    int bins = (mq == nq) ? h->qual_bins : h->qual_bins - 1;
    int count = 0;
    for (int i = 0; i < h->qual_bins; i++) {
        int v = h->qual_buf[i];                       /* uint8 promoted */
        if (v != (int)mq || v == (int)nq) { if (count < bins) out[count] = (uint8_t)v; count++; if (count > bins) break; }
    }
    return bins;
}

typedef struct { const uint8_t *name, *seq, *strand, *qual; uint32_t name_len, seq_len, strand_len, qual_len; } rec_t;

/* src/rfqheader.cpp:130-237.  reads in encounter order (PE: R1,R2,R1,R2...). */
static int make_quality_table(rfqo_header* h, const rec_t* reads, size_t s, char* err) {
    int table[128]; memset(table, 0, sizeof table);
    int ncount = 0; int8_t nbq = (int8_t)h->n_base_qual;
    for (size_t r = 0; r < s; r++) {
        const rec_t* rd = &reads[r];
        for (uint32_t i = 0; i < rd->seq_len; i++) {
            int8_t q = (int8_t)rd->qual[i];
            if (q < 0) { snprintf(err, 256, "bad quality value: %d", (int)q); return -1; }
            table[(int)q]++;
            uint8_t base = rd->seq[i];
            if (base == 'N') {
                if (ncount == 0) nbq = q;
                else if (nbq != q) { h->flags |= RFQO_H_N_POS; nbq = -1; }
                ncount++;
            }
            if (base != 'A' && base != 'T' && base != 'C' && base != 'G' && base != 'N') {
                if (base == 'a' || base == 't' || base == 'c' || base == 't')
                    snprintf(err, 256, "repaq doesn't support FASTQ with lowercase bases (a/t/c/g)\nbut we get:\n%.*s", (int)(rd->seq_len > 150 ? 150 : rd->seq_len), (const char*)rd->seq);
                else
                    snprintf(err, 256, "repaq only supports FASTQ with uppercase bases (A/T/C/G/N)\nbut we get:\n%.*s", (int)(rd->seq_len > 150 ? 150 : rd->seq_len), (const char*)rd->seq);
                return -1;
            }
            if (q == nbq && ncount > 0 && base != 'N') { h->flags |= RFQO_H_N_POS; nbq = -1; }
        }
    }
    if (ncount < 100) { h->flags |= RFQO_H_N_POS; nbq = -1; }
    int bins = 0, maxnum = 0; int major = 0; int has_n = 0;
    for (int i = 0; i < 128; i++) {
        if (table[i] > 0) { bins++; if (i == (int)nbq) has_n = 1; }
        if (table[i] > maxnum) { maxnum = table[i]; major = i; }
    }
    if (bins == 0) { snprintf(err, 256, "bad quality string, is this a valid FASTQ file?"); return -1; }
    else if (bins >= 64) h->flags |= RFQO_H_DONT_QUAL;
    if (!has_n) bins += 1;
    h->qual_bins = (uint8_t)bins;
    h->qual_buf[0] = (uint8_t)major;
    int cur = 1;
    for (int i = 0; i < 128; i++) { if (i == major) continue; if (table[i] > 0) h->qual_buf[cur++] = (uint8_t)i; }
    if (!has_n) h->qual_buf[bins - 1] = (uint8_t)nbq;
    if (bins <= 64) h->flags |= RFQO_H_QUAL_BY_COL;
    h->n_base_qual = (uint8_t)nbq;
    return 0;
}

static int name2_eq_replaced(const uint8_t* a, uint32_t alen, const uint8_t* b, uint32_t blen, uint8_t pos, uint8_t ch) {
    /* (a with a[pos]=ch, when ch != 0) == b.   pos >= alen: std::string operator[] past size — UB in the
     * reference; pos == alen only touches the terminator, so equality is unaffected: treat as no-op. */
    if (alen != blen) return 0;
    for (uint32_t i = 0; i < alen; i++) { uint8_t c = a[i]; if (ch != 0 && i == pos) c = ch; if (c != b[i]) return 0; }
    return 1;
}

/* src/rfqcodec.cpp:20-57 (SE) and :59-145 (PE; reads = R1,R2,R1,R2...) */
static int make_header(rfqo_header* h, const rec_t* reads, size_t s, int is_pe, char* err) {
    header_init(h);
    int has_ltxy = 1; uint32_t maxlen = 0;
    int support = 1; int dpos = 0; uint8_t dch = 0;
    for (size_t i = 0; i < s; i += (is_pe ? 2 : 1)) {
        rfqo_meta m1, m2;
        rfqo_parse_name(reads[i].name, reads[i].name_len, &m1);
        has_ltxy &= m1.ok; if (reads[i].seq_len > maxlen) maxlen = reads[i].seq_len;
        if (!is_pe) continue;
        rfqo_parse_name(reads[i + 1].name, reads[i + 1].name_len, &m2);
        has_ltxy &= m2.ok; if (reads[i + 1].seq_len > maxlen) maxlen = reads[i + 1].seq_len;
        if (!has_ltxy) support = 0;
        else if (support) {
            const uint8_t* n1 = reads[i].name + m1.name2_off; const uint8_t* n2 = reads[i + 1].name + m2.name2_off;
            if (i == 0) {
                if (m1.name2_len != m2.name2_len) support = 0;
                for (uint32_t p = 0; p < m1.name2_len; p++) {
                    /* reference indexes meta2.namePart2[p] even when shorter (reads the NUL / UB); guard */
                    uint8_t c2 = p < m2.name2_len ? n2[p] : 0;
                    if (n1[p] != c2) { dpos = (int)p; dch = c2; break; }
                }
            }
            if ((int)m1.name2_len < dpos) support = 0;
            else if (!name2_eq_replaced(n1, m1.name2_len, n2, m2.name2_len, (uint8_t)dpos, dch)) support = 0;
        }
    }
    if (is_pe && support) { h->support_interleaved = 1; h->name2_diff_pos = (uint8_t)dpos; h->name2_diff_char = dch; h->flags |= RFQO_H_PE_OVERLAP; }
    if (has_ltxy) h->flags |= RFQO_H_LANE | RFQO_H_TILE | RFQO_H_X | RFQO_H_Y | RFQO_H_NAME2;
    if (make_quality_table(h, reads, s, err)) return -1;
    if (is_pe) h->flags |= RFQO_H_PAIRED;
    h->read_len_bytes = maxlen > 255 ? 2 : 1;          /* second `if` is not `else if`: never 4 (src/rfqcodec.cpp:48-53) */
    return 0;
}

/* ------------------------------------------------------------------ chunk encode */
/* src/rfqcodec.cpp:163-586 + encodeSeqQual :588-623 + encodeQualByCol :712-765 + RfqChunk::write src/rfqchunk.cpp:230-311 */
static int encode_chunk(const rfqo_header* h, const rec_t* reads, size_t s_, int is_pe, uint16_t extra_flags, bb_t* out, char* err) {
    const uint32_t s = (uint32_t)s_;
    if (s == 0) return 0;
    rfqo_meta* meta = (rfqo_meta*)malloc(sizeof(rfqo_meta) * s);
    for (uint32_t i = 0; i < s; i++) rfqo_parse_name(reads[i].name, reads[i].name_len, &meta[i]);
    const rec_t* r0 = &reads[0]; const rfqo_meta* m0 = &meta[0];
    int readLenSame = 1, name1LenSame = 1, name2LenSame = 1, strandLenSame = 1, strandSame = 1, laneSame = 1, tileSame = 1, name1Same = 1, name2Same = 1;
    uint32_t totalReadLen = 0, totalName1 = 0, totalName2 = 0, totalStrand = 0;
    uint8_t* laneBuf = (uint8_t*)calloc(s, 1); uint16_t* tileBuf = (uint16_t*)calloc(s, 2);
    uint32_t* xBuf = (uint32_t*)calloc(s, 4); uint32_t* yBuf = (uint32_t*)calloc(s, 4);
    int canIl = is_pe && h->support_interleaved;
    const int encodeOverlap = canIl && (h->flags & RFQO_H_PE_OVERLAP);
    const uint8_t* lastName2 = NULL; uint32_t lastName2Len = 0; uint32_t lastX = 0, lastY = 0; uint16_t lastTile = 0; uint8_t lastLane = 0;
    const uint8_t* name20 = r0->name + m0->name2_off;
#define EQ(a, al, b, bl) ((al) == (bl) && memcmp((a), (b), (al)) == 0)
    for (uint32_t i = 0; i < s; i++) {
        const rec_t* r = &reads[i]; const rfqo_meta* m = &meta[i];
        const uint8_t* n2 = r->name + m->name2_off;
        readLenSame &= r0->seq_len == r->seq_len;
        name1LenSame &= m0->name1_len == m->name1_len;
        name2LenSame &= m0->name2_len == m->name2_len;
        strandLenSame &= r0->strand_len == r->strand_len;
        strandSame &= EQ(r0->strand, r0->strand_len, r->strand, r->strand_len);
        laneSame &= m0->lane == m->lane;
        tileSame &= m0->tile == m->tile;
        name1Same &= EQ(r0->name, m0->name1_len, r->name, m->name1_len);
        if (!canIl) name2Same &= EQ(name20, m0->name2_len, n2, m->name2_len);
        else if (i % 2 == 1) {
            if (!name2_eq_replaced(lastName2, lastName2Len, n2, m->name2_len, h->name2_diff_pos, h->name2_diff_char)) {
                canIl = 0; name2Same &= EQ(name20, m0->name2_len, n2, m->name2_len);
            }
        } else { lastName2 = n2; lastName2Len = m->name2_len; name2Same &= EQ(name20, m0->name2_len, n2, m->name2_len); }
        laneBuf[i] = m->lane; tileBuf[i] = m->tile; xBuf[i] = m->x; yBuf[i] = m->y;
        if (canIl) {
            if (i % 2 == 1) { canIl &= lastLane == m->lane; canIl &= lastTile == m->tile; canIl &= lastX == m->x; canIl &= lastY == m->y; }
            else { lastLane = m->lane; lastTile = m->tile; lastX = m->x; lastY = m->y; }
        }
        totalReadLen += r->seq_len; totalName1 += m->name1_len; totalName2 += m->name2_len; totalStrand += r->strand_len;
    }
    if (canIl) for (uint32_t p = 0; p < s / 2; p++) { laneBuf[p] = laneBuf[p * 2]; tileBuf[p] = tileBuf[p * 2]; xBuf[p] = xBuf[p * 2]; yBuf[p] = yBuf[p * 2]; }

    const uint32_t rlb = h->read_len_bytes;
    bb_t readLens = {0}, n1Lens = {0}, n2Lens = {0}, stLens = {0}, n1 = {0}, n2 = {0}, st = {0};
    uint8_t* seqO = (uint8_t*)calloc(totalReadLen ? totalReadLen : 1, 1); uint8_t* qualO = (uint8_t*)calloc(totalReadLen ? totalReadLen : 1, 1);
    int8_t* ovBuf = encodeOverlap ? (int8_t*)calloc(s / 2 + 1, 1) : NULL;
    uint32_t seqCopied = 0, qualCopied = 0;
    uint8_t* tmpSeq = NULL; uint8_t* tmpQual = NULL; uint32_t tmpCap = 0;
    for (uint32_t i = 0; i < s; i++) {
        const rec_t* r = &reads[i]; const rfqo_meta* m = &meta[i]; uint32_t rlen = r->seq_len;
        if (!readLenSame) { if (rlb == 1) bb_u8(&readLens, (uint8_t)rlen); else if (rlb == 2) bb_u16(&readLens, (uint16_t)rlen); else bb_u32(&readLens, rlen); }
        if (!name1Same) { bb_put(&n1, r->name, m->name1_len); if (!name1LenSame) bb_u8(&n1Lens, (uint8_t)m->name1_len); }
        if (!name2Same) { bb_put(&n2, r->name + m->name2_off, m->name2_len); if (!name2LenSame) bb_u8(&n2Lens, (uint8_t)m->name2_len); }
        if (!strandSame) { bb_put(&st, r->strand, r->strand_len); if (!strandLenSame) bb_u8(&stLens, (uint8_t)r->strand_len); }
        const uint8_t* sq = r->seq; const uint8_t* ql = r->qual; int ov = 0;
        if (canIl && i % 2 == 1) {
            if (rlen > tmpCap) { tmpCap = rlen * 2 + 16; tmpSeq = (uint8_t*)realloc(tmpSeq, tmpCap); tmpQual = (uint8_t*)realloc(tmpQual, tmpCap); }
            memcpy(tmpSeq, r->seq, rlen); memcpy(tmpQual, r->qual, rlen);
            rfqo_revcomp(tmpSeq, tmpQual, (int)rlen); sq = tmpSeq; ql = tmpQual;
            if (encodeOverlap) {
                /* reads[i-1] is an even read: never reverse-complemented */
                ov = rfqo_overlap(reads[i - 1].seq, (int)reads[i - 1].seq_len, sq, (int)rlen);
                int sh = (int8_t)h->overlap_shift;
                if (ov + sh > 127) ov = 0;
                if (ov + sh < -127) ov = 0;
                ovBuf[i / 2] = (int8_t)(ov + sh);
            }
        }
        if (ov == 0) { memcpy(seqO + seqCopied, sq, rlen); seqCopied += rlen; }
        else if (ov > 0) { memcpy(seqO + seqCopied, sq + ov, rlen - (uint32_t)ov); seqCopied += rlen - (uint32_t)ov; }
        else { memcpy(seqO + seqCopied, sq, rlen - (uint32_t)(-ov)); seqCopied += rlen - (uint32_t)(-ov); }
        memcpy(qualO + qualCopied, ql, rlen); qualCopied += rlen;
    }
    free(tmpSeq); free(tmpQual);

    /* 2-bit pack, src/rfqcodec.cpp:590-604 */
    uint32_t seqEncLen = (seqCopied + 3) / 4;
    uint8_t* seqEnc = (uint8_t*)calloc(seqEncLen ? seqEncLen : 1, 1);
    for (uint32_t i = 0; i < seqCopied; i++) {
        uint8_t v = 0;
        switch (seqO[i]) { case 'G': v = 0; break; case 'A': v = 1; break; case 'T': v = 2; break; case 'C': v = 3; break; default: break; }
        seqEnc[i >> 2] |= (uint8_t)(v << ((i & 3) * 2));
    }
    /* quality payload */
    bb_t qenc = {0};
    int rc = 0;
    if (h->flags & RFQO_H_DONT_QUAL) bb_put(&qenc, qualO, qualCopied);
    else if (h->flags & RFQO_H_QUAL_BY_COL) {
        uint8_t nv[256]; int bins = hdr_normal(h, nv);
        uint8_t* mask = (uint8_t*)calloc(qualCopied ? qualCopied : 1, 1);
        uint8_t* tmp = (uint8_t*)malloc((size_t)qualCopied * 4 + 16);
        size_t lens_at = qenc.n;
        for (int i = 0; i < bins; i++) bb_u32(&qenc, 0);
        for (int i = 0; i < bins; i++) {
            uint32_t l = rfqo_pos_encode(qualO, qualCopied, nv[i], tmp, mask);
            bb_put(&qenc, tmp, l);
            uint8_t* p = qenc.p + lens_at + 4 * (size_t)i; p[0] = (uint8_t)l; p[1] = (uint8_t)(l >> 8); p[2] = (uint8_t)(l >> 16); p[3] = (uint8_t)(l >> 24);
        }
        uint8_t mq = hdr_major(h);
        for (uint32_t i = 0; i < qualCopied; i++) if (!mask[i] && qualO[i] != mq) { bb_u8(&qenc, qualO[i]); bb_u32(&qenc, i); }
        free(mask); free(tmp);
        /* reference scratch is int(totalReadLen*1.5) bytes (src/rfqcodec.cpp:413): beyond that it overflows the heap (UB) */
        if (qenc.n > (size_t)((double)totalReadLen * 1.5)) { snprintf(err, 256, "quality payload exceeds the reference's 1.5x scratch buffer (reference UB, parity unpinned)"); rc = -1; }
    } else { snprintf(err, 256, "run-length quality coding is unreachable in repaq v0.5.1 encode"); rc = -1; }
    /* N positions over the trimmed sequence, src/rfqcodec.cpp:419-426 */
    bb_t npos = {0};
    if (!rc && (h->flags & RFQO_H_N_POS)) { bb_reserve(&npos, (size_t)seqCopied * 4 + 16); npos.n = rfqo_pos_encode(seqO, seqCopied, 'N', npos.p, NULL); }
    /* coordinates */
    bb_t xe = {0}, ye = {0}; uint32_t num = canIl ? s / 2 : s;
    if (!rc && (h->flags & RFQO_H_X)) { bb_reserve(&xe, (size_t)num * 3 + 4); int64_t l = rfqo_encode_coords(xBuf, num, xe.p); if (l < 0) { snprintf(err, 256, "The X/Y coordinate cannot be larger than 2M, but we get: %u", xBuf[-1 - l]); rc = -1; } else xe.n = (size_t)l; }
    if (!rc && (h->flags & RFQO_H_Y)) { bb_reserve(&ye, (size_t)num * 3 + 4); int64_t l = rfqo_encode_coords(yBuf, num, ye.p); if (l < 0) { snprintf(err, 256, "The X/Y coordinate cannot be larger than 2M, but we get: %u", yBuf[-1 - l]); rc = -1; } else ye.n = (size_t)l; }

    if (!rc) {
        uint16_t cf = 0;
        if (canIl) cf |= RFQO_C_PE_INTERLEAVED;
        if (readLenSame) cf |= RFQO_C_READ_LEN_SAME; if (name1LenSame) cf |= RFQO_C_NAME1_LEN_SAME; if (name2LenSame) cf |= RFQO_C_NAME2_LEN_SAME;
        if (strandLenSame) cf |= RFQO_C_STRAND_LEN_SAME; if (strandSame) cf |= RFQO_C_STRAND_SAME; if (laneSame) cf |= RFQO_C_LANE_SAME;
        if (tileSame) cf |= RFQO_C_TILE_SAME; if (name1Same) cf |= RFQO_C_NAME1_SAME; if (name2Same) cf |= RFQO_C_NAME2_SAME;
        uint32_t readLenBufSize = readLenSame ? rlb : rlb * s;
        uint32_t n1LenSize = name1LenSame ? 1 : s, n2LenSize = name2LenSame ? 1 : s, stLenSize = strandLenSame ? 1 : s;
        /* Q1: the tile size lands in mLaneBufSize, mTileBufSize stays 0 (src/rfqcodec.cpp:489-515) */
        uint32_t laneBufSizeBug = tileSame ? 2 : (canIl ? (2 * s) / 2 : 2 * s);
        uint32_t n1Size = name1Same ? m0->name1_len : totalName1, n2Size = name2Same ? m0->name2_len : totalName2, stSize = strandSame ? r0->strand_len : totalStrand;
        uint32_t msize = 4 + 4 + 2 + 4 + 4;
        msize += readLenBufSize + n1LenSize + n2LenSize + stLenSize;
        msize += laneBufSizeBug + 0 + n1Size + n2Size + stSize;
        msize += seqEncLen + (uint32_t)qenc.n;
        if (canIl && (h->flags & RFQO_H_PE_OVERLAP)) msize += s / 2;
        if (h->flags & RFQO_H_N_POS) msize += 4 + (uint32_t)npos.n;
        if (h->flags & RFQO_H_X) msize += 4 + (uint32_t)xe.n;
        if (h->flags & RFQO_H_Y) msize += 4 + (uint32_t)ye.n;
        /* RfqChunk::write, src/rfqchunk.cpp:230-311 */
        bb_u32(out, msize); bb_u32(out, s); bb_u16(out, (uint16_t)(cf | extra_flags)); bb_u32(out, seqEncLen); bb_u32(out, (uint32_t)qenc.n);
        if (h->flags & RFQO_H_N_POS) bb_u32(out, (uint32_t)npos.n);
        if (readLenSame) { uint32_t l0 = r0->seq_len; for (uint32_t b = 0; b < rlb; b++) bb_u8(out, (uint8_t)(l0 >> (8 * b))); } else bb_put(out, readLens.p, readLens.n);
        if (name1LenSame) bb_u8(out, (uint8_t)m0->name1_len); else bb_put(out, n1Lens.p, n1Lens.n);
        if (h->flags & RFQO_H_NAME2) { if (name2LenSame) bb_u8(out, (uint8_t)m0->name2_len); else bb_put(out, n2Lens.p, n2Lens.n); }
        if (strandLenSame) bb_u8(out, (uint8_t)r0->strand_len); else bb_put(out, stLens.p, stLens.n);
        uint32_t cnt = canIl ? s / 2 : s;
        if (h->flags & RFQO_H_LANE) { if (laneSame) bb_u8(out, m0->lane); else bb_put(out, laneBuf, cnt); }
        if (h->flags & RFQO_H_TILE) { if (tileSame) bb_u16(out, m0->tile); else for (uint32_t i = 0; i < cnt; i++) bb_u16(out, tileBuf[i]); }
        if (h->flags & RFQO_H_X) { bb_u32(out, (uint32_t)xe.n); bb_put(out, xe.p, xe.n); }
        if (h->flags & RFQO_H_Y) { bb_u32(out, (uint32_t)ye.n); bb_put(out, ye.p, ye.n); }
        if (name1Same) bb_put(out, r0->name, m0->name1_len); else bb_put(out, n1.p, n1.n);
        if (h->flags & RFQO_H_NAME2) { if (name2Same) bb_put(out, name20, m0->name2_len); else bb_put(out, n2.p, n2.n); }
        if (strandSame) bb_put(out, r0->strand, r0->strand_len); else bb_put(out, st.p, st.n);
        bb_put(out, seqEnc, seqEncLen);
        bb_put(out, qenc.p, qenc.n);
        if (canIl && (h->flags & RFQO_H_PE_OVERLAP)) bb_put(out, ovBuf, s / 2);
        if (h->flags & RFQO_H_N_POS) bb_put(out, npos.p, npos.n);
    }
#undef EQ
    free(meta); free(laneBuf); free(tileBuf); free(xBuf); free(yBuf); free(seqO); free(qualO); free(ovBuf); free(seqEnc);
    free(readLens.p); free(n1Lens.p); free(n2Lens.p); free(stLens.p); free(n1.p); free(n2.p); free(st.p); free(qenc.p); free(npos.p); free(xe.p); free(ye.p);
    return rc;
}

/* ------------------------------------------------------------------ FASTQ text reader */
/* src/fastqreader.cpp:31-46 (readToBuf), :94-156 (getLine), :166-196 (read) over an in-memory file */
typedef struct { const uint8_t* buf; size_t n; size_t bstart, bend, used; int eof_flag, no_lb; } reader_t;
static void reader_load(reader_t* r) {
    r->bstart = r->bend;
    size_t remain = r->n - r->bstart; size_t len = remain < FQ_BLOCK ? remain : FQ_BLOCK;
    r->bend = r->bstart + len; r->used = r->bstart;
    if (len < FQ_BLOCK) {
        r->eof_flag = 1;
        /* len == 0 (empty file, or size an exact multiple of 1 MiB): the reference reads mBuf[-1] — UB; with
         * glibc that byte is the top byte of the malloc size word (0), so the flag ends up set.  Parity unpinned. */
        if (len == 0 || r->buf[r->bend - 1] != '\n') r->no_lb = 1;
    }
}
static void reader_init(reader_t* r, const uint8_t* buf, size_t n) { memset(r, 0, sizeof *r); r->buf = buf; r->n = n; reader_load(r); }
static void reader_finish_line(reader_t* r, size_t end) {
    end++;                                              /* skip \n or \r */
    if (r->bend >= 1 && end < r->bend - 1 && r->buf[end] == '\n') end++;   /* "\r\n" (also swallows one blank line) */
    r->used = end;
}
static void reader_getline(reader_t* r, const uint8_t** p, size_t* len) {
    size_t start = r->used, end = start;
    while (end < r->bend && r->buf[end] != '\r' && r->buf[end] != '\n') end++;
    if (end < r->bend || (r->bend - r->bstart) < FQ_BLOCK) {
        *p = r->buf + start; *len = end >= start ? end - start : 0;
        reader_finish_line(r, end); return;
    }
    for (;;) {                                          /* line continues in the next 1 MiB block */
        reader_load(r);
        end = r->bstart;
        while (end < r->bend && r->buf[end] != '\r' && r->buf[end] != '\n') end++;
        if (end < r->bend || (r->bend - r->bstart) < FQ_BLOCK) {
            *p = r->buf + start; *len = end - start;
            reader_finish_line(r, end); return;
        }
    }
}
static int reader_read(reader_t* r, rec_t* o) {
    if (r->used >= r->bend && r->eof_flag) return 0;
    size_t l;
    reader_getline(r, &o->name, &l); o->name_len = (uint32_t)l;
    reader_getline(r, &o->seq, &l); o->seq_len = (uint32_t)l;
    reader_getline(r, &o->strand, &l); o->strand_len = (uint32_t)l;
    if (o->name_len == 0 || o->seq_len == 0 || o->strand_len == 0) return 0;
    reader_getline(r, &o->qual, &l); o->qual_len = (uint32_t)l;
    if (o->qual_len == 0) return 0;
    return 1;
}

/* ------------------------------------------------------------------ compress drivers */
typedef struct { rec_t* p; size_t n, cap; } recs_t;
static void recs_push(recs_t* v, const rec_t* r) { if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 16384; v->p = (rec_t*)realloc(v->p, v->cap * sizeof(rec_t)); } v->p[v->n++] = *r; }

/* Repaq::compress src/repaq.cpp:530-638 and compressPE :640-762 */
int rfqo_encode_file(const uint8_t* fq1, size_t n1, const uint8_t* fq2, size_t n2, int paired,
                     uint32_t chunk_bases, uint8_t** out, size_t* out_len, char* err) {
    reader_t ra, rb; reader_init(&ra, fq1, n1);
    if (paired == RFQO_PE_TWO_FILES) reader_init(&rb, fq2, n2);
    recs_t v = {0}; bb_t o = {0}; rfqo_header h; int have_h = 0; int rc = 0;
    uint32_t total = 0; err[0] = 0;
    const int is_pe = paired != RFQO_SE;
    for (;;) {
        rec_t a, b; int got = reader_read(&ra, &a);
        if (got && is_pe) got = reader_read(paired == RFQO_PE_INTERLEAVED ? &ra : &rb, &b);
        else if (!got && paired == RFQO_PE_TWO_FILES) { rec_t t; (void)reader_read(&rb, &t); } /* FastqReaderPair::read reads both sides (src/fastqreader.cpp:287-299) */
        int flush = 0;
        if (got) {
            if (a.qual_len < a.seq_len || (is_pe && b.qual_len < b.seq_len)) { snprintf(err, 256, "quality line shorter than sequence line (reference reads past the string: UB, parity unpinned)"); rc = -1; break; }
            recs_push(&v, &a); total += a.seq_len;
            if (is_pe) { recs_push(&v, &b); total += b.seq_len; }
            if (total >= chunk_bases) flush = 1;
        } else if (v.n > 0) flush = 1;
        if (flush) {
            if (!have_h) {
                if (make_header(&h, v.p, v.n, is_pe, err)) { rc = -1; break; }
                uint8_t hb[17 + 256]; size_t hl = rfqo_header_write(&h, hb); bb_put(&o, hb, hl); have_h = 1;
            }
            uint16_t ef = 0;
            if (ra.no_lb) ef |= RFQO_C_NO_LB;                                    /* src/repaq.cpp:571-572 */
            if (is_pe) { int nl2 = paired == RFQO_PE_TWO_FILES ? rb.no_lb : ra.no_lb; if (nl2) ef |= RFQO_C_NO_LB_R2; }   /* :683-692 */
            if (encode_chunk(&h, v.p, v.n, is_pe, ef, &o, err)) { rc = -1; break; }
            v.n = 0; total = 0;
        }
        if (!got) break;
    }
    free(v.p);
    if (rc) { free(o.p); *out = NULL; *out_len = 0; return rc; }
    *out = o.p; *out_len = o.n;
    return 0;
}

/* ------------------------------------------------------------------ chunk parse + decode */
typedef struct {
    uint32_t size_field, reads; uint16_t flags; uint32_t seq_size, qual_size, npos_size, x_size, y_size;
    const uint8_t *read_lens, *n1_lens, *n2_lens, *st_lens, *lanes, *tiles, *x, *y, *n1, *n2, *st, *seq, *qual, *ov, *npos;
    uint32_t n1_size, n2_size, st_size; size_t total;
} chunk_t;

/* RfqChunk::read, src/rfqchunk.cpp:161-228 (+ helpers :41-139).  Returns 0 ok, 1 clean EOF (mReads==0 / short), -1 error */
static int chunk_parse(const rfqo_header* h, const uint8_t* p, size_t n, chunk_t* c, char* err) {
    memset(c, 0, sizeof *c);
    size_t k = 0;
#define NEED(x) do { if (k + (size_t)(x) > n) { if (c->reads == 0 || k == 0) return 1; snprintf(err, 256, "truncated rfq chunk"); return -1; } } while (0)
    NEED(18);
    c->size_field = rd32(p); c->reads = rd32(p + 4); c->flags = rd16(p + 8); c->seq_size = rd32(p + 10); c->qual_size = rd32(p + 14); k = 18;
    if (c->reads == 0) return 1;
    if (h->flags & RFQO_H_N_POS) { NEED(4); c->npos_size = rd32(p + k); k += 4; }
    const uint32_t s = c->reads, rlb = h->read_len_bytes;
    if (rlb != 1 && rlb != 2 && rlb != 4) { snprintf(err, 256, "header incorrect: read length bytes should be 1/2/4"); return -1; }
    uint32_t cnt = (c->flags & RFQO_C_READ_LEN_SAME) ? 1 : s;
    NEED((size_t)cnt * rlb); c->read_lens = p + k; k += (size_t)cnt * rlb;
#define LENARR(ptr, sizevar, lenflag, sameflag) do { \
        uint32_t m_ = (c->flags & (lenflag)) ? 1 : s; NEED(m_); ptr = p + k; k += m_; \
        uint32_t sum_ = 0; for (uint32_t i_ = 0; i_ < m_; i_++) sum_ += ptr[i_]; \
        if ((c->flags & (lenflag)) && !(c->flags & (sameflag))) sum_ *= s; sizevar = sum_; } while (0)
    LENARR(c->n1_lens, c->n1_size, RFQO_C_NAME1_LEN_SAME, RFQO_C_NAME1_SAME);
    if (h->flags & RFQO_H_NAME2) LENARR(c->n2_lens, c->n2_size, RFQO_C_NAME2_LEN_SAME, RFQO_C_NAME2_SAME);
    LENARR(c->st_lens, c->st_size, RFQO_C_STRAND_LEN_SAME, RFQO_C_STRAND_SAME);
    uint32_t hcnt = (c->flags & RFQO_C_PE_INTERLEAVED) ? s / 2 : s;
    if (h->flags & RFQO_H_LANE) { uint32_t m = (c->flags & RFQO_C_LANE_SAME) ? 1 : hcnt; NEED(m); c->lanes = p + k; k += m; }
    if (h->flags & RFQO_H_TILE) { uint32_t m = (c->flags & RFQO_C_TILE_SAME) ? 1 : hcnt; NEED((size_t)m * 2); c->tiles = p + k; k += (size_t)m * 2; }
    if (h->flags & RFQO_H_X) { NEED(4); c->x_size = rd32(p + k); k += 4; NEED(c->x_size); c->x = p + k; k += c->x_size; }
    if (h->flags & RFQO_H_Y) { NEED(4); c->y_size = rd32(p + k); k += 4; NEED(c->y_size); c->y = p + k; k += c->y_size; }
    NEED(c->n1_size); c->n1 = p + k; k += c->n1_size;
    if (h->flags & RFQO_H_NAME2) { NEED(c->n2_size); c->n2 = p + k; k += c->n2_size; }
    NEED(c->st_size); c->st = p + k; k += c->st_size;
    NEED(c->seq_size); c->seq = p + k; k += c->seq_size;
    NEED(c->qual_size); c->qual = p + k; k += c->qual_size;
    if ((c->flags & RFQO_C_PE_INTERLEAVED) && (h->flags & RFQO_H_PE_OVERLAP)) { NEED(s / 2); c->ov = p + k; k += s / 2; }
    if (h->flags & RFQO_H_N_POS) { NEED(c->npos_size); c->npos = p + k; k += c->npos_size; }
    c->total = k;
#undef NEED
#undef LENARR
    return 0;
}

int64_t rfqo_chunk_table(const uint8_t* rfq, size_t n, uint64_t* offsets, size_t cap, char* err) {
    rfqo_header h; size_t used = 0; err[0] = 0;
    if (rfqo_header_read(rfq, n, &h, &used, err)) return -1;
    size_t k = used; int64_t nc = 0;
    for (;;) {
        chunk_t c; int r = chunk_parse(&h, rfq + k, n - k, &c, err);
        if (r < 0) return -1;
        if ((size_t)nc < cap) offsets[nc] = k;
        if (r == 1) break;
        k += c.total; nc++;
    }
    return nc;
}

static uint32_t read_len_at(const rfqo_header* h, const chunk_t* c, uint32_t i) {
    const uint8_t* p = c->read_lens + (size_t)i * h->read_len_bytes;
    return h->read_len_bytes == 1 ? p[0] : (h->read_len_bytes == 2 ? rd16(p) : rd32(p));
}
static size_t put_dec(uint8_t* dst, uint32_t v) { char t[12]; int k = 0; do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v); for (int i = 0; i < k; i++) dst[i] = (uint8_t)t[k - 1 - i]; return (size_t)k; }

/* RfqCodec::decodeChunk src/rfqcodec.cpp:1049-1260 + decodeSeqQual :826-917 + decodeQualByCol :1009-1047.
 * Appends FASTQ text of read r to outs[(split && r odd) ? 1 : 0]. */
static int decode_chunk(const rfqo_header* h, const chunk_t* c, int split, bb_t* outs, char* err) {
    const uint32_t s = c->reads;
    const int il = (c->flags & RFQO_C_PE_INTERLEAVED) != 0;
    const int encOv = il && (h->flags & RFQO_H_PE_OVERLAP);
    uint32_t* rl = (uint32_t*)malloc(sizeof(uint32_t) * (s ? s : 1)); uint64_t seqLen64 = 0;
    for (uint32_t i = 0; i < s; i++) { rl[i] = read_len_at(h, c, (c->flags & RFQO_C_READ_LEN_SAME) ? 0 : i); seqLen64 += rl[i]; }
    if (seqLen64 > 0xFFFFFFFFull) { free(rl); snprintf(err, 256, "chunk too large"); return -1; }
    const uint32_t seqLen = (uint32_t)seqLen64;
    uint8_t* seq = (uint8_t*)malloc(seqLen ? seqLen : 1); memset(seq, 'N', seqLen);
    uint8_t* qual = (uint8_t*)malloc(seqLen ? seqLen : 1); memset(qual, hdr_major(h), seqLen);
    int rc = 0;
    /* 2-bit unpack :833-853 */
    static const uint8_t B[4] = { 'G', 'A', 'T', 'C' };
    uint32_t dec = 0;
    for (uint32_t i = 0; i < c->seq_size && dec < seqLen; i++) for (int b = 0; b < 4 && dec < seqLen; b++) seq[dec++] = B[(c->seq[i] >> (b * 2)) & 3];
    if (h->flags & RFQO_H_N_POS) rfqo_pos_decode(c->npos, c->npos_size, 'N', seq, seqLen);
    if (encOv) {                                        /* :860-901 */
        uint8_t* dst = (uint8_t*)malloc(seqLen ? seqLen : 1); uint32_t sp = 0, dp = 0;
        for (uint32_t r = 0; r < s && !rc; r++) {
            uint32_t rlen = rl[r];
            if (r % 2 == 0) { memcpy(dst + dp, seq + sp, rlen); dp += rlen; sp += rlen; continue; }
            int ov = (int)(int8_t)c->ov[r / 2] - (int)(int8_t)h->overlap_shift;
            if (ov == 0) { memcpy(dst + dp, seq + sp, rlen); dp += rlen; sp += rlen; }
            else if (ov > 0) {
                if ((uint32_t)ov > rlen || (uint32_t)ov > sp) { rc = -1; break; }
                memcpy(dst + dp, seq + sp - ov, (size_t)ov); memcpy(dst + dp + ov, seq + sp, rlen - (uint32_t)ov); dp += rlen; sp += rlen - (uint32_t)ov;
            } else {
                uint32_t a = (uint32_t)(-ov), lastR = rl[r - 1];
                if (a > rlen || lastR > sp || a > lastR) { rc = -1; break; }
                memcpy(dst + dp, seq + sp, rlen - a); memcpy(dst + dp + rlen - a, seq + sp - lastR, a); dp += rlen; sp += rlen - a;
            }
        }
        if (rc) snprintf(err, 256, "corrupt overlap buffer");
        memcpy(seq, dst, seqLen); free(dst);
    }
    if (!rc) {
        if (h->flags & RFQO_H_DONT_QUAL) { for (uint32_t i = 0; i < c->qual_size && i < seqLen; i++) qual[i] = c->qual[i]; }
        else if (h->flags & RFQO_H_QUAL_BY_COL) {
            uint8_t nv[256]; int bins = hdr_normal(h, nv);
            size_t consumed = 4 * (size_t)bins;
            if (consumed > c->qual_size) { snprintf(err, 256, "corrupt quality buffer"); rc = -1; }
            for (int i = 0; i < bins && !rc; i++) {
                uint32_t l = rd32(c->qual + 4 * (size_t)i);
                if (consumed + l > c->qual_size) { snprintf(err, 256, "corrupt quality buffer"); rc = -1; break; }
                rfqo_pos_decode(c->qual + consumed, l, nv[i], qual, seqLen); consumed += l;
            }
            while (!rc && consumed + 5 <= c->qual_size) { uint8_t q = c->qual[consumed]; uint32_t pos = rd32(c->qual + consumed + 1); consumed += 5; if (pos < seqLen) qual[pos] = q; }
        } else if (c->qual_size) {
            /* decodeQualByRunLenCoding, src/rfqcodec.cpp:919-955 (legacy: v0.5.1 never writes it, App. C Q13).  One byte per run: bit 0 clear =
             * the major value, run = (byte >> 1) + 1 (majorQualNumBits is 7, src/rfqheader.cpp:255-257); bit 0 set = the value with "bit" code
             * byte & mask, run = (byte >> (8 - n)) + 1, n = normalQualNumBits (computeNormalQualBits, :117-128).  Code -> value is
             * mBit2QualTable (makeQualBitTable :103-115: entry i of the header's table has code 0, 1, 3, 5, ...; unlisted codes read the
             * zeroed table).  The outer while re-reads the buffer until `len` qualities are out. */
            int mx = (int)h->qual_bins * 2 - 3; if (mx < 1) mx = 1;
            int nq = mx >= 64 ? 1 : mx >= 32 ? 2 : mx >= 16 ? 3 : mx >= 8 ? 4 : mx >= 4 ? 5 : mx >= 2 ? 6 : 7;
            uint8_t bit2q[256]; memset(bit2q, 0, sizeof bit2q);
            for (int i = 0; i < (int)h->qual_bins; i++) bit2q[(uint8_t)(i ? 2 * i - 1 : 0)] = h->qual_buf[i];
            const uint8_t mask = (uint8_t)((1u << (8 - nq)) - 1u);
            uint32_t decoded = 0;
            while (decoded < seqLen) {
                for (uint32_t i = 0; i < c->qual_size; i++) {
                    const uint8_t e = c->qual[i]; uint8_t q; uint32_t num;
                    if ((e & 1) == 0) { q = 0; num = e >> 1; } else { q = e & mask; num = e >> (8 - nq); }
                    num += 1;
                    for (uint32_t f = decoded; f < decoded + num && f < seqLen; f++) qual[f] = bit2q[q];
                    decoded += num;
                    if (decoded >= seqLen) break;
                }
            }
        }
    }
    if (!rc && !(h->flags & RFQO_H_N_POS)) { uint8_t nq = h->n_base_qual; for (uint32_t i = 0; i < seqLen; i++) if (qual[i] == nq) seq[i] = 'N'; }   /* :1093-1100 */
    uint32_t xyNum = il ? s / 2 : s;
    uint32_t* xb = (uint32_t*)calloc(xyNum ? xyNum : 1, 4); uint32_t* yb = (uint32_t*)calloc(xyNum ? xyNum : 1, 4);
    if (!rc && (h->flags & RFQO_H_X)) rfqo_decode_coords(c->x, c->x_size, xb, xyNum);
    if (!rc && (h->flags & RFQO_H_Y)) rfqo_decode_coords(c->y, c->y_size, yb, xyNum);
    const uint8_t *cn1 = c->n1, *cn2 = c->n2, *cst = c->st; uint32_t cur = 0;
    uint8_t* tmp = NULL; uint32_t tmpCap = 0;
    for (uint32_t r = 0; r < s && !rc; r++) {
        bb_t* o = &outs[(split && (r & 1)) ? 1 : 0];
        uint32_t rlen = rl[r];
        /* name :1157-1231 */
        if (c->flags & RFQO_C_NAME1_SAME) bb_put(o, c->n1, c->n1_lens[0]);
        else { uint32_t l = (c->flags & RFQO_C_NAME1_LEN_SAME) ? c->n1_lens[0] : c->n1_lens[r]; bb_put(o, cn1, l); cn1 += l; }
        uint32_t xy = il ? r / 2 : r; uint8_t d[12];
        if (h->flags & RFQO_H_LANE) { bb_u8(o, ':'); bb_put(o, d, put_dec(d, (c->flags & RFQO_C_LANE_SAME) ? c->lanes[0] : c->lanes[xy])); }
        if (h->flags & RFQO_H_TILE) { bb_u8(o, ':'); bb_put(o, d, put_dec(d, rd16(c->tiles + 2 * (size_t)((c->flags & RFQO_C_TILE_SAME) ? 0 : xy)))); }
        if (h->flags & RFQO_H_X) { bb_u8(o, ':'); bb_put(o, d, put_dec(d, xb[xy])); }
        if (h->flags & RFQO_H_Y) { bb_u8(o, ':'); bb_put(o, d, put_dec(d, yb[xy])); }
        if (h->flags & RFQO_H_NAME2) {
            if (c->flags & RFQO_C_NAME2_SAME) {
                uint32_t l = c->n2_lens[0]; size_t at = o->n; bb_put(o, c->n2, l);
                if (il && (r & 1) && h->name2_diff_char != 0 && h->name2_diff_pos < l) o->p[at + h->name2_diff_pos] = h->name2_diff_char;
            } else { uint32_t l = (c->flags & RFQO_C_NAME2_LEN_SAME) ? c->n2_lens[0] : c->n2_lens[r]; bb_put(o, cn2, l); cn2 += l; }
        }
        bb_u8(o, '\n');
        const uint8_t* sq = seq + cur; const uint8_t* ql = qual + cur;
        if (il && (r & 1)) {                            /* :1248-1252 */
            if (rlen > tmpCap) { tmpCap = rlen * 2 + 16; tmp = (uint8_t*)realloc(tmp, (size_t)tmpCap * 2); }
            memcpy(tmp, sq, rlen); memcpy(tmp + tmpCap, ql, rlen); rfqo_revcomp(tmp, tmp + tmpCap, (int)rlen); sq = tmp; ql = tmp + tmpCap;
        }
        bb_put(o, sq, rlen); bb_u8(o, '\n');
        if (c->flags & RFQO_C_STRAND_SAME) bb_put(o, c->st, c->st_lens[0]);
        else { uint32_t l = (c->flags & RFQO_C_STRAND_LEN_SAME) ? c->st_lens[0] : c->st_lens[r]; bb_put(o, cst, l); cst += l; }
        bb_u8(o, '\n');
        bb_put(o, ql, rlen); bb_u8(o, '\n');
        cur += rlen;
    }
    free(tmp); free(rl); free(seq); free(qual); free(xb); free(yb);
    return rc;
}

/* Repaq::decompress src/repaq.cpp:262-333 / decompressPE :335-417.
 * The final '\n' of a stream is dropped when the LAST chunk carries its NO_LINE_BREAK bit.
 * Deliberate divergence: decompressPE's `continue` (src/repaq.cpp:389,400) loses data when a NON-last chunk carries
 * the bit (files < 1 MiB without trailing newline and > 1 chunk); this restatement keeps every read instead. */
/* Repaq::decompress / decompressPE AS WRITTEN (src/repaq.cpp:262-417), data loss included: a chunk that carries a NO_LINE_BREAK bit makes the loop read
 * the chunk BEHIND it to see whether it was the last one (:303-311, :376-387) - and when it was not, the loop goes on with `continue` (:322-325,
 * :389-392, :400-403).  decompressPE declares its chunk inside the loop (:347-349), so that is a fresh read: the chunk it peeked at is never decoded,
 * and it leaves the loop body before the R2 text of the flagged chunk is written when it is the R1 bit that is set (:389-392).  decompress (one
 * output) keeps the peeked chunk in a variable declared outside the loop (:281-286) and decodes it in the next iteration: nothing is lost there.
 * rfqo_decode_file (below) keeps every read instead; this function is what `--bug_compat` must reproduce, pinned against the reference binary in
 * tests/test_oracle_golden.py. */
int rfqo_decode_file_compat(const uint8_t* rfq, size_t n, int split_pe, uint8_t** out1, size_t* n1, uint8_t** out2, size_t* n2, char* err) {
    rfqo_header h; size_t used = 0; err[0] = 0;
    *out1 = NULL; *n1 = 0; if (out2) { *out2 = NULL; *n2 = 0; }
    if (n == 0) {
        if (split_pe) { snprintf(err, 256, "The input RFQ file was encoded by single-end FASTQ, you should not specify <out2>"); return -1; }
        return 0;
    }
    if (rfqo_header_read(rfq, n, &h, &used, err)) return -1;
    if (split_pe && !(h.flags & RFQO_H_PAIRED)) { snprintf(err, 256, "The input RFQ file was encoded by single-end FASTQ, you should not specify <out2>"); return -1; }
    bb_t outs[2] = { {0}, {0} }; size_t k = used;
    for (;;) {
        chunk_t c; int r = chunk_parse(&h, rfq + k, n - k, &c, err);
        if (r < 0) { free(outs[0].p); free(outs[1].p); return -1; }
        if (r == 1) break;
        bb_t t[2] = { {0}, {0} };
        if (decode_chunk(&h, &c, split_pe, t, err)) { free(t[0].p); free(t[1].p); free(outs[0].p); free(outs[1].p); return -1; }
        k += c.total;
        const int f1 = (c.flags & RFQO_C_NO_LB) != 0, f2 = split_pe && (c.flags & RFQO_C_NO_LB_R2) != 0;
        int last = 0;
        if (f1 || f2) {                                                     /* the peek: the next chunk is read - and, if there is one, lost */
            chunk_t nx; char e2[256]; const int r2 = chunk_parse(&h, rfq + k, n - k, &nx, e2);
            last = r2 != 0;
            if (!last && split_pe) k += nx.total;                           /* decompressPE reads afresh (:347-349); decompress keeps the peeked chunk and decodes it next (:282-286) */
        }
        int skip2 = 0;
        if (f1) { if (last) bb_put(&outs[0], t[0].p, t[0].n ? t[0].n - 1 : 0); else { bb_put(&outs[0], t[0].p, t[0].n); skip2 = split_pe; } }
        else bb_put(&outs[0], t[0].p, t[0].n);
        if (split_pe && !skip2) {
            if (f2 && last) bb_put(&outs[1], t[1].p, t[1].n ? t[1].n - 1 : 0); else bb_put(&outs[1], t[1].p, t[1].n);
        }
        free(t[0].p); free(t[1].p);
        if (!split_pe && f1 && last) break;                                 /* (:319-321) */
    }
    *out1 = outs[0].p; *n1 = outs[0].n;
    if (out2) { *out2 = outs[1].p; *n2 = outs[1].n; } else free(outs[1].p);
    return 0;
}
int rfqo_decode_file(const uint8_t* rfq, size_t n, int split_pe, uint8_t** out1, size_t* n1, uint8_t** out2, size_t* n2, char* err) {
    rfqo_header h; size_t used = 0; err[0] = 0;
    *out1 = NULL; *n1 = 0; if (out2) { *out2 = NULL; *n2 = 0; }
    if (n == 0) {   /* every istream::read fails and the constructor's defaults stand (valid magic, flags 0; src/rfqheader.cpp:7-17,19-43): no chunk follows */
        if (split_pe) { snprintf(err, 256, "The input RFQ file was encoded by single-end FASTQ, you should not specify <out2>"); return -1; }
        return 0;
    }
    if (rfqo_header_read(rfq, n, &h, &used, err)) return -1;
    if (split_pe && !(h.flags & RFQO_H_PAIRED)) { snprintf(err, 256, "The input RFQ file was encoded by single-end FASTQ, you should not specify <out2>"); return -1; }
    bb_t outs[2] = { {0}, {0} }; size_t k = used; uint16_t last_flags = 0; int any = 0;
    for (;;) {
        chunk_t c; int r = chunk_parse(&h, rfq + k, n - k, &c, err);
        if (r < 0) { free(outs[0].p); free(outs[1].p); return -1; }
        if (r == 1) break;
        if (decode_chunk(&h, &c, split_pe, outs, err)) { free(outs[0].p); free(outs[1].p); return -1; }
        last_flags = c.flags; any = 1; k += c.total;
    }
    if (any) {
        if ((last_flags & RFQO_C_NO_LB) && outs[0].n) outs[0].n--;
        if (split_pe && (last_flags & RFQO_C_NO_LB_R2) && outs[1].n) outs[1].n--;
    }
    *out1 = outs[0].p; *n1 = outs[0].n;
    if (out2) { *out2 = outs[1].p; *n2 = outs[1].n; } else free(outs[1].p);
    return 0;
}
