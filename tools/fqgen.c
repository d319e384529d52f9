/*
 * fqgen — deterministic synthetic FASTQ generator (test + bench input tooling).
 *
 * Integer-only (splitmix64), so the GPU box regenerates byte-identical inputs from a
 * (profile, seed, n_reads) triple; nothing here is derived from the reference sources.
 * Profiles follow SURVEY.md §8(d) / BASELINE.json configs[]:
 *   0 NOVA_SE150   NovaSeq-shaped single-end, fixed length 150
 *   1 NOVA_PE150   NovaSeq-shaped paired-end 2x150, insert ~N(300,60) clipped to [75,700]
 *   2 SE_VAR       single-end, variable length 100..150 skewed to 150 (configs[0] stand-in)
 *   3 BGI_PE100    BGI-style PE100: long non-Illumina names, many quality values, N runs
 *
 * Build: gcc -O2 -shared -fPIC tools/fqgen.c -o tools/libfqgen.so   (ctypes)
 *        gcc -O2 -DFQGEN_MAIN tools/fqgen.c -o tools/fqgen           (CLI)
 */
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t seed;
    uint64_t n_reads;          /* reads (SE) or pairs (PE) */
    int32_t  profile;          /* 0..3, see above */
    uint32_t n_rate_ppm;       /* N bases per million bases (NovaSeq profiles) */
    int32_t  no_trailing_newline; /* bit0: stream 1, bit1: stream 2 */
    int32_t  interleaved;      /* PE: R1,R2 alternate in stream 1 */
    int32_t  n_quals;          /* BGI profile: distinct quality values (13..40) */
    int32_t  reserved;
} fqgen_params;

typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t* r) {
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint32_t rng_below(rng_t* r, uint32_t n) { /* n < 2^31, multiplicative range reduction */
    return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32);
}

typedef struct { uint8_t* p; size_t n, cap; int overflow; } sink_t;
static inline void put(sink_t* s, const void* d, size_t len) {
    if (s->n + len > s->cap) { s->overflow = 1; s->n += len; return; }
    if (s->p) memcpy(s->p + s->n, d, len);
    s->n += len;
}
static inline void putc1(sink_t* s, char c) { put(s, &c, 1); }
static size_t put_u(char* dst, uint64_t v) { /* decimal, returns length */
    char t[24]; int k = 0; do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (int i = 0; i < k; i++) dst[i] = t[k - 1 - i];
    return (size_t)k;
}

static const char BASES[4] = {'A', 'C', 'G', 'T'};
static inline char comp(char c) {
    switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; default: return 'N'; }
}
static void rand_bases(rng_t* r, char* dst, int n) {
    int i = 0;
    while (i < n) {
        uint64_t w = rng_next(r);
        for (int k = 0; k < 32 && i < n; k++, i++) { dst[i] = BASES[w & 3]; w >>= 2; }
    }
}
/* NovaSeq 4-bin qualities: F ~92 %, : ~5 %, , ~3 % (byte thresholds 235/248 of 256) */
static void nova_quals(rng_t* r, char* dst, int n) {
    int i = 0;
    while (i < n) {
        uint64_t w = rng_next(r);
        for (int k = 0; k < 8 && i < n; k++, i++) {
            unsigned b = (unsigned)(w & 0xFF); w >>= 8;
            dst[i] = b < 235 ? 'F' : (b < 248 ? ':' : ',');
        }
    }
}
/* each base independently N with probability ppm/1e6, approximated per read by up to 3 draws */
static void nova_ns(rng_t* r, char* seq, char* qual, int n, uint32_t ppm) {
    if (!ppm) return;
    /* expected N per read = n*ppm/1e6; threshold on a 32-bit draw */
    uint64_t thr = ((uint64_t)n * ppm * 4294967296ull) / 1000000ull;
    if (thr > 0xFFFFFFFFull) thr = 0xFFFFFFFFull;
    for (int k = 0; k < 3; k++) {
        uint32_t u = (uint32_t)(rng_next(r) >> 32);
        if (u >= thr) break;
        int pos = (int)rng_below(r, (uint32_t)n);
        seq[pos] = 'N'; qual[pos] = '#';
    }
}

typedef struct { uint32_t lane, tile, x, y; } coord_t;
static void coord_step(rng_t* r, coord_t* c) {
    c->x += rng_below(r, 41);                 /* x monotone, steps 0..40 */
    if (c->x > 32000) {
        c->x = 1000 + rng_below(r, 41);
        c->y += 1 + rng_below(r, 18);         /* y advances when x wraps */
        if (c->y > 36000) {
            c->y = 1000;
            c->tile += 1;
            if (c->tile % 100 > 78) c->tile += 22;   /* 1101..1178, 1201..1278, ... */
            if (c->tile > 2678) { c->tile = 1101; c->lane = c->lane % 4 + 1; }
        }
    }
}
static size_t nova_name(char* dst, const coord_t* c, int mate) {
    static const char P[] = "@A00250:26:H3YTWDSXX:";
    size_t k = sizeof(P) - 1; memcpy(dst, P, k);
    k += put_u(dst + k, c->lane); dst[k++] = ':';
    k += put_u(dst + k, c->tile); dst[k++] = ':';
    k += put_u(dst + k, c->x);    dst[k++] = ':';
    k += put_u(dst + k, c->y);
    dst[k++] = ' '; dst[k++] = (char)('0' + mate);
    static const char S[] = ":N:0:ACTGTTCC";
    memcpy(dst + k, S, sizeof(S) - 1); k += sizeof(S) - 1;
    return k;
}
static void emit_record(sink_t* s, const char* name, size_t nl, const char* seq, const char* qual, int len, int last_no_nl) {
    put(s, name, nl); putc1(s, '\n');
    put(s, seq, (size_t)len); putc1(s, '\n');
    putc1(s, '+'); putc1(s, '\n');
    put(s, qual, (size_t)len);
    if (!last_no_nl) putc1(s, '\n');
}

static int insert_size(rng_t* r) { /* ~N(300,60) via 12 uniforms, integer only */
    int64_t acc = 0;
    uint64_t a = rng_next(r), b = rng_next(r), c = rng_next(r);
    for (int k = 0; k < 4; k++) { acc += (int64_t)(a & 0xFFFF); a >>= 16; acc += (int64_t)(b & 0xFFFF); b >>= 16; acc += (int64_t)(c & 0xFFFF); c >>= 16; }
    acc -= 6 * 65535;                     /* mean 0, sd = 65536 */
    int64_t d = (acc * 60) / 65536;       /* C division truncates toward zero: deterministic */
    int64_t v = 300 + d;
    if (v < 75) v = 75;
    if (v > 700) v = 700;
    return (int)v;
}

static void gen_nova(const fqgen_params* p, sink_t* s1, sink_t* s2) {
    rng_t r = { p->seed * 0x9E3779B97F4A7C15ull + 0x1234567ull };
    coord_t c = { 1, 1101, 1000, 1000 };
    char name[128], seq1[160], q1[160], seq2[160], q2[160], frag[720];
    int pe = p->profile == 1;
    for (uint64_t i = 0; i < p->n_reads; i++) {
        coord_step(&r, &c);
        int last = (i + 1 == p->n_reads);
        int len = 150;
        if (p->profile == 2) {            /* variable length 100..150, ~75 % at 150 */
            uint32_t u = rng_below(&r, 100);
            len = u < 75 ? 150 : 100 + (int)rng_below(&r, 51);
        }
        if (!pe) {
            rand_bases(&r, seq1, len); nova_quals(&r, q1, len); nova_ns(&r, seq1, q1, len, p->n_rate_ppm);
            size_t nl = nova_name(name, &c, 1);
            emit_record(s1, name, nl, seq1, q1, len, last && (p->no_trailing_newline & 1));
        } else {
            int ins = insert_size(&r);
            rand_bases(&r, frag, ins);
            for (int k = 0; k < 150; k++) seq1[k] = k < ins ? frag[k] : BASES[rng_below(&r, 4)];
            for (int k = 0; k < 150; k++) seq2[k] = k < ins ? comp(frag[ins - 1 - k]) : BASES[rng_below(&r, 4)];
            nova_quals(&r, q1, 150); nova_quals(&r, q2, 150);
            nova_ns(&r, seq1, q1, 150, p->n_rate_ppm); nova_ns(&r, seq2, q2, 150, p->n_rate_ppm);
            size_t nl = nova_name(name, &c, 1);
            emit_record(s1, name, nl, seq1, q1, 150, 0 + ((p->interleaved == 0) && last && (p->no_trailing_newline & 1)));
            nl = nova_name(name, &c, 2);
            sink_t* sb = p->interleaved ? s1 : s2;
            int nonl = p->interleaved ? (last && (p->no_trailing_newline & 1)) : (last && (p->no_trailing_newline & 2));
            emit_record(sb, name, nl, seq2, q2, 150, nonl);
        }
    }
}

static void gen_bgi(const fqgen_params* p, sink_t* s1, sink_t* s2) {
    rng_t r = { p->seed * 0xD1B54A32D192ED03ull + 0x7654321ull };
    int nq = p->n_quals < 2 ? 13 : (p->n_quals > 40 ? 40 : p->n_quals);
    char name[260], seq[2][104], q[2][104];
    static const char DESC[] = " length=100 platform=BGISEQ-500 flowcell=V300012345 sample=NA12878_lib07_runA library=PCRfree "
                               "operator=auto comment=synthetic_worst_case_name_padding_for_raw_name_copy_path_0123456789abcdefghij";
    uint32_t lane = 1, col = 1, row = 1; uint64_t serial = 0;
    for (uint64_t i = 0; i < p->n_reads; i++) {
        int last = (i + 1 == p->n_reads);
        serial += 1 + rng_below(&r, 7);
        if (serial > 9999999) { serial = 1; row++; if (row > 999) { row = 1; col++; if (col > 999) { col = 1; lane = lane % 4 + 1; } } }
        size_t k = 0;
        memcpy(name, "@V300012345L", 12); k = 12;
        name[k++] = (char)('0' + lane); name[k++] = 'C';
        name[k++] = (char)('0' + col / 100); name[k++] = (char)('0' + col / 10 % 10); name[k++] = (char)('0' + col % 10);
        name[k++] = 'R';
        name[k++] = (char)('0' + row / 100); name[k++] = (char)('0' + row / 10 % 10); name[k++] = (char)('0' + row % 10);
        { char t[8]; uint64_t v = serial; for (int d = 6; d >= 0; d--) { t[d] = (char)('0' + v % 10); v /= 10; } memcpy(name + k, t, 7); k += 7; }
        name[k++] = '/';
        size_t mate_pos = k; name[k++] = '1';
        /* description padding: total name length 60..200 */
        size_t want = 60 + rng_below(&r, 141);
        size_t dl = want > k ? want - k : 0; if (dl > sizeof(DESC) - 1) dl = sizeof(DESC) - 1;
        memcpy(name + k, DESC, dl); k += dl;
        for (int m = 0; m < 2; m++) {
            rand_bases(&r, seq[m], 100);
            for (int b = 0; b < 100; b++) q[m][b] = (char)('%' + rng_below(&r, (uint32_t)nq)); /* uniform over nq values from '%' */
            uint32_t u = rng_below(&r, 100);
            if (u < 5) {                       /* 5 % of reads: an N run of 1..60 */
                int rl = 1 + (int)rng_below(&r, 60), st = (int)rng_below(&r, (uint32_t)(100 - rl + 1));
                for (int b = st; b < st + rl; b++) { seq[m][b] = 'N'; q[m][b] = (rng_next(&r) & 1) ? '!' : '#'; }
            } else if (u < 6) {                /* 1 %: scattered N */
                for (int t = 0; t < 3; t++) { int b = (int)rng_below(&r, 100); seq[m][b] = 'N'; q[m][b] = (rng_next(&r) & 1) ? '!' : '#'; }
            }
        }
        name[mate_pos] = '1';
        emit_record(s1, name, k, seq[0], q[0], 100, (p->interleaved == 0) && last && (p->no_trailing_newline & 1));
        name[mate_pos] = '2';
        sink_t* sb = p->interleaved ? s1 : s2;
        int nonl = p->interleaved ? (last && (p->no_trailing_newline & 1)) : (last && (p->no_trailing_newline & 2));
        emit_record(sb, name, k, seq[1], q[1], 100, nonl);
    }
}

/* Returns 0 on success, 1 if a capacity was too small (n1/n2 still report the required sizes).
 * Pass out1 == NULL (cap ignored) to size the output only. */
int fqgen_generate(const fqgen_params* p, uint8_t* out1, size_t cap1, uint8_t* out2, size_t cap2, size_t* n1, size_t* n2) {
    sink_t s1 = { out1, 0, out1 ? cap1 : (size_t)-1, 0 }, s2 = { out2, 0, out2 ? cap2 : (size_t)-1, 0 };
    if (p->profile == 3) gen_bgi(p, &s1, &s2); else gen_nova(p, &s1, &s2);
    if (n1) *n1 = s1.n;
    if (n2) *n2 = s2.n;
    return (s1.overflow || s2.overflow) ? 1 : 0;
}

#ifdef FQGEN_MAIN
static void usage(void) {
    fprintf(stderr, "usage: fqgen --profile {0,1,2,3} --reads N [--seed S] [--nppm R] [--nquals Q] [--nonl MASK] [--interleaved] -o out1 [-O out2]\n");
    exit(2);
}
int main(int argc, char** argv) {
    fqgen_params p; memset(&p, 0, sizeof p); p.seed = 1; p.n_rate_ppm = 20; p.n_quals = 13;
    const char *o1 = NULL, *o2 = NULL;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--profile") && i + 1 < argc) p.profile = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--reads") && i + 1 < argc) p.n_reads = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) p.seed = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--nppm") && i + 1 < argc) p.n_rate_ppm = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--nquals") && i + 1 < argc) p.n_quals = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--nonl") && i + 1 < argc) p.no_trailing_newline = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--interleaved")) p.interleaved = 1;
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) o1 = argv[++i];
        else if (!strcmp(argv[i], "-O") && i + 1 < argc) o2 = argv[++i];
        else usage();
    }
    if (!o1 || !p.n_reads) usage();
    size_t n1 = 0, n2 = 0;
    fqgen_generate(&p, NULL, 0, NULL, 0, &n1, &n2);
    uint8_t* b1 = (uint8_t*)malloc(n1 ? n1 : 1); uint8_t* b2 = (uint8_t*)malloc(n2 ? n2 : 1);
    if (fqgen_generate(&p, b1, n1, b2, n2, &n1, &n2)) { fprintf(stderr, "fqgen: internal size mismatch\n"); return 1; }
    FILE* f = !strcmp(o1, "-") ? stdout : fopen(o1, "wb"); if (!f) { perror(o1); return 1; }
    fwrite(b1, 1, n1, f); if (f != stdout) fclose(f);
    if (n2) { if (!o2) { fprintf(stderr, "fqgen: profile writes two streams, give -O\n"); return 1; } f = fopen(o2, "wb"); if (!f) { perror(o2); return 1; } fwrite(b2, 1, n2, f); fclose(f); }
    return 0;
}
#endif
