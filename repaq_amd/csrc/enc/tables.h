// enc/tables.h - the batch's text, per-read and per-chunk tables
// Part of rfq_encode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
#include "../rfq_common.h"

struct Text {                    // the FASTQ streams of a batch and their line tables
    const uint8_t* fq[2];
    uint32_t n[2];
    const uint32_t* lo[2];       // lo[s][i] = start of line i; lo[s][i+1]-1 = its terminator (virtual at n for an unterminated tail)
    const uint32_t* ot[2];       // normalised text only (else null): offset of line i's terminator in the caller's text
    int paired;                  // RFQ_SE / RFQ_PE_TWO_FILES / RFQ_PE_INTERLEAVED
    uint32_t n_reads;            // reads in interleaved order (PE: 2 * pairs)
    uint32_t upr;                // reads per partition unit (1 SE, 2 PE)
};
// element s of the two-entry arrays above for a stream index that is only known per lane: a select between two kernel arguments (indexing
// the argument struct dynamically makes the compiler fetch the pointer from memory - a dependent load in front of every access)
__device__ __forceinline__ const uint8_t* t_fq(const Text& T, int s) { return s ? T.fq[1] : T.fq[0]; }
__device__ __forceinline__ const uint32_t* t_lo(const Text& T, int s) { return s ? T.lo[1] : T.lo[0]; }
__device__ __forceinline__ const uint32_t* t_ot(const Text& T, int s) { return s ? T.ot[1] : T.ot[0]; }
__device__ __forceinline__ uint32_t t_n(const Text& T, int s) { return s ? T.n[1] : T.n[0]; }
__device__ __forceinline__ void read_loc(const Text& T, uint32_t g, int& s, uint32_t& r) {
    if (T.paired == 1) { s = (int)(g & 1u); r = g >> 1; } else { s = 0; r = g; }
}
__device__ __forceinline__ uint32_t line_beg(const Text& T, uint32_t g, int k) { int s; uint32_t r; read_loc(T, g, s, r); return t_lo(T, s)[4 * (size_t)r + k]; }
__device__ __forceinline__ uint32_t line_len(const Text& T, uint32_t g, int k) { int s; uint32_t r; read_loc(T, g, s, r);
        const uint32_t* p = t_lo(T, s) + 4 * (size_t)r + k; return p[1] - 1 - p[0]; }
__device__ __forceinline__ const uint8_t* line_ptr(const Text& T, uint32_t g, int k) { int s; uint32_t r; read_loc(T, g, s, r);
        return t_fq(T, s) + t_lo(T, s)[4 * (size_t)r + k]; }

struct ReadTab {                 // per-read arrays, indexed by g (interleaved order)
    uint32_t* len;               // sequence length
    uint32_t* name1_len;
    uint32_t* name2_off;         // name2 = name[name2_off, name_len)
    uint32_t* x; uint32_t* y;
    uint16_t* tile; uint8_t* lane; uint8_t* ok;
    uint32_t* chunk;             // chunk id
    uint32_t* stored;            // bases kept in the sequence stream (after overlap trimming)
    uint8_t*  eq2;               // name2 == name2 of the chunk's read 0
    uint32_t* pq;                // exclusive prefix of len          (n_reads + 1 entries)
    U4*       pv;                // exclusive prefix of (name1_len, name2_len, strand_len, stored): only differences inside one chunk are ever used (pv[g] - pv[first[c]];
                                 // the chunk's totals: ChunkTab::ptot) - the tile path restarts it at 0 in every chunk (k_chunk_prefix), the byte-wise path scans the
                                 // whole batch
    uint32_t* sd;                // tile path: the stored-base component of pv alone (chunk-local; k_seqpack reads 4 bytes per read instead of one word of every 16); pv itself
                                 // is written only for chunks that store a name piece per read (k_chunk_prefix)
};

struct ChunkTab {                // per-chunk arrays
    uint32_t* first;             // first read of chunk c; first[n_chunks] = end
    uint32_t* flags;             // RfqChunk::mFlags (without line-break bits)
    uint32_t* il;                // final canBePeInterleaved
    uint32_t* ncount;            // 'N' bases in the stored sequence
    uint32_t* nmap;              // [c][NMAP_WORDS] bit b set: the chunk's stored bases contain an 'N' in 4096-base steps [b << shift, (b+1) << shift)
    uint32_t* scap;              // [c][MAX_STREAMS] scratch capacity of each stream
    uint64_t* soff;              // [c][MAX_STREAMS] scratch offset of each stream
    uint32_t* ssize;             // [c][MAX_STREAMS] bytes written by the stream coder
    uint32_t* xsize; uint32_t* ysize;
    uint64_t* qbase; uint64_t* sbase;   // 64-byte aligned bases of the chunk in qcat / scat
    uint64_t* img_size;          // bytes of the chunk image
    uint64_t* img_off;           // exclusive prefix (n_chunks + 1)
    U4*       ptot;              // the chunk's totals of (name1_len, name2_len, strand_len, stored): ReadTab::pv[g] - pv[first] is read g's offset inside the chunk
};
// an arena the host sized before the counts existed turned out too small (k_enc_totals): whoever would touch it leaves, the host repeats the batch with room
__device__ __forceinline__ bool enc_arena_small(const DevStatus* st) { return (st->err & (DE_SCRATCH_SMALL | DE_SCRATCHN_SMALL)) != 0u; }
