// rfq_common.h — shared definitions for the gfx950 RFQ engine: on-disk flag bits, the device-side header/chunk
// descriptors, wave64 primitives and a small multi-block scan.  Wave size is 64 everywhere (CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// ---- on-disk bits (reference src/rfqheader.h:24-42, src/rfqchunk.h:25-50) ----
#define H_LANE        (1u << 0)
#define H_TILE        (1u << 1)
#define H_X           (1u << 2)
#define H_Y           (1u << 3)
#define H_NAME2       (1u << 4)
#define H_PAIRED      (1u << 5)
#define H_PE_OVERLAP  (1u << 6)
#define H_QUAL_BY_COL (1u << 7)
#define H_DONT_QUAL   (1u << 8)
#define H_N_POS       (1u << 9)
#define C_READ_LEN_SAME   (1u << 0)
#define C_NAME1_LEN_SAME  (1u << 1)
#define C_NAME2_LEN_SAME  (1u << 2)
#define C_STRAND_LEN_SAME (1u << 3)
#define C_LANE_SAME       (1u << 4)
#define C_TILE_SAME       (1u << 5)
#define C_NAME1_SAME      (1u << 6)
#define C_NAME2_SAME      (1u << 7)
#define C_STRAND_SAME     (1u << 8)
#define C_PE_INTERLEAVED  (1u << 9)
#define C_NO_LB           (1u << 10)
#define C_NO_LB_R2        (1u << 11)

#define WAVE 64
#define NL_BLOCK_BYTES 16384u        // one workgroup of the newline pass covers 16 KiB = 256 lanes x 64 B
#define MAX_STREAMS 66               // stream slots of a chunk: [0,64) normal quality values (by-col coding implies <= 64 bins),
#define NPOS_SLOT 64                 //   slot 64 = N positions, slot 65 = exception records
#define EXC_SLOT 65

// device error bits accumulated in DevStatus::err
#define DE_HAS_CR          (1u << 0)  // '\r' in the text
#define DE_EMPTY_LINE      (1u << 1)  // an empty line inside the record range (reference truncates there)
#define DE_QUAL_SHORT      (1u << 2)  // quality line shorter than sequence line (reference reads past the string)
#define DE_BAD_QUAL        (1u << 3)  // quality byte >= 128 in chunk 0 ("bad quality value")
#define DE_BAD_BASE        (1u << 4)  // non-ACGTN base in chunk 0
#define DE_COORD_RANGE     (1u << 5)  // X/Y >= 2^21
#define DE_QUAL_OVERFLOW   (1u << 6)  // quality payload > 1.5 x bases: overflows the reference's scratch (UB)
#define DE_NO_QUAL_BINS    (1u << 7)  // "bad quality string"
#define DE_CORRUPT         (1u << 8)  // decode: inconsistent chunk image
#define DE_E3_RETRY        (1u << 29) // decode, k_dec_emit3: a tile's per-read name pieces do not fit its LDS tiles: the range is emitted again by the expanded path (k_dec_emit)
#define DE_INDEX_RETRY     (1u << 30) // encode, one-pass line index: more lines than the table sized in advance holds (or a wait that did not end): index in two passes
#define DE_CORRUPT_OV      (1u << 10) // decode, fused path: an overlap byte that exceeds a mate's length (k_dec_readtab2; "corrupt overlap buffer" - DE_CORRUPT is the quality / stream verdict there)
#define DE_INTERNAL        (1u << 11) // encode: two kernels disagree about an invariant they share (k_partition: reads of one length whose units are not) - a bug, reported as one
#define DE_SCRATCH_SMALL   (1u << 12) // encode: the stream scratch the host sized in advance (no read-back between gather and coder) is too small: the coders and the assembler leave, the host grows it and repeats the batch
#define DE_SCRATCHN_SMALL  (1u << 13) // ... the N-position streams' arena
#define DE_NEED_SCAN       (1u << 14) // encode: reads of several lengths - the closed-form prefixes do not apply, the host runs the scans and the partition again
#define DE_UNITS_GUESS     (1u << 15) // encode: the batch holds more units than the host sized its tables for without waiting for the index's totals: once more, with the totals
#define DE_TAIL_BLANK      (1u << 9)  // an empty line in the \n-only text right behind the encoded records: blank or empty is for the normaliser to say

// Device-resident file header + derived tables (RfqHeader, src/rfqheader.h:44-108)
struct DevHeader {
    uint8_t  bytes[17 + 255];   // on-disk image (RfqHeader::write)
    uint32_t len;               // 17 + qual_bins
    uint32_t flags;             // mFlags
    uint32_t read_len_bytes;
    uint32_t support_interleaved;
    uint32_t name2_diff_pos, name2_diff_char;
    uint32_t n_base_qual;       // 0..255 (0xFF == -1)
    int32_t  overlap_shift;     // -24
    uint32_t major;             // majorQual()
    uint32_t n_normal;          // normalQualBins()
    uint8_t  normal[256];       // normalQualBuf()
    uint8_t  stream_of[256];    // quality byte -> index into normal[] (0xFF = none)
    uint8_t  is_exception[256]; // 1: neither major nor a normal value -> 5-byte exception record
    // encode, match-mask mode: the coded values' streams by falling frequency in chunk 0 (k_dense_order): the first ones get planes built in LDS
    uint8_t  dense[4];
    uint32_t dense_valid;
    uint32_t valid;
};

// One record of device status shared by all kernels of a batch (zeroed per call)
struct DevStatus {
    uint32_t err;               // DE_* bits
    uint32_t err_read;          // read index (interleaved order) tied to the first data error, or value for coords
    uint64_t err_key;           // ordering key for "first error"
    uint64_t coord_key;         // (chunk << 34 | axis << 33 | index) of the first out-of-range X/Y, ~0 if none
    uint64_t total_scratch;     // bytes of stream scratch needed (quality value + exception streams)
    uint64_t total_scratch_n;   // ... by the N-position streams (their own arena: planned later, behind the sequence packer)
    uint64_t image_bound;       // upper bound of all chunk images (from stream capacities)
    uint32_t n_chunks;
    uint32_t max_chunk_reads;
    uint32_t max_chunk_bases;
    uint32_t n_units_used;      // units (reads or pairs) covered by emitted chunks
    uint64_t total_image;       // bytes of all chunk images
    uint64_t total_bases;
    uint32_t first_empty;       // first read (interleaved order) with an empty line, ~0 if none
    uint32_t max_rec;           // longest record (four lines with their terminators) in bytes
    uint32_t unit_bases;        // bases of every cut unit when they are all the same, else 0
    uint32_t max_len;           // longest read (bases)
    // k_index_totals (no read-back behind the line index): lines of each stream (an unterminated last line of a final batch included), the units they hold, and
    // the units the batch's kernels work on - min(true, what the host sized its tables for)
    uint32_t idx_lines[2], idx_units_true, idx_units;
};

struct U4 { uint32_t a, b, c, d; };
__device__ __host__ __forceinline__ U4 operator+(const U4& x, const U4& y) { U4 r; r.a = x.a + y.a; r.b = x.b + y.b; r.c = x.c + y.c; r.d = x.d + y.d; return r; }
__device__ __host__ __forceinline__ U4 operator-(const U4& x, const U4& y) { U4 r; r.a = x.a - y.a; r.b = x.b - y.b; r.c = x.c - y.c; r.d = x.d - y.d; return r; }

// the launch's dynamic LDS as an array `name` of `type` (the SIMT interpreter of the test build hands out a fixed 64 KB buffer)
#ifdef RFQ_SIMT_EMULATION
#define RFQ_DYN_SHARED(type, name) type* const name = (type*)emu::dyn_shared()
#else
#define RFQ_DYN_SHARED(type, name) extern __shared__ type name[]
#endif

// ---------------------------------------------------------------- wave64 primitives
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// CONTRACT of every wave_* / quad_* / lane_* primitive below: all 64 lanes of the wave call it together (full EXEC mask: call it outside lane-divergent branches, or
// under a condition that is the same in every lane).  The shuffle forms of the interpreter build read an idle lane's last value; a DPP move reads NOTHING from a lane
// that is switched off (the destination keeps `old`), so a call under divergent EXEC gives different answers on the GPU and under the interpreter - the interpreter
// aborts on a divergent rendezvous, the GPU does not.  tests/test_wave_primitives.py sweeps every primitive lane by lane on the GPU.
// Wave scans and reductions run on DPP (row_shr:1/2/4/8 inside the rows of 16 lanes, row_bcast:15 / row_bcast:31 across them): six VALU
// instructions with a DPP source operand per 32-bit scan.  The __shfl forms they replace went through ds_bpermute_b32 - address VALU + LDS crossbar
// + s_waitcnt per step - in kernels that are VALU-issue-bound (1,505 bpermute sites, no DPP).  The SIMT interpreter of the test build
// (RFQ_SIMT_EMULATION) has no DPP and keeps the shuffle forms; tests/test_wave_primitives.py sweeps both against a serial reference on the GPU.
#ifdef RFQ_SIMT_EMULATION
template <class T> __device__ __forceinline__ T wave_incl_sum(T v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) { T t = __shfl_up(v, (unsigned)d); if (l >= d) v = t + v; }
    return v;
}
template <class T> __device__ __forceinline__ T wave_incl_max(T v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) { T t = __shfl_up(v, (unsigned)d); if (l >= d && t > v) v = t; }
    return v;
}
template <class T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
template <class T> __device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { T t = __shfl_xor(v, d); if (t < v) v = t; }
    return v;
}
template <class T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { T t = __shfl_xor(v, d); if (t > v) v = t; }
    return v;
}
__device__ __forceinline__ uint32_t wave_and(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v &= __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v |= __shfl_xor(v, d);
    return v;
}
// lane l <- lane l-1 (lane 0 <- fill); the value of lane 63 in every lane
template <class T> __device__ __forceinline__ T wave_shr1(T v, T fill) { const T t = __shfl_up(v, 1u); return lane_id() ? t : fill; }
template <class T> __device__ __forceinline__ T wave_last(T v) { return __shfl(v, 63); }
// lane K of my quad (four consecutive lanes); the lane four in front of mine (callers only use it where that lane lies in the same row of 16)
template <class T> __device__ __forceinline__ T wave_read(T v, uint32_t lane) { return __shfl(v, (int)lane); }      // lane: the same in every lane of the wave
template <int K> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) { return (uint32_t)__shfl((int)v, (lane_id() & ~3) | K); }
__device__ __forceinline__ uint32_t lane_shr4(uint32_t v) { return (uint32_t)__shfl_up((int)v, 4u); }
#else
// v of the lane the DPP control names; lanes it names none for (or that row_mask leaves out) get `old`
template <int CTRL, int ROWS, class T> __device__ __forceinline__ T dpp_take(T old, T v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "dpp_take: 32- or 64-bit values");
    if constexpr (sizeof(T) == 4) {
        const int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, false);
        return __builtin_bit_cast(T, r);
    } else {
        const unsigned long long o = __builtin_bit_cast(unsigned long long, old), x = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)x, CTRL, ROWS, 0xF, false);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(x >> 32), CTRL, ROWS, 0xF, false);
        return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
    }
}
template <class T> __device__ __forceinline__ T wave_read63(T v) {
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    else { const unsigned long long x = __builtin_bit_cast(unsigned long long, v);
           const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), 63);
           return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo); }
}
// inclusive scan with an associative op whose identity is `id` (for an idempotent op - min, max, and, or - pass the value itself: op(v, v) = v)
// (the DPP move is taken into a temporary first: an OP that names its operand twice would otherwise repeat the move - and the compiler may put the
// second copy inside a divergent select, where the lanes it reads from are switched off)
#define RFQ_DPP_STEP(v, OP, ID, CTRL, ROWS) { const auto t_ = dpp_take<CTRL, ROWS>(ID, v); v = OP(v, t_); }
#define RFQ_DPP_SCAN(v, OP, ID)                                                                          \
    { RFQ_DPP_STEP(v, OP, ID, 0x111, 0xF) RFQ_DPP_STEP(v, OP, ID, 0x112, 0xF) RFQ_DPP_STEP(v, OP, ID, 0x114, 0xF) RFQ_DPP_STEP(v, OP, ID, 0x118, 0xF) \
      RFQ_DPP_STEP(v, OP, ID, 0x142, 0xA) RFQ_DPP_STEP(v, OP, ID, 0x143, 0xC) }
#define RFQ_OP_ADD(a, b) ((a) + (b))
#define RFQ_OP_MAX(a, b) ((b) > (a) ? (b) : (a))
#define RFQ_OP_MIN(a, b) ((b) < (a) ? (b) : (a))
#define RFQ_OP_AND(a, b) ((a) & (b))
#define RFQ_OP_OR(a, b) ((a) | (b))
template <class T> __device__ __forceinline__ T wave_incl_sum(T v) { RFQ_DPP_SCAN(v, RFQ_OP_ADD, T()) return v; }
template <class T> __device__ __forceinline__ T wave_incl_max(T v) { RFQ_DPP_SCAN(v, RFQ_OP_MAX, v) return v; }
template <class T> __device__ __forceinline__ T wave_sum(T v) { RFQ_DPP_SCAN(v, RFQ_OP_ADD, T()) return wave_read63(v); }
template <class T> __device__ __forceinline__ T wave_max(T v) { RFQ_DPP_SCAN(v, RFQ_OP_MAX, v) return wave_read63(v); }
template <class T> __device__ __forceinline__ T wave_min(T v) { RFQ_DPP_SCAN(v, RFQ_OP_MIN, v) return wave_read63(v); }
__device__ __forceinline__ uint32_t wave_and(uint32_t v) { RFQ_DPP_SCAN(v, RFQ_OP_AND, v) return wave_read63(v); }
__device__ __forceinline__ uint32_t wave_or(uint32_t v) { RFQ_DPP_SCAN(v, RFQ_OP_OR, v) return wave_read63(v); }
// lane l <- lane l-1 (lane 0 <- fill): wave_shr:1; the value of lane 63 in every lane (an SGPR)
template <class T> __device__ __forceinline__ T wave_shr1(T v, T fill) { return dpp_take<0x138, 0xF>(fill, v); }
template <class T> __device__ __forceinline__ T wave_last(T v) { return wave_read63(v); }
template <class T> __device__ __forceinline__ T wave_read(T v, uint32_t lane) { static_assert(sizeof(T) == 4, "wave_read: 32-bit values");
        return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), __builtin_amdgcn_readfirstlane((int)lane))); }
// quad_perm:[K,K,K,K]
template <int K> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K | (K << 2) | (K << 4) | (K << 6),
        0xF, 0xF, true); }
__device__ __forceinline__ uint32_t lane_shr4(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true); }             // row_shr:4
#endif
__device__ __forceinline__ U4 wave_incl_sum(U4 v) {
    v.a = wave_incl_sum(v.a); v.b = wave_incl_sum(v.b); v.c = wave_incl_sum(v.c); v.d = wave_incl_sum(v.d);
    return v;
}

// Exclusive prefix sum over a workgroup of NT threads (NT multiple of 64, <= 1024); returns the exclusive prefix of
// this thread and writes the workgroup total to *total.  Every thread of the workgroup must call it.
template <class T> __device__ __forceinline__ T block_excl_sum(T v, T* total) {
    __shared__ T wsum[16];
    __shared__ T wtot;
    const int l = lane_id(), w = wave_id(), nw = (int)(blockDim.x >> 6);
    T inc = wave_incl_sum(v);
    if (l == 63) wsum[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { T run = T(); for (int i = 0; i < nw; i++) { T t = wsum[i]; wsum[i] = run; run = run + t; } wtot = run; }
    __syncthreads();
    T res = (wsum[w] + inc) - v;
    if (total) *total = wtot;
    __syncthreads();
    return res;
}


// little-endian stores to unaligned byte addresses
__device__ __forceinline__ void st_u16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
__device__ __forceinline__ void st_u32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
__device__ __forceinline__ uint32_t ld_u16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// complement rule of Read::changeToReverseComplement (src/read.cpp:77-115): either case maps to the upper-case complement, anything else to N
__device__ __forceinline__ uint8_t comp_base(uint8_t b) {
    switch (b) { case 'A': case 'a': return 'T'; case 'T': case 't': return 'A'; case 'C': case 'c': return 'G'; case 'G': case 'g': return 'C'; default: return 'N'; }
}

// 16 / 8 / 4 consecutive bytes starting at an arbitrary LDS byte address, as little-endian words: ONE ds_read_b128 / _b64 / _b32 at a byte-granular address (gfx950 LDS runs in
// unaligned access mode; the compiler emits the wide read for an align-1 type).  What that costs was measured in round 5 (tools/micro/lds_align.hip, profiles/r05_lds_align.txt): the
// LDS serves an access that is not naturally aligned ONE LANE PER CLOCK - 64 clocks of the CU's LDS pipeline per wave instruction, whatever its width (16 bytes per lane at
// record-strided byte offsets: 15.8 B/clk/CU; the same rows 16-aligned: 81; unaligned 16-byte WRITES: 10).  Aligned dwords + a funnel shift per word (lds_funnel; round 2's form of
// these helpers) are 2.5 - 3.5 times faster on the LDS side and cost two to nine more instructions per read: they pay where a kernel waits for the LDS (k_overlap's window words), not
// where it is bound by its instructions (k_gather2: +0.4 ms with them) or by its stores (k_dec_emit3: no change) - those keep the one-instruction form.
struct __attribute__((packed, aligned(1))) LdsU16 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(1))) LdsU4 { uint32_t a; };
__device__ __forceinline__ void lds_get16(const uint8_t* base, uint32_t a, uint32_t (&w)[4]) {
    const LdsU16 v = *(const LdsU16*)(base + a); w[0] = v.a; w[1] = v.b; w[2] = v.c; w[3] = v.d;
}
__device__ __forceinline__ uint32_t lds_get4(const uint8_t* base, uint32_t a) { return ((const LdsU4*)(base + a))->a; }
struct __attribute__((packed, aligned(1))) LdsU8 { unsigned long long a; };
__device__ __forceinline__ unsigned long long lds_get8(const uint8_t* base, uint32_t a) { return ((const LdsU8*)(base + a))->a; }
__device__ __forceinline__ uint32_t lds_funnel(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((unsigned long long)hi << 32) | lo) >> sh); }   // sh < 32: v_alignbit_b32
// a wave-uniform value the compiler cannot prove uniform (it was loaded through a vector address, or derives from the wave's index): into an SGPR
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return ((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v); }
template <class P> __device__ __forceinline__ P* uniptr(P* p) { return (P*)(uintptr_t)uni64((uint64_t)(uintptr_t)p); }
#ifdef RFQ_SIMT_EMULATION
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
#else
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }      // both factors < 2^24
#endif
// v_dot4_u32_u8: sum of the four byte products + c, one full-rate instruction.  With flag bytes (0x80 or 0) or 2-bit codes in the bytes of `a` and powers of two in the bytes
// of `b` it gathers a bit (field) of every byte into adjacent bits - what took a shift-or cascade (8 instructions per 8 flags) or a v_mul_lo_u32 (a quarter of the rate) before.
#ifdef RFQ_SIMT_EMULATION
__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xFFu) * ((b >> (8 * i)) & 0xFFu); return c; }
#else
__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
#endif
__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24); }
// byte-wise full mask (0xFF per byte of w equal to the pattern byte)
__device__ __forceinline__ uint32_t eq_bytes_full(uint32_t w, uint32_t pat) {
    const uint32_t v = w ^ pat; const uint32_t t = ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);   // 0x80 per equal byte
    return (t >> 7) * 0xFFu;
}
// complement of four packed bases, comp_base() semantics (either case -> upper-case complement, anything else -> 'N').
// Case-folded, bits 1-2 of a base are a perfect hash (A 0, C 1, T 2, G 3): v_perm_b32 looks the four bytes up in a 4-entry
// complement table in ONE instruction, a second look-up in the identity table tells which bytes really were A/C/G/T.
__device__ __forceinline__ uint32_t comp4(uint32_t w) {
    const uint32_t u = w & 0xDFDFDFDFu;                                   // fold case: u == 'A' exactly for 'A' and 'a' (bit 5 is the only one dropped)
    const uint32_t idx = (u >> 1) & 0x03030303u;
    const uint32_t c = __builtin_amdgcn_perm(0u, 0x43414754u, idx);      // [A,C,T,G] -> T,G,A,C
    const uint32_t o = __builtin_amdgcn_perm(0u, 0x47544341u, idx);      // [A,C,T,G] -> A,C,T,G
    const uint32_t ok = eq_bytes_full(o, u);
    return (c & ok) | (0x4E4E4E4Eu & ~ok);
}

// LDS hand-off between lanes of ONE wave (rows private to the wave): order the wave's LDS writes before its later LDS reads.
// (DS operations of a wave execute in order; this pins the compiler's ordering and is free at run time.)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
