/*
 * rfq_hip.h — C-ABI of librfq_hip.so, the MI355X (gfx950) engine for repaq's RfqCodec path.
 *
 * The reference (OpenGene/repaq v0.5.1) has no FFI: its seam is the C++ class RfqCodec
 * (src/rfqcodec.h:17-43) driven by Repaq::compress / compressPE / decompress / decompressPE (src/repaq.cpp:262-762).  A GPU engine cannot take
 * vector<Read*>, so each entry point below replaces one RfqCodec/Repaq call at BATCH granularity over raw bytes:
 * FASTQ text in HBM -> .rfq chunk images in HBM and back, bit-identical to what the reference writes.
 *
 * Conventions
 *   - plain pointers + sizes, no C++/torch types; every d_* pointer is DEVICE memory (hipMalloc or a torch CUDA
 *     tensor's data_ptr()); h_* pointers are host memory.  Input pointers (FASTQ text, .rfq image) may have any alignment and
 *     any length: a stream of 4 GiB or more is worked through in slices inside the call, the result is the one image / the
 *     one text.  A FASTQ pointer that is not 16-byte aligned is rounded down inside the call: the (up to 15) bytes in front of it are READ
 *     (never interpreted), so they must belong to the same allocation - true of any pointer into a hipMalloc'ed buffer or a torch tensor.
 *     Caller-provided OUTPUT buffers must be 16-byte aligned.
 *   - return 0 (RFQ_OK) or a negative RFQ_E_* code; rfq_last_error(ctx) then holds the reference's error_exit text
 *     (src/util.h:246-249) where the reference has one for the condition.
 *   - one rfq_ctx per (host thread, GPU); distinct contexts may be used concurrently.  A context owns its workspace
 *     and its result buffers; result pointers stay valid until the next call on the same context.
 *   - there is NO CPU fallback: without a usable GPU rfq_create fails with RFQ_E_NO_DEVICE.
 */
#ifndef RFQ_HIP_H
#define RFQ_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RFQ_OK              0
#define RFQ_E_NO_DEVICE    -1   /* no GPU / hip runtime error at create */
#define RFQ_E_HIP          -2   /* a hip call failed (message has the hip error string) */
#define RFQ_E_ARG          -3   /* bad argument */
#define RFQ_E_TEXT         -4   /* (no longer returned: '\r' line ends and blank lines take the normalising path, src/fastqreader.cpp:94-196) */
#define RFQ_E_DATA         -5   /* the reference would error_exit on this input (message = its text) */
#define RFQ_E_FORMAT       -6   /* not a valid .rfq / different ALGORITHM_VER (src/rfqheader.cpp:23-25,40-42) */
#define RFQ_E_UNPINNED     -7   /* input is in a reference-UB zone: refused rather than guessed */
#define RFQ_E_NOSPACE      -8   /* caller-provided output buffer too small (required size is reported) */
#define RFQ_E_STATE        -9   /* call order (e.g. encode continuation without a header) */

/* how the FASTQ streams pair up — Repaq::run, src/repaq.cpp:12-20 */
#define RFQ_SE             0    /* -i           : compress()    */
#define RFQ_PE_TWO_FILES   1    /* -i/-I        : compressPE()  */
#define RFQ_PE_INTERLEAVED 2    /* --interleaved_in             */

#define RFQ_HEADER_MAX     (17 + 255)

typedef struct rfq_ctx rfq_ctx;

/* the library is built with -fvisibility=hidden: these entry points are its whole dynamic symbol table */
#define RFQ_API __attribute__((visibility("default")))

/* RfqCodec::RfqCodec / ~RfqCodec (src/rfqcodec.cpp:9-14).  device_id: HIP ordinal. */
RFQ_API int         rfq_create(rfq_ctx** out, int device_id);
RFQ_API void        rfq_destroy(rfq_ctx* ctx);
RFQ_API const char* rfq_last_error(const rfq_ctx* ctx);
/* all work of this context is enqueued on `hip_stream` (a hipStream_t; NULL = the context's own stream) */
RFQ_API int         rfq_set_stream(rfq_ctx* ctx, void* hip_stream);

/* RfqCodec::setHeader (src/rfqcodec.cpp:16-18) from the on-disk header bytes (RfqHeader::read, src/rfqheader.cpp:19-43).
 * mSupportInterleaved is not stored on disk; it is re-derived from BIT_ENCODE_PE_BY_OVERLAP, which makeHeader sets
 * exactly when it is true (src/rfqcodec.cpp:117-122). */
RFQ_API int rfq_set_header(rfq_ctx* ctx, const uint8_t* h_header, size_t header_len);
/* the header currently set / made: RfqHeader::write (src/rfqheader.cpp:84-97) */
RFQ_API int rfq_get_header(rfq_ctx* ctx, uint8_t* h_out /* >= RFQ_HEADER_MAX */, size_t* header_len);
RFQ_API void rfq_clear_header(rfq_ctx* ctx);

typedef struct {
    const uint8_t* d_fq1; size_t n1;      /* stream 1 (R1, or the only / interleaved stream)                          */
    const uint8_t* d_fq2; size_t n2;      /* stream 2 (R2) for RFQ_PE_TWO_FILES, else NULL/0                          */
    int32_t  paired;                      /* RFQ_SE / RFQ_PE_TWO_FILES / RFQ_PE_INTERLEAVED                            */
    uint32_t chunk_bases;                 /* Options::chunkSize = max(100,-k)*1000 (src/main.cpp:69); any value >= 1   */
    int32_t  final;                       /* 1: this is the end of the input: also emit the tail chunk
                                             (src/repaq.cpp:590-624).  0: stop after the last full chunk and report how
                                             many bytes were consumed so the caller can carry the rest into the next
                                             batch.                                                                   */
    int32_t  emit_header;                 /* 1: prefix the result with the file header (first batch of a file)        */
    /* line-break bits (src/repaq.cpp:571-572,683-692): a chunk whose last record ends at absolute
     * file offset >= nolb_from{1,2} gets BIT_HAS_NO_LINE_BREAK_AT_END{,_R2}.  The caller (the Repaq::compress counterpart)
     * passes file_off = offset of this batch in the file and nolb_from = start of the final 1 MiB reader block when the
     * file lacks a trailing '\n', or UINT64_MAX.  */
    uint64_t file_off1, file_off2;
    uint64_t nolb_from1, nolb_from2;
    uint8_t* d_out; size_t out_cap;       /* optional caller buffer for the .rfq bytes; NULL = context-owned result    */
    int32_t  flush_all;                   /* 1: the batch ends exactly on a chunk boundary found by rfq_scan_batch: encode
                                             every record of it (like final) without treating its end as the end of the
                                             input (an unterminated last line is not a line; the 1 MiB reader-block rule
                                             still sees the file continue).  For workers of a multi-GPU host queue.     */
    uint32_t carry_bases;                 /* plan pass of a share of a larger input (rfq_scan_batch): bases the chunk that is open at the start of
                                             this text has already taken from the text in front of it (< chunk_bases); the first chunk
                                             closes at chunk_bases - carry_bases.  0 for an encode (a range starts on a chunk boundary).   */
} rfq_encode_args;

typedef struct {
    const uint8_t* d_rfq;                 /* device pointer to [header?][chunk][chunk]...                              */
    size_t   rfq_len;
    uint32_t n_chunks;
    uint64_t n_reads;                     /* reads encoded (PE: both mates counted, like RfqChunk::mReads)            */
    uint64_t n_bases;
    size_t   consumed1, consumed2;        /* bytes of each stream covered by the emitted chunks                        */
    const uint64_t* h_chunk_off;          /* host array [n_chunks+1]: byte offset of each chunk image in d_rfq         */
    int32_t  input_ended;                 /* 1: the reader met an empty line inside a record: FastqReader::read returns NULL
                                             there (src/fastqreader.cpp:180-191), so the result already holds the tail chunk
                                             and the caller must not feed the rest of the input                          */
    int32_t  reserved;
} rfq_encode_result;

/* RfqCodec::makeHeader (first call without a header; src/rfqcodec.cpp:20-145) + RfqCodec::encodeChunk + RfqChunk::write
 * for every chunk of the batch (src/rfqcodec.cpp:147-586, src/rfqchunk.cpp:230-311), with the chunk cut rule of
 * Repaq::compress (src/repaq.cpp:546-553): cut after the read that brings the running base count to >= chunk_bases. */
RFQ_API int rfq_encode_batch(rfq_ctx* ctx, const rfq_encode_args* args, rfq_encode_result* res);

/* The plan pass of a chunk-parallel encode: line index, read lengths and the cut rule of Repaq::compress
 * (src/repaq.cpp:546-553) only — no header, no coding.  h_end1/2[c] = offset in the caller's stream(s) just past the last
 * record of chunk c (for RFQ_PE_TWO_FILES one array per file, else h_end2 is NULL); a host work queue hands the byte ranges
 * [h_end[c0-1], h_end[c1-1]) to other contexts / GPUs, which encode them with flush_all = 1 and the header of the first
 * range (rfq_get_header -> rfq_set_header); the concatenated images equal the one-shot image.  Takes the same arguments as
 * rfq_encode_batch (final = 0: only full chunks are planned, consumed* says where the next scan starts). */
typedef struct {
    uint32_t n_chunks;
    uint64_t n_reads;
    size_t   consumed1, consumed2;
    const uint64_t* h_end1;               /* host arrays [n_chunks], valid until the next call on the context           */
    const uint64_t* h_end2;
    int32_t  input_ended;                 /* as in rfq_encode_result                                                   */
    uint32_t unit_bases;                  /* bases of every cut unit (a read, or a pair) when they are all the same, else 0: with equal units the
                                             chunk cuts of a share follow from the number of units in front of it (repaq_amd/dist.py)      */
} rfq_scan_result;
RFQ_API int rfq_scan_batch(rfq_ctx* ctx, const rfq_encode_args* args, rfq_scan_result* res);

typedef struct {
    const uint8_t* d_rfq; size_t n;       /* .rfq bytes in HBM                                                          */
    int32_t  has_header;                  /* 1: the image starts with the file header (it is parsed and set)           */
    int32_t  split_pe;                    /* 1: decompressPE (even reads -> out1, odd -> out2; src/repaq.cpp:367-373);
                                             0: decompress (everything to out1 in chunk order; Q14)                    */
    int32_t  final;                       /* 1: the image ends the file: apply the NO_LINE_BREAK bits of the last chunk
                                             (src/repaq.cpp:301-328,375-413)                                           */
    int32_t  bug_compat;                  /* 1: Repaq::decompressPE as it stands (src/repaq.cpp:330-417): with split_pe = 1, a chunk that carries a
                                             NO_LINE_BREAK bit and is not the image's last makes the reference's loop lose the chunk behind it (and, with
                                             the R1 bit, the flagged chunk's R2 text).  0 (default): every read is kept.  With final = 0 a flagged chunk
                                             at the very end of the range stays unconsumed (what follows it decides).  Repaq::decompress (split_pe = 0,
                                             :262-328) keeps the chunk it peeks at and loses nothing: the flag changes nothing there.               */
    uint8_t* d_out1; size_t cap1;         /* optional caller buffers (16-byte aligned); NULL = context-owned results.  cap must cover the text plus its last line break (a
                                             file that ends without one is trimmed in the result's count, not in the buffer).  When the call fails, what the buffers hold is
                                             unspecified: with caller buffers the text is written before the host has looked at the image's verdict.               */
    uint8_t* d_out2; size_t cap2;
    /* optional chunk index (the .rfq format has none: RfqChunk::read finds chunk c+1 only by parsing chunk c, src/rfqchunk.cpp:161-228,
     * a dependent load per chunk).  A host that has the offsets - it encoded the image (rfq_encode_result.h_chunk_off), or it walked
     * the chunk headers while the image was on its way to the GPU - passes them: h_chunk_off[0 .. n_chunk_off] = byte offset of every
     * chunk in d_rfq and, last, the end of the last chunk.  Every extent is still verified on the device by a full parse of its chunk;
     * a table that does not verify is ignored (the chain is walked instead).  NULL / 0 = walk.                                      */
    const uint64_t* h_chunk_off; uint32_t n_chunk_off; uint32_t reserved3;
} rfq_decode_args;

typedef struct {
    const uint8_t* d_fq1; size_t n1;
    const uint8_t* d_fq2; size_t n2;
    uint32_t n_chunks;
    uint64_t n_reads, n_bases;
    size_t   consumed;                    /* bytes of the image covered by whole chunks                                */
} rfq_decode_result;

/* RfqChunk::read + RfqCodec::decodeChunk + Read::toString for every chunk of the image
 * (src/rfqchunk.cpp:161-228, src/rfqcodec.cpp:826-1260, src/read.cpp:170-172).  Images whose header lacks
 * BIT_ENCODE_QUAL_BY_COL (legacy run-length quality coding, src/rfqcodec.cpp:919-955; v0.5.1 never writes one) decode too. */
RFQ_API int rfq_decode_batch(rfq_ctx* ctx, const rfq_decode_args* args, rfq_decode_result* res);

/* stage timings of the last batch call, in milliseconds, measured with HIP events on the context's stream.
 * names[i] is a static string; returns the number of stages written (<= cap). */
RFQ_API int rfq_last_timings(const rfq_ctx* ctx, const char** names, float* ms, int cap);

/* device-memory helpers for hosts that do not bring their own allocator (the C++ driver, ctypes tests): thin wrappers
 * over hipMalloc / hipFree / hipMemcpy on the context's device.  Buffers from rfq_dev_malloc are 256-byte aligned. */
RFQ_API int rfq_dev_malloc(rfq_ctx* ctx, void** d_ptr, size_t n);
RFQ_API int rfq_dev_free(rfq_ctx* ctx, void* d_ptr);
RFQ_API int rfq_copy_h2d(rfq_ctx* ctx, void* d_dst, const void* h_src, size_t n);
/* The same without waiting: the copy is queued on the context's copy stream (h_src page-locked: rfq_host_alloc) and runs beside whatever the
 * context's own stream does.  *ticket (optional) identifies it: rfq_copy_done says whether it has finished (the host buffer may be reused),
 * rfq_copy_sync waits for everything queued so far - call it before handing the destination to rfq_encode_batch / rfq_decode_batch. */
RFQ_API int rfq_copy_h2d_async(rfq_ctx* ctx, void* d_dst, const void* h_src, size_t n, uint64_t* ticket);
RFQ_API int rfq_copy_done(rfq_ctx* ctx, uint64_t ticket);      /* 1: finished, 0: still running, < 0: error */
RFQ_API int rfq_copy_sync(rfq_ctx* ctx);
RFQ_API int rfq_copy_d2h(rfq_ctx* ctx, void* h_dst, const void* d_src, size_t n);
RFQ_API int rfq_copy_d2d(rfq_ctx* ctx, void* d_dst, const void* d_src, size_t n);
/* d_dst on ctx's GPU <- d_src on src_ctx's GPU (hipMemcpyPeerAsync): how a worker of a multi-GPU host queue pulls its byte range */
RFQ_API int rfq_copy_peer(rfq_ctx* ctx, void* d_dst, const rfq_ctx* src_ctx, const void* d_src, size_t n);
/* page-locked host buffers (hipHostMalloc): H2D / D2H copies from them run at full PCIe rate */
RFQ_API int rfq_host_alloc(rfq_ctx* ctx, void** h_ptr, size_t n);
RFQ_API int rfq_host_free(rfq_ctx* ctx, void* h_ptr);
/* the same for memory the caller owns (hipHostRegister / hipHostUnregister): buffers a host filled before the context existed */
RFQ_API int rfq_host_register(rfq_ctx* ctx, void* h_ptr, size_t n);
RFQ_API int rfq_host_unregister(rfq_ctx* ctx, void* h_ptr);

/* --compare on the device (Repaq::compare / comparePE, src/repaq.cpp:36-233): the first offset at which two device texts differ,
 * *first_diff = n when they are identical.  A decoded batch equal byte for byte to the same span of the FASTQ text passes the
 * reference's four per-read tests (name, sequence, strand, quality; :85-108) for every read in it; the host cuts records only in
 * a batch that differs, to word the reference's message. */
RFQ_API int rfq_compare_bytes(rfq_ctx* ctx, const void* d_a, const void* d_b, size_t n, uint64_t* first_diff);

/* Test / diagnostic switches of one context: name = the RFQ_* environment variable of the same meaning (RFQ_GATHER=old, RFQ_QUAL=bytes|masks, RFQ_CODER=list|mask, RFQ_INDEX=2pass,
 * RFQ_IDX_TILES, RFQ_STREAMS=1, RFQ_SLICE_BYTES, RFQ_SLICE_BASES, RFQ_WALK=exact, RFQ_GW_SHIFT, RFQ_MATERIALISE=1, RFQ_TRACE, RFQ_G2_PAD, RFQ_SP_PAD; see RfqOpts in
 * repaq_amd/csrc/rfq_ctx.h), value NULL or "" = default.  Every switch selects another formulation of the same, bit-identical result - they exist so that
 * the tests can pin each one (tests/test_gpu_formulations.py forces every one of them on the GPU).  Unknown names and values that are not of the switch's
 * form or range are RFQ_E_ARG.  The environment is read once, by rfq_create; the batch calls never call getenv. */
RFQ_API int rfq_set_option(rfq_ctx* ctx, const char* name, const char* value);
/* the switch's current value in the form rfq_set_option takes ("" = default), so that a caller can put back what it found */
RFQ_API int rfq_get_option(const rfq_ctx* ctx, const char* name, char* out, size_t cap);
/* the switches' names: i = 0, 1, ... until NULL */
RFQ_API const char* rfq_option_name(int i);

/* self test of the wave-level scans / reductions every kernel is built on (DPP row shifts and broadcasts on gfx950): h_in holds 64 * n_waves lane
 * values, h_out receives 12 u64 per lane (see k_selftest_wave in rfq_api.hip); tests/test_wave_primitives.py checks them against a serial reference. */
RFQ_API int rfq_selftest_wave(rfq_ctx* ctx, const uint64_t* h_in, uint32_t n_waves, uint64_t* h_out);

/* library / build info: "rfq_hip <version> gfx950" (or "... simt-emulation" for the test build) */
RFQ_API const char* rfq_version(void);

#ifdef __cplusplus
}
#endif
#endif
