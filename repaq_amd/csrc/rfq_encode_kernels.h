// rfq_encode_kernels.h — gfx950 kernels of the FASTQ -> RFQ encode path (included by rfq_encode.hip only).
//
// Pipeline (one batch = whole FASTQ stream(s) resident in HBM; reference call sites in brackets):
//   index    k_nl_bitmap -> scan -> k_line_offsets            newline bitmap, global line ranks, line-start table
//            [FastqReader::getLine/read, src/fastqreader.cpp:94-196, for '\n'-terminated text]
//   table    k_read_table                                     per-read lengths + FastqMeta::parse [src/fastqmeta.cpp:22-80]
//   cut      scan + k_partition                               chunk boundaries [Repaq::compress, src/repaq.cpp:546-553]
//   header   k_hdr_*                                          RfqCodec::makeHeader + makeQualityTable on chunk 0
//   chunk    k_chunk_flags, k_overlap, scans, k_gather, k_stream_plan, k_pos_coder, k_coords, k_chunk_layout,
//            k_assemble*                                      RfqCodec::encodeChunk + RfqChunk::write
#pragma once
// The kernels live in enc/*.h by stage (one 244 KB header was unreviewable); this file is the order they are included in.
#include "enc/tables.h"                       // the batch's text, per-read and per-chunk tables
#include "enc/index.h"                        // line index (one pass with a decoupled look-back; two-pass fallback) and the text normaliser of the slow path
#include "enc/reads_cut.h"                    // name parse, read lengths, chunk partition
#include "enc/hdr_from_chunk0.h"              // the file header from chunk 0 (makeHeader, makeQualityTable)
#include "enc/chunk_flags_overlap.h"          // per-chunk flag words, interleave test, the overlap search, stored prefixes
#include "enc/gather_bytes.h"                 // byte-wise gather (reads that do not fit a tile) + the counters both gathers share
#include "enc/gather_tiles.h"                 // tile gather k_gather2 (names parsed, match masks, bases packed where they stand), sequence packer, stream plan
#include "enc/pos_coder.h"                    // position coder: a wave per (chunk, value streams, segment)
#include "enc/pos_coder_list.h"               // position coder for many value streams: work follows the coded positions
#include "enc/coords.h"                       // coordinate coder (encodeCoords)
#include "enc/assemble.h"                     // chunk layout (incl. the mSize bug) and image assembly
