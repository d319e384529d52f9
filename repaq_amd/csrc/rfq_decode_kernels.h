// rfq_decode_kernels.h — gfx950 kernels of the RFQ -> FASTQ decode path (included by rfq_decode.hip only).
//
//   walk     k_dec_walk                 RfqChunk::read for every chunk: section offsets (the reader ignores mSize,
//                                       src/rfqchunk.cpp:161-228), running read base
//   table    k_dec_readtab + scans      per-read lengths, overlap values, stored lengths, name/strand piece lengths
//   streams  k_dec_unpack, k_dec_pos, k_dec_except, k_dec_coords
//                                       decodeSeqQual / decodeSingleQualByCol / decodeQualByCol / decodeCoords
//                                       (src/rfqcodec.cpp:826-1047,1332-1389)
//   text     k_dec_textlen + scan, k_dec_emit
//                                       decodeChunk's per-read loop + Read::toString (src/rfqcodec.cpp:1141-1254, src/read.cpp:170)
#pragma once
// The kernels live in dec/*.h by stage; this file is the order they are included in.
#include "dec/chunk_walk.h"                   // chunk descriptors, RfqChunk::read per chunk, the index-less walk (exact, guess and verify), verification
#include "dec/read_table.h"                   // per-read table, chunk bases, 2-bit unpack of the expanded path
#include "dec/pos_streams.h"                  // token-boundary automaton; position streams of the expanded path (summary, link, emit)
#include "dec/pos_lists.h"                    // fused path: position lists + cell index (sum2, link2, off, list); exception records, RLE, prefill, coordinates
#include "dec/text_len.h"                     // name middles and text lengths
#include "dec/emit_expanded.h"                // text emission of the expanded path (tile emitter k_dec_emit)
#include "dec/emit_tiles.h"                   // text emission, fused path: k_dec_emit3 (fixed tiles, no output tile)
