// ref_harness — unit-level access to the reference's private codec members, for golden vectors and
// differential tests of oracle/rfq_oracle.c.  This file is OUR code; it is compiled against the reference
// sources where they lie (/root/reference/src, -iquote) into oracle/_ref/ (git-ignored).  Nothing is copied.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <iostream>
#define private public
#include "rfqcodec.h"
#undef private
#include "fastqmeta.h"
#include "fastqreader.h"
#include <fstream>

std::string command;   // referenced by the reference's util.h error path

static std::vector<unsigned char> slurp() {
    std::vector<unsigned char> v; unsigned char buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, stdin)) > 0) v.insert(v.end(), buf, buf + n);
    return v;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_harness coords|decoords N|pos Q|overlap|parse|rle_image R1 R2|- OUT\n"); return 2; }
    std::string cmd = argv[1];
    RfqCodec codec;
    if (cmd == "coords") {            // stdin: u32 LE values -> stdout: encoded stream
        std::vector<unsigned char> in = slurp(); uint32 n = in.size() / 4;
        std::vector<uint8> out(n * 3 + 8);
        uint32 l = codec.encodeCoords((uint32*)in.data(), out.data(), n);
        fwrite(out.data(), 1, l, stdout);
    } else if (cmd == "decoords") {   // argv[2]=count; stdin: stream -> stdout: u32 LE values
        std::vector<unsigned char> in = slurp(); uint32 n = atoi(argv[2]);
        std::vector<uint32> out(n + 64, 0);
        codec.decodeCoords(in.data(), in.size(), out.data(), n);
        fwrite(out.data(), 4, n, stdout);
    } else if (cmd == "pos") {        // argv[2]=byte value; stdin: buffer -> stdout: position stream
        std::vector<unsigned char> in = slurp(); uint8 q = (uint8)atoi(argv[2]);
        std::vector<uint8> out(in.size() * 4 + 16);
        uint32 l = codec.encodeSingleQualByCol(in.data(), q, out.data(), in.size(), NULL);
        fwrite(out.data(), 1, l, stdout);
    } else if (cmd == "overlap") {    // stdin: two lines -> prints overlap
        std::string a, b; std::getline(std::cin, a); std::getline(std::cin, b);
        printf("%d\n", codec.overlap(a, b));
    } else if (cmd == "parse") {      // stdin: one name per line -> ok|name1|lane|tile|x|y|name2
        std::string s;
        while (std::getline(std::cin, s)) {
            FastqMeta m = FastqMeta::parse(s);
            printf("%d|%s|%d|%d|%u|%u|%s\n", (int)m.hasLaneTileXY, m.namePart1.c_str(), (int)m.lane, (int)m.tile, m.x, m.y, m.namePart2.c_str());
        }
    } else if (cmd == "rle_image") {  // argv[2]=R1.fq argv[3]=R2.fq or "-" argv[4]=out.rfq: ONE chunk coded with the legacy run-length quality coder.
        // RfqCodec::encodeSeqQual takes that branch when the header carries neither BIT_ENCODE_QUAL_BY_COL nor BIT_DONT_ENCODE_QUAL
        // (src/rfqcodec.cpp:612-621); v0.5.1's makeQualityTable never leaves a header that way (App. C Q13), so the flag is cleared here
        // on the header the reference itself made from the reads.  Everything else - header fields, chunk layout - is the reference's.
        std::string r1 = argv[2], r2 = argv[3]; std::ofstream out(argv[4], std::ios::binary);
        if (r2 == "-") {
            FastqReader reader(r1); std::vector<Read*> reads; Read* r;
            while ((r = reader.read()) != NULL) reads.push_back(r);
            RfqHeader* h = codec.makeHeader(reads); h->mFlags &= ~BIT_ENCODE_QUAL_BY_COL; codec.setHeader(h);
            RfqChunk* c = codec.encodeChunk(reads); h->write(out); c->write(out);
        } else {
            FastqReaderPair reader(r1, r2); std::vector<ReadPair*> pairs; ReadPair* p;
            while ((p = reader.read()) != NULL) pairs.push_back(p);
            RfqHeader* h = codec.makeHeader(pairs); h->mFlags &= ~BIT_ENCODE_QUAL_BY_COL; codec.setHeader(h);
            RfqChunk* c = codec.encodeChunk(pairs); h->write(out); c->write(out);
        }
    } else return 2;
    return 0;
}
