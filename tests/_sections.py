"""Test infrastructure: the sections of a .rfq image, cut exactly as RfqHeader::read / RfqChunk::read cut them (src/rfqheader.cpp:19-43, src/rfqchunk.cpp:161-228;
the reader ignores mSize and derives every extent from flags + length arrays).  Used to compare single sections of an image the HIP path wrote with the known answers
of the reference's private codec members (tests/golden/unit.json): a section's bytes ARE that member's output - the x section is encodeCoords' stream, a quality value's
stream is encodeSingleQualByCol's, the overlap section is overlap() + the header's shift, lanes / tiles / names are FastqMeta::parse's fields."""
import struct

H_LANE, H_TILE, H_X, H_Y, H_NAME2, H_PAIRED, H_PE_OVERLAP, H_QUAL_BY_COL, H_DONT_QUAL, H_N_POS = (1 << i for i in range(10))
C_READ_LEN_SAME, C_NAME1_LEN_SAME, C_NAME2_LEN_SAME, C_STRAND_LEN_SAME, C_LANE_SAME, C_TILE_SAME, C_NAME1_SAME, C_NAME2_SAME, C_STRAND_SAME, C_PE_INTERLEAVED = (1 << i for i in range(10))


class Header:
    def __init__(self, b: bytes):
        assert b[:3] == b"RFQ" and b[8] == 2, "not a v2 .rfq"
        self.read_len_bytes = b[9]; self.flags = b[10] | (b[11] << 8); self.name2_diff_pos = b[12]; self.name2_diff_char = b[13]
        self.n_base_qual = struct.unpack("b", b[14:15])[0]; self.overlap_shift = struct.unpack("b", b[15:16])[0]; self.bins = b[16]
        self.qual_table = b[17:17 + self.bins]; self.len = 17 + self.bins
        # majorQual / normalQualBuf (src/rfqheader.cpp:263,308-328): the table's first entry is the major value; the others - all of them when the major value
        # is also the N quality - get a stream each, in table order
        mq = struct.unpack("b", self.qual_table[:1])[0] if self.bins else 0
        nb = self.bins if mq == self.n_base_qual else max(0, self.bins - 1)
        self.normal = [v for v in self.qual_table if struct.unpack("b", bytes([v]))[0] != mq or struct.unpack("b", bytes([v]))[0] == self.n_base_qual][:nb]


class Chunk:
    """one chunk at rfq[k:]; .total = its size; sections as bytes"""
    def __init__(self, h: Header, rfq: bytes, k: int):
        p = rfq[k:]; u32 = lambda o: struct.unpack_from("<I", p, o)[0]
        self.size_field, self.reads, = u32(0), u32(4); self.flags = struct.unpack_from("<H", p, 8)[0]; self.seq_size, self.qual_size = u32(10), u32(14)
        q = 18; self.npos_size = 0
        if h.flags & H_N_POS:
            self.npos_size = u32(q); q += 4
        self.fixed = q                                   # bytes of the fixed fields (size, reads, flags, seq / quality / N-position sizes)
        s, fl, rlb = self.reads, self.flags, h.read_len_bytes
        cnt = 1 if fl & C_READ_LEN_SAME else s
        fmt = {1: "B", 2: "<H", 4: "<I"}[rlb]
        self.read_lens = [struct.unpack_from(fmt, p, q + i * rlb)[0] for i in range(cnt)]; q += cnt * rlb
        self.marks = [self.fixed, q]                     # offsets at which a section ends (hostile-image tests cut and scribble there)

        def lenarr(lenflag, sameflag):
            nonlocal q
            m = 1 if fl & lenflag else s
            arr = list(p[q:q + m]); q += m
            tot = sum(arr)
            if (fl & lenflag) and not (fl & sameflag):
                tot *= s
            return arr, tot
        self.n1_lens, n1_size = lenarr(C_NAME1_LEN_SAME, C_NAME1_SAME); self.marks.append(q)
        self.n2_lens, n2_size = (lenarr(C_NAME2_LEN_SAME, C_NAME2_SAME) if h.flags & H_NAME2 else ([], 0)); self.marks.append(q)
        self.st_lens, st_size = lenarr(C_STRAND_LEN_SAME, C_STRAND_SAME); self.marks.append(q)
        self.len_arrays_end = q                          # fixed fields + the four length arrays: what RfqChunk::read sizes everything else from
        hc = s // 2 if fl & C_PE_INTERLEAVED else s
        self.lanes = self.tiles = []; self.x = self.y = b""
        if h.flags & H_LANE:
            m = 1 if fl & C_LANE_SAME else hc; self.lanes = list(p[q:q + m]); q += m
        if h.flags & H_TILE:
            m = 1 if fl & C_TILE_SAME else hc; self.tiles = list(struct.unpack_from("<%dH" % m, p, q)); q += 2 * m
        if h.flags & H_X:
            n = u32(q); q += 4; self.x = p[q:q + n]; q += n
        if h.flags & H_Y:
            n = u32(q); q += 4; self.y = p[q:q + n]; q += n
        self.marks.append(q); self.coords_end = q
        self.n1 = p[q:q + n1_size]; q += n1_size; self.marks.append(q)
        self.n2 = b""
        if h.flags & H_NAME2:
            self.n2 = p[q:q + n2_size]; q += n2_size
        self.st = p[q:q + st_size]; q += st_size; self.marks.append(q)
        self.seq = p[q:q + self.seq_size]; q += self.seq_size; self.marks.append(q)
        self.qual_off = q
        self.qual = p[q:q + self.qual_size]; q += self.qual_size; self.marks.append(q)
        self.ov = b""
        if (fl & C_PE_INTERLEAVED) and (h.flags & H_PE_OVERLAP):
            self.ov = p[q:q + s // 2]; q += s // 2
        self.npos = p[q:q + self.npos_size] if h.flags & H_N_POS else b""; q += self.npos_size
        self.total = q; self.marks.append(q); self.marks = sorted(set(self.marks))
        assert q <= len(p), "truncated chunk"
        self.h = h

    def quality_streams(self):
        """{quality value: its position stream}, [exception records] of a BY_COL payload: u32 LE len[bins] | streams | (q, u32 LE pos) records (src/rfqcodec.cpp:712-765)"""
        h = self.h; nn = len(h.normal); lens = struct.unpack_from("<%dI" % nn, self.qual, 0); o = 4 * nn; out = {}
        for v, n in zip(h.normal, lens):
            out[v] = self.qual[o:o + n]; o += n
        rest = self.qual[o:]
        return out, [(rest[i], struct.unpack_from("<I", rest, i + 1)[0]) for i in range(0, len(rest) - 4, 5)]


def parse(rfq: bytes):
    h = Header(rfq); k = h.len; chunks = []
    while len(rfq) - k >= 18:
        c = Chunk(h, rfq, k)
        if c.reads == 0:
            break
        c.off = k                                        # where the chunk starts in the image
        chunks.append(c); k += c.total
    return h, chunks
