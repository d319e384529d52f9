"""The reference's private-member known answers (tests/golden/unit.json: encodeCoords, encodeSingleQualByCol, overlap, FastqMeta::parse - src/rfqcodec.cpp:1262-1330,
625-710, 1391-1438, src/fastqmeta.cpp:22-80) checked at the BOUNDARY: each vector becomes a small FASTQ whose image holds that member's output verbatim
in one section (tests/_sections.py), the HIP path encodes it through the C-ABI, and the section is compared with the reference's bytes.  Shared by the GPU test and
its interpreter twin."""
import json
import os
import random

import _engine as E
import _oracle as O
import _sections as S

UNIT = json.load(open(os.path.join(E.ROOT, "tests", "golden", "unit.json")))
_COMP = {65: 84, 84: 65, 67: 71, 71: 67}


def _rc(s: bytes) -> bytes:
    return bytes(_COMP.get(b, 78) for b in reversed(s))


def _se(names, seqs, quals):
    return b"".join(b"%s\n%s\n+\n%s\n" % (n, s, q) for n, s, q in zip(names, seqs, quals))


def check_coords(codec, opts=None):
    """k_coords: reads whose names carry the vector's values as X (and, reversed, as Y): the x / y sections are encodeCoords' streams"""
    n_checked = 0
    for v in UNIT["coords"]:
        vals = v["values"]; rnd = random.Random(len(vals))
        seqs = [bytes(rnd.choice(b"ACGT") for _ in range(30)) for _ in vals]
        fq = _se([b"@M:1:FC:1:7:%d:%d 1:N:0:A" % (x, x) for x in vals], seqs, [b"F" * 30] * len(vals))
        with E._Options(codec, opts or {}):
            rfq = E.encode(codec, fq, b"", O.SE, 1_000_000)
        h, ch = S.parse(rfq)
        assert len(ch) == 1 and (h.flags & S.H_X) and ch[0].x.hex() == v["hex"] and ch[0].y.hex() == v["hex"], (vals[:8], ch[0].x.hex(), v["hex"])
        n_checked += 1
    return n_checked


def check_pos(codec, opts=None):
    """k_pos_coder (mask planes by default, byte streams / the list coder under the switches): SE reads of 150 whose concatenated qualities are the vector's
    buffer - the value's stream in the quality section is encodeSingleQualByCol's output; the one N vector goes through the N-position section"""
    from cases import pos_buffers
    import hashlib
    n_checked = 0
    for (buf, q), v in zip(pos_buffers(), UNIT["pos"]):
        assert hashlib.md5(buf).hexdigest() == v["buf_md5"] and q == v["q"]
        rnd = random.Random(len(buf))
        # (two trailing reads of plain 'F' qualities: a stream ends with its last match, so they add nothing to it - but they keep 'F' the major value of the short
        # vectors and the payload of a six-base file under the reference's 1.5 x bases scratch buffer)
        pad_s = [bytes(rnd.choice(b"ACGT") for _ in range(150)) for _ in range(2)]; pad_q = [b"F" * 150] * 2
        if q == ord("N"):                                                    # the sequence IS the buffer (fewer than 100 N in chunk 0: their positions are coded, src/rfqheader.cpp:186-190)
            parts_s, parts_q = [buf], [b"F" * len(buf)]
        else:
            parts_q = [buf[i:i + 150] for i in range(0, len(buf), 150)]; parts_s = [bytes(rnd.choice(b"ACGT") for _ in p_) for p_ in parts_q]
        parts_s += pad_s; parts_q += pad_q
        fq = _se([b"@M:1:FC:1:7:%d:%d 1:N:0:A" % (1000 + i, 2000 + i) for i in range(len(parts_q))], parts_s, parts_q)
        with E._Options(codec, opts or {}):
            rfq = E.encode(codec, fq, b"", O.SE, 1_000_000)
        h, ch = S.parse(rfq); assert len(ch) == 1
        if q == ord("N"):
            assert h.flags & S.H_N_POS and ch[0].npos.hex() == v["hex"], (ch[0].npos.hex(), v["hex"])
            n_checked += 1
        elif q in h.normal and (h.flags & S.H_QUAL_BY_COL) and not (h.flags & S.H_DONT_QUAL):
            streams, _ = ch[0].quality_streams()
            assert streams[q].hex() == v["hex"], (len(buf), streams[q].hex()[:60], v["hex"][:60])
            n_checked += 1
        # (else: the value is the chunk's major one, or absent - it has no stream in a file; the vector pins the oracle only)
    return n_checked


def check_overlap(codec, opts=None):
    """k_overlap (+ the clamp into the overlap byte, src/rfqcodec.cpp:376-383): pairs whose R2 file record is the reverse complement of the vector's second string -
    encodeChunk reverse-complements the mate before it calls overlap() - the chunk's overlap section is overlap() - shift per pair"""
    r1s = [v["r1"].encode() for v in UNIT["overlap"]]; r2s = [_rc(v["r2"].encode()) for v in UNIT["overlap"]]
    names1 = [b"@M:1:FC:1:7:%d:%d 1:N:0:A" % (1000 + i, 2000 + i) for i in range(len(r1s))]; names2 = [n.replace(b" 1:N", b" 2:N") for n in names1]
    fq1 = _se(names1, r1s, [b"F" * len(s) for s in r1s]); fq2 = _se(names2, r2s, [b"F" * len(s) for s in r2s])
    with E._Options(codec, opts or {}):
        rfq = E.encode(codec, fq1, fq2, O.PE_TWO_FILES, 1_000_000)
    h, ch = S.parse(rfq)
    assert len(ch) == 1 and (h.flags & S.H_PE_OVERLAP) and (ch[0].flags & S.C_PE_INTERLEAVED) and len(ch[0].ov) == len(r1s)
    import struct
    got = [struct.unpack("b", bytes([b]))[0] - h.overlap_shift for b in ch[0].ov]
    assert got == [v["ov"] for v in UNIT["overlap"]], got
    return len(got)


def check_parse(codec, opts=None):
    """FastqMeta::parse (g2_parse inside k_gather2; k_read_table for chunk 0 / under RFQ_GATHER=old): one file per vector - the header's flags say whether the name was
    taken apart, name1 / lane / tile / X / Y / name2 come back from their sections; fields encodeCoords refuses (>= 2^21) come back in its error text"""
    from repaq_amd import RfqError
    n_checked = 0
    for v in UNIT["parse"]:
        name = v["name"].encode("latin-1"); ok, n1, lane, tile, x, y, n2 = v["ref"].split("|", 6)
        fq = _se([name], [b"ACGTACGTACGTAAAC"], [b"FFFFFFFFFFFFFFFF"])
        try:
            with E._Options(codec, opts or {}):
                rfq = E.encode(codec, fq, b"", O.SE, 1_000_000)
        except RfqError as e:                                               # encodeCoords' error_exit names the first offender: X before Y
            bad = int(x) if int(x) >= (1 << 21) else int(y)
            assert ok == "1" and bad >= (1 << 21) and e.message.strip() == "The X/Y coordinate cannot be larger than 2M, but we get: %d" % bad, (v, e.message)
            n_checked += 1; continue
        h, ch = S.parse(rfq); c = ch[0]
        assert bool(h.flags & S.H_LANE) == (ok == "1"), v
        assert c.n1 == n1.encode("latin-1"), (v, c.n1)
        if ok == "1":
            assert c.lanes == [int(lane)] and c.tiles == [int(tile)] and O.decode_coords(c.x, 1) == [int(x)] and O.decode_coords(c.y, 1) == [int(y)] and c.n2 == n2.encode("latin-1"), (v, c.lanes, c.tiles, c.x.hex(), c.y.hex(), c.n2)
        n_checked += 1
    return n_checked
