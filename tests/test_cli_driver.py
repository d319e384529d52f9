"""The C++ host driver (repaq's -c/-d/--compare command line over the C-ABI).  CPU: linked against the SIMT-emulation
test library to exercise flag handling, batching with carry-over and the compare JSON; GPU: the real binary."""
import json
import os
import subprocess

import pytest

import _engine as E
import _oracle as O

EMU_BIN = os.path.join(E.EMU_DIR, "repaq_hip_emu")
GPU_BIN = os.path.join(E.ROOT, "repaq_amd", "bin", "repaq_hip")


def _run(binary, args, **kw):
    return subprocess.run([binary] + args, capture_output=True, **kw)


def _suite(binary, tmp_path, reads_se, pairs, batch_mb):
    fq1, _ = O.gen(O.NOVA_SE150, reads_se, seed=41, nonl=1)
    p = tmp_path / "a.fq"; p.write_bytes(fq1)
    out = tmp_path / "a.rfq"
    r = _run(binary, ["-c", "-i", str(p), "-o", str(out), "-k", "100", "--batch_mb", str(batch_mb)])
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == O.encode_file(fq1, b"", O.SE, 100_000)
    back = tmp_path / "back.fq"
    r = _run(binary, ["-d", "-i", str(out), "-o", str(back)])
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == fq1
    r = _run(binary, ["-p", "-i", str(p), "-r", str(out), "-j", str(tmp_path / "cmp.json")])
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout)
    assert j["result"] == "passed" and j["fastq_reads"] == reads_se and j["rfq_reads"] == reads_se and j["fastq_bases"] == reads_se * 150
    assert json.loads((tmp_path / "cmp.json").read_text()) == j
    # reader quirks through the batching loop (consumed offsets map back through the normalised text): CRLF line ends, and an
    # empty line mid-file after which nothing is read (src/fastqreader.cpp:94-196)
    crlf = fq1.replace(b"\n", b"\r\n")
    pc = tmp_path / "crlf.fq"; pc.write_bytes(crlf)
    r = _run(binary, ["-c", "-i", str(pc), "-o", str(out), "-k", "100", "--batch_mb", str(batch_mb)])
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == O.encode_file(crlf, b"", O.SE, 100_000)
    cut = fq1.index(b"\n@", len(fq1) * 2 // 3) + 1
    stop = fq1[:cut] + b"\n\n" + fq1[cut:]
    ps = tmp_path / "stop.fq"; ps.write_bytes(stop)
    r = _run(binary, ["-c", "-i", str(ps), "-o", str(out), "-k", "100", "--batch_mb", str(batch_mb)])
    assert r.returncode == 0, r.stderr
    want = O.encode_file(stop, b"", O.SE, 100_000)
    assert out.read_bytes() == want and len(want) < len(O.encode_file(fq1, b"", O.SE, 100_000))
    # PE: -i/-I -> one .rfq, -o/-O back
    a, b = O.gen(O.NOVA_PE150, pairs, seed=42)
    pa, pb = tmp_path / "r1.fq", tmp_path / "r2.fq"; pa.write_bytes(a); pb.write_bytes(b)
    pe = tmp_path / "pe.rfq"
    r = _run(binary, ["-i", str(pa), "-I", str(pb), "-o", str(pe), "-k", "100", "--batch_mb", str(batch_mb)])   # compress is the default mode
    assert r.returncode == 0, r.stderr
    assert pe.read_bytes() == O.encode_file(a, b, O.PE_TWO_FILES, 100_000)
    o1, o2 = tmp_path / "o1.fq", tmp_path / "o2.fq"
    assert _run(binary, ["-d", "-i", str(pe), "-o", str(o1), "-O", str(o2)]).returncode == 0
    assert (o1.read_bytes(), o2.read_bytes()) == (a, b)
    # .gz text in and out (zlib, src/fastqreader.cpp:31-37, src/writer.cpp:39-51), the .rfq.xz wrapper (external xz, src/main.cpp:134-177),
    # --stdin / --stdout, and interleaved text for a PE image decoded to STDOUT
    import gzip
    import shutil
    pg = tmp_path / "a.fq.gz"; pg.write_bytes(gzip.compress(fq1, 1))
    og = tmp_path / "ag.rfq"
    assert _run(binary, ["-c", "-i", str(pg), "-o", str(og), "-k", "100", "--batch_mb", str(batch_mb)]).returncode == 0
    assert og.read_bytes() == O.encode_file(fq1, b"", O.SE, 100_000)
    bg = tmp_path / "back.fq.gz"
    assert _run(binary, ["-d", "-i", str(og), "-o", str(bg), "--batch_mb", str(batch_mb)]).returncode == 0
    assert gzip.decompress(bg.read_bytes()) == fq1
    # the .gz the driver writes is blocked gzip (members of <= 64 KiB with their size in a 'BC' extra field, an empty member at the end: what bgzip
    # writes), deflated and - read back - inflated on all I/O threads; any gzip reader sees the same text, the reference included; a plain .gz
    # and a file that turns from blocked into plain gzip half way go through zlib's one-stream reader
    z = bg.read_bytes(); off = 0; members = 0
    while off < len(z):
        assert z[off:off + 4] == b"\x1f\x8b\x08\x04" and z[off + 10:off + 16] == b"\x06\x00BC\x02\x00", "member %d at %d is not a BGZF member" % (members, off)
        off += (z[off + 16] | (z[off + 17] << 8)) + 1; members += 1
    assert off == len(z) and members >= 2 and z[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    og2 = tmp_path / "ag2.rfq"
    assert _run(binary, ["-c", "-i", str(bg), "-o", str(og2), "-k", "100", "--batch_mb", str(batch_mb), "--block_mb", "1"]).returncode == 0
    assert og2.read_bytes() == og.read_bytes()
    mixed = tmp_path / "mixed.fq.gz"; cutz = 0
    for _ in range(max(1, members // 2)):
        cutz += (z[cutz + 16] | (z[cutz + 17] << 8)) + 1
    mixed.write_bytes(z[:cutz] + gzip.compress(fq1[len(gzip.decompress(z[:cutz])):], 1))
    assert _run(binary, ["-c", "-i", str(mixed), "-o", str(og2), "-k", "100", "--batch_mb", str(batch_mb)]).returncode == 0
    assert og2.read_bytes() == og.read_bytes()
    ez = tmp_path / "empty_back.fq.gz"; pe0 = tmp_path / "e0.rfq"; pe0.write_bytes(b"")
    assert _run(binary, ["-d", "-i", str(pe0), "-o", str(ez)]).returncode == 0 and gzip.decompress(ez.read_bytes()) == b""
    if O.have_ref():
        rr = tmp_path / "ref_from_blocked.rfq"
        assert subprocess.run([O.REF_BIN, "-c", "-i", str(bg), "-o", str(rr), "-k", "100"], capture_output=True).returncode == 0
        assert rr.read_bytes() == og.read_bytes()
    if shutil.which("xz"):
        ox = tmp_path / "a.rfq.xz"
        r = _run(binary, ["-c", "-i", str(p), "-o", str(ox), "-k", "100", "-z", "1", "--batch_mb", str(batch_mb)])
        assert r.returncode == 0, r.stderr
        assert subprocess.run(["xz", "-d", "-c", str(ox)], capture_output=True, check=True).stdout == og.read_bytes()
        bx = tmp_path / "backx.fq"
        assert _run(binary, ["-d", "-i", str(ox), "-o", str(bx), "--batch_mb", str(batch_mb)]).returncode == 0
        assert bx.read_bytes() == fq1
    r = _run(binary, ["-c", "--stdin", "--stdout", "-k", "100", "--batch_mb", str(batch_mb)], input=fq1)
    assert r.returncode == 0 and r.stdout == og.read_bytes(), r.stderr
    r = _run(binary, ["-d", "--stdin", "--stdout", "--batch_mb", str(batch_mb)], input=og.read_bytes())
    assert r.returncode == 0 and r.stdout == fq1, r.stderr
    r = _run(binary, ["-d", "-i", str(pe), "--stdout", "--batch_mb", str(batch_mb)])
    assert r.returncode == 0 and r.stdout == O.decode_file(pe.read_bytes(), False)
    # -v / -f: every (tenth) batch is decoded again on a second context and compared with its text; silent and byte-identical output when
    # the codec is right, for plain text (device compare), for CRLF text (record cutter) and for pairs (src/repaq.cpp:430-528, App. C Q18)
    ov = tmp_path / "v.rfq"
    for flag, src, extra, ref in (("-v", p, [], og.read_bytes()), ("-f", p, [], og.read_bytes()), ("-v", pc, [], O.encode_file(crlf, b"", O.SE, 100_000)),
                                  ("-v", ps, [], want), ("-v", pa, ["-I", str(pb)], pe.read_bytes()), ("-v", p, ["--devices", "0,0"], og.read_bytes())):
        r = _run(binary, ["-c", flag, "-i", str(src), "-o", str(ov), "-k", "100", "--batch_mb", str(batch_mb)] + extra)
        assert r.returncode == 0 and r.stderr == b"", r.stderr
        assert ov.read_bytes() == ref
    # --devices a,b,c: the input's batches dealt round robin to one context per listed device (each uploads its own batches), every worker plans its own
    # text (rfq_scan_batch), encodes its whole chunks (flush_all) and hands the rest to the next; written in order — the image must be the one-shot image whatever the split
    om = tmp_path / "multi.rfq"
    for src, ref in ((p, og.read_bytes()), (pc, O.encode_file(crlf, b"", O.SE, 100_000)), (ps, want), (p, og.read_bytes())):
        r = _run(binary, ["-c", "-i", str(src), "-o", str(om), "-k", "100", "--batch_mb", str(batch_mb), "--devices", "0,0,0"])
        assert r.returncode == 0, r.stderr
        assert om.read_bytes() == ref
    r = _run(binary, ["-c", "-i", str(pa), "-I", str(pb), "-o", str(om), "-k", "100", "--batch_mb", str(batch_mb), "--devices", "0,0"])
    assert r.returncode == 0, r.stderr
    assert om.read_bytes() == pe.read_bytes()
    r = _run(binary, ["-c", "--stdin", "--stdout", "-k", "100", "--batch_mb", str(4 * batch_mb), "--devices=0,0,0,0"], input=fq1)
    assert r.returncode == 0 and r.stdout == og.read_bytes(), r.stderr
    # -d --devices a,b,c: the image's chunk headers walked on the host as the blocks go by, ranges of whole chunks pulled by one worker per device
    # (each uploads its own range), ordered writer - the text must be the one-device text; an image without a final line break too
    for src, split, ref in ((og, False, fq1), (pe, True, None)):
        o1 = tmp_path / "multi_back_1.fq"; o2 = tmp_path / "multi_back_2.fq"
        r = _run(binary, ["-d", "-i", str(src), "-o", str(o1)] + (["-O", str(o2)] if split else []) + ["--batch_mb", str(batch_mb), "--devices", "0,0,0"])
        assert r.returncode == 0, r.stderr
        if split:
            assert o1.read_bytes() == pa.read_bytes() and o2.read_bytes() == pb.read_bytes()
        else:
            assert o1.read_bytes() == ref
    # bytes behind the chain's end (an mReads == 0 record, then anything: the reference stops reading there, src/repaq.cpp:287-292) and an implausible chunk
    # header in mid-image: the host's walk cannot index those - the multi-device path then hands the rest to the device's own walk, as the one-device path
    # does, and writes the same text / gives the same error (it used to exit with "cannot index" after it had opened its outputs)
    padded = tmp_path / "padded.rfq"; padded.write_bytes(og.read_bytes() + bytes(64))
    texts = []
    for extra in ([], ["--devices", "0,0"]):
        o1 = tmp_path / "padded_back.fq"
        r = _run(binary, ["-d", "-i", str(padded), "-o", str(o1), "--batch_mb", str(batch_mb)] + extra)
        texts.append((r.returncode, o1.read_bytes() if r.returncode == 0 else r.stderr))
    assert texts[0] == texts[1] and texts[0] == (0, fq1), texts[0][0]
    # a file of exactly 1 MiB without a final line break: only the chunk that holds the last record carries the line-break bit (ADVICE r1;
    # the driver computes the threshold itself, one-shot and under --devices)
    from cases import CASES
    em = CASES["se_exact_mib_no_final_newline"]["fq1"]; pm = tmp_path / "mib.fq"; pm.write_bytes(em)
    for extra in ([], ["--devices", "0,0,0"]):
        r = _run(binary, ["-c", "-i", str(pm), "-o", str(om), "-k", "100", "--batch_mb", str(batch_mb)] + extra)
        assert r.returncode == 0, r.stderr
        assert om.read_bytes() == O.encode_file(em, b"", O.SE, 100_000)
    # the input stops at an empty line (two blank lines behind the last record, which ends shortly before the reader's last 1 MiB block; no final line
    # break): the tail chunk's line-break bit follows from how far the readers' last attempt got - one-shot and under --devices
    lines = fq1.split(b"\n"); recs = [lines[4 * i: 4 * i + 4] for i in range(len(lines) // 4)]
    cut = (1 << 20) - 37; acc = bytearray(); i = 0
    while len(acc) + 2 * 360 <= cut:                                             # (a record of this profile is < 360 bytes)
        acc += b"\n".join(recs[i % len(recs)]) + b"\n"; i += 1
    pad = cut - len(acc) - sum(len(x) + 1 for x in recs[i % len(recs)])          # the next record, lengthened to end exactly at `cut`
    nm, sq, st_, ql = recs[i % len(recs)]
    acc += nm + b"p" * (pad & 1) + b"\n" + sq + b"A" * (pad >> 1) + b"\n" + st_ + b"\n" + ql + b"F" * (pad >> 1) + b"\n"
    assert len(acc) == cut
    tail_case = bytes(acc) + b"\n\n" + b"\n".join(recs[0])                      # two blank lines, then an unterminated record in the last block
    pt = tmp_path / "stops.fq"; pt.write_bytes(tail_case)
    want_t = O.encode_file(tail_case, b"", O.SE, 100_000)
    for extra in ([], ["--devices", "0,0,0"]):
        r = _run(binary, ["-c", "-i", str(pt), "-o", str(om), "-k", "100", "--batch_mb", str(batch_mb)] + extra)
        assert r.returncode == 0, r.stderr
        assert om.read_bytes() == want_t, extra
    # --devices with batches smaller than a chunk and a carry larger than the room a batch leaves for it: names of ~210 bytes on reads of 3 bases make a
    # chunk of 100 k bases ~7.4 MB of text - several batches without a whole chunk hand their text on, the buffer of the one that meets the chunk's end grows
    import random
    rnd = random.Random(77)
    longn = b"".join(b"@" + (b"N%05d_" % i) + b"x" * 200 + b"\n" + bytes(rnd.choice(b"ACGT") for _ in range(3)) + b"\n+\n" + b"F:," + b"\n" for i in range(80000))
    pl = tmp_path / "longnames.fq"; pl.write_bytes(longn)
    want_l = O.encode_file(longn, b"", O.SE, 100_000)
    for extra in (["--devices", "0,0,0"], ["--devices", "0,0"]):
        r = _run(binary, ["-c", "-i", str(pl), "-o", str(om), "-k", "100", "--batch_mb", "1"] + extra)
        assert r.returncode == 0, r.stderr
        assert om.read_bytes() == want_l, extra
    # --devices, the reader stops at an empty line early in the input (src/fastqreader.cpp:180-191): the batch that meets it ends the image, the batches
    # behind it - already uploaded to their devices - contribute nothing; and two files of different length (the pair reader stops with the shorter one)
    early = fq1[: len(fq1) // 5]; early = early[: early.rfind(b"\n@") + 1] + b"\n\n" + fq1[len(fq1) // 5:]
    pe_ = tmp_path / "early_stop.fq"; pe_.write_bytes(early)
    want_e = O.encode_file(early, b"", O.SE, 100_000)
    assert 0 < len(want_e) < len(og.read_bytes()) // 2
    for extra in ([], ["--devices", "0,0,0"]):
        r = _run(binary, ["-c", "-i", str(pe_), "-o", str(om), "-k", "100", "--batch_mb", "1"] + extra)
        assert r.returncode == 0, r.stderr
        assert om.read_bytes() == want_e, extra
    fa, fb = pa.read_bytes(), pb.read_bytes(); cut_b = fb[: fb.rfind(b"\n@", 0, (2 * len(fb)) // 3) + 1]
    pbs = tmp_path / "r2_short.fq"; pbs.write_bytes(cut_b)
    want_s = O.encode_file(fa, cut_b, O.PE_TWO_FILES, 100_000)
    for extra in ([], ["--devices", "0,0,0"]):
        r = _run(binary, ["-c", "-i", str(pa), "-I", str(pbs), "-o", str(om), "-k", "100", "--batch_mb", "1"] + extra)
        assert r.returncode == 0, r.stderr
        assert om.read_bytes() == want_s, extra
    # the other way round and by a lot: R1 a tenth of R2.  The input ends with R1 (FastqReaderPair::read, src/fastqreader.cpp:287-299); the batch that holds the last of
    # R1 finds that out by a final plan and ends the image there - the batches behind it (R2 text only, no whole chunk in any of them) used to be read, uploaded and
    # carried from batch to batch in full: the trace shows how many batches became resident
    cut_a = fa[: fa.rfind(b"\n@", 0, len(fa) // 10) + 1]
    pas = tmp_path / "r1_short.fq"; pas.write_bytes(cut_a)
    want_a = O.encode_file(cut_a, fb, O.PE_TWO_FILES, 100_000)
    for extra in ([], ["--devices", "0,0,0"]):
        r = _run(binary, ["-c", "-i", str(pas), "-I", str(pb), "-o", str(om), "-k", "100", "--batch_mb", "1"] + extra + ["--trace"])
        assert r.returncode == 0, r.stderr
        assert om.read_bytes() == want_a, extra
        # (one-device: the batch that uses R1 up ends the input; --devices: the ingestion runs up to 2 D + 1 batches ahead of the workers before one of them says stop)
        assert r.stderr.count(b"compress: batch resident") <= 2 + (len(cut_a) >> 20) + (7 if extra else 0), (extra, r.stderr.count(b"compress: batch resident"), len(fb) >> 20)
    # an empty input leaves an empty .rfq, which decodes to an empty FASTQ (RfqHeader defaults, src/rfqheader.cpp:7-17)
    pz = tmp_path / "empty.fq"; pz.write_bytes(b""); oz = tmp_path / "empty.rfq"; bz = tmp_path / "empty_back.fq"
    assert _run(binary, ["-c", "-i", str(pz), "-o", str(oz)]).returncode == 0 and oz.read_bytes() == b""
    assert _run(binary, ["-d", "-i", str(oz), "-o", str(bz)]).returncode == 0 and bz.read_bytes() == b""
    # a corrupted base makes --compare fail with the reference's message shape
    bad = bytearray(a); k = bad.index(b"\n") + 5; bad[k] = ord("A") if bad[k] != ord("A") else ord("C")
    pbad = tmp_path / "r1_bad.fq"; pbad.write_bytes(bytes(bad))
    r = _run(binary, ["-p", "-i", str(pbad), "-I", str(pb), "-r", str(pe)])
    j = json.loads(r.stdout)
    assert j["result"] == "failed" and "different sequence in the 0 pair" in j["msg"]     # comparePE words rfqReads / 2 (src/repaq.cpp:195)
    # flag validation texts (src/options.cpp:36-111)
    r = _run(binary, ["-c", "-i", str(p)])
    assert r.returncode == 255 and b"Please specify output file by <out1>" in r.stderr
    r = _run(binary, ["-c", "-d", "-i", str(p), "-o", str(out)])
    assert r.returncode == 255 and b"you can only choose any one mode" in r.stderr


def _compare_suite(binary, tmp_path, batch_mb):
    """--compare (src/repaq.cpp:36-259): batches equal on the device move only counters; a differing batch is cut into records to word
    the reference's message.  Every scenario's JSON is checked against the compiled reference binary when it is present."""
    import gzip
    fq, _ = O.gen(O.NOVA_SE150, 6000, seed=77, nonl=1)
    recs = fq.split(b"\n"); assert len(recs) == 24000
    def edit(read, field, fn):
        r = list(recs); r[read * 4 + field] = fn(r[read * 4 + field]); return b"\n".join(r)
    flip = lambda b: (b"A" if b[:1] != b"A" else b"C") + b[1:]
    a, b = O.gen(O.NOVA_PE150, 2500, seed=78)
    brecs = b.split(b"\n")
    b_bad = list(brecs); b_bad[2100 * 4 + 3] = flip(b_bad[2100 * 4 + 3]); b_bad = b"\n".join(b_bad)
    b_short = b"\n".join(brecs[:2000 * 4]) + b"\n"
    rfq_se = O.encode_file(fq, b"", O.SE, 100_000); rfq_pe = O.encode_file(a, b, O.PE_TWO_FILES, 100_000)
    (tmp_path / "se.rfq").write_bytes(rfq_se); (tmp_path / "pe.rfq").write_bytes(rfq_pe)
    # "\r" as the last byte of the reader's first 1 MiB block, its "\n" opening the next: that "\n" reads as an empty line and ends the input
    crlf = fq.replace(b"\n", b"\r\n"); at = crlf.index(b"\r\n", (1 << 20) - 400)
    lines = crlf.split(b"\r\n"); d = (1 << 20) - 1 - at; k = 0
    while d:                                               # push that "\r" onto the block's last byte by padding a few names
        lines[k * 4] += b"p" * min(d, 100); d -= min(d, 100); k += 1
    split = b"\r\n".join(lines)
    assert split[(1 << 20) - 1:(1 << 20) + 1] == b"\r\n"
    rfq_split = O.encode_file(split, b"", O.SE, 100_000); n_split = len(O.decode_file(rfq_split, False).split(b"\n")) // 4
    assert 0 < n_split < 6000
    (tmp_path / "split.rfq").write_bytes(rfq_split)
    scen = {                                           # name -> (fq1, fq2 | None, rfq file, passed, message fragment)
        "same": (fq, None, "se.rfq", True, ""),
        "crlf": (fq.replace(b"\n", b"\r\n"), None, "se.rfq", True, ""),
        "with_final_newline": (fq + b"\n", None, "se.rfq", True, ""),
        "name_late": (edit(5000, 0, lambda x: x + b"x"), None, "se.rfq", False, "different name in the 5001 read"),
        "seq_first": (edit(0, 1, flip), None, "se.rfq", False, "different sequence in the 1 read"),
        "strand_mid": (edit(3000, 2, lambda x: x + b"k"), None, "se.rfq", False, "different strand in the 3001 read"),
        "qual_last": (edit(5999, 3, flip), None, "se.rfq", False, "different quality in the 6000 read"),
        "fastq_longer": (fq + b"\n" + b"\n".join(recs[:4]) + b"\n", None, "se.rfq", False, "The FASTQ file has more reads than the RFQ file. The FASTQ file has >= 6001"),
        "fastq_shorter": (b"\n".join(recs[:5990 * 4]) + b"\n", None, "se.rfq", False, "The RFQ file has more reads than the FASTQ file. The RFQ file has >= 5991"),
        "pe_same": (a, b, "pe.rfq", True, ""),
        "pe_r2_quality": (a, b_bad, "pe.rfq", False, "different quality in the 2101 pair"),
        "pe_r2_shorter": (a, b_short, "pe.rfq", False, "The RFQ file has more reads than the FASTQ file. The RFQ file has >= 2000 pairs, while the FASTQ file only has 2000 pairs"),
        "gz": (gzip.compress(fq, 1), None, "se.rfq", True, ""),
        "pe_r1_longer_only": (a + b"\n".join(a.split(b"\n")[:4]) + b"\n", b, "pe.rfq", True, ""),      # FastqReaderPair::read needs both mates
        "pe_both_longer": (a + b"\n".join(a.split(b"\n")[:4]) + b"\n", b + b"\n".join(brecs[:4]) + b"\n", "pe.rfq", False, "The FASTQ file has >= 2500 pairs, while the RFQ file only has 2500 pairs"),
        "one_blank_line_is_skipped": (b"\n".join(recs[:3000 * 4]) + b"\n\n" + b"\n".join(recs[3000 * 4:]), None, "se.rfq", True, ""),   # getLine swallows one "\n" after a terminator
        "two_blank_lines_stop_reader": (b"\n".join(recs[:3000 * 4]) + b"\n\n\n" + b"\n".join(recs[3000 * 4:]), None, "se.rfq", False, "The RFQ file has >= 3001 reads, while the FASTQ file only has 3000 reads"),
        "crlf_cut_by_reader_block": (split, None, "split.rfq", True, ""),
        "truncated_last_quality": (fq[:-7], None, "se.rfq", False, "different quality in the 6000 read"),
    }
    for name, (f1, f2, rfq, passed, frag) in scen.items():
        p1 = tmp_path / ("c1.fq.gz" if name == "gz" else "c1.fq"); p1.write_bytes(f1)
        args = ["-p", "-i", str(p1), "-r", str(tmp_path / rfq), "--batch_mb", str(batch_mb)]
        if f2 is not None:
            p2 = tmp_path / "c2.fq"; p2.write_bytes(f2); args += ["-I", str(p2)]
        r = _run(binary, args)
        assert r.returncode == 0, (name, r.stderr)
        j = json.loads(r.stdout)
        assert (j["result"] == "passed") == passed and frag in j["msg"], (name, j)
        if passed:
            n = n_split if rfq == "split.rfq" else 6000 if f2 is None else 5000
            assert j["fastq_reads"] == j["rfq_reads"] == n and j["fastq_bases"] == j["rfq_bases"] == n * 150, (name, j)
        if O.have_ref():
            ref = subprocess.run([O.REF_BIN] + [x for x in args if x not in ("--batch_mb", str(batch_mb))], capture_output=True)
            assert json.loads(ref.stdout) == j, (name, ref.stdout, r.stdout)



def _io_pipeline_suite(binary, tmp_path, reads):
    """The driver's parallel I/O: a regular file read by several pread threads in 1 MiB staging blocks (block i handed out before i + 1),
    device batches of several blocks with carry-over, the output written by several pwrite threads at each piece's offset, an output
    path that already holds a longer file (truncated), an exact-multiple-of-the-block file size, and a sequential source (stdin) beside it."""
    fq1, _ = O.gen(O.NOVA_SE150, reads, seed=43)
    want = O.encode_file(fq1, b"", O.SE, 100_000)
    p = tmp_path / "io.fq"; p.write_bytes(fq1)
    out = tmp_path / "io.rfq"; out.write_bytes(b"x" * (len(want) + 12345))          # stale, longer content
    for extra in (["--batch_mb", "2", "--block_mb", "1", "--io_threads", "8", "--write_threads", "3"],
                  ["--batch_mb", "1", "--block_mb", "1", "--io_threads", "2"], ["--batch_mb", "64", "--io_threads", "5"]):
        r = _run(binary, ["-c", "-i", str(p), "-o", str(out), "-k", "100"] + extra)
        assert r.returncode == 0, r.stderr
        assert out.read_bytes() == want, extra
    back = tmp_path / "io_back.fq"; back.write_bytes(b"y" * (len(fq1) + 777))
    r = _run(binary, ["-d", "-i", str(out), "-o", str(back), "--batch_mb", "8", "--block_mb", "1", "--io_threads", "4", "--write_threads", "4"])
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == fq1
    # a file whose size is an exact multiple of the staging block (the last block is full; no line break at the end: src/fastqreader.cpp:31-46)
    n = (len(fq1) >> 20) << 20
    cut = fq1[:n]
    if cut and cut[-1:] != b"\n":
        pm = tmp_path / "mib.fq"; pm.write_bytes(cut)
        r = _run(binary, ["-c", "-i", str(pm), "-o", str(out), "-k", "100", "--batch_mb", "2", "--block_mb", "1", "--io_threads", "8"])
        assert r.returncode == 0, r.stderr
        assert out.read_bytes() == O.encode_file(cut, b"", O.SE, 100_000)
    r = _run(binary, ["-c", "--stdin", "-o", str(out), "-k", "100", "--batch_mb", "2", "--block_mb", "1"], input=fq1)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == want
    r = _run(binary, ["-p", "-i", str(p), "-r", str(out), "--batch_mb", "2", "--block_mb", "1", "--io_threads", "8"])
    assert r.returncode == 0 and json.loads(r.stdout)["result"] == "passed", r.stdout


def _bug_compat_suite(binary, tmp_path, names, batch_mb):
    """`-d --bug_compat` writes what the reference binary writes (tests/golden/compat.json: sizes and md5 of ITS decode output, made by make_golden_compat.py):
    the chunk behind a non-last NO_LINE_BREAK chunk lost, the flagged chunk's R2 text too when it is the R1 bit; plain `-d` keeps every read."""
    import hashlib
    import json
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "compat.json")))
    for name in names:
        g = G[name]
        fq1, fq2 = O.gen(g["profile"], g["reads"], seed=g["seed"], **g["kw"])
        rfq = O.encode_file(fq1, fq2, g["paired"], 100_000)
        assert hashlib.md5(rfq).hexdigest() == g["rfq_md5"]
        p = tmp_path / (name + ".rfq"); p.write_bytes(rfq)
        split = g.get("split", g["paired"] != O.SE)
        o1 = tmp_path / (name + "_1.fq"); o2 = tmp_path / (name + "_2.fq")
        outs = ["-o", str(o1)] + (["-O", str(o2)] if split else [])
        r = _run(binary, ["-d", "-i", str(p)] + outs + ["--bug_compat", "--batch_mb", str(batch_mb)])
        assert r.returncode == 0, r.stderr
        texts = [o1.read_bytes()] + ([o2.read_bytes()] if split else [])
        assert [len(t) for t in texts] == g["ref_decode_len"] and [hashlib.md5(t).hexdigest() for t in texts] == g["ref_decode_md5"], name
        r = _run(binary, ["-d", "-i", str(p)] + outs + ["--batch_mb", str(batch_mb)])
        assert r.returncode == 0, r.stderr
        if split or g["paired"] == O.SE:
            assert [o1.read_bytes()] + ([o2.read_bytes()] if split else []) == ([fq1, fq2] if split else [fq1]), name
        else:
            assert hashlib.md5(o1.read_bytes()).hexdigest() == g["ref_decode_md5"][0], name   # (one output: the reference loses nothing there)


def test_cli_bug_compat_on_simt_emulation(tmp_path):
    E.build_emu()
    assert os.path.exists(EMU_BIN)
    _bug_compat_suite(EMU_BIN, tmp_path, ["pe_nonl_r1_small", "se_nonl_small", "pe_nonl_r1_one_output"], batch_mb=1)


@pytest.mark.gpu
def test_cli_bug_compat_on_gpu(tmp_path):
    assert os.path.exists(GPU_BIN), "repaq_hip is built by __graft_entry__.build()"
    for mb in (256, 2):                                                         # one call per image; a streaming caller's slices
        _bug_compat_suite(GPU_BIN, tmp_path, ["se_nonl", "se_nonl_small", "pe_nonl_r1_one_output", "pe_nonl_both_one_output", "pe_nonl_r2", "pe_nonl_r1", "pe_nonl_both", "pe_nonl_r1_small", "bgi_nonl_both"], batch_mb=mb)


@pytest.mark.skipif(__import__("shutil").which("xz") is None, reason="no external xz here: the .rfq.xz legs inside the CLI suites (src/main.cpp:134-177) did NOT run")
def test_xz_is_present_so_the_rfq_xz_legs_ran():
    """the CLI suites skip their .rfq.xz leg quietly when xz is missing; this test makes that visible as a reported skip"""
    assert subprocess.run(["xz", "--version"], capture_output=True).returncode == 0


@pytest.mark.gpu
@pytest.mark.skipif(__import__("shutil").which("xz") is None, reason="no external xz on the GPU box: the .rfq.xz legs inside the GPU CLI suites did NOT run")
def test_xz_is_present_on_the_gpu_box_so_the_rfq_xz_legs_ran():
    """the same check inside the -m gpu run: a GPU box without xz shows up as a reported skip there too"""
    assert subprocess.run(["xz", "--version"], capture_output=True).returncode == 0


def test_cli_io_pipeline_on_simt_emulation(tmp_path):
    E.build_emu()
    assert os.path.exists(EMU_BIN)
    _io_pipeline_suite(EMU_BIN, tmp_path, reads=9000)


@pytest.mark.gpu
def test_cli_io_pipeline_on_gpu(tmp_path):
    assert os.path.exists(GPU_BIN), "repaq_hip is built by __graft_entry__.build()"
    _io_pipeline_suite(GPU_BIN, tmp_path, reads=120000)


def _block_fuzz_through_cli(binary, tmp_path, seeds):
    """tests/_fuzz.py::block_case inputs through the streaming driver in small batches (thresholds known late, carry-over, reading that
    stops at an empty line mid-stream), the image against the oracle, then a streamed decode against the oracle's decode."""
    import _fuzz as F
    done = 0
    for seed in seeds:
        fq1, fq2, paired, cb = F.block_case(seed)
        try:
            want = O.encode_file(fq1, fq2, paired, cb)
        except O.OracleError:
            continue                                                     # an input both sides refuse
        d = tmp_path / ("s%d" % seed); d.mkdir()
        p1 = d / "a.fq"; p1.write_bytes(fq1); out = d / "o.rfq"
        args = ["-c", "-i", str(p1), "-o", str(out), "-k", str(cb // 1000), "--batch_mb", str(1 + seed % 3), "--block_mb", "1", "--io_threads", str(1 + seed % 4)]
        if paired == O.PE_TWO_FILES:
            p2 = d / "b.fq"; p2.write_bytes(fq2); args += ["-I", str(p2)]
        if paired == O.PE_INTERLEAVED:
            args += ["--interleaved_in"]
        r = _run(binary, args)
        assert r.returncode == 0, (seed, r.stderr)
        assert out.read_bytes() == want, "block seed %d through the driver" % seed
        split = paired != O.SE
        a1, a2 = d / "x1.fq", d / "x2.fq"
        r = _run(binary, ["-d", "-i", str(out), "-o", str(a1)] + (["-O", str(a2)] if split else []) + ["--batch_mb", "8", "--block_mb", "1"])
        assert r.returncode == 0, (seed, r.stderr)
        exp = O.decode_file(want, split)
        if split:
            assert (a1.read_bytes(), a2.read_bytes()) == tuple(exp), seed
        else:
            assert a1.read_bytes() == (exp if isinstance(exp, bytes) else exp[0]), seed
        done += 1
    assert done >= len(seeds) // 2


def test_cli_block_fuzz_on_simt_emulation(tmp_path):
    E.build_emu()
    _block_fuzz_through_cli(EMU_BIN, tmp_path, range(8))


@pytest.mark.gpu
def test_cli_block_fuzz_on_gpu(tmp_path):
    assert os.path.exists(GPU_BIN), "repaq_hip is built by __graft_entry__.build()"
    _block_fuzz_through_cli(GPU_BIN, tmp_path, range(40))

def test_cli_on_simt_emulation(tmp_path):
    E.build_emu()
    subprocess.check_call(["make", "-s", "-C", E.EMU_DIR, "all"])
    _suite(EMU_BIN, tmp_path, reads_se=4000, pairs=1500, batch_mb=1)


def test_cli_compare_on_simt_emulation(tmp_path):
    E.build_emu()
    subprocess.check_call(["make", "-s", "-C", E.EMU_DIR, "all"])
    _compare_suite(EMU_BIN, tmp_path, batch_mb=1)


@pytest.mark.gpu
def test_cli_on_gpu(tmp_path):
    import __graft_entry__ as g
    g.build_host_tools()
    assert os.path.exists(GPU_BIN)
    _suite(GPU_BIN, tmp_path, reads_se=60000, pairs=30000, batch_mb=8)


@pytest.mark.gpu
def test_cli_compare_on_gpu(tmp_path):
    import __graft_entry__ as g
    g.build_host_tools()
    _compare_suite(GPU_BIN, tmp_path, batch_mb=1)


@pytest.mark.gpu
def test_cli_devices_on_two_physical_gpus(tmp_path):
    """--devices 0,1: chunk ranges planned on GPU 0, pulled GPU-to-GPU (rfq_copy_peer = hipMemcpyPeerAsync) and encoded on both; the image is
    the one-shot image.  Needs two GPUs: skipped on the one-GPU test box (the same path runs there as --devices 0,0 in test_cli_on_gpu)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    assert os.path.exists(GPU_BIN)
    a, b = O.gen(O.NOVA_PE150, 60000, seed=45)
    pa, pb = tmp_path / "r1.fq", tmp_path / "r2.fq"; pa.write_bytes(a); pb.write_bytes(b)
    one, two = tmp_path / "one.rfq", tmp_path / "two.rfq"
    assert _run(GPU_BIN, ["-c", "-i", str(pa), "-I", str(pb), "-o", str(one), "-k", "100"]).returncode == 0
    r = _run(GPU_BIN, ["-c", "-i", str(pa), "-I", str(pb), "-o", str(two), "-k", "100", "--batch_mb", "8", "--devices", "0,1"])
    assert r.returncode == 0, r.stderr
    assert two.read_bytes() == one.read_bytes() == O.encode_file(a, b, O.PE_TWO_FILES, 100_000)


def _serve_suite(binary, tmp_path, pairs):
    """repaq_hip --serve: several jobs in ONE process (the HIP runtime and the device's main context stay up), one command line per line of stdin - compress, decompress,
    compress another input under the same context (another header), jobs with several writers; every output equals the one-shot run's / the oracle's."""
    fq1, fq2 = O.gen(O.NOVA_PE150, pairs, seed=51, nonl=2)
    se, _ = O.gen(O.SE_VAR, pairs, seed=52)
    pa, pb, ps = tmp_path / "s_1.fq", tmp_path / "s_2.fq", tmp_path / "s_se.fq"
    pa.write_bytes(fq1); pb.write_bytes(fq2); ps.write_bytes(se)
    o1, o2, b1, b2, bs = tmp_path / "s_pe.rfq", tmp_path / "s_se.rfq", tmp_path / "s_b1.fq", tmp_path / "s_b2.fq", tmp_path / "s_bse.fq"
    jobs = ["-c -i %s -I %s -o %s -k 100 --batch_mb 1" % (pa, pb, o1),
            "# a comment line, and a blank one:", "",
            "repaq_hip -d -i %s -o %s -O %s --batch_mb 1" % (o1, b1, b2),
            '-c -i "%s" -o %s -k 100 --batch_mb 2 --write_threads 4' % (ps, o2),
            "-d -i %s -o %s --write_threads 3" % (o2, bs)]
    r = subprocess.run([binary, "--serve"], input=("\n".join(jobs) + "\n").encode(), capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stderr.count(b"[serve] job ") == 4, r.stderr
    assert o1.read_bytes() == O.encode_file(fq1, fq2, O.PE_TWO_FILES, 100_000)
    assert (b1.read_bytes(), b2.read_bytes()) == (fq1, fq2)
    assert o2.read_bytes() == O.encode_file(se, b"", O.SE, 100_000) and bs.read_bytes() == se
    # a job that cannot run ends the server with the reference's text and status
    r = subprocess.run([binary, "--serve"], input=b"-c -i /nonexistent/x.fq -o /tmp/x.rfq\n", capture_output=True)
    assert r.returncode == 255 and b"ERROR:" in r.stderr
    r = subprocess.run([binary, "--serve"], input=b"-d --stdin -o /tmp/x.fq\n", capture_output=True)
    assert r.returncode == 255 and b"--serve" in r.stderr


def test_cli_serve_on_simt_emulation(tmp_path):
    E.build_emu()
    _serve_suite(EMU_BIN, tmp_path, 260)


@pytest.mark.gpu
def test_cli_serve_on_gpu(tmp_path):
    _serve_suite(GPU_BIN, tmp_path, 40000)
