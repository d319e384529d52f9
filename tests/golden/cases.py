"""Hand-built edge-case FASTQ inputs shared by the golden maker (make_golden.py), the oracle-vs-reference
differential test and the GPU parity tests.  Each case: name -> dict(fq1, fq2, paired, k).
Everything is synthesised here; no reference data is involved."""
import random

SE, PE2, PEI = 0, 1, 2


def rec(name, seq, qual, strand="+"):
    return (name, seq, strand, qual)


def fastq(records, eol="\n", final_eol=True):
    out = []
    for (n, s, st, q) in records:
        out += [n, s, st, q]
    txt = eol.join(out) + (eol if final_eol else "")
    return txt.encode("latin-1")


def illumina(i, mate=1, lane=1, tile=1101, x=None, y=None, idx="ACGT"):
    x = 1000 + 7 * i if x is None else x
    y = 2000 + (i // 50) if y is None else y
    return "@A00250:26:H3YTWDSXX:%d:%d:%d:%d %d:N:0:%s" % (lane, tile, x, y, mate, idx)


def rseq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def rqual(rng, n, vals="FFFFFFFFFFFFFFFFFFFF::,"):
    return "".join(rng.choice(vals) for _ in range(n))


def comp(s):
    m = {"A": "T", "T": "A", "C": "G", "G": "C"}
    return "".join(m.get(c, "N") for c in reversed(s))


def build_cases():
    C = {}
    rng = random.Random(12345)

    # --- SURVEY.md Appendix D.5 / D.6 ---
    k1 = [rec("@A00250:26:H3YTWDSXX:1:1101:1000:2000 1:N:0:ACGT", "ACGTNACGTACGTACGTACGTAAAA", "FFFF#FFF:FFFFFFFFFF,FFFFF"),
          rec("@A00250:26:H3YTWDSXX:1:1101:1032:2000 1:N:0:ACGT", "GGGGGGGGTTTTTTTTCCCCCCCCA", "FFFFFFFFFFFFFFFFFFFFFFFF:")]
    k2 = [rec("@A00250:26:H3YTWDSXX:1:1101:1000:2000 2:N:0:ACGT", "CCCCTTTTACGTACGTACGTACGTN", ",FFFFFFFFFFFFFFFFFFFFFFF#"),
          rec("@A00250:26:H3YTWDSXX:1:1101:1032:2000 2:N:0:ACGT", "ACACACACACACACACACACACACA", "FFFFFFFFFFFFFFFFFFFFFFFFF")]
    C["d5_tiny_se"] = dict(fq1=fastq(k1), paired=SE)
    C["d6_tiny_pe"] = dict(fq1=fastq(k1), fq2=fastq(k2), paired=PE2)
    inter = [k1[0], k2[0], k1[1], k2[1]]
    C["d6_tiny_pe_interleaved_in"] = dict(fq1=fastq(inter), paired=PEI)

    # --- line endings / trailing newline / truncation quirks (src/fastqreader.cpp:94-196) ---
    base = [rec(illumina(i), rseq(rng, 40), rqual(rng, 40)) for i in range(30)]
    C["se_crlf"] = dict(fq1=fastq(base, eol="\r\n"), paired=SE)
    C["se_cr_only"] = dict(fq1=fastq(base, eol="\r"), paired=SE)
    C["se_no_final_newline"] = dict(fq1=fastq(base, final_eol=False), paired=SE)
    C["se_crlf_no_final"] = dict(fq1=fastq(base, eol="\r\n", final_eol=False), paired=SE)
    blank = fastq(base[:10]) + b"\n" + fastq(base[10:])
    C["se_blank_line_after_record"] = dict(fq1=blank, paired=SE)       # "\n\n": one blank line is swallowed
    blank2 = fastq(base[:10]) + b"\n\n" + fastq(base[10:])
    C["se_two_blank_lines_truncate"] = dict(fq1=blank2, paired=SE)      # reader stops at the empty name line
    # more reader quirks: pairs, mixed line ends, 1 MiB block edges (`end < mBufDataLen - 1`, src/fastqreader.cpp:94-156)
    qr = random.Random(4242)                # own generator: the cases below keep their bytes
    C["pe_crlf"] = dict(fq1=fastq(base, eol="\r\n"), fq2=fastq([rec(illumina(i, mate=2), rseq(qr, 40), rqual(qr, 40)) for i in range(30)], eol="\r\n"), paired=PE2)
    r2q = [rec(illumina(i, mate=2), rseq(qr, 40), rqual(qr, 40)) for i in range(30)]
    C["pe_crlf_r2_only"] = dict(fq1=fastq(base), fq2=fastq(r2q, eol="\r\n"), paired=PE2)
    C["il_crlf"] = dict(fq1=fastq([r for ab in zip(base, r2q) for r in ab], eol="\r\n"), paired=PEI)
    C["pe_three_newlines_in_r2"] = dict(fq1=fastq(base), fq2=fastq(r2q[:17]) + b"\n\n" + fastq(r2q[17:]), paired=PE2)     # pairs stop at the empty line of R2
    mixed = b"".join(l.encode("latin-1") + qr.choice([b"\n", b"\r\n", b"\r", b"\n", b"\r\n"]) for r in base for l in r)
    C["se_mixed_line_ends"] = dict(fq1=mixed, paired=SE)
    C["se_leading_newline"] = dict(fq1=b"\n" + fastq(base), paired=SE)                                                   # empty name line at offset 0: nothing is read
    C["se_blank_line_inside_record"] = dict(fq1=fastq(base[:5]) + base[5][0].encode() + b"\n\n" + base[5][1].encode() + b"\n+\n" + base[5][3].encode() + b"\n" + fastq(base[6:]), paired=SE)
    bigrng = random.Random(99)
    bigrecs = [rec(illumina(i, tile=1101 + i // 4000), rseq(bigrng, 40), rqual(bigrng, 40)) for i in range(9000)]

    def shifted(eol, want_pos, term):
        """CRLF/LF text of bigrecs whose first `term` byte at or after `want_pos` is moved exactly onto `want_pos` by padding read 0's name."""
        txt = fastq(bigrecs, eol=eol)
        p0 = txt.index(term, want_pos - 120)
        pad = want_pos - p0
        while pad < 0:
            p0 = txt.index(term, p0 - 200 if p0 > 200 else 0); pad = want_pos - p0    # (not reached: lines are < 120 bytes)
        recs = [rec(bigrecs[0][0] + "P" * pad, bigrecs[0][1], bigrecs[0][3])] + bigrecs[1:]
        out = fastq(recs, eol=eol)
        assert out[want_pos:want_pos + len(term)] == term, (out[want_pos - 2:want_pos + 3], pad)
        return out
    MiB = 1 << 20
    C["se_crlf_cr_is_last_byte_of_block"] = dict(fq1=shifted("\r\n", MiB - 1, b"\r\n"), paired=SE, k=100)     # '\n' opens the next block: empty line, reader stops
    C["se_crlf_cr_is_second_last_byte_of_block"] = dict(fq1=shifted("\r\n", MiB - 2, b"\r\n"), paired=SE, k=100)   # '\n' is the block's last byte: not swallowed
    C["se_crlf_cr_is_third_last_byte_of_block"] = dict(fq1=shifted("\r\n", MiB - 3, b"\r\n"), paired=SE, k=100)   # ordinary "\r\n": whole file, 4 chunks
    lf = shifted("\n", MiB - 1, b"\n")
    C["se_blank_line_at_block_edge"] = dict(fq1=lf[:MiB] + b"\n" + lf[MiB:], paired=SE, k=100)               # "\n\n" split by the block edge: the second one is an empty line
    lf2 = fastq(bigrecs)
    cut = lf2.index(b"\n@", 600000) + 1
    C["se_three_newlines_mid_file"] = dict(fq1=lf2[:cut] + b"\n\n" + lf2[cut:], paired=SE, k=100)            # chunks before the empty line + a truncated tail chunk
    # a file of exactly 1 MiB without a final line break: the reader's last full block does not raise the flag, the empty
    # read that follows the unterminated last line does (src/fastqreader.cpp:31-46) - only the chunk holding the last record carries the bit
    def exact_mib(recs, mate_fix=None):
        m = len(recs)
        while len(fastq(recs[:m], final_eol=False)) > MiB:
            m -= 1
        need = MiB - len(fastq(recs[:m], final_eol=False))
        out = fastq([rec(recs[0][0] + "P" * need, recs[0][1], recs[0][3])] + recs[1:m], final_eol=False)
        assert len(out) == MiB and out[-1:] != b"\n"
        return out, m
    em, em_n = exact_mib(bigrecs)
    C["se_exact_mib_no_final_newline"] = dict(fq1=em, paired=SE, k=100)
    r2big = [rec(illumina(i, mate=2, tile=1101 + i // 4000), rseq(bigrng, 40), rqual(bigrng, 40)) for i in range(em_n)]
    C["pe_exact_mib_r1_no_final_newline"] = dict(fq1=em, fq2=fastq(r2big), paired=PE2, k=100)
    # a TRUNCATED last record and no final line break (an interrupted copy): the dropped record's bytes are still read - the reader has its final
    # block loaded (or has met the empty read of an exact-MiB file) before the tail chunk is written, so the tail chunk carries the bit although its
    # own last record ended with a line break long before.  Found by the driver's I/O test in round 2; the reference binary fixes the expectation.
    def truncated(recs, size):
        full = fastq(recs)
        i = size                                                                          # the last record whose name and 17 bases fit: < one record of padding
        while True:
            i = full.rfind(b"\n@", 0, i) + 1; j = full.index(b"\n", i) + 1               # a record start, its sequence line
            if j + 17 <= size: break
            i -= 1
        body = full[:j + 17]                                                              # 17 bases of that sequence line, no line break
        pad = size - len(body); assert 0 <= pad < 200
        k = full.index(b"\n")
        out = full[:k] + b"P" * pad + body[k:]
        assert len(out) == size and out[-1:] != b"\n"
        return out
    C["se_truncated_record_alone_in_last_block"] = dict(fq1=truncated(bigrecs, MiB + 30), paired=SE, k=100)
    C["se_truncated_record_exact_mib"] = dict(fq1=truncated(bigrecs, MiB), paired=SE, k=100)
    r2long = r2big + [rec(illumina(i, mate=2, tile=1101 + i // 4000), rseq(bigrng, 40), rqual(bigrng, 40)) for i in range(em_n, em_n + 400)]
    C["pe_truncated_r2_into_last_block"] = dict(fq1=fastq(bigrecs[:em_n]), fq2=truncated(r2long, MiB + 40), paired=PE2, k=100)
    C["pe_truncated_r1_exact_mib"] = dict(fq1=truncated(bigrecs, MiB), fq2=fastq(r2big), paired=PE2, k=100)
    # exactly 1 MiB WITH its final line break: the flag still goes up - at the empty read behind the last full block the reader tests the byte
    # in front of its buffer (src/fastqreader.cpp:41-45) - so the tail chunk carries the bit (and the reference's own decode drops the last line break)
    def exact_mib_nl(recs):
        m = len(recs)
        while len(fastq(recs[:m])) > MiB:
            m -= 1
        need = MiB - len(fastq(recs[:m]))
        out = fastq([rec(recs[0][0] + "P" * need, recs[0][1], recs[0][3])] + recs[1:m])
        assert len(out) == MiB and out[-1:] == b"\n"
        return out
    C["se_exact_mib_with_final_newline"] = dict(fq1=exact_mib_nl(bigrecs), paired=SE, k=100)
    # "\r\n" laid across R2's first block edge is an empty line: reading stops there (src/fastqreader.cpp:112-114,180-191) - but FastqReaderPair::read
    # has already taken R1's record of that pair, and that record ends inside R1's final block: the tail chunk carries R1's bit
    def crlf_on_edge(recs):
        full = fastq(recs)
        e = full.rfind(b"\n", 0, MiB - 200)                                               # a line end somewhere before the edge ...
        k = full.index(b"\n")
        out = full[:k] + b"P" * (MiB - 1 - e) + full[k:]                                   # ... moved onto the edge's last byte by padding the first name
        assert out[MiB - 1:MiB] == b"\n"
        return out[:MiB - 1] + b"\r\n" + out[MiB:]
    r1edge = fastq(bigrecs[:em_n] + bigrecs[:30], final_eol=False)
    assert MiB < len(r1edge) < MiB + 8000
    C["pe_crlf_across_r2_block_edge_r1_tail_in_last_block"] = dict(fq1=r1edge, fq2=crlf_on_edge(r2long), paired=PE2, k=100)
    C["se_partial_last_record"] = dict(fq1=fastq(base) + b"@partial\nACGT\n", paired=SE)
    C["se_single_read"] = dict(fq1=fastq(base[:1]), paired=SE)
    r2 = [rec(illumina(i, mate=2), rseq(rng, 40), rqual(rng, 40)) for i in range(30)]
    C["pe_r2_no_final_newline"] = dict(fq1=fastq(base), fq2=fastq(r2, final_eol=False), paired=PE2)
    C["pe_r1_no_final_newline"] = dict(fq1=fastq(base, final_eol=False), fq2=fastq(r2), paired=PE2)
    C["pe_r2_shorter_file"] = dict(fq1=fastq(base), fq2=fastq(r2[:20]), paired=PE2)

    # --- name parsing quirks (src/fastqmeta.cpp:22-80) ---
    names = [
        "@A:B:C:4:55:66:77 1:N:0:X", "@A:B:C:D:E rest of it", "@A:B:C:1:2:3:4:5:6 tail", "@A:B:C:1:2:3:4", "@A:B:C:1:2:3",
        "@noColonsAtAll", "@ leading space", "@A:B:C:007:0012:+33:-4 x", "@A:B:C:300:70000:99999999999:4294967299 big",
        "@A:B:C:1:2: 3:\t4 ws", "@A:B:C:::: empty", "@A:B:C:1:2:3:4\ttab 1:N", "@A:B:C:1x:2y:3z:4w q", "@a b:c:d:e:f:g:h",
        "@A:B:C:1:2:3 onlysix:colons", "@A:B:C:12:1101:2047:2048 1:N:0:ACGT+TTGA", "@:::::::", "@A:B:C:1:2:3:9223372036854775808 of",
    ]
    recs = [rec(n, rseq(rng, 30), rqual(rng, 30)) for n in names]
    C["se_name_quirks"] = dict(fq1=fastq(recs), paired=SE)
    # each quirk name alone decides hasLaneTileXY for its own file
    for j, n in enumerate(names):
        C["se_name_quirk_%02d" % j] = dict(fq1=fastq([rec(n, rseq(rng, 20), rqual(rng, 20)) for _ in range(3)]), paired=SE)
    mixed = [rec(illumina(i), rseq(rng, 30), rqual(rng, 30)) for i in range(10)] + [rec("@V300012345L1C001R0010000005/1 desc", rseq(rng, 30), rqual(rng, 30))]
    C["se_mixed_illumina_then_bgi"] = dict(fq1=fastq(mixed), paired=SE)
    C["se_mixed_bgi_then_illumina"] = dict(fq1=fastq(mixed[::-1]), paired=SE)
    longname = "@" + "L" * 300 + ":1:2:3:4:5:6 tail"
    C["se_name_over_255"] = dict(fq1=fastq([rec(longname, rseq(rng, 20), rqual(rng, 20)), rec(illumina(1), rseq(rng, 20), rqual(rng, 20))]), paired=SE)
    strands = [rec(illumina(i), rseq(rng, 25), rqual(rng, 25), strand=("+" if i % 3 else "+" + illumina(i)[1:])) for i in range(12)]
    C["se_strand_varies"] = dict(fq1=fastq(strands), paired=SE)
    C["se_name1_varies"] = dict(fq1=fastq([rec(illumina(i).replace("A00250", "A0025%d" % (i % 3)) if i % 2 else illumina(i).replace("A00250", "AX"), rseq(rng, 25), rqual(rng, 25)) for i in range(12)]), paired=SE)
    C["se_lane_tile_vary"] = dict(fq1=fastq([rec(illumina(i, lane=1 + i % 3, tile=1101 + i % 5), rseq(rng, 25), rqual(rng, 25)) for i in range(40)]), paired=SE)
    C["se_name2_varies"] = dict(fq1=fastq([rec(illumina(i, idx="ACGT" if i % 4 else "TTTTTT"), rseq(rng, 25), rqual(rng, 25)) for i in range(20)]), paired=SE)

    # --- coordinates (src/rfqcodec.cpp:1262-1330) ---
    xs = [1000] * 40 + [1001, 1065, 1130, 1130, 40000, 7, 7, 7, 2097151, 32767, 32768, 32832, 32833] + [5] * 70 + [6]
    C["se_coord_patterns"] = dict(fq1=fastq([rec(illumina(i, x=x, y=(i * 977) % 50000), rseq(rng, 20), rqual(rng, 20)) for i, x in enumerate(xs)]), paired=SE)
    C["se_coord_too_large"] = dict(fq1=fastq([rec(illumina(0, x=2097152), rseq(rng, 20), rqual(rng, 20))]), paired=SE)

    # --- read lengths ---
    C["se_varlen"] = dict(fq1=fastq([rec(illumina(i), rseq(rng, 5 + (i * 37) % 120), "F" * (5 + (i * 37) % 120)) for i in range(60)]), paired=SE)
    C["se_len_over_255"] = dict(fq1=fastq([rec(illumina(i), rseq(rng, 200 + 100 * (i % 3)), rqual(rng, 200 + 100 * (i % 3))) for i in range(9)]), paired=SE)
    C["se_len_over_65535"] = dict(fq1=fastq([rec(illumina(0), rseq(rng, 66000), rqual(rng, 66000)), rec(illumina(1), rseq(rng, 100), rqual(rng, 100))]), paired=SE)

    # --- bases / N rules (src/rfqheader.cpp:130-237) ---
    def with_n(s, q, positions, nq="#"):
        s = list(s); q = list(q)
        for p in positions:
            s[p] = "N"; q[p] = nq
        return "".join(s), "".join(q)
    many_n = []
    for i in range(60):
        s, q = with_n(rseq(rng, 50), rqual(rng, 50), [3, 17, 40])
        many_n.append(rec(illumina(i), s, q))
    C["se_implied_n_path"] = dict(fq1=fastq(many_n), paired=SE)                 # >=100 N, all '#': mNBaseQual stays '#'
    mn2 = list(many_n); s, q = with_n(rseq(rng, 50), rqual(rng, 50), [5], nq=",")
    mn2[30] = rec(illumina(30), s, q)
    C["se_n_two_quals"] = dict(fq1=fastq(mn2), paired=SE)                        # second N quality -> ENCODE_N_POS
    mn3 = list(many_n); mn3[40] = rec(illumina(40), "ACGT" * 12 + "AC", "F" * 20 + "#" + "F" * 29)
    C["se_nqual_on_non_n_after_first_n"] = dict(fq1=fastq(mn3), paired=SE)
    mn4 = [rec(illumina(0), "ACGT" * 12 + "AC", "#" * 50)] + many_n
    C["se_nqual_on_non_n_before_first_n"] = dict(fq1=fastq(mn4), paired=SE)      # not detected by the reference: stays implied-N
    C["se_all_n_reads"] = dict(fq1=fastq([rec(illumina(i), "N" * 40, "#" * 40) for i in range(5)]), paired=SE)
    C["se_lowercase_error"] = dict(fq1=fastq([rec(illumina(0), "ACGTacgtACGT", "FFFFFFFFFFFF")]), paired=SE)
    C["se_lowercase_g_error"] = dict(fq1=fastq([rec(illumina(0), "ACGTgACGT", "FFFFFFFFF")]), paired=SE)
    C["se_iupac_error"] = dict(fq1=fastq([rec(illumina(0), "ACGTRYACGT", "FFFFFFFFFF")]), paired=SE)

    # --- quality tables ---
    q70 = "".join(chr(33 + i) for i in range(70))
    C["se_70_quality_values"] = dict(fq1=fastq([rec(illumina(i), rseq(rng, 70), q70[i % 7:] + q70[: i % 7]) for i in range(40)]), paired=SE)   # DONT_ENCODE_QUAL
    q63 = "".join(chr(33 + i) for i in range(63))
    C["se_63_quality_values"] = dict(fq1=fastq([rec(illumina(i), rseq(rng, 63), q63[i % 5:] + q63[: i % 5]) for i in range(40)]), paired=SE)
    q64 = "".join(chr(33 + i) for i in range(64))
    C["se_64_quality_values"] = dict(fq1=fastq([rec(illumina(i), rseq(rng, 64), q64) for i in range(10)]), paired=SE)
    C["se_one_quality_value"] = dict(fq1=fastq([rec(illumina(i), rseq(rng, 30), "I" * 30) for i in range(10)]), paired=SE)
    C["se_tie_major"] = dict(fq1=fastq([rec(illumina(i), rseq(rng, 30), "I" * 15 + "5" * 15) for i in range(10)]), paired=SE)
    # position-coder patterns: matches at 0,1,2; runs > 32; gaps > 128 and > 16384
    big = ["F"] * 40000
    for p in [0, 1, 2, 3, 203, 204] + list(range(300, 340)) + [20300] + list(range(25000, 25100)) + [39999]:
        big[p] = ","
    for p in [1, 5, 6, 7, 8, 30000, 30001]:
        big[p] = ":" if big[p] == "F" else big[p]
    bigq = "".join(big)
    C["se_pos_coder_patterns"] = dict(fq1=fastq([rec(illumina(i), rseq(rng, 4000), bigq[i * 4000:(i + 1) * 4000]) for i in range(10)]), paired=SE)
    C["se_qual_starts_with_normal_run"] = dict(fq1=fastq([rec(illumina(0), rseq(rng, 80), "," * 40 + "F" * 40), rec(illumina(1), rseq(rng, 80), "F" * 80), rec(illumina(2), rseq(rng, 80), "F" * 80)]), paired=SE)

    # --- paired-end: overlap / interleave rules (src/rfqcodec.cpp:59-145, 212-287, 371-403, 1391-1438) ---
    def pair(i, ins, rl=60, n_at=None, name2a="1:N:0:ACGT", name2b="2:N:0:ACGT", xb=None):
        frag = rseq(rng, max(ins, 1))
        s1 = (frag[:rl] + rseq(rng, rl))[:rl]
        s2 = (comp(frag)[:rl] + rseq(rng, rl))[:rl]
        q1, q2 = rqual(rng, rl), rqual(rng, rl)
        if n_at is not None:
            s1, q1 = with_n(s1, q1, [n_at])
        n1 = illumina(i).rsplit(" ", 1)[0] + " " + name2a
        n2 = illumina(i, x=xb).rsplit(" ", 1)[0] + " " + name2b
        return rec(n1, s1, q1), rec(n2, s2, q2)
    pe = [pair(i, ins) for i, ins in enumerate([100, 108, 109, 120, 60, 61, 40, 30, 20, 12, 11, 200, 70, 119, 118, 90] * 3)]
    C["pe_overlap_sweep"] = dict(fq1=fastq([a for a, _ in pe]), fq2=fastq([b for _, b in pe]), paired=PE2)
    C["pe_overlap_sweep_interleaved_in"] = dict(fq1=fastq([r for ab in pe for r in ab]), paired=PEI)
    pe_n = [pair(i, 80, n_at=50) for i in range(20)]
    C["pe_overlap_with_n"] = dict(fq1=fastq([a for a, _ in pe_n]), fq2=fastq([b for _, b in pe_n]), paired=PE2)
    pe_bad = [pair(i, 90, name2b="2:Y:1:ACGT") for i in range(10)]
    C["pe_name2_two_diffs"] = dict(fq1=fastq([a for a, _ in pe_bad]), fq2=fastq([b for _, b in pe_bad]), paired=PE2)          # no interleave support
    pe_same = [pair(i, 90, name2b="1:N:0:ACGT") for i in range(10)]
    C["pe_name2_identical"] = dict(fq1=fastq([a for a, _ in pe_same]), fq2=fastq([b for _, b in pe_same]), paired=PE2)        # diff char '\0'
    flip = [pair(i, 90) for i in range(12)]
    flip[7] = pair(7, 90, name2b="2:N:0:TTTT")
    C["pe_interleave_flips_mid_chunk_name2"] = dict(fq1=fastq([a for a, _ in flip]), fq2=fastq([b for _, b in flip]), paired=PE2)   # Q12
    flip2 = [pair(i, 90) for i in range(12)]
    flip2[5] = pair(5, 90, xb=4242)
    C["pe_interleave_flips_mid_chunk_x"] = dict(fq1=fastq([a for a, _ in flip2]), fq2=fastq([b for _, b in flip2]), paired=PE2)
    flip3 = [pair(i, 90, name2a="1:N:0:ACGT" if i % 5 else "1:N:0:GGGG", name2b="2:N:0:ACGT" if i % 5 else "2:N:0:GGGG") for i in range(12)]
    flip3[8] = pair(8, 90, name2b="2:N:0:CCCC")
    C["pe_interleave_flip_with_name2_variation"] = dict(fq1=fastq([a for a, _ in flip3]), fq2=fastq([b for _, b in flip3]), paired=PE2)
    short = []
    for i, (l1, l2) in enumerate([(30, 30), (11, 30), (30, 11), (12, 12), (5, 5), (40, 25), (25, 40)] * 2):
        frag = rseq(rng, 45)
        short.append((rec(illumina(i), frag[:l1], rqual(rng, l1)), rec(illumina(i, mate=2), comp(frag)[:l2], rqual(rng, l2))))
    C["pe_unequal_lengths"] = dict(fq1=fastq([a for a, _ in short]), fq2=fastq([b for _, b in short]), paired=PE2)
    bgi = [(rec("@V300012345L1C001R00100%05d/1" % i, rseq(rng, 50), rqual(rng, 50, "%&'()*+,-./0123")),
            rec("@V300012345L1C001R00100%05d/2" % i, rseq(rng, 50), rqual(rng, 50, "%&'()*+,-./0123"))) for i in range(25)]
    C["pe_bgi_names"] = dict(fq1=fastq([a for a, _ in bgi]), fq2=fastq([b for _, b in bgi]), paired=PE2)
    # long homopolymer pairs: many candidate overlaps, forward-first / smallest-o rule
    homo = [(rec(illumina(i), "A" * 50, "F" * 50), rec(illumina(i, mate=2), "T" * 50, "F" * 50)) for i in range(4)]
    homo += [(rec(illumina(9), "ACGTACGTACGTACGTAAAA", "F" * 20), rec(illumina(9, mate=2), comp("ACGTACGTACGTAAAATTTT"), "F" * 20))]   # D.4: overlap 16
    C["pe_homopolymer_overlap"] = dict(fq1=fastq([a for a, _ in homo]), fq2=fastq([b for _, b in homo]), paired=PE2)
    return C


CASES = build_cases()


def pos_buffers():
    """(buffer, q) inputs for the position-coder known-answer vectors in unit.json (SURVEY.md D.3 first)."""
    rng = random.Random(777001)
    d3 = bytearray(b"F" * 40000)
    for p in [0, 1, 2, 3, 203, 204] + list(range(300, 340)) + [20300]:
        d3[p] = ord(",")
    sets = [(bytes(d3), ord(","))]
    for dens in (2, 10, 50, 300, 5000):
        sets.append((bytes(rng.choice(b",F") if rng.randrange(dens) == 0 else ord("F") for _ in range(30000)), ord(",")))
    sets += [(b",,,,,F", ord(",")), (b"F,,,,,", ord(",")), (b",", ord(",")), (b"F", ord(",")), (b"," * 100, ord(",")),
             (b"F" + b"," * 100, ord(",")), (b"FF" + b"," * 64 + b"F", ord(",")), (b"NNACGTNNNN" * 7, ord("N"))]
    return sets
