"""The GPU Zstandard decoder's record layer restated in Python (test infrastructure only).

flagstat_zstd_kernels.hip decodes a frame in four kernels: `zstd_prepare` (headers, literals, tables), `zstd_chain` (the
serial walk of the FSE states), `zstd_records` (sequence RECORDS, one CHECKPOINT per 64 records) and `zstd_execute` (records
-> output bytes).  This file follows the bit arithmetic of the walk (16-byte windows over a backward stream, packed FSE
entries) and, exactly, the record / checkpoint layout the execution kernel reads: the split of long runs, repeat offsets
resolved per block with an unknown incoming history and replayed afterwards in frame order -- so that the layout and its
corner cases are pinned on the CPU against libzstd (tests/test_zstd_model.py) before the device code runs.
zstd_model.py is the plain RFC 8878 restatement this one is checked against."""
import struct

import zstd_model as zm

REC_LL = 16383
REC_ML = 16383
REC_REP = 1 << 31
REC_FLAG = 1 << 30
MAX_BLOCKS = 256
LANES = 8

OK, BAD_HEADER, BAD_BLOCK, BAD_LITERALS, BAD_HUFFMAN, BAD_SEQUENCES, BAD_BITSTREAM, BAD_OFFSET, BAD_SIZE, NO_TABLE = range(10)
UNSUPPORTED, DICTIONARY, CHECKSUM, TOO_MANY_BLOCKS, TRAILING, TOO_LARGE = 64, 65, 66, 67, 68, 69


class Fail(Exception):
    def __init__(self, code, why=""):
        Exception.__init__(self, "%d %s" % (code, why))
        self.code = code


def layout(max_dst_len):
    rec_stride = ((max_dst_len // 3 + 64) & ~63) + 96 * MAX_BLOCKS
    return {"rec_stride": rec_stride, "ck_stride": rec_stride // 64, "lit_stride": (max_dst_len + 64 + 15) & ~15}


def fse_entry_table(counts, log, xbits):
    """packed entries: symbol | nb << 6 | extra bits << 10 | base << 16"""
    t = zm.build_fse_table(counts, log)
    return [s | (nb << 6) | (xbits(s) << 10) | (base << 16) for s, nb, base in t]


def window16(data, pos):
    """the kernel's window: 16 bytes [byte - 15, byte + 1) of the stream as a 128-bit integer, and the index t of the
    first bit NOT below pos in it (bits [0, t) are unread)"""
    byte = pos >> 3
    lo = byte - 15
    chunk = bytes(data[max(lo, 0):byte + 1])
    if lo < 0:
        chunk = bytes(-lo) + chunk
    chunk = chunk + bytes(16 - len(chunk))
    return int.from_bytes(chunk, "little"), (pos & 7) + 120


def top64(v, t):
    """bits [t - 64, t) of the 128-bit v for t >= 65, else bits [0, t) top-aligned"""
    hi, lo = v >> 64, v & ((1 << 64) - 1)
    if t >= 65:
        return ((hi << (128 - t)) | (lo >> (t - 64))) & ((1 << 64) - 1)
    return (lo << (64 - t)) & ((1 << 64) - 1)


def take(w, n):
    v = (w >> 1) >> (63 - n)
    return v, (w << n) & ((1 << 64) - 1)


class Block:
    pass


def entropy_stage(frame, dst_len):
    """-> dict(nslots, ck, recs, lits, out_len) or raises Fail(code)"""
    frame = bytes(frame)
    n = len(frame)
    if dst_len > (1 << 26) or n >= (1 << 27):
        raise Fail(TOO_LARGE)
    lay = layout(dst_len)
    # ---- frame header
    if n < 6:
        raise Fail(BAD_HEADER)
    magic = struct.unpack_from("<I", frame, 0)[0]
    if magic != zm.MAGIC:
        raise Fail(UNSUPPORTED)
    fhd = frame[4]
    p = 5
    fcs_flag, single, did_flag = fhd >> 6, (fhd >> 5) & 1, fhd & 3
    if fhd & 8:
        raise Fail(BAD_HEADER)
    if fhd & 4:
        raise Fail(CHECKSUM)
    if did_flag:
        raise Fail(DICTIONARY)
    if not single:
        if (frame[p] >> 3) > 21:
            raise Fail(BAD_HEADER, "window above 2^31 bytes")
        p += 1
    fcs_bytes = (1 if single else 0, 2, 4, 8)[fcs_flag]
    if p + fcs_bytes > n:
        raise Fail(BAD_HEADER)
    if fcs_bytes:
        content = int.from_bytes(frame[p:p + fcs_bytes], "little") + (256 if fcs_bytes == 2 else 0)
        if content != dst_len:
            raise Fail(BAD_HEADER)
        p += fcs_bytes
    recs = [0] * lay["rec_stride"]
    ck = [[0, 0, 0] for _ in range(lay["ck_stride"])]
    lits = bytearray(lay["lit_stride"])
    hist = [1, 4, 8]
    carry_huf = None
    carry_tab = [None, None, None]
    rec_top = lit_top = out_top = 0
    nblk = 0
    last = False
    while not last:
        # ---- one pass: up to LANES blocks
        blocks = []
        while len(blocks) < LANES and not last:
            if p + 3 > n:
                raise Fail(BAD_BLOCK)
            bh = int.from_bytes(frame[p:p + 3], "little")
            b = Block()
            last, b.type, b.size = bool(bh & 1), (bh >> 1) & 3, bh >> 3
            b.at = p + 3
            if b.type == 3:
                raise Fail(BAD_BLOCK)
            span = 1 if b.type == 1 else b.size
            if b.at + span > n or (b.type != 1 and b.size > zm.BLOCK_MAX) or (b.type == 1 and b.size > zm.BLOCK_MAX):
                raise Fail(BAD_BLOCK)
            p = b.at + span
            blocks.append(b)
            nblk += 1
            if nblk > MAX_BLOCKS:
                raise Fail(TOO_MANY_BLOCKS)
        # ---- every lane parses its block's section headers
        for b in blocks:
            b.nseq = 0
            b.def_huf = None
            b.defs = [None, None, None]      # None: leaves the table as it is; "repeat"; or (mode, position)
            b.lit_regen = 0
            if b.type != 2:
                continue
            end = b.at + b.size
            if b.size < 2:
                raise Fail(BAD_BLOCK)
            b0 = frame[b.at]
            b.lt, fmt = b0 & 3, (b0 >> 2) & 3
            if b.lt < 2:
                if fmt in (0, 2):
                    size, hdr = b0 >> 3, 1
                elif fmt == 1:
                    size, hdr = int.from_bytes(frame[b.at:b.at + 2], "little") >> 4, 2
                else:
                    size, hdr = int.from_bytes(frame[b.at:b.at + 3], "little") >> 4, 3
                b.lit_regen, b.lit_data = size, b.at + hdr
                b.lit_end = b.lit_data + (size if b.lt == 0 else 1)
                b.streams = 0
            else:
                if fmt < 2:
                    h = int.from_bytes(frame[b.at:b.at + 3], "little")
                    regen, comp, hdr = (h >> 4) & 1023, (h >> 14) & 1023, 3
                elif fmt == 2:
                    h = int.from_bytes(frame[b.at:b.at + 4], "little")
                    regen, comp, hdr = (h >> 4) & 16383, (h >> 18) & 16383, 4
                else:
                    h = int.from_bytes(frame[b.at:b.at + 5], "little")
                    regen, comp, hdr = (h >> 4) & 262143, (h >> 22) & 262143, 5
                b.streams = 1 if fmt == 0 else 4
                b.lit_regen, b.lit_data = regen, b.at + hdr
                b.lit_end = b.lit_data + comp
                if b.lt == 2:
                    b.def_huf = b.lit_data
            if b.lit_end > end or b.lit_regen > zm.BLOCK_MAX:
                raise Fail(BAD_LITERALS)
            q = b.lit_end
            if q >= end:
                raise Fail(BAD_SEQUENCES)
            s0 = frame[q]
            if s0 == 0:
                if q + 1 != end:
                    raise Fail(BAD_SEQUENCES)
                continue
            if s0 < 128:
                b.nseq, q = s0, q + 1
            elif s0 < 255:
                if q + 2 > end:
                    raise Fail(BAD_SEQUENCES)
                b.nseq, q = ((s0 - 128) << 8) + frame[q + 1], q + 2
            else:
                if q + 3 > end:
                    raise Fail(BAD_SEQUENCES)
                b.nseq, q = frame[q + 1] + (frame[q + 2] << 8) + 0x7F00, q + 3
            if q >= end:
                raise Fail(BAD_SEQUENCES)
            if b.nseq > zm.BLOCK_MAX // 3:
                raise Fail(BAD_SEQUENCES)
            modes = frame[q]
            q += 1
            if modes & 3:
                raise Fail(BAD_SEQUENCES)
            for t, (mode, max_sym, max_log) in enumerate((((modes >> 6) & 3, 35, 9), ((modes >> 4) & 3, 31, 8), ((modes >> 2) & 3, 52, 9))):
                if mode == 3:
                    b.defs[t] = "repeat"
                else:
                    b.defs[t] = (mode, q)
                    if mode == 1:
                        if q >= end:
                            raise Fail(BAD_SEQUENCES)
                        q += 1
                    elif mode == 2:
                        try:
                            _, _, q = zm.read_fse_counts(frame[:end], q, max_sym, max_log)
                        except zm.ZstdError:
                            raise Fail(BAD_SEQUENCES)
            b.bits_at, b.end = q, end
        # ---- sources of repeated tables
        for i, b in enumerate(blocks):
            if b.type != 2:
                continue
            if b.lt == 3:
                src = carry_huf
                for j in range(i - 1, -1, -1):
                    if blocks[j].def_huf is not None:
                        src = blocks[j].def_huf
                        break
                if src is None:
                    raise Fail(NO_TABLE)
                b.huf_src = src
            elif b.lt == 2:
                b.huf_src = b.def_huf
            b.src = [None, None, None]
            for t in range(3):
                if b.defs[t] == "repeat":
                    src = carry_tab[t]
                    for j in range(i - 1, -1, -1):
                        if isinstance(blocks[j].defs[t], tuple):
                            src = blocks[j].defs[t]
                            break
                    if src is None:
                        raise Fail(NO_TABLE)
                    b.src[t] = src
                else:
                    b.src[t] = b.defs[t]
        for b in blocks:
            if b.def_huf is not None:
                carry_huf = b.def_huf
            for t in range(3):
                if isinstance(b.defs[t], tuple):
                    carry_tab[t] = b.defs[t]
        # ---- scratch placement
        for b in blocks:
            b.rec_cap = 64 if b.type != 2 else (b.nseq + 40 + 63) & ~63
            b.rec_at = rec_top
            rec_top += b.rec_cap
            b.nlit = b.lit_regen if b.type == 2 else (b.size if b.type == 0 else (1 if b.size else 0))
            b.lit_at = lit_top
            lit_top += b.nlit
        if rec_top > lay["rec_stride"] or lit_top > lay["lit_stride"] - 16:
            raise Fail(BAD_SIZE)
        # ---- literals
        for b in blocks:
            if b.type == 0:
                lits[b.lit_at:b.lit_at + b.size] = frame[b.at:b.at + b.size]
            elif b.type == 1:
                if b.size:
                    lits[b.lit_at] = frame[b.at]
            elif b.lt == 0:
                lits[b.lit_at:b.lit_at + b.nlit] = frame[b.lit_data:b.lit_data + b.nlit]
            elif b.lt == 1:
                lits[b.lit_at:b.lit_at + b.nlit] = bytes([frame[b.lit_data]]) * b.nlit
            else:
                try:
                    max_bits, table, after, _ = zm.read_huffman_tree(frame, b.huf_src, b.lit_end if b.lt == 2 else n)
                except zm.ZstdError:
                    raise Fail(BAD_HUFFMAN)
                start = after if b.lt == 2 else b.lit_data
                try:
                    if b.streams == 1:
                        out = zm.huffman_decode_stream(frame, start, b.lit_end, max_bits, table, b.nlit)
                    else:
                        if start + 6 > b.lit_end:
                            raise zm.ZstdError("jump table")
                        s1, s2, s3 = struct.unpack_from("<HHH", frame, start)
                        start += 6
                        per = (b.nlit + 3) // 4
                        e = [start, start + s1, start + s1 + s2, start + s1 + s2 + s3, b.lit_end]
                        if e[3] >= b.lit_end or per * 3 > b.nlit:
                            raise zm.ZstdError("jump table")
                        out = bytearray()
                        for k in range(4):
                            out += zm.huffman_decode_stream(frame, e[k], e[k + 1], max_bits, table, per if k < 3 else b.nlit - 3 * per)
                except zm.ZstdError:
                    raise Fail(BAD_LITERALS)
                lits[b.lit_at:b.lit_at + b.nlit] = out
        # ---- sequences -> records
        for b in blocks:
            b.nrec = 0
            b.out = 0
            b.lit_pos = 0
            b.n_sym = 0
            b.hist = None    # final history if it became fully known inside the block

            def put(w0, ll, ml, b=b):
                if (b.nrec & 63) == 0:
                    ck[(b.rec_at + b.nrec) // 64][0:2] = [b.out, b.lit_pos]
                assert b.nrec < b.rec_cap and ll + ml > 0
                recs[b.rec_at + b.nrec] = w0 | ((ll | (ml << 14)) << 32)
                b.nrec += 1
                b.out += ll + ml
                b.lit_pos += ll

            def literal_run(count, b=b):
                while count:
                    piece = min(count, REC_LL)
                    put(REC_FLAG, piece, 0)
                    count -= piece

            if b.type == 0:
                literal_run(b.size)
                continue
            if b.type == 1:
                if b.size:
                    left = b.size - 1
                    piece = min(left, REC_ML)
                    put((1 if piece else 0) | REC_FLAG, 1, piece)
                    left -= piece
                    while left:
                        piece = min(left, REC_ML)
                        put(1 | REC_FLAG, 0, piece)
                        left -= piece
                continue
            if b.nseq:
                tabs = []
                for t, (default, dlog, max_sym, max_log, xb) in enumerate(((zm.LL_DEFAULT, 6, 35, 9, lambda s: zm.LL_BITS[s] if s < 36 else 0),
                                                                            (zm.OF_DEFAULT, 5, 31, 8, lambda s: s),
                                                                            (zm.ML_DEFAULT, 6, 52, 9, lambda s: zm.ML_BITS[s] if s < 53 else 0))):
                    mode, at = b.src[t]
                    if mode == 0:
                        tabs.append((fse_entry_table(default, dlog, xb), dlog))
                    elif mode == 1:
                        sym = frame[at]
                        if sym > max_sym:
                            raise Fail(BAD_SEQUENCES)
                        tabs.append(([sym | (xb(sym) << 10)], 0))
                    else:
                        try:
                            log, counts, _ = zm.read_fse_counts(frame, at, max_sym, max_log)
                            tabs.append((fse_entry_table(counts, log, xb), log))
                        except zm.ZstdError:
                            raise Fail(BAD_SEQUENCES)
                (tl, logl), (to, logo), (tm, logm) = tabs
                if b.end <= b.bits_at or frame[b.end - 1] == 0:
                    raise Fail(BAD_BITSTREAM)
                start_bit = 8 * b.bits_at
                pos = 8 * (b.end - 1) + frame[b.end - 1].bit_length() - 1
                v, t = window16(frame, pos)
                w = top64(v, t)
                sl, w = take(w, logl)
                so, w = take(w, logo)
                sm, w = take(w, logm)
                pos -= logl + logo + logm
                if pos < start_bit:
                    raise Fail(BAD_BITSTREAM)
                r = [0, 0, 0]
                known = [False, False, False]
                all_known = False
                for i in range(b.nseq):
                    el, eo, em = tl[sl], to[so], tm[sm]
                    oc, mlb, llb = (eo >> 10) & 31, (em >> 10) & 31, (el >> 10) & 31
                    if oc > 26:
                        raise Fail(BAD_OFFSET)
                    lastseq = i + 1 == b.nseq
                    nbl, nbm, nbo = (0, 0, 0) if lastseq else ((el >> 6) & 15, (em >> 6) & 15, (eo >> 6) & 15)
                    ext = oc + mlb + llb
                    v, t = window16(frame, pos)
                    w1 = top64(v, t)
                    obits, w1 = take(w1, oc)
                    mbits, w1 = take(w1, mlb)
                    lbits, w1 = take(w1, llb)
                    w2 = top64(v, t - ext)
                    bl, w2 = take(w2, nbl)
                    bm, w2 = take(w2, nbm)
                    bo, w2 = take(w2, nbo)
                    pos -= ext + nbl + nbm + nbo
                    if pos < start_bit:
                        raise Fail(BAD_BITSTREAM)
                    ofv = (1 << oc) + obits
                    mlv = zm.ML_BASE[em & 63] + mbits
                    llv = zm.LL_BASE[el & 63] + lbits
                    sl, sm, so = (el >> 16) + bl, (em >> 16) + bm, (eo >> 16) + bo
                    if b.lit_pos + llv > b.nlit:
                        raise Fail(BAD_LITERALS)
                    # ---- repeat offsets: history slots may still be unknown at the start of a block
                    rep = ofv <= 3
                    if not rep:
                        off = ofv - 3
                        r = [off, r[0], r[1]]
                        known = [True, known[0], known[1]]
                        w0 = off
                    else:
                        idx = ofv - 1 + (1 if llv == 0 else 0)
                        if idx == 0:
                            off, offk = r[0], known[0]
                        elif idx == 1:
                            off, offk = r[1], known[1]
                            r = [r[1], r[0], r[2]]
                            known = [known[1], known[0], known[2]]
                        elif idx == 2:
                            off, offk = r[2], known[2]
                            r = [r[2], r[0], r[1]]
                            known = [known[2], known[0], known[1]]
                        else:
                            off, offk = r[0] - 1, known[0]
                            r = [off, r[0], r[1]]
                            known = [offk, known[0], known[1]]
                        if all_known:
                            if off == 0:
                                raise Fail(BAD_OFFSET)
                            w0 = off
                        else:
                            w0 = REC_REP | (REC_FLAG if llv == 0 else 0) | ofv
                    # ---- records: long runs split
                    while llv > REC_LL:
                        put(REC_FLAG, REC_LL, 0)
                        llv -= REC_LL
                    piece = min(mlv, REC_ML)
                    put(w0, llv, piece)
                    mlv -= piece
                    cont = (w0 & ~REC_FLAG) | REC_FLAG if not (w0 & REC_REP) else None
                    while mlv:
                        piece = min(mlv, REC_ML)
                        if cont is None:
                            # (a split match with a still unknown offset: the continuation repeats "offset 1 of the history
                            # after this sequence", which is what it just became)
                            put(REC_REP | REC_FLAG | 0x10, 0, piece)
                        else:
                            put(cont, 0, piece)
                        mlv -= piece
                    if not all_known and all(known):
                        all_known = True
                        b.n_sym = b.nrec
                if pos != start_bit:
                    raise Fail(BAD_BITSTREAM)
                if not all_known:
                    b.n_sym = b.nrec
                else:
                    b.hist = list(r)
            literal_run(b.nlit - b.lit_pos)
            if b.out > zm.BLOCK_MAX:
                raise Fail(BAD_SIZE)
        # ---- replay of the unknown prefixes, block after block; checkpoints made absolute
        for b in blocks:
            for j in range(b.rec_at, b.rec_at + b.n_sym):
                w = recs[j]
                w0, w1 = w & 0xFFFFFFFF, w >> 32
                if w0 & REC_REP:
                    code = w0 & 0xFF
                    if code == 0x10:
                        off = hist[0]
                    else:
                        idx = code - 1 + (1 if w0 & REC_FLAG else 0)
                        if idx == 0:
                            off = hist[0]
                        elif idx == 1:
                            off = hist[1]
                            hist = [hist[1], hist[0], hist[2]]
                        elif idx == 2:
                            off = hist[2]
                            hist = [hist[2], hist[0], hist[1]]
                        else:
                            off = hist[0] - 1
                            hist = [off, hist[0], hist[1]]
                        if off == 0:
                            raise Fail(BAD_OFFSET)
                    recs[j] = off | (w1 << 32)
                elif not (w0 & REC_FLAG):
                    hist = [w0, hist[0], hist[1]]
            if b.hist is not None:
                hist = list(b.hist)
            b.out_at = out_top
            out_top += b.out
            for s in range(b.rec_at // 64, (b.rec_at + b.rec_cap) // 64):
                first = s * 64 - b.rec_at
                if first < b.nrec:
                    ck[s] = [ck[s][0] + b.out_at, ck[s][1] + b.lit_at, min(64, b.nrec - first)]
                else:
                    ck[s] = [b.out_at + b.out, b.lit_at + b.nlit, 0]
    if p != n:
        raise Fail(TRAILING)
    if out_top != dst_len:
        raise Fail(BAD_SIZE)
    return {"nslots": rec_top // 64, "ck": ck, "recs": recs, "lits": bytes(lits), "out_len": out_top, "nblk": nblk}


def execute_stage(stage, dst_len):
    out = bytearray()
    lits = stage["lits"]
    for s in range(stage["nslots"]):
        o, lp, nv = stage["ck"][s]
        if nv == 0:
            continue
        if o != len(out):
            raise Fail(BAD_SIZE, "checkpoint %d says %d, output is at %d" % (s, o, len(out)))
        for j in range(nv):
            w = stage["recs"][s * 64 + j]
            w0, w1 = w & 0xFFFFFFFF, w >> 32
            ll, ml, off = w1 & 16383, (w1 >> 14) & 16383, w0 & 0x3FFFFFFF
            out += lits[lp:lp + ll]
            lp += ll
            if ml:
                if off == 0 or off > len(out):
                    raise Fail(BAD_OFFSET)
                st = len(out) - off
                for k in range(ml):
                    out.append(out[st + k])
    if len(out) != dst_len:
        raise Fail(BAD_SIZE)
    return bytes(out)


def decode(frame, dst_len):
    return execute_stage(entropy_stage(frame, dst_len), dst_len)
