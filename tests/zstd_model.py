"""Zstandard frame decoder restated in plain Python (RFC 8878), test infrastructure only.

The GPU decoder for .zst block files (libflagstats_amd/csrc/flagstat_zstd_kernels.hip) is written against this model,
and this model is pinned against the image's libzstd over compressor outputs of every level (tests/test_zstd_model.py).
Besides the decoded bytes it returns what each stage produced -- literals, (literal length, match length, offset)
triples, table modes -- so the device stages can be checked one by one.  The reference calls ZSTD_decompress on every
block payload (benchmark/flagstats.cpp:636-682)."""
import struct

MAGIC = 0xFD2FB528
BLOCK_MAX = 1 << 17

LL_BASE = list(range(16)) + [16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
LL_BITS = [0] * 16 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
ML_BASE = list(range(3, 35)) + [35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539]
ML_BITS = [0] * 32 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
LL_DEFAULT = [4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1]
ML_DEFAULT = [1, 4, 3, 2, 2, 2, 2, 2, 2] + [1] * 37 + [-1] * 7
OF_DEFAULT = [1, 1, 1, 1, 1, 1, 2, 2, 2] + [1] * 15 + [-1] * 5
assert len(LL_BASE) == 36 and len(ML_BASE) == 53 and len(LL_DEFAULT) == 36 and len(ML_DEFAULT) == 53 and len(OF_DEFAULT) == 29
assert sum(abs(x) for x in LL_DEFAULT) == 64 and sum(abs(x) for x in ML_DEFAULT) == 64 and sum(abs(x) for x in OF_DEFAULT) == 32


class ZstdError(Exception):
    pass


class FwdBits:
    """little-endian bit reader, first bit = bit 0 of the first byte (FSE table descriptions)"""

    def __init__(self, data, pos):
        self.data, self.bit = data, pos * 8

    def peek(self, n):
        v = 0
        byte, sh = self.bit >> 3, self.bit & 7
        chunk = int.from_bytes(self.data[byte:byte + 8], "little")
        return (chunk >> sh) & ((1 << n) - 1)

    def read(self, n):
        v = self.peek(n)
        self.bit += n
        return v

    def end_byte(self):
        return (self.bit + 7) >> 3


class BackBits:
    """backward bit stream: the last byte's highest set bit ends it; reads take the top remaining bits"""

    def __init__(self, data, begin, end):
        if end <= begin or data[end - 1] == 0:
            raise ZstdError("empty or unterminated backward bit stream")
        self.value = int.from_bytes(data[begin:end], "little")
        self.left = (end - begin) * 8 - (8 - data[end - 1].bit_length()) - 1  # bits below the end mark

    def read(self, n):
        self.left -= n
        if self.left >= 0:
            return (self.value >> self.left) & ((1 << n) - 1)
        # past the start: zeros are supplied (the caller decides whether that is an error)
        have = self.left + n
        if have <= 0:
            return 0
        return (self.value & ((1 << have) - 1)) << (n - have)


def read_fse_counts(data, pos, max_symbol, max_log):
    """FSE table description -> (accuracy log, normalised counts, position after)"""
    br = FwdBits(data, pos)
    log = br.read(4) + 5
    if log > max_log:
        raise ZstdError("FSE accuracy log %d above %d" % (log, max_log))
    remaining = (1 << log) + 1
    threshold = 1 << log
    nbits = log + 1
    counts = []
    while remaining > 1 and len(counts) <= max_symbol:
        mx = (2 * threshold - 1) - remaining
        low = br.peek(nbits - 1)
        if low < mx:
            v = low
            br.bit += nbits - 1
        else:
            v = br.peek(nbits)
            if v >= threshold:
                v -= mx
            br.bit += nbits
        v -= 1
        remaining -= abs(v)
        if remaining < 1:
            raise ZstdError("FSE table description: probabilities above the table size")
        counts.append(v)
        if v == 0:
            while True:
                rep = br.read(2)
                counts.extend([0] * rep)
                if rep != 3:
                    break
        while remaining < threshold:
            nbits -= 1
            threshold >>= 1
    if remaining != 1 or len(counts) > max_symbol + 1:
        raise ZstdError("corrupt FSE table description")
    if br.end_byte() > len(data):
        raise ZstdError("FSE table description past the input")
    return log, counts, br.end_byte()


def build_fse_table(counts, log):
    """-> list of (symbol, nbits, base) per state"""
    size = 1 << log
    symbol = [0] * size
    high = size - 1
    nxt = [0] * len(counts)
    for s, c in enumerate(counts):
        if c == -1:
            symbol[high] = s
            high -= 1
            nxt[s] = 1
        else:
            nxt[s] = c
    step = (size >> 1) + (size >> 3) + 3
    pos = 0
    for s, c in enumerate(counts):
        for _ in range(max(c, 0)):
            symbol[pos] = s
            pos = (pos + step) & (size - 1)
            while pos > high:
                pos = (pos + step) & (size - 1)
    if pos != 0:
        raise ZstdError("FSE spread did not close")
    table = []
    for u in range(size):
        s = symbol[u]
        x = nxt[s]
        nxt[s] += 1
        nb = log - (x.bit_length() - 1)
        table.append((s, nb, (x << nb) - size))
    return table


def rle_table(sym):
    return [(sym, 0, 0)]


def huffman_table_from_weights(weights):
    """weights of symbols 0..n-2 given, the last one implied -> (max bits, list of (symbol, nbits) per code prefix)"""
    total = sum((1 << (w - 1)) for w in weights if w > 0)
    if total == 0:
        raise ZstdError("Huffman weights all zero")
    max_bits = total.bit_length()  # smallest power of two strictly above total
    left = (1 << max_bits) - total
    if left & (left - 1):
        raise ZstdError("Huffman weights do not complete to a power of two")
    weights = list(weights) + [left.bit_length()]
    if max_bits > 11:
        raise ZstdError("Huffman tree deeper than 11 bits")
    table = [None] * (1 << max_bits)
    at = 0
    for w in range(1, max_bits + 1):
        for s, ws in enumerate(weights):
            if ws == w:
                n = 1 << (w - 1)
                for k in range(n):
                    table[at + k] = (s, max_bits + 1 - w)
                at += n
    assert at == 1 << max_bits
    return max_bits, table, weights


def read_huffman_tree(data, pos, end):
    """Huffman tree description -> (max_bits, table, position after, info)"""
    if pos >= end:
        raise ZstdError("Huffman tree description missing")
    hb = data[pos]
    pos += 1
    if hb >= 128:
        n = hb - 127
        nbytes = (n + 1) // 2
        if pos + nbytes > end:
            raise ZstdError("direct Huffman weights past the section")
        weights = []
        for i in range(n):
            b = data[pos + i // 2]
            weights.append(b >> 4 if i % 2 == 0 else b & 15)
        pos += nbytes
        kind = "direct"
    else:
        if pos + hb > end or hb < 2:
            raise ZstdError("FSE-compressed Huffman weights past the section")
        log, counts, after = read_fse_counts(data[:pos + hb], pos, 12, 6)
        table = build_fse_table(counts, log)
        bits = BackBits(data, after, pos + hb)
        s1 = bits.read(log)
        s2 = bits.read(log)
        weights = []
        while True:
            sym, nb, base = table[s1]
            weights.append(sym)
            s1 = base + bits.read(nb)
            if bits.left < 0:
                weights.append(table[s2][0])
                break
            sym, nb, base = table[s2]
            weights.append(sym)
            s2 = base + bits.read(nb)
            if bits.left < 0:
                weights.append(table[s1][0])
                break
            if len(weights) > 255:
                raise ZstdError("too many Huffman weights")
        pos += hb
        kind = "fse"
    if len(weights) > 255:
        raise ZstdError("too many Huffman weights")
    max_bits, table, allw = huffman_table_from_weights(weights)
    return max_bits, table, pos, {"kind": kind, "weights": allw}


def huffman_decode_stream(data, begin, end, max_bits, table, count):
    bits = BackBits(data, begin, end)
    out = bytearray()
    for _ in range(count):
        sym, nb = table[bits.read(max_bits)] if True else (0, 0)
        bits.left += max_bits - nb
        out.append(sym)
    if bits.left != 0:
        raise ZstdError("Huffman stream not consumed exactly (left %d bits)" % bits.left)
    return out


def decode_literals(data, pos, end, state):
    """literals section at data[pos:end) -> (literals, position after, info)"""
    if pos >= end:
        raise ZstdError("literals section missing")
    b0 = data[pos]
    kind = b0 & 3
    fmt = (b0 >> 2) & 3
    info = {"type": ("raw", "rle", "compressed", "treeless")[kind]}
    if kind < 2:
        if fmt in (0, 2):
            size, hdr = b0 >> 3, 1
        elif fmt == 1:
            size, hdr = int.from_bytes(data[pos:pos + 2], "little") >> 4, 2
        else:
            size, hdr = int.from_bytes(data[pos:pos + 3], "little") >> 4, 3
        pos += hdr
        if kind == 0:
            if pos + size > end:
                raise ZstdError("raw literals past the block")
            return bytes(data[pos:pos + size]), pos + size, info
        if pos + 1 > end:
            raise ZstdError("RLE literal missing")
        return bytes([data[pos]]) * size, pos + 1, info
    if fmt < 2:
        h = int.from_bytes(data[pos:pos + 3], "little")
        regen, comp, hdr = (h >> 4) & 1023, (h >> 14) & 1023, 3
    elif fmt == 2:
        h = int.from_bytes(data[pos:pos + 4], "little")
        regen, comp, hdr = (h >> 4) & 16383, (h >> 18) & 16383, 4
    else:
        h = int.from_bytes(data[pos:pos + 5], "little")
        regen, comp, hdr = (h >> 4) & 262143, (h >> 22) & 262143, 5
    streams = 1 if fmt == 0 else 4
    pos += hdr
    if pos + comp > end:
        raise ZstdError("compressed literals past the block")
    if regen > BLOCK_MAX:
        raise ZstdError("literals larger than a block")
    lend = pos + comp
    if kind == 2:
        max_bits, table, pos, tinfo = read_huffman_tree(data, pos, lend)
        state["huf"] = (max_bits, table)
        info["tree"] = tinfo
    else:
        if state.get("huf") is None:
            raise ZstdError("treeless literals without an earlier tree")
        max_bits, table = state["huf"]
    info.update(streams=streams, regen=regen, comp=comp, max_bits=max_bits)
    if streams == 1:
        lit = huffman_decode_stream(data, pos, lend, max_bits, table, regen)
    else:
        if pos + 6 > lend:
            raise ZstdError("jump table past the section")
        s1, s2, s3 = struct.unpack_from("<HHH", data, pos)
        pos += 6
        per = (regen + 3) // 4
        b = [pos, pos + s1, pos + s1 + s2, pos + s1 + s2 + s3, lend]
        if b[3] >= lend or per * 3 > regen:
            raise ZstdError("jump table inconsistent")
        lit = bytearray()
        for k in range(4):
            lit += huffman_decode_stream(data, b[k], b[k + 1], max_bits, table, per if k < 3 else regen - 3 * per)
    return bytes(lit), lend, info


def seq_table(mode, data, pos, end, default, default_log, max_symbol, max_log, prev):
    """-> (table, accuracy log, position after)"""
    if mode == 0:
        return build_fse_table(default, default_log), default_log, pos
    if mode == 1:
        if pos >= end:
            raise ZstdError("RLE symbol missing")
        if data[pos] > max_symbol:
            raise ZstdError("RLE symbol out of range")
        return rle_table(data[pos]), 0, pos + 1
    if mode == 2:
        log, counts, after = read_fse_counts(data[:end], pos, max_symbol, max_log)
        return build_fse_table(counts, log), log, after
    if prev is None:
        raise ZstdError("repeat mode without an earlier table")
    return prev[0], prev[1], pos


def decode_sequences(data, pos, end, state):
    """sequences section at data[pos:end) -> (list of (ll, ml, offset value), info); offset value still holds repeat codes"""
    if pos >= end:
        raise ZstdError("sequences section missing")
    b0 = data[pos]
    if b0 == 0:
        if pos + 1 != end:
            raise ZstdError("bytes after an empty sequences section")
        return [], {"nseq": 0}
    if b0 < 128:
        nseq, pos = b0, pos + 1
    elif b0 < 255:
        if pos + 2 > end:
            raise ZstdError("sequence count past the block")
        nseq, pos = ((b0 - 128) << 8) + data[pos + 1], pos + 2
    else:
        if pos + 3 > end:
            raise ZstdError("sequence count past the block")
        nseq, pos = data[pos + 1] + (data[pos + 2] << 8) + 0x7F00, pos + 3
    if pos >= end:
        raise ZstdError("compression modes missing")
    modes = data[pos]
    pos += 1
    if modes & 3:
        raise ZstdError("reserved bits of the compression modes set")
    llm, ofm, mlm = modes >> 6, (modes >> 4) & 3, (modes >> 2) & 3
    ll = seq_table(llm, data, pos, end, LL_DEFAULT, 6, 35, 9, state.get("ll"))
    of = seq_table(ofm, data, ll[2], end, OF_DEFAULT, 5, 31, 8, state.get("of"))
    ml = seq_table(mlm, data, of[2], end, ML_DEFAULT, 6, 52, 9, state.get("ml"))
    state["ll"], state["of"], state["ml"] = ll[:2], of[:2], ml[:2]
    pos = ml[2]
    bits = BackBits(data, pos, end)
    sl, so, sm = bits.read(ll[1]), bits.read(of[1]), bits.read(ml[1])
    seqs = []
    for i in range(nseq):
        lsym, lnb, lbase = ll[0][sl]
        osym, onb, obase = of[0][so]
        msym, mnb, mbase = ml[0][sm]
        if osym > 31:
            raise ZstdError("offset code above 31")
        ofv = (1 << osym) + bits.read(osym)
        mlv = ML_BASE[msym] + bits.read(ML_BITS[msym])
        llv = LL_BASE[lsym] + bits.read(LL_BITS[lsym])
        seqs.append((llv, mlv, ofv))
        if i + 1 < nseq:
            sl = lbase + bits.read(lnb)
            sm = mbase + bits.read(mnb)
            so = obase + bits.read(onb)
        if bits.left < 0:
            raise ZstdError("sequence bit stream over-read")
    if bits.left != 0:
        raise ZstdError("sequence bit stream not consumed exactly (left %d bits)" % bits.left)
    return seqs, {"nseq": nseq, "modes": (llm, ofm, mlm), "logs": (ll[1], of[1], ml[1])}


def resolve_offsets(seqs, rep):
    """repeat codes -> actual offsets; rep = [r1, r2, r3] is updated in place"""
    out = []
    for ll, ml, ofv in seqs:
        if ofv > 3:
            off = ofv - 3
            rep[2], rep[1], rep[0] = rep[1], rep[0], off
        else:
            idx = ofv - 1 + (1 if ll == 0 else 0)
            if idx == 0:
                off = rep[0]
            else:
                off = rep[idx] if idx < 3 else rep[0] - 1
                if off == 0:
                    raise ZstdError("repeat offset of zero")
                if idx > 1:
                    rep[2] = rep[1]
                rep[1] = rep[0]
                rep[0] = off
        out.append((ll, ml, off))
    return out


def execute(out, literals, seqs, window_base=0):
    lp = 0
    for ll, ml, off in seqs:
        if lp + ll > len(literals):
            raise ZstdError("sequence wants more literals than the block has")
        out += literals[lp:lp + ll]
        lp += ll
        if off > len(out) - window_base:
            raise ZstdError("offset %d reaches before the frame (%d bytes decoded)" % (off, len(out)))
        start = len(out) - off
        if off >= ml:
            out += out[start:start + ml]
        else:
            for k in range(ml):
                out.append(out[start + k])
    out += literals[lp:]


def parse_frame_header(data, pos=0):
    """-> dict(window, content_size, checksum, header_end)"""
    if len(data) < pos + 6 or struct.unpack_from("<I", data, pos)[0] != MAGIC:
        raise ZstdError("not a Zstandard frame")
    fhd = data[pos + 4]
    p = pos + 5
    fcs_flag, single, checksum, did_flag = fhd >> 6, (fhd >> 5) & 1, (fhd >> 2) & 1, fhd & 3
    if fhd & 8:
        raise ZstdError("reserved bit of the frame header set")
    window = None
    if not single:
        wd = data[p]
        p += 1
        if (wd >> 3) > 21:
            raise ZstdError("window above 2^31 bytes (libzstd: frame requires too much memory)")
        base = 1 << (10 + (wd >> 3))
        window = base + (base >> 3) * (wd & 7)
    did_bytes = (0, 1, 2, 4)[did_flag]
    did = int.from_bytes(data[p:p + did_bytes], "little")
    p += did_bytes
    fcs_bytes = (1 if single else 0, 2, 4, 8)[fcs_flag]
    content = None
    if fcs_bytes:
        content = int.from_bytes(data[p:p + fcs_bytes], "little") + (256 if fcs_bytes == 2 else 0)
        p += fcs_bytes
    if p > len(data):
        raise ZstdError("frame header past the input")
    if single:
        window = content
    return {"window": window, "content_size": content, "checksum": bool(checksum), "dict_id": did, "header_end": p}


def decode_frame(data, detail=None):
    """One frame (the whole of `data`) -> decoded bytes; `detail`, if a list, receives one dict per block."""
    data = bytes(data)
    hdr = parse_frame_header(data)
    if hdr["dict_id"]:
        raise ZstdError("dictionaries are not supported")
    p = hdr["header_end"]
    out = bytearray()
    state = {}
    rep = [1, 4, 8]
    while True:
        if p + 3 > len(data):
            raise ZstdError("block header past the input")
        bh = int.from_bytes(data[p:p + 3], "little")
        p += 3
        last, btype, bsize = bh & 1, (bh >> 1) & 3, bh >> 3
        d = {"type": ("raw", "rle", "compressed", "reserved")[btype], "at": p - 3, "size": bsize, "out_at": len(out)}
        if btype == 3:
            raise ZstdError("reserved block type")
        if btype == 0:
            if p + bsize > len(data):
                raise ZstdError("raw block past the input")
            out += data[p:p + bsize]
            p += bsize
        elif btype == 1:
            if p + 1 > len(data):
                raise ZstdError("RLE block past the input")
            out += bytes([data[p]]) * bsize
            p += 1
        else:
            if bsize > BLOCK_MAX or p + bsize > len(data):
                raise ZstdError("compressed block past the input")
            if bsize < 2:
                raise ZstdError("compressed block too small")
            end = p + bsize
            lit, after, linfo = decode_literals(data, p, end, state)
            seqs, sinfo = decode_sequences(data, after, end, state)
            res = resolve_offsets(seqs, rep)
            before = len(out)
            execute(out, lit, res)
            if len(out) - before > BLOCK_MAX:
                raise ZstdError("block decodes to more than 128 KiB")
            d.update(literals=linfo, sequences=sinfo, nlit=len(lit), lit_bytes=lit, seqs=seqs, resolved=res)
            p = end
        d["out_len"] = len(out) - d["out_at"]
        if detail is not None:
            detail.append(d)
        if last:
            break
    if hdr["checksum"]:
        p += 4
    if p != len(data):
        raise ZstdError("bytes after the frame")
    if hdr["content_size"] is not None and hdr["content_size"] != len(out):
        raise ZstdError("frame content size does not match")
    return bytes(out)
