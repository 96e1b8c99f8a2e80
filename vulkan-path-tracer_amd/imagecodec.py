"""PNG and JPEG texture decoding, the numpy twin of host/ImageCodec.cpp (same arithmetic, byte-identical by test).

Stands in for stb_image behind VulkanHelper::AssetImporter::ImportTexture (reference call sites PathTracer.cpp:812-836 and
:239,259,279,299,319): everything comes out as RGBA8 [h, w, 4], rows top to bottom, as stbi_load(path, 4) hands it to
LoadTexture.  JPEG follows stb_image's published arithmetic (fixed-point IDCT with 12-bit constants, triangle-filter chroma
upsampling for 2x1 / 1x2 / 2x2 and nearest otherwise, 20-bit fixed-point YCbCr), because the decoder decides the texels; PNG
follows its conventions where the format leaves a choice (16-bit -> high byte, sub-byte grey scaled by 255 / (2^depth - 1),
tRNS colour key compared at the file's depth).  Parity with stb_image itself is unpinned (it is not in the reference tree and
cannot be run here); tests/test_image_codecs.py holds both twins against PIL.
"""
import struct
import zlib

import numpy as np

_PNG_SIG = b"\x89PNG\r\n\x1a\n"
MAX_TEXELS = 1 << 28


MAX_JPEG_SCANS = 64   # a progressive file of real encoders has about ten

# ---------------------------------------------------------------------------------------------------- PNG
def _png_unfilter(raw, off, w, h, channels, depth, path):
    """One (sub)image: filtered scanlines at raw[off:] -> samples uint16 [h, w, channels], bytes consumed."""
    bits = channels * depth
    row = (w * bits + 7) // 8
    bpp = bits // 8 if bits >= 8 else 1
    if len(raw) - off < (row + 1) * h:
        raise ValueError("bad PNG scanline: %s" % path)
    lines = np.frombuffer(raw, np.uint8, (row + 1) * h, off).reshape(h, row + 1)
    img = np.zeros((h, row), np.uint8)
    zero = np.zeros(row, np.int32)
    for y in range(h):
        ft, src = int(lines[y, 0]), lines[y, 1:].astype(np.int32)
        up = img[y - 1].astype(np.int32) if y else zero
        if ft > 4:
            raise ValueError("bad PNG scanline: %s" % path)
        if ft == 0:
            img[y] = src
        elif ft == 2:
            img[y] = (src + up) & 255
        else:  # filters 1, 3, 4 depend on the byte to the left: sequential
            cur = [0] * row
            s, u = src.tolist(), up.tolist()
            for x in range(row):
                a = cur[x - bpp] if x >= bpp else 0
                c = u[x - bpp] if x >= bpp else 0
                b = u[x]
                if ft == 1:
                    pr = a
                elif ft == 3:
                    pr = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (s[x] + pr) & 255
            img[y] = cur
    n = w * channels
    if depth == 8:
        smp = img[:, :n].astype(np.uint16)
    elif depth == 16:
        smp = (img[:, 0:2 * n:2].astype(np.uint16) << 8) | img[:, 1:2 * n:2]
    else:
        i = np.arange(n) * depth
        smp = ((img[:, i >> 3] >> (8 - depth - (i & 7))) & ((1 << depth) - 1)).astype(np.uint16)
    return smp.reshape(h, w, channels), (row + 1) * h


def decode_png(b, path="<memory>"):
    """Every colour type and bit depth of the format, palette, tRNS, Adam7 -> uint8 [h, w, 4]."""
    if len(b) < 33 or b[:8] != _PNG_SIG:
        raise ValueError("not a PNG: %s" % path)
    p, idat, hdr, plte, trns = 8, [], None, None, b""
    while p + 12 <= len(b):
        n, typ = struct.unpack(">I4s", b[p:p + 8])
        if p + 12 + n > len(b):
            break
        if typ == b"IHDR":
            if n < 13:
                raise ValueError("bad PNG header: %s" % path)
            hdr = struct.unpack(">IIBBBBB", b[p + 8:p + 21])
        elif typ == b"PLTE":
            if n > 768 or n % 3:
                raise ValueError("bad PNG palette: %s" % path)
            plte = np.frombuffer(b, np.uint8, n, p + 8).reshape(-1, 3)
        elif typ == b"tRNS":
            trns = b[p + 8:p + 8 + n]
        elif typ == b"IDAT":
            idat.append(b[p + 8:p + 8 + n])
        elif typ == b"IEND":
            break
        p += 12 + n
    if hdr is None:
        raise ValueError("bad PNG header: %s" % path)
    w, h, depth, ctype, _, _, interlace = hdr
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}.get(ctype, 0)
    ok = {0: (1, 2, 4, 8, 16), 3: (1, 2, 4, 8)}.get(ctype, (8, 16))
    if channels == 0 or depth not in ok or interlace > 1 or w == 0 or h == 0:
        raise ValueError("unsupported PNG format: %s" % path)
    if w * h > MAX_TEXELS:
        raise ValueError("PNG larger than 2^28 texels: %s" % path)
    key = None
    palette = None
    if ctype == 3:
        if plte is None or len(plte) == 0:
            raise ValueError("PNG palette missing: %s" % path)
        palette = np.full((len(plte), 4), 255, np.uint8)
        palette[:, :3] = plte
        k = min(len(trns), len(plte))
        palette[:k, 3] = np.frombuffer(trns, np.uint8, k)
    elif trns and ctype in (0, 2):
        if len(trns) < channels * 2:
            raise ValueError("bad PNG tRNS: %s" % path)
        key = np.array(struct.unpack(">%dH" % channels, trns[:channels * 2]), np.uint16)
    # the inflated size is known from the header: inflate at most that much (a few KB of IDAT must not expand into gigabytes), and a
    # stream too short to hold it (deflate expands at most ~1032 : 1) is rejected before anything is allocated — as the C++ twin does
    passes = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))
    bits = channels * depth
    if not interlace:
        total = ((w * bits + 7) // 8 + 1) * h
    else:
        total = sum((((w - xo + xs - 1) // xs * bits + 7) // 8 + 1) * ((h - yo + ys - 1) // ys) for xo, yo, xs, ys in passes if w > xo and h > yo)
    data = b"".join(idat)
    if len(data) * 1032 + 1024 < total:
        raise ValueError("PNG inflate failed: %s" % path)
    try:
        z = zlib.decompressobj()
        raw = z.decompress(data, total + 1)
    except zlib.error:
        raise ValueError("PNG inflate failed: %s" % path)
    if len(raw) != total or not z.eof:
        raise ValueError("PNG inflate failed: %s" % path)
    if not interlace:
        img, used = _png_unfilter(raw, 0, w, h, channels, depth, path)
    else:
        img = np.zeros((h, w, channels), np.uint16)
        off = 0
        for xo, yo, xs, ys in passes:
            if w <= xo or h <= yo:
                continue
            pw, ph = (w - xo + xs - 1) // xs, (h - yo + ys - 1) // ys
            sub, n = _png_unfilter(raw, off, pw, ph, channels, depth, path)
            off += n
            img[yo::ys, xo::xs] = sub
        used = off
    if used != len(raw):
        raise ValueError("PNG inflate failed: %s" % path)
    scale = {1: 255, 2: 85, 4: 17}.get(depth, 1)
    to8 = (lambda v: (v >> 8).astype(np.uint8)) if depth == 16 else (lambda v: (v * scale).astype(np.uint8))
    out = np.full((h, w, 4), 255, np.uint8)
    if ctype == 0:
        out[..., :3] = to8(img)
        if key is not None:
            out[..., 3] = np.where(img[..., 0] == key[0], 0, 255)
    elif ctype == 2:
        out[..., :3] = to8(img)
        if key is not None:
            out[..., 3] = np.where((img == key).all(-1), 0, 255)
    elif ctype == 3:
        if int(img.max()) >= len(palette):
            raise ValueError("PNG palette index out of range: %s" % path)
        out[:] = palette[img[..., 0]]
    elif ctype == 4:
        out[..., :3] = to8(img[..., :1]); out[..., 3] = to8(img[..., 1])
    else:
        out[:] = to8(img)
    return out


# ---------------------------------------------------------------------------------------------------- JPEG
_ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63] + [63] * 15)
_ZZ = _ZIGZAG.tolist()


def _i16(v):
    """C's (short) conversion."""
    v &= 0xffff
    return v - 0x10000 if v >= 0x8000 else v


class _Huff:
    def __init__(self, counts, vals):
        self.lookup = {}
        code, k = 0, 0
        for length in range(1, 17):
            for _ in range(counts[length - 1]):
                self.lookup[(length, code)] = vals[k]
                code += 1; k += 1
            if code > (1 << length):
                raise ValueError("bad code lengths")
            code <<= 1


class _Comp:
    pass


class _Jpeg:
    def __init__(self, data, path):
        self.d, self.p, self.path = data, 0, path
        self.bitbuf, self.bitcnt, self.marker, self.nomore = 0, 0, -1, False
        self.dequant = [np.zeros(64, np.int32) for _ in range(4)]
        self.hdc, self.hac = [None] * 4, [None] * 4
        self.comp = []
        self.restart_interval = 0
        self.adobe, self.jfif, self.rgb, self.progressive = -1, False, False, False
        self.eob_run = 0

    def fail(self, m):
        raise ValueError("JPEG: %s: %s" % (m, self.path))

    def get8(self):
        if self.p < len(self.d):
            v = self.d[self.p]; self.p += 1
            return v
        return 0

    def get16(self):
        a = self.get8()
        return (a << 8) | self.get8()

    # ---- entropy-coded data
    def grow(self):
        while self.bitcnt <= 24:
            b = 0 if self.nomore else self.get8()
            if b == 0xff and not self.nomore:
                c = self.get8()
                while c == 0xff:
                    c = self.get8()
                if c != 0:
                    self.marker, self.nomore, b = c, True, 0
            self.bitbuf |= b << (24 - self.bitcnt)
            self.bitcnt += 8

    def bits(self, n):
        if n == 0:
            return 0
        if self.bitcnt < n:
            self.grow()
        v = (self.bitbuf >> (32 - n)) & ((1 << n) - 1)
        self.bitbuf = (self.bitbuf << n) & 0xffffffff
        self.bitcnt -= n
        return v

    def decode(self, h):
        if self.bitcnt < 16:
            self.grow()
        lk = h.lookup
        for length in range(1, 17):
            v = lk.get((length, self.bitbuf >> (32 - length)))
            if v is not None:
                self.bitbuf = (self.bitbuf << length) & 0xffffffff
                self.bitcnt -= length
                return v
        return -1

    def extend_receive(self, n):
        if n == 0:
            return 0
        v = self.bits(n)
        return v - (1 << n) + 1 if v < (1 << (n - 1)) else v

    def reset_entropy(self):
        self.bitbuf, self.bitcnt, self.nomore, self.marker = 0, 0, False, -1
        for c in self.comp:
            c.dc_pred = 0
        self.todo = self.restart_interval if self.restart_interval else 0x7fffffff
        self.eob_run = 0

    def block_baseline(self, c):
        data = [0] * 64
        dq = self.dq_list[c.tq]
        hdc, hac = self.hdc[c.td], self.hac[c.ta]
        if hdc is None or hac is None:
            self.fail("missing huffman table")
        t = self.decode(hdc)
        if t < 0 or t > 15:
            self.fail("bad huffman code")
        c.dc_pred += self.extend_receive(t)
        data[0] = _i16(c.dc_pred * dq[0])
        k = 1
        while k < 64:
            rs = self.decode(hac)
            if rs < 0:
                self.fail("bad huffman code")
            s, r = rs & 15, rs >> 4
            if s == 0:
                if rs != 0xf0:
                    break
                k += 16
            else:
                k += r
                z = _ZZ[k]; k += 1
                data[z] = _i16(self.extend_receive(s) * dq[z])
        return data

    def block_prog_dc(self, data, c):
        if self.spec_end != 0:
            self.fail("can't merge dc and ac")
        if self.succ_high == 0:
            data[:] = 0
            hdc = self.hdc[c.td]
            if hdc is None:
                self.fail("missing huffman table")
            t = self.decode(hdc)
            if t < 0 or t > 15:
                self.fail("bad huffman code")
            c.dc_pred += self.extend_receive(t)
            data[0] = _i16(c.dc_pred * (1 << self.succ_low))
        elif self.bits(1):
            data[0] = _i16(int(data[0]) + (1 << self.succ_low))

    def block_prog_ac(self, data, c):
        if self.spec_start == 0:
            self.fail("can't merge dc and ac")
        hac = self.hac[c.ta]
        if hac is None:
            self.fail("missing huffman table")
        if self.succ_high == 0:
            shift = self.succ_low
            if self.eob_run:
                self.eob_run -= 1
                return
            k = self.spec_start
            while True:
                rs = self.decode(hac)
                if rs < 0:
                    self.fail("bad huffman code")
                s, r = rs & 15, rs >> 4
                if s == 0:
                    if r < 15:
                        self.eob_run = 1 << r
                        if r:
                            self.eob_run += self.bits(r)
                        self.eob_run -= 1
                        break
                    k += 16
                else:
                    k += r
                    z = _ZZ[k]; k += 1
                    data[z] = _i16(self.extend_receive(s) * (1 << shift))
                if k > self.spec_end:
                    break
            return
        bitv = _i16(1 << self.succ_low)

        def refine(z):
            q = int(data[z])
            if self.bits(1) and (q & bitv) == 0:
                data[z] = _i16(q + bitv) if q > 0 else _i16(q - bitv)
        if self.eob_run:
            self.eob_run -= 1
            for k in range(self.spec_start, self.spec_end + 1):
                if data[_ZZ[k]] != 0:
                    refine(_ZZ[k])
            return
        k = self.spec_start
        while True:
            rs = self.decode(hac)
            if rs < 0:
                self.fail("bad huffman code")
            s, r = rs & 15, rs >> 4
            if s == 0:
                if r < 15:
                    self.eob_run = (1 << r) - 1
                    if r:
                        self.eob_run += self.bits(r)
                    r = 64
            else:
                if s != 1:
                    self.fail("bad huffman code")
                s = bitv if self.bits(1) else -bitv
            while k <= self.spec_end:
                z = _ZZ[k]; k += 1
                if data[z] != 0:
                    refine(z)
                else:
                    if r == 0:
                        data[z] = _i16(s)
                        break
                    r -= 1
            if k > self.spec_end:
                break

    # ---- segments
    def read_dqt(self, length):
        while length > 0:
            q = self.get8()
            prec, t = q >> 4, q & 15
            if prec not in (0, 1) or t > 3:
                self.fail("bad DQT")
            for i in range(64):
                self.dequant[t][_ZZ[i]] = self.get16() if prec else self.get8()
            length -= 129 if prec else 65
        if length != 0:
            self.fail("bad DQT length")

    def read_dht(self, length):
        while length > 0:
            q = self.get8()
            tc, th = q >> 4, q & 15
            if tc > 1 or th > 3:
                self.fail("bad DHT")
            counts = [self.get8() for _ in range(16)]
            n = sum(counts)
            if n > 256:
                self.fail("bad DHT")
            vals = [self.get8() for _ in range(n)]
            try:
                h = _Huff(counts, vals)
            except ValueError:
                self.fail("bad code lengths")
            (self.hdc if tc == 0 else self.hac)[th] = h
            length -= 17 + n
        if length != 0:
            self.fail("bad DHT length")

    def read_sof(self, length, prog):
        self.progressive = prog
        if length < 11:
            self.fail("bad SOF length")
        if self.get8() != 8:
            self.fail("only 8-bit JPEG is supported")
        self.img_y, self.img_x, n = self.get16(), self.get16(), self.get8()
        if self.img_x <= 0 or self.img_y <= 0:
            self.fail("empty JPEG")
        if self.img_x * self.img_y > MAX_TEXELS:
            self.fail("JPEG larger than 2^28 texels")
        if n not in (1, 3):
            self.fail("only 1- and 3-component JPEG is supported")
        if length != 8 + 3 * n:
            self.fail("bad SOF length")
        rgbn = 0
        for i in range(n):
            c = _Comp()
            c.id = self.get8()
            if n == 3 and c.id == b"RGB"[i]:
                rgbn += 1
            q = self.get8()
            c.h, c.v, c.tq = q >> 4, q & 15, self.get8()
            if not (1 <= c.h <= 4 and 1 <= c.v <= 4) or c.tq > 3:
                self.fail("bad SOF component")
            c.td = c.ta = 0; c.dc_pred = 0
            self.comp.append(c)
        self.rgb = rgbn == 3
        self.h_max = max(c.h for c in self.comp); self.v_max = max(c.v for c in self.comp)
        for c in self.comp:
            if self.h_max % c.h or self.v_max % c.v:
                self.fail("bad sampling factors")
        self.mcu_w, self.mcu_h = self.h_max * 8, self.v_max * 8
        self.mcu_x = (self.img_x + self.mcu_w - 1) // self.mcu_w
        self.mcu_y = (self.img_y + self.mcu_h - 1) // self.mcu_h
        for c in self.comp:
            c.x = (self.img_x * c.h + self.h_max - 1) // self.h_max
            c.y = (self.img_y * c.v + self.v_max - 1) // self.v_max
            c.w2, c.h2 = self.mcu_x * c.h * 8, self.mcu_y * c.v * 8
            c.bw, c.bh = c.w2 // 8, c.h2 // 8
            c.coeff = np.zeros((c.bh, c.bw, 64), np.int32)   # (dequantised) coefficients per block, natural order
            c.seen = np.zeros((c.bh, c.bw), bool)              # blocks a scan has delivered

    def read_sos(self, length):
        self.scan_n = self.get8()
        if not (1 <= self.scan_n <= len(self.comp)) or length != 6 + 2 * self.scan_n:
            self.fail("bad SOS")
        self.order = []
        for _ in range(self.scan_n):
            cid, q = self.get8(), self.get8()
            which = [k for k, c in enumerate(self.comp) if c.id == cid]
            if not which:
                self.fail("bad SOS component")
            c = self.comp[which[-1]]
            c.td, c.ta = q >> 4, q & 15
            if c.td > 3 or c.ta > 3:
                self.fail("bad SOS tables")
            self.order.append(which[-1])
        self.spec_start, self.spec_end = self.get8(), self.get8()
        q = self.get8()
        self.succ_high, self.succ_low = q >> 4, q & 15
        if self.progressive:
            if self.spec_start > 63 or self.spec_end > 63 or self.spec_start > self.spec_end or self.succ_high > 13 or self.succ_low > 13:
                self.fail("bad SOS")
        else:
            if self.spec_start != 0 or self.succ_high != 0 or self.succ_low != 0:
                self.fail("bad SOS")
            self.spec_end = 63

    def restart_if_due(self):
        self.todo -= 1
        if self.todo > 0:
            return
        if self.bitcnt < 24:
            self.grow()
        if 0xd0 <= self.marker <= 0xd7:
            self.reset_entropy()

    def do_block(self, c, bx, by):
        if not self.progressive:
            c.coeff[by, bx] = self.block_baseline(c)
            c.seen[by, bx] = True
        elif self.spec_start == 0:
            self.block_prog_dc(c.coeff[by, bx], c)
        else:
            self.block_prog_ac(c.coeff[by, bx], c)

    def decode_scan(self):
        self.reset_entropy()
        self.dq_list = [d.tolist() for d in self.dequant]
        if self.scan_n == 1:
            c = self.comp[self.order[0]]
            w, h = (c.x + 7) >> 3, (c.y + 7) >> 3
            for j in range(h):
                for i in range(w):
                    self.do_block(c, i, j)
                    self.restart_if_due()
                    if self.nomore and self.marker >= 0 and not (0xd0 <= self.marker <= 0xd7) and self.bitcnt <= 0:
                        return
            return
        for j in range(self.mcu_y):
            for i in range(self.mcu_x):
                for k in self.order:
                    c = self.comp[k]
                    for y in range(c.v):
                        for x in range(c.h):
                            if self.progressive and self.spec_start != 0:
                                self.fail("can't merge dc and ac")
                            self.do_block(c, i * c.h + x, j * c.v + y)
                self.restart_if_due()

    def next_marker(self):
        if self.marker >= 0:
            m, self.marker = self.marker, -1
            return m
        x = self.get8()
        if x != 0xff:
            return -1
        while x == 0xff:
            x = self.get8()
        return x

    def decode_file(self):
        if self.get8() != 0xff or self.get8() != 0xd8:
            self.fail("not a JPEG")
        have_sof = have_scan = False
        scans = 0
        while True:
            m = self.next_marker()
            while m < 0 and self.p < len(self.d):
                m = self.next_marker()
            if m < 0 or m == 0xd9:
                break
            if m == 0xda:
                if not have_sof:
                    self.fail("SOS before SOF")
                scans += 1
                if scans > MAX_JPEG_SCANS:   # every scan walks every MCU of the image: bounded work for a crafted file (the C++ twin has the same cap)
                    self.fail("too many scans")
                self.read_sos(self.get16())
                self.decode_scan()
                have_scan = True
                if self.marker < 0:   # the scan's data ends at the next marker: skip to it
                    d, n = self.d, len(self.d)
                    while self.p < n:
                        b = d[self.p]; self.p += 1
                        if b == 0xff:
                            while self.p < n and d[self.p] == 0xff:
                                self.p += 1
                            if self.p < n and d[self.p] != 0:
                                self.marker = d[self.p]; self.p += 1
                                break
                self.nomore = False
                continue
            if 0xd0 <= m <= 0xd7:
                continue
            length = self.get16() - 2
            if length < 0 or self.p + length > len(self.d):
                self.fail("bad segment length")
            seg_end = self.p + length
            if m == 0xdb:
                self.read_dqt(length)
            elif m == 0xc4:
                self.read_dht(length)
            elif m in (0xc0, 0xc1, 0xc2):
                if have_sof:
                    self.fail("two SOF segments")
                self.read_sof(length + 2, m == 0xc2)
                have_sof = True
            elif m == 0xdd:
                if length != 2:
                    self.fail("bad DRI")
                self.restart_interval = self.get16()
            elif m == 0xee and length >= 12 and self.d[self.p:self.p + 5] == b"Adobe":
                self.adobe = self.d[self.p + 11]
            elif m == 0xe0 and length >= 5 and self.d[self.p:self.p + 5] == b"JFIF\0":
                self.jfif = True
            elif 0xc3 <= m <= 0xcf and m not in (0xc4, 0xc8, 0xcc):
                self.fail("unsupported JPEG coding process (lossless / hierarchical / arithmetic)")
            self.p = seg_end
        if not have_sof or not have_scan:
            self.fail("no image data")


def _idct_blocks(coef):
    """stb_image's integer IDCT on int32 coefficient blocks [..., 64] (natural order) -> uint8 [..., 8, 8]."""
    d = coef.reshape(coef.shape[:-1] + (8, 8)).astype(np.int32)

    def one_d(s0, s1, s2, s3, s4, s5, s6, s7):
        p2, p3 = s2, s6
        p1 = (p2 + p3) * 2217
        t2 = p1 + p3 * -7567
        t3 = p1 + p2 * 3135
        p2, p3 = s0, s4
        t0 = (p2 + p3) * 4096
        t1 = (p2 - p3) * 4096
        x0, x3, x1, x2 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
        t0, t1, t2, t3 = s7, s5, s3, s1
        p3, p4, p1, p2 = t0 + t2, t1 + t3, t0 + t3, t1 + t2
        p5 = (p3 + p4) * 4816
        t0, t1, t2, t3 = t0 * 1223, t1 * 8410, t2 * 12586, t3 * 6149
        p1 = p5 + p1 * -3685
        p2 = p5 + p2 * -10497
        p3 = p3 * -8034
        p4 = p4 * -1597
        return x0, x1, x2, x3, t0 + p1 + p3, t1 + p2 + p4, t2 + p2 + p3, t3 + p1 + p4
    # columns: d[..., row, col]; the transform runs down each column
    x0, x1, x2, x3, t0, t1, t2, t3 = one_d(*[d[..., r, :] for r in range(8)])
    x0, x1, x2, x3 = x0 + 512, x1 + 512, x2 + 512, x3 + 512
    v = np.stack([(x0 + t3) >> 10, (x1 + t2) >> 10, (x2 + t1) >> 10, (x3 + t0) >> 10, (x3 - t0) >> 10, (x2 - t1) >> 10, (x1 - t2) >> 10, (x0 - t3) >> 10], axis=-2)
    x0, x1, x2, x3, t0, t1, t2, t3 = one_d(*[v[..., :, c] for c in range(8)])
    k = 65536 + (128 << 17)
    x0, x1, x2, x3 = x0 + k, x1 + k, x2 + k, x3 + k
    o = np.stack([(x0 + t3) >> 17, (x1 + t2) >> 17, (x2 + t1) >> 17, (x3 + t0) >> 17, (x3 - t0) >> 17, (x2 - t1) >> 17, (x1 - t2) >> 17, (x0 - t3) >> 17], axis=-1)
    return np.clip(o, 0, 255).astype(np.uint8)


def _resample_row(near, far, w, hs, vs):
    n, f = near[:w].astype(np.int32), far[:w].astype(np.int32)
    if hs == 1 and vs == 1:
        return near[:w].copy()
    if hs == 1 and vs == 2:
        return ((3 * n + f + 2) >> 2).astype(np.uint8)
    if hs == 2 and vs == 1:
        out = np.zeros(2 * w, np.int32)
        if w == 1:
            out[:] = n[0]
            return out.astype(np.uint8)
        out[0] = n[0]; out[1] = (n[0] * 3 + n[1] + 2) >> 2
        m = 3 * n[1:w - 1] + 2
        out[2:2 * w - 2:2] = (m + n[0:w - 2]) >> 2
        out[3:2 * w - 2:2] = (m + n[2:w]) >> 2
        out[2 * w - 2] = (n[w - 2] * 3 + n[w - 1] + 2) >> 2; out[2 * w - 1] = n[w - 1]
        return out.astype(np.uint8)
    if hs == 2 and vs == 2:
        t = 3 * n + f
        out = np.zeros(2 * w, np.int32)
        if w == 1:
            out[:] = (t[0] + 2) >> 2
            return out.astype(np.uint8)
        out[0] = (t[0] + 2) >> 2
        out[1:2 * w - 1:2] = (3 * t[:-1] + t[1:] + 8) >> 4
        out[2:2 * w:2] = (3 * t[1:] + t[:-1] + 8) >> 4
        out[2 * w - 1] = (t[w - 1] + 2) >> 2
        return out.astype(np.uint8)
    return np.repeat(near[:w], hs)


def decode_jpeg(b, path="<memory>"):
    """Baseline / progressive Huffman JPEG, 8 bit, 1 or 3 components -> uint8 [h, w, 4] (alpha 255)."""
    J = _Jpeg(bytes(b), path)
    J.decode_file()
    W, H = J.img_x, J.img_y
    planes = []
    for c in J.comp:
        coef = c.coeff
        if J.progressive:   # dequantise at the end, with (short) wrap-around as the C code has it
            coef = ((coef * J.dequant[c.tq][None, None, :] + 0x8000) & 0xffff) - 0x8000
            done = np.zeros_like(c.seen)
            done[:(c.y + 7) >> 3, :(c.x + 7) >> 3] = True   # finish covers exactly the blocks the image covers
        else:
            done = c.seen
        px = _idct_blocks(coef)                                  # [bh, bw, 8, 8]
        px[~done] = 0                                            # blocks no scan delivered stay 0
        planes.append(px.transpose(0, 2, 1, 3).reshape(c.h2, c.w2))
    out = np.full((H, W, 4), 255, np.uint8)
    rows = []
    for k, c in enumerate(J.comp):
        hs, vs = J.h_max // c.h, J.v_max // c.v
        w_lores = (W + hs - 1) // hs
        ystep, ypos, l0, l1 = vs >> 1, 0, 0, 0
        comp_rows = np.zeros((H, w_lores * hs), np.uint8)
        for j in range(H):
            bot = ystep >= (vs >> 1)
            near, far = (planes[k][l1], planes[k][l0]) if bot else (planes[k][l0], planes[k][l1])
            comp_rows[j] = _resample_row(near, far, w_lores, hs, vs)
            ystep += 1
            if ystep >= vs:
                ystep = 0; l0 = l1
                ypos += 1
                if ypos < c.y:
                    l1 += 1
        rows.append(comp_rows[:, :W])
    if len(J.comp) == 1:
        out[..., :3] = rows[0][..., None]
    elif J.rgb or (J.adobe == 0 and not J.jfif):
        out[..., 0], out[..., 1], out[..., 2] = rows
    else:
        y = (rows[0].astype(np.int64) << 20) + (1 << 19)
        cb, cr = rows[1].astype(np.int64) - 128, rows[2].astype(np.int64) - 128
        r = y + cr * (5743 << 8)
        g = y + cr * -(2925 << 8) + ((cb * -(1410 << 8)) & -65536)   # the C code masks the low 16 bits of this term
        bl = y + cb * (7258 << 8)
        out[..., 0] = np.clip(r >> 20, 0, 255); out[..., 1] = np.clip(g >> 20, 0, 255); out[..., 2] = np.clip(bl >> 20, 0, 255)
    return out


def decode_image(b, path="<memory>"):
    if b[:4] == b"\x89PNG":
        return decode_png(b, path)
    if b[:2] == b"\xff\xd8":
        return decode_jpeg(b, path)
    raise ValueError("unsupported image format (need PNG or JPEG): %s" % path)


def load_image(path):
    return decode_image(open(path, "rb").read(), path)
