"""Image files of the reference's asset path, without third-party decoders: Radiance .hdr environment maps
(PathTracer.cpp:1137-1164 via the absent VulkanHelper::AssetImporter), PNG / JPEG textures (imagecodec.py; glTF's two core
image formats) and the PNG export (Editor::SaveToFile, Editor.cpp:815-843).  host/SceneLoader.cpp + host/ImageCodec.cpp hold
the same readers/writers in C++; the tests compare the two byte for byte."""
import struct
import zlib

import numpy as np

from .imagecodec import decode_image, decode_jpeg, decode_png, load_image  # noqa: F401  (the LDR texture decoders)

_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def load_hdr(path):
    """Radiance RGBE ('-Y h +X w', flat or new-RLE scanlines) -> float32 [h, w, 4], rows top to bottom, A = 1.
    value = mantissa * 2^(e - 136), 0 when e == 0 (stb_image's convention)."""
    b = open(path, "rb").read()
    p = b.index(b"\n") + 1
    if not (b.startswith(b"#?RADIANCE") or b.startswith(b"#?RGBE")):
        raise ValueError("not a Radiance HDR file: %s" % path)
    fmt = False
    while True:
        e = b.index(b"\n", p)
        line, p = b[p:e], e + 1
        if not line:
            break
        fmt |= line == b"FORMAT=32-bit_rle_rgbe"
    if not fmt:
        raise ValueError("unsupported HDR format (need 32-bit_rle_rgbe): %s" % path)
    e = b.index(b"\n", p)
    tok, p = b[p:e].split(), e + 1
    if len(tok) != 4 or tok[0] != b"-Y" or tok[2] != b"+X":
        raise ValueError("unsupported HDR orientation (need -Y h +X w): %s" % path)
    h, w = int(tok[1]), int(tok[3])
    rgbe = np.zeros((h, w, 4), np.uint8)
    for y in range(h):
        if w < 8 or w >= 32768 or b[p] != 2 or b[p + 1] != 2 or (b[p + 2] & 0x80):
            rgbe[y] = np.frombuffer(b, np.uint8, w * 4, p).reshape(w, 4)
            p += w * 4
            continue
        if ((b[p + 2] << 8) | b[p + 3]) != w:
            raise ValueError("bad HDR scanline width: %s" % path)
        p += 4
        for c in range(4):
            x = 0
            while x < w:
                n = b[p]; p += 1
                if n > 128:
                    n -= 128
                    rgbe[y, x:x + n, c] = b[p]; p += 1
                else:
                    rgbe[y, x:x + n, c] = np.frombuffer(b, np.uint8, n, p); p += n
                if n == 0 or x + n > w:
                    raise ValueError("bad HDR run: %s" % path)
                x += n
    out = np.ones((h, w, 4), np.float32)
    scale = np.ldexp(np.float32(1.0), rgbe[..., 3].astype(np.int32) - 136).astype(np.float32)
    out[..., :3] = rgbe[..., :3].astype(np.float32) * scale[..., None]
    out[rgbe[..., 3] == 0, :3] = 0.0
    return out


def rgbe_encode(rgb):
    """float [h, w, 3] -> uint8 [h, w, 4] (Ward's float2rgbe: mantissa = v * 256 / 2^e with frexp of the largest channel)."""
    rgb = np.asarray(rgb, np.float32)
    m = rgb.max(axis=2)
    mant, ex = np.frexp(m)
    scale = np.where(m < 1e-32, 0.0, mant * 256.0 / np.maximum(m, 1e-38)).astype(np.float32)
    out = np.zeros(rgb.shape[:2] + (4,), np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(m < 1e-32, 0, ex + 128).astype(np.uint8)
    return out


def save_hdr(path, rgb, rle=True):
    """Writes a Radiance .hdr; rle=True uses new-style RLE scanlines when the width allows (8 <= w < 32768)."""
    q = rgbe_encode(rgb)
    h, w = q.shape[:2]
    out = bytearray(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + ("-Y %d +X %d\n" % (h, w)).encode())
    for y in range(h):
        if not rle or w < 8 or w >= 32768:
            out += q[y].tobytes()
            continue
        out += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            row = q[y, :, c]
            x = 0
            while x < w:
                run = 1
                while x + run < w and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 4:
                    out += bytes([128 + run, int(row[x])]); x += run
                    continue
                e = x  # literal span up to the next run of >= 4 (or 128 bytes)
                while e < w and e - x < 128:
                    r = 1
                    while e + r < w and r < 4 and row[e + r] == row[e]:
                        r += 1
                    if r >= 4:
                        break
                    e += 1
                out += bytes([e - x]) + row[x:e].tobytes(); x = e
    open(path, "wb").write(bytes(out))


def load_png(path):
    """Any PNG (imagecodec.decode_png) -> uint8 [h, w, 4]."""
    return decode_png(open(path, "rb").read(), path)


def save_png(path, rgba):
    """uint8 [h, w, 4] -> RGBA PNG (filter 0 rows, one zlib stream: what host/SceneLoader.cpp SavePNG writes)."""
    rgba = np.ascontiguousarray(rgba, np.uint8)
    h, w = rgba.shape[:2]
    raw = np.zeros((h, w * 4 + 1), np.uint8)
    raw[:, 1:] = rgba.reshape(h, w * 4)

    def chunk(typ, data):
        body = typ + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)
    open(path, "wb").write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)) +
                           chunk(b"IEND", b""))
