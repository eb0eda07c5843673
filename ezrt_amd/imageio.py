"""Image writers of the asset path: 8-bit PNG as part 1 saved its renders
(P1/main.cpp:176-194 via svpng: truecolour, no filtering) and PFM for the float frame buffer.
The frame buffer keeps the GL convention (row 0 = bottom); PNG rows go top first, PFM rows
bottom first (its native order)."""
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path, rgb8, bottom_up=True):
    """rgb8: uint8 [H, W, 3]; bottom_up=True flips a GL-convention frame to PNG's top-first rows."""
    a = np.ascontiguousarray(rgb8, np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("write_png wants uint8 [H, W, 3]")
    if bottom_up:
        a = a[::-1]
    h, w = a.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * 3)], axis=1).tobytes()
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(_chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(_chunk(b"IEND", b""))


def quantize_p1(rgb):
    """Part 1's 8-bit quantisation: clamp(pow(x, 1/2.2) * 255, 0, 255) truncated (P1/main.cpp:187-189).
    float64 pow as the C++ `pow(double, double)` call there; not on any parity path."""
    x = np.asarray(rgb, np.float32).astype(np.float64)
    v = np.power(np.maximum(x, 0.0), 1.0 / 2.2) * 255.0
    return np.clip(v, 0.0, 255.0).astype(np.uint8)


def write_pfm(path, img):
    """float32 [H, W, 3] (or [H, W, 4]: alpha dropped), little-endian, rows bottom first."""
    a = np.ascontiguousarray(np.asarray(img, np.float32)[..., :3])
    h, w = a.shape[:2]
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(a.astype("<f4").tobytes())


def read_pfm(path):
    with open(path, "rb") as f:
        if f.readline().strip() != b"PF":
            raise ValueError("not a colour PFM")
        w, h = (int(t) for t in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(w * h * 12), "<f4" if scale < 0 else ">f4")
    return data.reshape(h, w, 3).astype(np.float32)


def write_hdr_rgbe(path, rgbe):
    """uint8 [H, W, 4] RGBE texels -> a Radiance .hdr file HDRLoader::load reads back exactly
    (P5/lib/hdrloader.cpp:50-118).  New-style scanlines (2, 2, W_hi, W_lo, then the four channels, each
    as literal chunks of <= 128 bytes) for 8 <= W < 32768, the flat old format otherwise."""
    a = np.ascontiguousarray(np.asarray(rgbe, np.uint8))
    h, w, c = a.shape
    if c != 4:
        raise ValueError("expected [H, W, 4] RGBE")
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w)
    if w < 8 or w >= 32768:
        body = a.tobytes()
    else:
        planes = np.transpose(a, (0, 2, 1))                      # [H, 4, W]
        full, rest = divmod(w, 128)
        parts = []
        if full:
            blk = planes[:, :, :full * 128].reshape(h, 4, full, 128)
            cnt = np.full((h, 4, full, 1), 128, np.uint8)
            parts.append(np.concatenate([cnt, blk], axis=3).reshape(h, 4, full * 129))
        if rest:
            cnt = np.full((h, 4, 1), rest, np.uint8)
            parts.append(np.concatenate([cnt, planes[:, :, full * 128:]], axis=2))
        chan = np.concatenate(parts, axis=2).reshape(h, -1)      # the four channels of a scanline, back to back
        mark = np.tile(np.array([2, 2, (w >> 8) & 255, w & 255], np.uint8), (h, 1))
        body = np.concatenate([mark, chan], axis=1).tobytes()
    with open(path, "wb") as f:
        f.write(head)
        f.write(body)
