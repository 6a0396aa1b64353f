"""Writes a traced running-mean image to disk the way the reference shows it on screen.

The reference's display path is GL: gamma correction `rgb = pow(rgb, 1/gamma)` with gamma 2.2
(GLSL/postProcess.fs:20-28, Renderer/PostProcessor.hpp:50) and then the framebuffer's clamp to [0, 1]
when the float colour is written to the 8-bit window (GLSL/screenOutput.fs:15-21 composites the debug
layer, which a headless run does not have).  Image row 0 is the BOTTOM row (GL convention,
Renderer/TraceMethods.hpp:30), so rows are flipped for file formats whose first row is the top.
"""
import struct
import zlib

import numpy as np


def to_display(image, gamma=2.2):
    """(H, W, 4) float running mean -> (H, W, 3) uint8, top row first."""
    rgb = np.asarray(image, dtype=np.float32)[..., :3]
    rgb = np.power(np.maximum(rgb, 0.0), np.float32(1.0 / gamma))
    rgb = np.clip(rgb, 0.0, 1.0)
    return (rgb[::-1] * 255.0 + 0.5).astype(np.uint8)


def write_ppm(path, image, gamma=2.2):
    px = to_display(image, gamma)
    h, w, _ = px.shape
    with open(path, "wb") as fh:
        fh.write(b"P6\n%d %d\n255\n" % (w, h))
        fh.write(px.tobytes())


def write_png(path, image, gamma=2.2):
    px = to_display(image, gamma)
    h, w, _ = px.shape
    raw = b"".join(b"\x00" + px[y].tobytes() for y in range(h))

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n")
        fh.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        fh.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        fh.write(chunk(b"IEND", b""))


def primary_ids_to_png(path, prim):
    """False-colour view of a primary-hit id image (0xFFFFFFFF = miss -> black)."""
    p = np.asarray(prim, dtype=np.uint32)
    h = (p * np.uint32(2654435761)) & np.uint32(0xFFFFFF)
    rgb = np.stack([(h >> 16) & 0xFF, (h >> 8) & 0xFF, h & 0xFF], axis=-1).astype(np.float32) / 255.0
    rgb[p == 0xFFFFFFFF] = 0.0
    img = np.concatenate([rgb, np.ones(rgb.shape[:2] + (1,), dtype=np.float32)], axis=-1)
    write_png(path, img, gamma=1.0)
