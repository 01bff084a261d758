"""Test helpers: build RIFF/WAVE and AIFF / AIFF-C files byte by byte (no audio library in the image)."""
import struct

import numpy as np


def _chunk_le(tag, body):
    return tag + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")


def _chunk_be(tag, body):
    return tag + struct.pack(">I", len(body)) + body + (b"\0" if len(body) & 1 else b"")


def sample_bytes(pcm, kind, big_endian=False):
    """pcm: float array (frames, ch) in [-1, 1) -> interleaved bytes of kind s16/s24/s32/f32/f64 and the values the
    meter should see (as the numpy array a TrackSpec would take; s24 as packed little-endian uint8)."""
    if kind == "f32":
        a = pcm.astype(np.float32)
        return a.astype(">f4" if big_endian else "<f4").tobytes(), a
    if kind == "f64":
        a = pcm.astype(np.float64)
        return a.astype(">f8" if big_endian else "<f8").tobytes(), a
    if kind == "s16":
        a = np.clip(np.round(pcm * 32768.0), -32768, 32767).astype(np.int16)
        return a.astype(">i2" if big_endian else "<i2").tobytes(), a
    if kind == "s32":
        a = np.clip(np.round(pcm * 2147483648.0), -2147483648, 2147483647).astype(np.int32)
        return a.astype(">i4" if big_endian else "<i4").tobytes(), a
    if kind == "s24":
        v = np.clip(np.round(pcm * 8388608.0), -8388608, 8388607).astype(np.int32)
        le = np.stack([v & 0xff, (v >> 8) & 0xff, (v >> 16) & 0xff], axis=-1).astype(np.uint8)  # (frames, ch, 3)
        raw = le[..., ::-1] if big_endian else le
        return raw.tobytes(), le.reshape(-1)
    raise ValueError(kind)


_BITS = {"s16": 16, "s24": 24, "s32": 32, "f32": 32, "f64": 64}


def wav_bytes(pcm, fs, kind, extensible_mask=None, extra_chunks=(), data_size_override=None):
    data, seen = sample_bytes(pcm, kind)
    ch = pcm.shape[1]
    bits = _BITS[kind]
    tag = 3 if kind in ("f32", "f64") else 1
    block = ch * bits // 8
    if extensible_mask is None:
        fmt = struct.pack("<HHIIHH", tag, ch, fs, fs * block, block, bits)
    else:
        guid = struct.pack("<H", tag) + bytes.fromhex("000000001000800000aa00389b71")
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, ch, fs, fs * block, block, bits, 22, bits, extensible_mask) + guid
    body = b"WAVE" + _chunk_le(b"fmt ", fmt)
    for tg, bd in extra_chunks:
        body += _chunk_le(tg, bd)
    dchunk = b"data" + struct.pack("<I", len(data) if data_size_override is None else data_size_override) + data
    if len(data) & 1:
        dchunk += b"\0"
    body += dchunk
    return b"RIFF" + struct.pack("<I", len(body)) + body, seen


def _extended(x):
    """80-bit IEEE extended, big endian (AIFF sample rate)."""
    import math
    if x == 0:
        return b"\0" * 10
    m, e = math.frexp(x)  # x = m * 2^e, 0.5 <= m < 1
    mant = int(m * (1 << 64))
    return struct.pack(">HQ", e - 1 + 16383, mant)


def aiff_bytes(pcm, fs, kind, aifc=None, ssnd_offset=0, ssnd_first=False):
    """aifc: None (plain AIFF, big endian) or a compression tag: b'NONE', b'twos', b'sowt', b'fl32', b'FL32', b'fl64'."""
    big = aifc != b"sowt"
    data, seen = sample_bytes(pcm, kind, big_endian=big)
    ch, frames = pcm.shape[1], pcm.shape[0]
    comm = struct.pack(">hIh", ch, frames, _BITS[kind]) + _extended(float(fs))
    if aifc is not None:
        comm += aifc + b"\x00\x00"  # empty pascal string (count byte + pad)
    ssnd = struct.pack(">II", ssnd_offset, 0) + b"\xaa" * ssnd_offset + data
    chunks = [_chunk_be(b"COMM", comm), _chunk_be(b"SSND", ssnd)]
    if ssnd_first:
        chunks.reverse()
    body = (b"AIFC" if aifc is not None else b"AIFF")
    if aifc is not None:
        body += _chunk_be(b"FVER", struct.pack(">I", 0xA2805140))
    body += b"".join(chunks)
    return b"FORM" + struct.pack(">I", len(body)) + body, seen
