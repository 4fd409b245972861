"""Kaldi feature matrices from ark files: what the reference reads through the third-party `kaldi_io` package
(`kaldi_io.read_mat(b[1]['input'][0]['feat'])`, src/utils/data.py:169; the package is not vendored in the reference and not pinned
in its requirements).  This file restates the published Kaldi on-disk formats (kaldi/src/matrix/kaldi-matrix.cc `Matrix::Read`,
compressed-matrix.{h,cc}) - parity unpinned by a reference fixture: the reference ships no ark file; the tests pin the reader on
byte strings built by hand from the format description and on round trips through the writer below.

Formats read (an rxfilename "path.ark:offset" seeks to the entry's data, as espnet's feats.scp / data.json 'feat' fields do):
  binary  "\\0B" + "FM " | "DM " : "\\4" int32 rows "\\4" int32 cols, then rows*cols float32 | float64, row-major
          "\\0B" + "CM "         : float32 min, float32 range, int32 rows, int32 cols; per column 4 x uint16 (percentiles 0 / 25 / 75 /
                                   100 on the [min, min + range] scale), then one byte per element, column-major, piecewise linear
                                   between the column's percentiles (0..64 | 64..192 | 192..255)
          "\\0B" + "CM2" | "CM3" : same 16-byte header, then uint16 | uint8 per element, row-major, linear on [min, min + range]
  text    " [ r0c0 r0c1 ...\\n r1c0 ... ]"
"""
import struct

import numpy as np


def _open(rx):
    """-> (file object positioned at the entry, needs_close)"""
    if hasattr(rx, "read"):
        return rx, False
    path, offset = rx, None
    if ":" in rx:
        head, tail = rx.rsplit(":", 1)
        if tail.isdigit():
            path, offset = head, int(tail)
    f = open(path, "rb")
    if offset is not None:
        f.seek(offset)
    return f, True


def _need(f, n):
    b = f.read(n)
    if len(b) != n:
        raise EOFError("kaldi_ark: truncated file (wanted %d bytes, got %d)" % (n, len(b)))
    return b


def _read_int32(f):
    if _need(f, 1) != b"\x04":
        raise ValueError("kaldi_ark: expected a 4-byte integer marker")
    return struct.unpack("<i", _need(f, 4))[0]


def _read_compressed(f, token):
    vmin, vrange, rows, cols = struct.unpack("<ffii", _need(f, 16))
    vmin, vrange = np.float32(vmin), np.float32(vrange)
    if token == b"CM ":
        hdr = np.frombuffer(_need(f, cols * 8), dtype="<u2").reshape(cols, 4).astype(np.float32)
        data = np.frombuffer(_need(f, cols * rows), dtype=np.uint8).reshape(cols, rows)
        p = vmin + vrange * np.float32(1.52590218966964e-05) * hdr                     # Uint16ToFloat
        p0, p25, p75, p100 = (p[:, i:i + 1] for i in range(4))
        v = data.astype(np.float32)
        low = p0 + (p25 - p0) * v * np.float32(1.0 / 64.0)
        mid = p25 + (p75 - p25) * (v - np.float32(64.0)) * np.float32(1.0 / 128.0)
        high = p75 + (p100 - p75) * (v - np.float32(192.0)) * np.float32(1.0 / 63.0)
        return np.where(data <= 64, low, np.where(data <= 192, mid, high)).T.astype(np.float32).copy()
    if token == b"CM2":
        data = np.frombuffer(_need(f, rows * cols * 2), dtype="<u2").reshape(rows, cols).astype(np.float32)
        return (vmin + vrange * np.float32(1.52590218966964e-05) * data).astype(np.float32)
    data = np.frombuffer(_need(f, rows * cols), dtype=np.uint8).reshape(rows, cols).astype(np.float32)
    return (vmin + vrange * np.float32(1.0 / 255.0) * data).astype(np.float32)


def _read_text(f, first):
    buf = first
    while b"]" not in buf:
        chunk = f.read(4096)
        if not chunk:
            raise EOFError("kaldi_ark: text matrix without ']'")
        buf += chunk
    body = buf.split(b"]", 1)[0].decode().replace("[", " ")
    rows = [np.array(line.split(), dtype=np.float32) for line in body.split("\n") if line.strip()]
    return np.vstack(rows) if rows else np.zeros((0, 0), np.float32)


def read_mat_fd(f):
    """the matrix at the current position of a binary file object (after the key, if any)"""
    first = _need(f, 2)
    if first != b"\x00B":
        return _read_text(f, first)
    token = _need(f, 3)
    if token in (b"CM ", b"CM2", b"CM3"):
        return _read_compressed(f, token)
    if token in (b"FM ", b"DM "):
        rows, cols = _read_int32(f), _read_int32(f)
        dt, size = ("<f4", 4) if token == b"FM " else ("<f8", 8)
        return np.frombuffer(_need(f, rows * cols * size), dtype=dt).reshape(rows, cols).copy()
    raise ValueError("kaldi_ark: unknown matrix token %r" % token)


def read_mat(rx):
    """kaldi_io.read_mat: `rx` is "file.ark:offset", a plain path, or an open binary file -> float32 / float64 ndarray [rows, cols]"""
    f, close = _open(rx)
    try:
        return read_mat_fd(f)
    finally:
        if close:
            f.close()


def _read_key(f):
    key = b""
    while True:
        c = f.read(1)
        if not c:
            return None
        if c == b" ":
            return key.decode()
        key += c


def read_mat_ark(rx):
    """kaldi_io.read_mat_ark: iterate (key, matrix) over a whole ark file"""
    f, close = _open(rx)
    try:
        while True:
            key = _read_key(f)
            if not key:
                return
            yield key, read_mat_fd(f)
    finally:
        if close:
            f.close()


def read_mat_scp(path):
    """kaldi_io.read_mat_scp: iterate (key, matrix) over a feats.scp ("key file.ark:offset" per line)"""
    with open(path) as scp:
        for line in scp:
            if line.strip():
                key, rx = line.split(None, 1)
                yield key, read_mat(rx.strip())


# ---- writers (tests, and tools that dump synthetic features in the layout the loader reads) --------------------------------------
def write_mat(f, m, key=""):
    """append `m` (float32 / float64 [rows, cols]) as a binary FM / DM entry -> the offset of its data (what goes after ':' in an scp)"""
    m = np.ascontiguousarray(m)
    if key:
        f.write(key.encode() + b" ")
    offset = f.tell()
    token = b"FM " if m.dtype == np.float32 else b"DM "
    if m.dtype not in (np.float32, np.float64):
        raise TypeError("kaldi_ark.write_mat: float32 or float64")
    f.write(b"\x00B" + token + b"\x04" + struct.pack("<i", m.shape[0]) + b"\x04" + struct.pack("<i", m.shape[1]))
    f.write(m.astype("<f4" if m.dtype == np.float32 else "<f8").tobytes())
    return offset


def write_mat_compressed(f, m, key="", method=1):
    """append `m` as a CM (method 1: per-column percentiles + one byte per element), CM2 (2: uint16) or CM3 (3: uint8) entry, the
    encodings of Kaldi's CompressedMatrix -> the offset of its data"""
    m = np.asarray(m, np.float32)
    rows, cols = m.shape
    if key:
        f.write(key.encode() + b" ")
    offset = f.tell()
    vmin = np.float32(m.min()) if m.size else np.float32(0)
    vmax = np.float32(m.max()) if m.size else np.float32(0)
    vrange = np.float32(vmax - vmin)
    if not vrange > 0:                       # a constant matrix: any positive range decodes it
        vrange = np.float32(1e-5) * max(abs(vmin), np.float32(1.0))
    head = struct.pack("<ffii", float(vmin), float(vrange), rows, cols)

    def to_u16(x):
        return np.clip(np.floor((x - vmin) / vrange * np.float32(65535.0) + np.float32(0.499)), 0, 65535).astype("<u2")

    if method == 2:
        f.write(b"\x00BCM2" + head + to_u16(m).tobytes())
    elif method == 3:
        f.write(b"\x00BCM3" + head + np.clip(np.floor((m - vmin) / vrange * np.float32(255.0) + np.float32(0.499)), 0, 255).astype(np.uint8).tobytes())
    else:
        srt = np.sort(m, axis=0)
        q = np.stack([srt[0], srt[rows // 4], srt[(3 * rows) // 4], srt[rows - 1]], 1)                       # [cols, 4]
        u = to_u16(q).astype(np.int64)
        # percentiles must be strictly increasing on the uint16 scale for the three linear pieces to invert
        u[:, 1] = np.minimum(np.maximum(u[:, 1], u[:, 0] + 1), 65533)
        u[:, 2] = np.minimum(np.maximum(u[:, 2], u[:, 1] + 1), 65534)
        u[:, 3] = np.maximum(u[:, 3], u[:, 2] + 1)
        u[:, 0] = np.minimum(u[:, 0], u[:, 1] - 1)
        p = vmin + vrange * np.float32(1.52590218966964e-05) * u.astype(np.float32)
        p0, p25, p75, p100 = (p[:, i:i + 1] for i in range(4))
        x = m.T                                                                                               # [cols, rows]

        def frac(num, den):      # (a constant column's percentiles can coincide in float32: every byte of the piece decodes alike)
            return np.divide(num, den, out=np.zeros_like(num), where=den > 0)

        low = np.floor(frac(x - p0, p25 - p0) * 64 + 0.5)
        mid = 64 + np.floor(frac(x - p25, p75 - p25) * 128 + 0.5)
        high = 192 + np.floor(frac(x - p75, p100 - p75) * 63 + 0.5)
        b = np.where(x < p25, np.clip(low, 0, 64), np.where(x < p75, np.clip(mid, 64, 192), np.clip(high, 192, 255))).astype(np.uint8)
        f.write(b"\x00BCM " + head + u.astype("<u2").tobytes() + b.tobytes())
    return offset

