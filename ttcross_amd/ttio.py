"""Host-side reader/writer of the reference's raw tensor-train stream file (lib/ttio.f90).

Layout (little endian, no record markers -- Fortran `access='stream'`):
  128-byte header `tthead` (lib/ttio.f90:10-17): 'TT      ', ver(2) = (1, 0), inf(4) = (tt_size = 2048, 0, 0, 0),
  comment*64, i(8) with i(1:2) = (l, m) [comment and i(3:8) are never set by the reference];
  l, m            int32   (lib/ttio.f90:76)
  n(l:m), r(l-1:m) int32  (lib/ttio.f90:77)
  cores l..m      float64, each r(k-1) x n(k) x r(k) column-major, concatenated (lib/ttio.f90:65-69, 78).
The device engine has the same pair behind the C-ABI (ttx_write / ttx_read, include/ttx.h); this module is file
plumbing for hosts without a GPU (inspecting or producing files) and holds no numerics.
"""
import struct

import numpy as np

HEADER_BYTES = 128
TT_SIZE = 2048
# bytes of the header the reference leaves uninitialised (comment, i(3:8)): ignore them when comparing files
UNSET = [(32, 96), (104, 128)]


class TTFileError(ValueError):
    pass


def write_tt(path, cores, l=1):
    """dtt_write: cores = list of (r(k-1), n(k), r(k)) arrays."""
    cores = [np.asarray(c, dtype=np.float64) for c in cores]
    d = len(cores)
    if d < 1 or any(c.ndim != 3 for c in cores):
        raise TTFileError("dtt_write: tt structure has invalid size")
    n = [c.shape[1] for c in cores]
    r = [cores[0].shape[0]] + [c.shape[2] for c in cores]
    for k in range(1, d):
        if cores[k].shape[0] != r[k]:
            raise TTFileError(f"dtt_write: ranks of cores {k} and {k + 1} do not match")
    m = l + d - 1
    head = b"TT      " + struct.pack("<2i4i", 1, 0, TT_SIZE, 0, 0, 0) + bytes(64) + struct.pack("<8i", l, m, 0, 0, 0, 0, 0, 0)
    with open(path, "wb") as f:
        f.write(head)
        f.write(struct.pack("<2i", l, m))
        f.write(np.asarray(n + r, dtype="<i4").tobytes())
        for c in cores:
            f.write(np.asfortranarray(c).ravel(order="F").astype("<f8").tobytes())


def read_tt(path):
    """dtt_read: returns (l, n, r, cores); the reference's checks (lib/ttio.f90:236-251) raise TTFileError."""
    with open(path, "rb") as f:
        b = f.read()
    if len(b) < HEADER_BYTES + 8:
        raise TTFileError("dtt_read: error reading header")
    if b[:2] != b"TT":
        raise TTFileError("dtt_read: not TT header in file")
    ver = struct.unpack_from("<2i", b, 8)
    if ver[0] != 1:
        raise TTFileError(f"dtt_read: not correct version of TT file: {ver}")
    l, m = struct.unpack_from("<2i", b, HEADER_BYTES)
    if l < 0 or m < l or m > TT_SIZE:
        raise TTFileError(f"dtt_read: read strange l,m: {l} {m}")
    d = m - l + 1
    off = HEADER_BYTES + 8
    if len(b) < off + 4 * (2 * d + 1):
        raise TTFileError("dtt_read: error reading nr")
    n = np.frombuffer(b, dtype="<i4", count=d, offset=off).astype(np.int32)
    r = np.frombuffer(b, dtype="<i4", count=d + 1, offset=off + 4 * d).astype(np.int32)
    off += 4 * (2 * d + 1)
    cores = []
    for k in range(d):
        sz = int(r[k]) * int(n[k]) * int(r[k + 1])
        if sz <= 0 or len(b) < off + 8 * sz:
            raise TTFileError("dtt_read: error reading cores")
        cores.append(np.frombuffer(b, dtype="<f8", count=sz, offset=off).reshape((r[k], n[k], r[k + 1]), order="F").copy())
        off += 8 * sz
    return l, n, r, cores


def same_file(a, b):
    """byte equality of two stream files, ignoring the header bytes the reference never sets"""
    x, y = bytearray(open(a, "rb").read()), bytearray(open(b, "rb").read())
    if len(x) != len(y):
        return False
    for lo, hi in UNSET:
        x[lo:hi] = bytes(hi - lo)
        y[lo:hi] = bytes(hi - lo)
    return x == y
