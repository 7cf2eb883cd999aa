"""Kaldi-style token / basic-type stream I/O (text and binary) as kaldi_native_io's io-funcs do it --
the wire conventions the reference's Write/Read methods are built on (csrc/transition-model.cc:37-116,
csrc/hmm-topology.cc:23-282, csrc/context-dep.cc:45-83, csrc/event-map.cc).

  token            ASCII + one space, in both modes
  int32 / uint32   text: decimal + space;  binary: one size byte (+4 signed, -4 unsigned) + little-endian
  float            text: shortest repr + space;  binary: size byte 4 + IEEE-754 little-endian
  integer vector   text: "[ a b c ]\\n";  binary: size byte 4, int32 count, raw int32
  float vector     text: " [ a b c ]\\n" (6 significant digits);  binary: "FV " + int32 count (basic type) + raw float32
  files            binary files start with "\\0B"
"""
import io
import struct
from typing import List

import numpy as np

from ._lib import KhgError


def _fmt_float(x: float) -> str:
    """C++ ostream << float with default precision (6 significant digits)."""
    s = "%.6g" % float(x)
    return s


class Writer:
    def __init__(self, binary: bool):
        self.binary = binary
        self.buf = io.BytesIO()

    def raw(self, s):
        self.buf.write(s if isinstance(s, bytes) else s.encode("ascii"))

    def token(self, t: str):
        self.raw(t + " ")

    def nl(self):
        if not self.binary:
            self.raw("\n")

    def int32(self, v: int):
        if self.binary:
            self.raw(struct.pack("<bi", 4, int(v)))
        else:
            self.raw(f"{int(v)} ")

    def uint32(self, v: int):
        if self.binary:
            self.raw(struct.pack("<bI", -4, int(v)))
        else:
            self.raw(f"{int(v)} ")

    def float32(self, v: float):
        if self.binary:
            self.raw(struct.pack("<bf", 4, float(v)))
        else:
            self.raw(_fmt_float(v) + " ")

    def int_vector(self, v):
        v = [int(x) for x in v]
        if self.binary:
            self.raw(struct.pack("<bi", 4, len(v)))
            self.raw(np.asarray(v, "<i4").tobytes())
        else:
            self.raw("[ " + "".join(f"{x} " for x in v) + "]\n")

    def float_vector(self, v):
        v = np.asarray(v, np.float32)
        if self.binary:
            self.token("FV")
            self.int32(v.shape[0])
            self.raw(v.astype("<f4").tobytes())
        else:
            self.raw(" [ " + "".join(_fmt_float(x) + " " for x in v) + "]\n")

    def getvalue(self) -> bytes:
        return self.buf.getvalue()


class Reader:
    def __init__(self, data: bytes, binary: bool):
        self.d = data
        self.i = 0
        self.binary = binary

    @staticmethod
    def from_file_bytes(data: bytes) -> "Reader":
        """Kaldi Input: a leading "\\0B" selects binary mode."""
        if data[:2] == b"\0B":
            return Reader(data[2:], True)
        return Reader(data, False)

    def _skip_ws(self):
        while self.i < len(self.d) and self.d[self.i: self.i + 1].isspace():
            self.i += 1

    def peek(self) -> str:
        if not self.binary:
            self._skip_ws()
        if self.i >= len(self.d):
            raise KhgError("unexpected end of stream")
        return chr(self.d[self.i])

    def token(self) -> str:
        self._skip_ws()
        j = self.i
        while j < len(self.d) and not self.d[j: j + 1].isspace():
            j += 1
        if j == self.i:
            raise KhgError("ReadToken: unexpected end of stream")
        t = self.d[self.i: j].decode("ascii")
        self.i = min(j + 1, len(self.d))      # consume exactly one trailing space
        return t

    def expect(self, t: str):
        got = self.token()
        if got != t:
            raise KhgError(f"Expected token \"{t}\", got instead \"{got}\".")

    def _basic(self, size_byte: int, fmt: str):
        if self.d[self.i] != (size_byte & 0xFF):
            raise KhgError(f"ReadBasicType: expected size byte {size_byte}, got {self.d[self.i]} at offset {self.i}")
        (v,) = struct.unpack_from(fmt, self.d, self.i + 1)
        self.i += 1 + struct.calcsize(fmt)
        return v

    def int32(self) -> int:
        if self.binary:
            return self._basic(4, "<i")
        return int(self.token())

    def uint32(self) -> int:
        if self.binary:
            return self._basic(-4, "<I")
        return int(self.token())

    def float32(self) -> float:
        if self.binary:
            return self._basic(4, "<f")
        return float(self.token())

    def int_vector(self) -> List[int]:
        if self.binary:
            if self.d[self.i] != 4:
                raise KhgError("ReadIntegerVector: expected to see type of size 4")
            (n,) = struct.unpack_from("<i", self.d, self.i + 1)
            self.i += 5
            v = np.frombuffer(self.d, "<i4", n, self.i).tolist()
            self.i += 4 * n
            return v
        self.expect("[")
        out = []
        while True:
            t = self.token()
            if t == "]":
                return out
            out.append(int(t))

    def float_vector(self) -> np.ndarray:
        if self.binary:
            self.expect("FV")
            n = self.int32()
            v = np.frombuffer(self.d, "<f4", n, self.i).astype(np.float32)
            self.i += 4 * n
            return v
        self.expect("[")
        out = []
        while True:
            t = self.token()
            if t == "]":
                return np.asarray(out, np.float32)
            out.append(float(t))


def write_file(filename: str, binary: bool, payload: bytes):
    with open(filename, "wb") as fh:
        if binary:
            fh.write(b"\0B")
        fh.write(payload)


def read_file(filename: str) -> Reader:
    with open(filename, "rb") as fh:
        return Reader.from_file_bytes(fh.read())
