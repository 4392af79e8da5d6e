"""Marshalled forms of share rows.

Two layers, both the reference's own formats:

* element layer -- `field.to_bytes / from_bytes` (finfields.py:91-102): byte_length little-endian bytes per
  element.  For every field of the device path that IS the limb layout in HBM (8 / 12 / 16 bytes; GF(2^8)
  elements are 1-byte limbs padded to the reference's 2-byte elements), so `FieldArray.to_wire()` is one
  device-to-pinned-host copy and no Python integer exists on either side;
* message layer -- the frame `<qI{n}s` that `asyncoro.MessageExchanger.send` writes (asyncoro.py:54-64):
  program counter (8 bytes, signed, little-endian) | payload size (4 bytes, unsigned) | payload;
  `FrameReader` is the receiving side (asyncoro.py:66-106: accumulate, split off whole frames).

Under mpyc_amd.install() the reference's own asyncoro does the framing and `pickle.dumps(row)` the
marshalling (runtime.py:484,571,655): FieldArray.__reduce__ / HostView.__reduce__ make that pickle carry the
element-layer bytes instead of a graph of PyLongs.  The functions here are the same two layers for callers
that drive the engine without the reference's runtime (bench, multi-GPU party-major exchange over sockets).
Golden vectors: tests/golden/wire.json, produced by the reference's own to_bytes and send().
"""
from __future__ import annotations

import struct
from typing import Iterator, List, Optional, Tuple

_HEADER = struct.Struct('<qI')
MAX_PAYLOAD = 2**32 - 1


def frame(pc: int, payload: bytes) -> bytes:
    """One message as asyncoro.MessageExchanger.send writes it (asyncoro.py:54-64)."""
    n = len(payload)
    if n > MAX_PAYLOAD:
        raise ValueError('payload exceeds the 4-byte size field of the frame')      # struct.error in the reference
    return _HEADER.pack(pc, n) + bytes(payload)


def frame_rows(pc: int, rows) -> bytes:
    """Frame of one share row or several rows back to back (element-layer bytes concatenated)."""
    if hasattr(rows, 'to_wire'):
        return frame(pc, rows.to_wire())
    return frame(pc, b''.join(r.to_wire() for r in rows))


class FrameReader:
    """Receiving side (asyncoro.py:66-106): feed() byte chunks of any size, iterate complete (pc, payload) pairs."""

    def __init__(self):
        self._buf = bytearray()

    def feed(self, data: bytes) -> List[Tuple[int, bytes]]:
        self._buf.extend(data)
        out = []
        buf = self._buf
        while len(buf) >= 12:
            pc, size = _HEADER.unpack_from(buf)
            if len(buf) < 12 + size:
                break
            out.append((pc, bytes(buf[12:12 + size])))
            del buf[:12 + size]
        return out

    @property
    def pending(self) -> int:
        """bytes received that do not form a whole frame yet"""
        return len(self._buf)


def unframe(stream: bytes) -> Iterator[Tuple[int, bytes]]:
    r = FrameReader()
    yield from r.feed(stream)
    if r.pending:
        raise ValueError(f'{r.pending} trailing bytes do not form a whole frame')


def marshal(row) -> bytes:
    """field.to_bytes of a device share row (finfields.py:91-95) -- straight from limbs."""
    return row.to_wire()


def unmarshal(field, data: bytes, shape: Optional[tuple] = None, check: bool = True):
    """field.from_bytes (finfields.py:97-102) into a device array of `field` (reduced like field.array(...) when
    the bytes come from a peer)."""
    return field.array.from_wire(data, shape, check=check)
