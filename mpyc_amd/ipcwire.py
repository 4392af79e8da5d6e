"""Device-side wire for CO-LOCATED parties (include/ffgpu.h, section "device-side wire").

The reference's runtime marshals the share rows of the np path with `pickle.dumps(row)` and moves the bytes through
its own asyncio TCP mesh (runtime.py:484-495 `_distribute`, :655-661 `_reshare`, :571-577 `output`;
asyncoro.py:54-106).  mpyc_amd does not replace that mesh.  At n = 10^7 a row is 80 MB: a gate among three local
parties costs 0.5-0.7 s of framing and socket copies against 0.4 ms of kernels (profiles/r03_api_path.md).  When every
party is a process on the SAME node -- one per GPU, or several on one GPU -- the rows need not leave the device(s):

  exporter   FieldArray.__reduce__ (finfields.py), when called from the runtime's own marshalling, returns a
             descriptor instead of the limb bytes: (pid, export id, 64-byte hipIpc handle of the allocation, offset,
             size) -- a few hundred bytes on the TCP mesh.  A device clone of the row is parked in `_pending` so that
             neither the allocator can recycle it nor the caller write to it.
  receiver   `_array_from_ipc` (the unpickle hook) opens the handle (ffgpu_ipc_open; opened allocations are cached),
             copies the row device-to-device into its own memory (ffgpu_ipc_read; over xGMI between GPUs),
             synchronises, and ACKNOWLEDGES with an 8-byte datagram to the exporter's abstract UNIX socket (its address ends
             in a random tag and travels in the descriptor: only receivers of descriptors can acknowledge).
             Rows read through a cached mapping are checked against the descriptor's canary (first and last 16 bytes)
             before the copy; a mismatch drops the mapping and reopens the handle.
  release    the exporter counts how many times a descriptor left the process (`Runtime._send_message` is wrapped
             to look for the descriptor's token in small payloads) and drops the parked buffer when as many
             acknowledgements have arrived -- `output` sends ONE marshalled share to up to t peers (runtime.py:571-577),
             `_distribute` unmarshals the sender's own row locally (runtime.py:490-508; an interprocess handle cannot
             be opened by the process that made it: the descriptor then resolves to the parked buffer itself).

Scope of the switch: descriptors are only produced while the RUNTIME marshals (`mpyc.runtime.pickle` is replaced by a
shim that raises a flag around `dumps`); `pickle.dumps(array)` anywhere else keeps producing the limb bytes.
MPYC_AMD_IPC_WIRE=1 switches the wire on, =0 off; unset, it is on exactly when the runtime itself launched every party
on this machine (`-M<m>` without `-P`/`-C`).  A descriptor is meaningless on another host; receiving one there fails
loudly in ffgpu_ipc_open.  What is parked is a device CLONE of the row taken at marshal time (the semantics of
pickle.dumps); a descriptor is recognised as this process's own by the random tag of its acknowledgement address, never
by the pid.
"""
import atexit
import ctypes
import os
import socket
import struct
import sys
import time
from collections import OrderedDict

import torch

from . import _ffi

_MODE = os.environ.get('MPYC_AMD_IPC_WIRE', 'auto')          # '1' on, '0' off, 'auto': on when every party is on this host
ENABLED = _MODE == '1'
_auto = _MODE == 'auto'
_LOOPBACK = ('localhost', '127.0.0.1', '::1', '')
MIN_BYTES = int(os.environ.get('MPYC_AMD_IPC_WIRE_MIN', str(1 << 16)))      # smaller rows travel inline
TOKEN = b'MPYCAMD-IPC:'
HANDLE_BYTES = 64
ESTALE = 6                # FFGPU_ESTALE
OPEN_CACHE = 16           # exporter allocations kept mapped (the caching allocator hands out the same segments again)

_in_transport = 0         # > 0 while the runtime's pickle.dumps runs (set by the shim below)
_hooked = False
_pending = {}             # export id -> [tensor, times sent, acknowledgements, resolved locally]
_next_id = 0
_sock = None              # this process's acknowledgement socket (abstract namespace: no file to clean up)
_ack_socks = {}           # exporter's tagged address -> connected datagram socket
_opened = OrderedDict()   # (exporter's tagged address, handle bytes) -> (base pointer, ctx)
stats = {'exported': 0, 'imported': 0, 'local': 0, 'released': 0, 'inline': 0}


def enable(on=True):
    global ENABLED, _auto
    ENABLED = bool(on)
    _auto = False


def resolve_auto():
    """MPYC_AMD_IPC_WIRE unset: the wire is switched on only when the runtime itself started every party as a process on
    THIS machine -- `-M<m>` without `-P`/`-C` (runtime.py:5154-5190: party 0 spawns parties m-1..1 with the same command
    line).  Loopback addresses given with `-P localhost:...` say nothing about where the peer runs (containers sharing a
    loopback, SSH-forwarded ports): such runs keep the reference's byte wire unless MPYC_AMD_IPC_WIRE=1 is set on every
    party.  Called once the runtime has been imported (mpyc_amd.install -> arrayGF); explicit '0' / '1' always win."""
    global ENABLED, _auto
    if not _auto:
        return ENABLED
    rt = sys.modules.get('mpyc.runtime')
    mpc = getattr(rt, 'mpc', None)
    parties = getattr(mpc, 'parties', None)
    if parties is None:
        return ENABLED
    _auto = False
    opts = getattr(mpc, 'options', None)
    launched_here = bool(getattr(opts, 'M', None)) and not getattr(opts, 'parties', None) and not getattr(opts, 'config', None)
    hosts = [getattr(pty, 'host', None) for pty in parties]
    ENABLED = (launched_here and len(parties) > 1 and all(h is None or h in _LOOPBACK for h in hosts)
               and torch.cuda.is_available())
    return ENABLED


_sock_addr = None         # abstract-namespace address of this process's acknowledgement socket; it ends in a random
#                           tag and travels inside the descriptors, so only their receivers can acknowledge


def _ack_socket():
    global _sock, _sock_addr
    if _sock is None:
        import secrets
        s = socket.socket(socket.AF_UNIX, socket.SOCK_DGRAM)
        _sock_addr = b'\0mpyc_amd_ipc_%d_%s' % (os.getpid(), secrets.token_hex(8).encode())
        s.bind(_sock_addr)
        s.setblocking(False)
        _sock = s
        atexit.register(_linger)
    return _sock


def drain():
    """Collect acknowledgements; release every parked buffer all of whose receivers have copied it."""
    if _ack_backlog:
        _flush_acks()
    if _sock is None:
        return
    while True:
        try:
            msg = _sock.recv(65536)
        except (BlockingIOError, InterruptedError):
            break
        if len(msg) % 8 == 0:
            for (eid,) in struct.iter_unpack('<Q', msg):
                ent = _pending.get(eid)
                if ent is not None:
                    ent[2] += 1
    for eid in [e for e, ent in _pending.items() if ent[2] >= ent[1] and (ent[1] > 0 or ent[3])]:
        del _pending[eid]
        stats['released'] += 1


def _linger():
    """At interpreter exit: give receivers a moment to finish copying rows this process still owes them."""
    t0 = time.time()
    while (any(ent[2] < ent[1] for ent in _pending.values()) or any(_ack_backlog.values())) and time.time() - t0 < 5.0:
        drain()
        time.sleep(0.002)


# ---- the two hooks into the runtime (both looked up on the imported module; absent -> the wire stays off) -----------
class _PickleShim:
    """Stands in for the `pickle` module inside mpyc.runtime: dumps raises the transport flag around the real call."""

    def __init__(self, real):
        self._real = real

    def dumps(self, obj, *args, **kwargs):
        global _in_transport
        _in_transport += 1
        try:
            return self._real.dumps(obj, *args, **kwargs)
        finally:
            _in_transport -= 1

    def __getattr__(self, name):
        return getattr(self._real, name)


def ensure_runtime_hooks():
    """Install the marshalling flag and the send counter on the imported mpyc.runtime.  Returns True when both are in
    place (then, and only then, descriptors are produced)."""
    global _hooked
    if _hooked:
        return True
    rt = sys.modules.get('mpyc.runtime')
    if rt is None or not hasattr(rt, 'Runtime') or not hasattr(rt.Runtime, '_send_message') or not hasattr(rt, 'pickle'):
        return False
    orig_send = rt.Runtime._send_message

    def _send_message(self, peer_pid, data):
        if _pending:
            # every descriptor in the payload, whatever else travels with it (mpc.transfer pickles arbitrary objects);
            # bytes.find runs at memchr speed, and with the wire on the large rows are descriptors, not bytes
            at = data.find(TOKEN)
            while at >= 0:
                try:
                    ent = _pending.get(int(data[at + len(TOKEN):at + len(TOKEN) + 16], 16))
                except ValueError:
                    ent = None
                if ent is not None:
                    ent[1] += 1
                at = data.find(TOKEN, at + 1)
        return orig_send(self, peer_pid, data)

    rt.Runtime._send_message = _send_message
    if not isinstance(rt.pickle, _PickleShim):
        rt.pickle = _PickleShim(rt.pickle)
    _hooked = True
    return True


# ---- exporter ---------------------------------------------------------------------------------------------------------
def want_descriptor(ctx, nbytes):
    return (ENABLED and _in_transport > 0 and nbytes >= MIN_BYTES and getattr(ctx, 'torch_device', None) is not None
            and _hooked)


def export(ctx, t):
    """Park a SNAPSHOT of the device tensor t and return its descriptor (plain picklable values).  pickle.dumps copies
    the bytes at marshal time (runtime.py:484,571,655); a peer reads the parked buffer later, so what is parked must not
    be storage the caller can still write to (`mpc.output(a)` followed by `np_update(a, ...)`): one device-to-device
    copy, on the stream the export waits for."""
    global _next_id
    _ack_socket()
    drain()
    t = t.clone(memory_format=torch.contiguous_format)          # on torch's current stream = ctx._stream() below
    handle = ctypes.create_string_buffer(HANDLE_BYTES)
    canary = ctypes.create_string_buffer(32)
    offset = ctypes.c_ulonglong()
    nbytes = t.numel() * t.element_size()
    _ffi.check(ctx._L.ffgpu_ipc_export(ctx._h, t.data_ptr(), nbytes, handle, ctypes.byref(offset), canary, ctx._stream()), 'ipc_export')
    _next_id += 1
    eid = _next_id
    _pending[eid] = [t, 0, 0, False]
    if len(_pending) > 1024:            # descriptors that were made but never sent nor resolved (nothing in the runtime does that)
        for old in [e for e, ent in _pending.items() if e < eid - 1024 and ent[1] == 0 and not ent[3]]:
            del _pending[old]
    stats['exported'] += 1
    # canary = first and last 16 bytes of the row: a receiver that reads through a CACHED mapping has them compared with
    # what the mapping shows (ffgpu_ipc_read) -- should an exporter ever free an allocation and get the same handle bytes
    # for a new one, the stale mapping is detected (share rows are uniformly random), dropped and reopened
    return (os.getpid(), TOKEN + b'%016x' % eid, handle.raw, int(offset.value), nbytes, str(t.dtype).replace('torch.', ''),
            tuple(t.shape), _sock_addr, canary.raw)


# ---- receiver ---------------------------------------------------------------------------------------------------------
def fetch(ctx, desc, reduce_n=None):
    """The row behind a descriptor as a tensor in THIS party's memory.  reduce_n: the row holds that many field elements
    of ctx's field -- they are reduced to canonical form on the way (one pass instead of copy + reduce)."""
    pid, token, handle, offset, nbytes, dtype, shape, ack_addr, canary = desc
    eid = int(token[len(TOKEN):], 16)
    if _sock_addr is not None and ack_addr == _sock_addr:
        # this process's own descriptor (the address ends in a 64-bit random tag: a peer with the same pid and export
        # counter -- another container, a forwarded port -- never matches).  An interprocess handle cannot be opened by
        # the process that made it: the parked snapshot itself is the row.
        ent = _pending.get(eid)
        if ent is None:
            raise RuntimeError('device-side wire: this process no longer holds the buffer of its own descriptor')
        # ALWAYS a copy: peers that were (or are about to be) sent the same descriptor read the parked snapshot later, and the
        # local caller may write to what it gets here in place (runtime.py:384-396 loads the one marshalled row locally too);
        # the parked buffer itself is never handed out.  One device copy of a row: microseconds.
        t = ent[0].clone()
        ent[3] = True
        stats['local'] += 1
        drain()
        return t
    pid = ack_addr                    # peers are told apart by their tagged address from here on
    key = (ack_addr, handle)
    t = torch.empty(shape, dtype=getattr(torch, dtype), device=ctx.torch_device)
    for attempt in (0, 1):
        got = _opened.get(key)
        cached = got is not None
        if not cached:
            base = ctypes.c_void_p()
            rc = ctx._L.ffgpu_ipc_open(ctx._h, handle, ctypes.byref(base))
            if rc != 0:
                raise RuntimeError('MPYC_AMD_IPC_WIRE: cannot open the device buffer of party process %d (%s) -- the '
                                   'device-side wire needs every party on the same node with its GPUs visible to the others; '
                                   'set MPYC_AMD_IPC_WIRE=0 on every party to use the byte wire'
                                   % (desc[0], ctx._L.ffgpu_last_hip_error().decode() or ctx._L.ffgpu_strerror(rc).decode()))
            got = _opened[key] = (base.value, ctx)
            stats['opened'] = stats.get('opened', 0) + 1
            while len(_opened) > OPEN_CACHE:
                _, (old, octx) = _opened.popitem(last=False)
                octx._L.ffgpu_ipc_close(octx._h, old)
        else:
            _opened.move_to_end(key)
        check = canary if (cached or attempt) else None
        if reduce_n is not None:
            rc = ctx._L.ffgpu_ipc_read_reduced(ctx._h, got[0], offset, t.data_ptr(), reduce_n, check, ctx._stream())
        else:
            rc = ctx._L.ffgpu_ipc_read(ctx._h, got[0], offset, t.data_ptr(), nbytes, check, ctx._stream())
        if rc != ESTALE:
            _ffi.check(rc, 'ipc_read')
            break
        if attempt:
            raise RuntimeError('device-side wire: the row read through a fresh mapping does not match its descriptor')
        _opened.pop(key)                                # a stale mapping: unmap, open the handle afresh, read again
        ctx._L.ffgpu_ipc_close(ctx._h, got[0])
        stats['stale'] = stats.get('stale', 0) + 1
    _acknowledge(pid, ack_addr, eid)
    stats['imported'] += 1
    return t


_ack_backlog = {}         # exporter's tagged address -> [acknowledgements its socket would not take yet]


def _acknowledge(pid, ack_addr, eid):
    """Tell the exporter that row `eid` has been copied.  Never blocks: a datagram socket whose receiver has a full queue
    (the exporter collects acknowledgements when it next exports or imports) would otherwise stall this party inside a
    message handler -- and two parties stalled on each other's queues would never drain them.  What does not fit is kept
    and sent with the next acknowledgement or the next drain()."""
    s = _ack_socks.get(pid)
    if s is None:
        s = socket.socket(socket.AF_UNIX, socket.SOCK_DGRAM)
        s.setblocking(False)
        try:
            s.connect(ack_addr)
        except OSError:                 # the exporter is gone (its memory with it; our copy is complete)
            s.close()
            return
        _ack_socks[pid] = s
    _ack_backlog.setdefault(pid, []).append(struct.pack('<Q', eid))
    _flush_acks(pid)


def _flush_acks(pid=None):
    for q in ([pid] if pid is not None else list(_ack_backlog)):
        pend, s = _ack_backlog.get(q), _ack_socks.get(q)
        while pend and s is not None:
            batch = pend[:4096]         # one datagram carries every acknowledgement that is waiting (queues hold few datagrams)
            try:
                s.send(b''.join(batch))
            except (ConnectionRefusedError, ConnectionResetError, FileNotFoundError, BrokenPipeError):
                _ack_socks.pop(q, None)             # the exporter is gone (its memory with it)
                pend.clear()
                break
            except OSError:             # EAGAIN / ENOBUFS: its queue is full -- later
                break
            del pend[:len(batch)]
