"""Host logic of the device-side wire (mpyc_amd/ipcwire.py) without a GPU: the marshalling flag is only up inside the
runtime's pickle.dumps, descriptors that leave the process are counted by the wrapped Runtime._send_message, a parked
buffer is released exactly when as many acknowledgements have come in (or when the exporter resolved its own
descriptor locally and nobody else got it), and anything else stays parked.  libffgpu's interprocess entry points are
replaced by host-memory stand-ins here; the real thing runs in tests/test_api_path.py -m gpu."""
import ctypes
import os
import pickle
import socket
import struct
import sys
import types

import pytest
import torch


class FakeLib:
    def ffgpu_ipc_export(self, h, ptr, nbytes, handle, offset, canary, stream):
        ctypes.memmove(handle, struct.pack('<Q', ptr) + bytes(56), 64)
        offset._obj.value = 0
        return 0

    def ffgpu_last_hip_error(self):
        return b''


class FakeCtx:
    torch_device = torch.device('cpu')
    _h = None
    _L = FakeLib()

    def _stream(self):
        return None


@pytest.fixture
def wire(monkeypatch):
    from mpyc_amd import ipcwire
    rt = types.ModuleType('mpyc.runtime')
    sent = []

    class Runtime:
        def _send_message(self, peer_pid, data):
            sent.append((peer_pid, data))
    rt.Runtime = Runtime
    rt.pickle = pickle
    monkeypatch.setitem(sys.modules, 'mpyc.runtime', rt)
    for name, val in (('ENABLED', True), ('_auto', False), ('_hooked', False), ('_pending', {}), ('_next_id', 0), ('_in_transport', 0), ('_sock', None), ('_sock_addr', None),
                      ('stats', {'exported': 0, 'imported': 0, 'local': 0, 'released': 0, 'inline': 0})):
        monkeypatch.setattr(ipcwire, name, val)
    yield ipcwire, rt, sent
    if ipcwire._sock is not None:
        ipcwire._sock.close()


def ack(ipcwire, eid):
    s = socket.socket(socket.AF_UNIX, socket.SOCK_DGRAM)
    s.connect(ipcwire._sock_addr)
    s.send(struct.pack('<Q', eid))
    s.close()


class Row:
    """stands for a FieldArray: __reduce__ asks the wire whether to ship a descriptor"""

    def __init__(self, t):
        self.t = t

    def __reduce__(self):
        from mpyc_amd import ipcwire
        ctx = FakeCtx()
        ipcwire.ensure_runtime_hooks()
        if ipcwire.want_descriptor(ctx, self.t.numel() * 8):
            return (tuple, (ipcwire.export(ctx, self.t),))
        return (bytes, (b'inline',))


def test_descriptor_only_inside_runtime_marshalling(wire):
    ipcwire, rt, sent = wire
    row = Row(torch.arange(20000, dtype=torch.int64))
    assert ipcwire.ensure_runtime_hooks() and isinstance(rt.pickle, ipcwire._PickleShim)
    assert pickle.loads(pickle.dumps(row)) == b'inline'                      # a user's pickle: the data
    blob = rt.pickle.dumps(row)                                              # the runtime's marshal: a descriptor
    desc = pickle.loads(blob)
    assert desc[0] == os.getpid() and desc[1].startswith(ipcwire.TOKEN) and len(desc[2]) == 64 and desc[4] == 160000
    assert ipcwire._in_transport == 0 and len(blob) < 400
    small = Row(torch.arange(10, dtype=torch.int64))
    assert pickle.loads(rt.pickle.dumps(small)) == b'inline'                 # below MIN_BYTES
    ipcwire.enable(False)
    assert pickle.loads(rt.pickle.dumps(row)) == b'inline'
    ipcwire.enable(True)


def test_release_after_as_many_acknowledgements_as_sends(wire):
    ipcwire, rt, sent = wire
    assert ipcwire.ensure_runtime_hooks()            # (install() does this when the first field is made)
    r = rt.Runtime()
    rows = [Row(torch.full((20000,), j, dtype=torch.int64)) for j in range(3)]
    blobs = [rt.pickle.dumps(x) for x in rows]
    ids = sorted(ipcwire._pending)
    assert len(ids) == 3 and all(ent[1] == 0 for ent in ipcwire._pending.values())
    r._send_message(1, blobs[1])                                             # _distribute: row j to party j ...
    r._send_message(2, blobs[2])
    assert [ipcwire._pending[i][1] for i in ids] == [0, 1, 1] and len(sent) == 2
    ipcwire.drain()
    assert len(ipcwire._pending) == 3                                        # nothing acknowledged yet, row 0 unresolved
    rows[0].t.fill_(77)                                                      # the caller writes to its array after marshalling ...
    parked = ipcwire._pending[ids[0]][0]
    own = ipcwire.fetch(FakeCtx(), pickle.loads(blobs[0]))                   # ... and the own row is unmarshalled locally
    assert own.data_ptr() != parked.data_ptr()       # never the parked snapshot itself: peers holding the descriptor read it later
    assert own is not rows[0].t and bool((own == 0).all()) and ipcwire.stats['local'] == 1     # the snapshot of marshal time
    assert ids[0] not in ipcwire._pending and len(ipcwire._pending) == 2     # released: nobody else holds its descriptor
    ack(ipcwire, ids[1])
    ipcwire.drain()
    assert sorted(ipcwire._pending) == [ids[2]]
    # output(): ONE marshalled share goes to two peers -- two acknowledgements before the buffer may be recycled
    share = rt.pickle.dumps(Row(torch.ones(20000, dtype=torch.int64)))
    sid = max(ipcwire._pending)
    r._send_message(1, share)
    r._send_message(2, share)
    ack(ipcwire, sid)
    ipcwire.drain()
    assert sid in ipcwire._pending and ipcwire._pending[sid][1:3] == [2, 1]
    ack(ipcwire, sid)
    ack(ipcwire, ids[2])
    ipcwire.drain()
    assert not ipcwire._pending and ipcwire.stats['released'] == 4
    with pytest.raises(RuntimeError):
        ipcwire.fetch(FakeCtx(), pickle.loads(blobs[0]))                     # a second local resolution: the buffer is gone
    # a descriptor inside a LARGE payload (mpc.transfer of an object holding an array and other data) is counted too
    big = rt.pickle.dumps(Row(torch.ones(20000, dtype=torch.int64)))
    bid = max(ipcwire._pending)
    r._send_message(1, b'x' * 100000 + big + b'y' * 100000)
    assert ipcwire._pending[bid][1] == 1
    ack(ipcwire, bid)
    ipcwire.drain()
    assert not ipcwire._pending


def test_wire_stays_off_without_the_runtime_hooks(wire, monkeypatch):
    ipcwire, rt, sent = wire
    del rt.Runtime._send_message                                             # an upstream without the choke point
    assert ipcwire.ensure_runtime_hooks() is False
    assert pickle.loads(rt.pickle.dumps(Row(torch.arange(20000, dtype=torch.int64)))) == b'inline'


def test_auto_mode_follows_the_party_list(wire, monkeypatch):
    """MPYC_AMD_IPC_WIRE unset: on exactly when the runtime launched more than one party on this machine itself (-M<m>
    without -P / -C): loopback addresses alone (containers, forwarded ports) do not switch it on."""
    ipcwire, rt, sent = wire
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)

    class P:
        def __init__(self, host):
            self.host = host
    for hosts, want in ((['localhost'] * 3, True), (['localhost'], False), (['localhost', '10.0.0.7', 'localhost'], False),
                        ([None, None], True), (['127.0.0.1', ''], True)):
        monkeypatch.setattr(ipcwire, '_auto', True)
        monkeypatch.setattr(ipcwire, 'ENABLED', False)
        rt.mpc = types.SimpleNamespace(parties=[P(h) for h in hosts], options=types.SimpleNamespace(M=len(hosts), parties=None, config=None))
        assert ipcwire.resolve_auto() is want, hosts
    for opts in (types.SimpleNamespace(M=None, parties=['localhost:11365', 'localhost:11366'], config=None),
                 types.SimpleNamespace(M=None, parties=None, config='run.ini'), None):
        monkeypatch.setattr(ipcwire, '_auto', True)
        monkeypatch.setattr(ipcwire, 'ENABLED', False)
        rt.mpc = types.SimpleNamespace(parties=[P('localhost'), P('localhost')], options=opts)
        assert ipcwire.resolve_auto() is False, opts
    monkeypatch.setattr(ipcwire, '_auto', False)                  # an explicit setting is not re-evaluated
    monkeypatch.setattr(ipcwire, 'ENABLED', False)
    rt.mpc = types.SimpleNamespace(parties=[P('localhost')] * 3)
    assert ipcwire.resolve_auto() is False


class RemoteLib(FakeLib):
    """host-memory stand-in for the receiver's entry points: a `handle` is the address of the exporter's buffer"""

    def __init__(self):
        self.opens, self.closes, self.stale_next = 0, 0, False

    def ffgpu_ipc_open(self, h, handle, base):
        self.opens += 1
        base._obj.value = struct.unpack('<Q', handle[:8])[0]
        return 0

    def ffgpu_ipc_close(self, h, base):
        self.closes += 1
        return 0

    def _read(self, base, offset, dst, nbytes, expect):
        if expect is not None and self.stale_next:
            self.stale_next = False
            return 6                                      # FFGPU_ESTALE: the cached mapping shows something else
        ctypes.memmove(dst, base + offset, nbytes)
        return 0

    def ffgpu_ipc_read(self, h, base, offset, dst, nbytes, expect, stream):
        return self._read(base, offset, dst, nbytes, expect)

    def ffgpu_ipc_read_reduced(self, h, base, offset, dst, n, expect, stream):
        return self._read(base, offset, dst, 8 * n, expect)

    def ffgpu_strerror(self, rc):
        return b'error %d' % rc


def test_remote_fetch_acknowledges_and_survives_a_stale_mapping(wire, monkeypatch):
    """The receiver's side with a peer simulated in this process (while it fetches, this process answers to another tagged
    address -- the pid in the descriptor is the receiver's OWN, as for a peer in another container): the handle is opened
    once and cached, every fetch is acknowledged to the exporter's socket, a cached mapping that fails the canary check is
    unmapped, reopened and read again, and the exporter releases the buffer after the acknowledgement."""
    ipcwire, rt, sent = wire
    assert ipcwire.ensure_runtime_hooks()
    monkeypatch.setattr(ipcwire, '_opened', type(ipcwire._opened)())
    monkeypatch.setattr(ipcwire, '_ack_socks', {})
    lib = RemoteLib()

    class Ctx(FakeCtx):
        _L = lib
    r = rt.Runtime()
    buf = torch.arange(60000, dtype=torch.int64) * 3                       # ONE allocation holding three rows
    base_handle = struct.pack('<Q', buf.data_ptr()) + bytes(56)
    for j in range(3):
        row = Row(buf[20000 * j:20000 * (j + 1)])
        blob = rt.pickle.dumps(row)
        r._send_message(1, blob)
        desc = list(pickle.loads(blob))
        assert desc[0] == os.getpid()                                      # same pid, same export counter as a peer could have
        desc[2], desc[3] = base_handle, 8 * 20000 * j                      # (handle of the allocation, offset of the row)
        if j == 2:
            lib.stale_next = True                                          # the cached mapping fails the canary check once
        mine = ipcwire._sock_addr
        ipcwire._sock_addr = mine[:-4] + b'beef'                           # "another process": identity is the tagged address
        try:
            t = ipcwire.fetch(Ctx(), tuple(desc), reduce_n=20000 if j == 1 else None)
        finally:
            ipcwire._sock_addr = mine
        assert torch.equal(t, row.t), j
    assert lib.opens == 2 and lib.closes == 1                              # opened once, cached, reopened after the stale read
    assert ipcwire.stats.get('stale') == 1
    assert ipcwire.stats['imported'] == 3
    ipcwire.drain()
    assert not ipcwire._pending and ipcwire.stats['released'] == 3


def test_acknowledgements_never_block(wire, monkeypatch):
    """2000 acknowledgements to an exporter that is not collecting them (a datagram queue holds a few hundred): none of the
    sends may block, what does not fit waits in the backlog, and after the exporter drains everything arrives."""
    ipcwire, rt, sent = wire
    monkeypatch.setattr(ipcwire, '_ack_socks', {})
    monkeypatch.setattr(ipcwire, '_ack_backlog', {})
    ipcwire._ack_socket()
    n = 2000
    for eid in range(1, n + 1):
        ipcwire._pending[eid] = [None, 1, 0, False]
    import time
    t0 = time.time()
    for eid in range(1, n + 1):
        ipcwire._acknowledge(4242, ipcwire._sock_addr, eid)
    assert time.time() - t0 < 5.0
    assert len(ipcwire._ack_backlog[4242]) > 0                     # the queue was full at some point
    for _ in range(50):
        ipcwire.drain()                                            # collects, then flushes the backlog into the freed queue
        if not ipcwire._pending:
            break
    assert not ipcwire._pending and not ipcwire._ack_backlog[4242] and ipcwire.stats['released'] == n
