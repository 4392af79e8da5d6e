"""Where the time of a sliced PRSS call goes (engine.prss_streamed): squeeze alone into pageable / pinned memory, squeeze +
upload, the whole call, for several slice sizes.  m = 7, t = 3 (20 subset keys), 28 bytes per draw, n = 10^7."""
import ctypes, itertools, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from mpyc_amd import finfields, thresha, engine, _ffi
F = finfields.GF(2**61 - 1)
m, t, i = 7, 3, 2
keys = {S: bytes([sum(S) % 256]) * 16 + bytes(S) for S in itertools.combinations(range(m), m - t) if i in S}
prfs = {S: thresha.PRF(k, F.order) for S, k in keys.items()}
msgs = [p.key + b'uci' for p in prfs.values()]
k = len(msgs)
L = _ffi.lib()
n = 10_000_000
per = 28


def squeeze_loop(dst_ptrs_of_slice, slice_bytes, total):
    keep = [ctypes.create_string_buffer(mg, len(mg)) for mg in msgs]
    mp = (ctypes.c_void_p * k)(*[ctypes.addressof(b) for b in keep]); ml = (ctypes.c_size_t * k)(*[len(mg) for mg in msgs])
    h = ctypes.c_void_p(); assert L.ffgpu_shake128_open(mp, ml, k, ctypes.byref(h)) == 0
    t0 = time.perf_counter(); done = 0; c = 0
    while done < total:
        nb = min(slice_bytes, total - done)
        op = (ctypes.c_void_p * k)(*dst_ptrs_of_slice(c))
        assert L.ffgpu_shake128_squeeze(h, op, nb, 0) == 0
        done += nb; c += 1
    dt = time.perf_counter() - t0
    L.ffgpu_shake128_close(h)
    return dt


for sb in (2 << 20, 8 << 20, 32 << 20):
    pitch = sb
    page = np.zeros(2 * k * pitch, dtype=np.uint8)
    pin = torch.zeros(2 * k * pitch, dtype=torch.uint8).pin_memory()
    for name, base in (('pageable', page.ctypes.data), ('pinned', pin.data_ptr())):
        dt = squeeze_loop(lambda c: [base + (c & 1) * k * pitch + j * pitch for j in range(k)], sb, n * per)
        print(f'slice {sb >> 20} MiB: squeeze only into {name}: {dt*1e3:.0f} ms = {k*n*per/dt/1e9:.1f} GB/s')
    del page, pin
    thresha.np_pseudorandom_share(F, m, i, prfs, b'warm', 1000)
    engine.FieldContext.PRSS_SLICE_BYTES = sb
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        thresha.np_pseudorandom_share(F, m, i, prfs, b'uci', n)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'slice {sb >> 20} MiB: whole call {dt*1e3:.0f} ms ({n/dt/1e6:.1f} M shares/s)')
