"""mpyc_amd -- MI355X-native finite-field / secret-sharing engine behind MPyC's
field-array (mpyc.finfields) and threshold-sharing (mpyc.thresha) interfaces.

Layout:
    csrc/          HIP kernels (gfx950) + the C ABI -> libffgpu.so   (include/ffgpu.h)
    _ffi.py        ctypes binding of the C ABI (no fallback path)
    engine.py      field context + device-resident limb arrays
    finfields.py   host-side mirror of mpyc.finfields GF()/array types on the engine
    thresha.py     host-side mirror of mpyc.thresha (np_)random_split / (np_)recombine
    install()      substitution of both into an importable mpyc
"""
__version__ = '0.3.0'

list_path_min = 256     # install(): thresha.random_split / recombine (the per-element list path, thresha.py:23-44,
#                         88-116) go to the device from this many secrets on; below it the reference's own
#                         pure-Python function runs (a kernel launch + two PCIe hops per scalar is not a speed-up)
_installed = None


def install():
    """Substitute the GPU array type and sharing functions into an importable `mpyc`
    (INTEGRATION.md section 2).  Must run before any field is created: mpyc caches array types
    per field (finfields.py:45,347).  Returns the list of substituted names.

    * `finfields.arrayGF` (finfields.py:45-60) is wrapped: for every prime field of up to 128 bits, the 129..192-bit primes 2^k - c (c < 2^31), and every
      GF(2^n), n <= 128, the array type derives from BOTH mpyc_amd.finfields.FieldArray (behaviour, device
      storage) and mpyc's own finfields.FiniteFieldArray (so `isinstance(x, finfields.FiniteFieldArray)`,
      sectypes.py:1372, holds); its inherited `value` slot is shadowed by the lazy device-backed property.
      Fields the device path does not cover (other wide primes, odd-characteristic extension fields) keep the
      reference's own array classes.
    * `thresha.np_random_split / np_recombine / np_pseudorandom_share(_0)` are replaced for those fields; the
      list-path functions from `list_path_min` secrets on.
    * PRSS keeps the reference's SHAKE128 PRF (bit-exact) unless MPYC_AMD_PRSS_PRF=chacha is set for EVERY party: then the
      draws come from a counter-mode PRF expanded on the device (mpyc_amd/thresha.py).
    """
    global _installed
    if _installed is not None:
        return list(_installed)
    import functools
    from mpyc import finfields, thresha          # ImportError if mpyc is not installed
    from . import finfields as gff, thresha as gth

    class DeviceFieldArray(gff.FieldArray, finfields.FiniteFieldArray):
        """field.array base class under install(): mpyc_amd behaviour on mpyc's own class hierarchy."""
        __slots__ = gff._STORAGE_SLOTS

    DeviceFieldArray._mix_types = (int, gff.np.integer)
    orig_arrayGF = finfields.arrayGF

    def supported(field):
        try:
            ops = gff._fops(field)
        except NotImplementedError:
            return False
        return ops.modulus.bit_length() <= 129 if ops.binary else gff.device_supports_prime(ops.modulus)

    picked = []

    def pick_device():
        """One party process per GPU: with several GPUs visible, party i computes on GPU i mod count (MPYC_AMD_DEVICE=
        <index> pins one, MPYC_AMD_DEVICE=current keeps torch's current device).  Decided once, when the first field is
        made -- mpyc.runtime has parsed the party index by then."""
        if picked:
            return
        import os
        import sys
        import torch
        want = os.environ.get('MPYC_AMD_DEVICE', 'party')
        if not torch.cuda.is_available() or want == 'current':
            picked.append(True)
            return
        if want.isdigit():
            picked.append(True)
            torch.cuda.set_device(int(want))
            return
        rt = sys.modules.get('mpyc.runtime')
        mpc = getattr(rt, 'mpc', None)
        if mpc is None:
            return                      # (a field made while mpyc.runtime is still being imported: decide at the next one)
        picked.append(True)
        count = torch.cuda.device_count()
        if count > 1 and len(getattr(mpc, 'parties', ())) > 1:
            torch.cuda.set_device(mpc.pid % count)

    @functools.cache
    def arrayGF(field, modulus):
        from . import ipcwire
        pick_device()
        if ipcwire.resolve_auto():
            ipcwire.ensure_runtime_hooks()       # (fields are made after mpyc.runtime is imported, before any gate runs)
        if not supported(field):
            return orig_arrayGF(field, modulus)
        array = type(f'Array{field.__name__}', (DeviceFieldArray,), {'__slots__': ()})
        array.field = field
        return array

    finfields.arrayGF = arrayGF
    gth.register_shake_prf(thresha.PRF)           # the runtime's PRF objects: expanded by the engine's host threads IF a
    #                                               known-answer check shows this mpyc's PRF is the shake_128 one
    finfields.DeviceFieldArray = DeviceFieldArray
    done = ['finfields.arrayGF']

    def on_device(field):
        arr = getattr(field, 'array', None)
        return arr is not None and issubclass(arr, gff.FieldArray)

    def route(name, min_len=None):
        ref_fn, gpu_fn = getattr(thresha, name), getattr(gth, name)

        @functools.wraps(ref_fn)
        def fn(field, *args, **kwargs):
            if not on_device(field):
                return ref_fn(field, *args, **kwargs)
            if min_len is not None:
                first = args[0]
                n = len(first) if name == 'random_split' else len(first[0][1])
                if n < list_path_min:
                    return ref_fn(field, *args, **kwargs)
            return gpu_fn(field, *args, **kwargs)
        fn.reference = ref_fn
        setattr(thresha, name, fn)
        done.append('thresha.' + name)

    for name in ('np_random_split', 'np_recombine', 'np_pseudorandom_share', 'np_pseudorandom_share_0'):
        route(name)
    for name in ('random_split', 'recombine'):
        route(name, min_len=list_path_min)
    _installed = done
    import os
    if os.environ.get('MPYC_AMD_TRACE_INSTALL') == '1':
        print(f'mpyc_amd.install: {len(done)} names substituted (pid {os.getpid()})', flush=True)
    return list(done)
