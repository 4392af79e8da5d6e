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
__version__ = '0.4.0'

list_path_min = 256     # install(): thresha.random_split / recombine (the per-element list path, thresha.py:23-44,
#                         88-116) go to the device from this many secrets on; below it the reference's own
#                         pure-Python function runs (a kernel launch + two PCIe hops per scalar is not a speed-up)
_installed = None
prss_confirm_timeout = 30.0     # seconds a party in production-mode PRSS waits at start-up for its peers' mode tags
_start_hooked = []


def _hook_prss_confirmation():
    """Production-mode PRSS (mpyc_amd.thresha.prss_prf = 'chacha') is only correct when EVERY party runs it with the same
    round count: the holders of a subset key must expand it with one PRF, or the shares are inconsistent and openings are
    silently wrong.  So Runtime.start (runtime.py:231-299) is followed by one mpc.transfer of the mode tag among the
    parties -- in production mode only: in the default (parity) mode nothing is added, and engine parties interoperate
    with plain-mpyc parties.  A peer with another tag, or one that never answers (it runs the default mode and sends no
    tag), ends the start-up with a RuntimeError.  Hooked as soon as mpyc.runtime is imported (install() and every arrayGF
    call try)."""
    if _start_hooked:
        return True
    import sys
    rt = sys.modules.get('mpyc.runtime')
    if rt is None or not hasattr(rt, 'Runtime') or not hasattr(rt.Runtime, 'start'):
        return False
    from . import thresha as gth
    orig_start = rt.Runtime.start

    async def start(self):
        await orig_start(self)
        await _confirm_prss_mode(self, gth)
    start.__doc__ = orig_start.__doc__
    rt.Runtime.start = start
    _start_hooked.append(True)
    mpc = getattr(rt, 'mpc', None)
    if mpc is not None and getattr(mpc, 'start_time', None) and len(mpc.parties) > 1 and gth.prss_mode_tag() != 'shake':
        import logging
        logging.warning('mpyc_amd: production-mode PRSS (%s) installed after the runtime started: the parties have NOT '
                        'confirmed that they all run it', gth.prss_mode_tag())
    return True


def _hook_after_runtime_import():
    """mpyc.runtime is not imported yet (install() usually runs first): hook Runtime.start right after its import."""
    import sys

    class AfterRuntimeImport:
        @staticmethod
        def find_spec(name, path=None, target=None):
            if name != 'mpyc.runtime':
                return None
            sys.meta_path.remove(AfterRuntimeImport)        # the real spec comes from the remaining finders
            import importlib.util
            spec = importlib.util.find_spec(name)
            if spec is not None and spec.loader is not None and hasattr(spec.loader, 'exec_module'):
                run_module = spec.loader.exec_module

                def exec_module(module):
                    run_module(module)
                    _hook_prss_confirmation()
                spec.loader.exec_module = exec_module
            return spec
    sys.meta_path.insert(0, AfterRuntimeImport)


async def _confirm_prss_mode(runtime, gth):
    tag = gth.prss_mode_tag()
    if tag == 'shake' or len(runtime.parties) == 1 or getattr(runtime.options, 'no_prss', False):
        return
    import asyncio
    import logging
    logging.info(f'mpyc_amd: PRSS production mode {tag}: confirming with the other parties')
    try:
        tags = await asyncio.wait_for(asyncio.ensure_future(runtime.transfer(('mpyc_amd prss mode', tag))), prss_confirm_timeout)
    except asyncio.TimeoutError:
        raise RuntimeError(f'mpyc_amd: this party runs PRSS in production mode ({tag}) but a peer did not confirm its mode '
                           f'within {prss_confirm_timeout:.0f} s: set MPYC_AMD_PRSS_PRF / MPYC_AMD_PRSS_ROUNDS identically for '
                           'every party') from None
    theirs = [t[1] if isinstance(t, tuple) and len(t) == 2 and t[0] == 'mpyc_amd prss mode' else repr(t)[:40] for t in tags]
    if any(t != tag for t in theirs):
        raise RuntimeError(f'mpyc_amd: the parties disagree on the PRSS PRF: {theirs} (this party: {tag}); shares of the '
                           'same subset key would be inconsistent')


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
        _hook_prss_confirmation()
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
    if not _hook_prss_confirmation():
        _hook_after_runtime_import()
    import os
    if os.environ.get('MPYC_AMD_TRACE_INSTALL') == '1':
        print(f'mpyc_amd.install: {len(done)} names substituted (pid {os.getpid()})', flush=True)
    return list(done)
