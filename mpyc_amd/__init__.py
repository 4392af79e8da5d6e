"""mpyc_amd -- MI355X-native finite-field / secret-sharing engine behind MPyC's
field-array (mpyc.finfields) and threshold-sharing (mpyc.thresha) interfaces.

Layout:
    csrc/          HIP kernels (gfx950) + the C ABI -> libffgpu.so   (include/ffgpu.h)
    _ffi.py        ctypes binding of the C ABI (no fallback path)
    engine.py      field context + device-resident limb arrays
    finfields.py   host-side mirror of mpyc.finfields GF()/array types on the engine
    thresha.py     host-side mirror of mpyc.thresha (np_)random_split / (np_)recombine
"""
__version__ = '0.1.0'
