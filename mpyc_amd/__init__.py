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


def install():
    """Substitute the GPU array type and sharing functions into an importable `mpyc`
    (INTEGRATION.md section 2).  Must run before any field is created: mpyc caches array types
    per field (finfields.py:45,347).  Returns the list of substituted names."""
    from mpyc import finfields, thresha          # ImportError if mpyc is not installed
    from . import finfields as gff, thresha as gth
    finfields.PrimeFieldArray = gff.FieldArray
    finfields.BinaryFieldArray = gff.FieldArray
    done = ['finfields.PrimeFieldArray', 'finfields.BinaryFieldArray']
    for name in ('np_random_split', 'np_recombine', 'random_split', 'recombine', '_recombination_vector'):
        setattr(thresha, name, getattr(gth, name))
        done.append('thresha.' + name)
    return done
