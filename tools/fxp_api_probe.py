"""Secure FIXED-POINT multiplication through the reference runtime (np_multiply + np_trunc: random bits from PRSS, masked opening,\ninteger arithmetic on .value) on SecFxp(32) arrays (80-bit field): API_MODE=ref | gpu, API_N elements, optional -M3."""
import os, sys, time
MODE=os.environ.get('API_MODE','ref')
if MODE!='ref':
    ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path[:0]=[ROOT+'/tests',ROOT]
    import mpyc_amd; mpyc_amd.install()
    if MODE=='cpuctx':
        from cpuctx import use_cpu_contexts; use_cpu_contexts()
import numpy as np
if MODE!='ref':
    import mpyc_amd.finfields as _gff, mpyc_amd.thresha as _gth
    if os.environ.get('FXP_LAZY_PRODUCTS')=='0': _gff.lazy_products=False
    if os.environ.get('FXP_NO_STREAM'): _gth.PRSS_STREAM_MIN=1<<62
from mpyc.runtime import mpc
N=int(os.environ.get('API_N','2000'))
if os.environ.get('FXP_CHECK'):
    from mpyc import asyncoro, runtime as _rt
    from mpyc.asyncoro import Future

    def _ints(x):
        return np.asarray(x.value if hasattr(x, 'value') else x, dtype=object)

    def _rep(name, got, want):
        bad = np.nonzero(got != want)[0]
        print(f'CHECK {name}: bad {len(bad)}', [(int(i), hex(int(got[i])), hex(int(want[i]))) for i in bad[:3]], flush=True)

    TOUCH = set(os.environ.get('FXP_TOUCH', '').split(','))

    @asyncoro.mpc_coro
    async def my_trunc(self, a, f=None, l=None):
        """runtime.py:838-873 verbatim, plus optional early materialisation of ONE intermediate (FXP_TOUCH)"""
        n = a.size
        sftype = type(a)
        await self.returnType((sftype, a.shape))
        Zp = sftype.sectype.field
        l = l or sftype.sectype.bit_length
        if f is None:
            f = sftype.frac_length
        if not os.environ.get('FXP_REF_L'):      # the reference adds f only for SCALAR fixed-point types (issubclass test, runtime.py:853)
            l += f
        k = self.options.sec_param
        r_bits = await self.np_random_bits(Zp, f * n)
        if 'bits' in TOUCH: _ints(r_bits)
        ar_modf = np.sum(r_bits.value.reshape((n, f)) << np.arange(f), axis=1)
        ar_modf = ar_modf.reshape(a.shape)
        r_divf = self._np_randoms(Zp, n, 1 << k + l - f)
        if 'rdiv' in TOUCH: _ints(r_divf)
        r_divf = r_divf.value
        r_divf = r_divf.reshape(a.shape)
        a = await self.gather(a)
        if 'prod' in TOUCH: _ints(a)
        ar_modf += a.value
        cc = Zp.array(ar_modf + (1 << l-1) + (r_divf << f))
        if 'cc' in TOUCH: _ints(cc)
        c = await self.output(cc)
        if 'c' in TOUCH: _ints(c)
        c = c.value & ((1<<f) - 1)
        pre = Zp.array(ar_modf - c)
        if 'pre' in TOUCH: _ints(pre)
        y = pre >> f
        if 'y' in TOUCH: _ints(y)
        return y
    _rt.Runtime.np_trunc = my_trunc
async def main():
    await mpc.start()
    secfxp=mpc.SecFxp(32)
    rng=np.random.default_rng(5)
    xa=rng.uniform(-100,100,N); xb=rng.uniform(-100,100,N)
    a=mpc.input(secfxp.array(xa), senders=0); b=mpc.input(secfxp.array(xb), senders=0)
    await mpc.gather(a,b)
    prof=None
    if os.environ.get('FXP_CPROFILE') and mpc.pid==0:
        import cProfile; prof=cProfile.Profile(); prof.enable()
    t0=time.perf_counter()
    c=a*b
    y=await mpc.output(c)
    if os.environ.get('FXP_RAW'):
        yr = await mpc.output(c, raw=True)
        pmod = secfxp.field.modulus
        vals = yr.value if MODE == 'ref' else np.asarray(yr.value)
        flo = np.asarray(y, dtype=float)
        badi = np.nonzero(np.abs(flo - xa*xb) > 0.01)[0][:6]
        for i in badi.tolist():
            print('RAW idx', i, 'field value', hex(int(vals[i])), 'p - v', hex(pmod - int(vals[i])), 'float out', flo[i], 'want', xa[i]*xb[i],
                  'limbs', (yr._dev.t[i].tolist() if MODE != 'ref' else None), 'to_ints', hex(yr._dev.to_ints()[i]) if MODE != 'ref' else None)
        good = 5
        print('RAW good idx', good, hex(int(vals[good])), 'float', flo[good], 'want', xa[good]*xb[good])
    dt=time.perf_counter()-t0
    if prof is not None:
        prof.disable()
        import pstats, io
        st=io.StringIO(); pstats.Stats(prof,stream=st).sort_stats('tottime').print_stats(22); print(st.getvalue()[:4500])
    diff=np.abs(np.asarray(y,dtype=float)-xa*xb)
    err=float(np.max(diff))
    bad=np.nonzero(diff>0.01)[0]
    if len(bad):
        print('BAD count', len(bad), 'first', bad[:8].tolist(), 'last', bad[-8:].tolist(), 'min', int(bad.min()), 'max', int(bad.max()),
              'values', np.asarray(y,dtype=float)[bad[:4]].tolist(), 'want', (xa*xb)[bad[:4]].tolist())
    print('RESULT', MODE, N, round(dt,4), 'maxerr', err, 'field bits', secfxp.field.order.bit_length())
    await mpc.shutdown()
mpc.run(main())
