"""Minimal stand-ins for `mitsuba` and `drjit` that let the reference's OWN test functions (read from
/root/reference at test time, never copied) run against this repo's implementations: every `mi.*` object used by
the selected tests is backed by the CPU oracle (backend "oracle") or by the product's HAR_HD code compiled for the
host (backend "product"), `dr.*` array helpers are numpy.  Test infrastructure only."""
import ctypes as C
import sys
import types

import numpy as np


def make_modules(backend, O=None, H=None):
    """backend = "oracle" (needs O = oracle.oracle) or "product" (needs H = host harness CDLL + O for scene building)"""
    dr = types.ModuleType("drjit")
    dr.pi = np.float32(np.pi); dr.inv_pi = np.float32(1 / np.pi)
    dr.cos = lambda x: np.cos(np.asarray(x, np.float32)).astype(np.float32)
    dr.sin = lambda x: np.sin(np.asarray(x, np.float32)).astype(np.float32)
    dr.full = lambda t, v, n: np.full(n, v, np.float32)
    # drjit.linspace: fma(arange(n), step, start) in float32 -- one rounding per element, so 20 steps to float32(pi / 2) end BELOW pi / 2 (np.linspace's end point
    # lies above it: its cosine is negative, i.e. a ray from the other side of the interface; src/render/tests/test_fresnel.py:70-75 depends on the difference)
    dr.linspace = lambda t, a, b, n, endpoint=True: (np.arange(n, dtype=np.float64) * float(np.float32((float(b) - float(a)) / (n - 1 if endpoint else n)))
                                                       + float(np.float32(a))).astype(np.float32)

    def meshgrid(a, b):
        x, y = np.meshgrid(a, b)
        return x.ravel(), y.ravel()
    dr.meshgrid = meshgrid

    def allclose(a, b, rtol=1e-5, atol=1e-8):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        return bool(np.allclose(a, b, rtol=rtol, atol=atol))
    dr.allclose = allclose
    dr.abs = np.abs
    dr.sqrt = lambda x: np.sqrt(np.asarray(x, np.float64))
    dr.acos = lambda x: np.arccos(np.clip(np.asarray(x, np.float64), -1.0, 1.0))
    dr.all = lambda x, axis=None: bool(np.all(x))
    dr.zeros = lambda t, n=1: np.zeros(n, np.float32)

    mi = types.ModuleType("mitsuba")
    mi.Float = lambda x: np.asarray(x, np.float32)

    class Vector3f(list):
        def __init__(self, *a):
            super().__init__(a[0] if len(a) == 1 else a)
    mi.Vector3f = Vector3f

    class MicrofacetType:
        Beckmann = 0; GGX = 1
    mi.MicrofacetType = MicrofacetType

    def _vec(v, n=None):
        v = [np.asarray(c, np.float32).reshape(-1) for c in v]
        n = n or max(c.size for c in v)
        return np.stack([np.broadcast_to(c, n) for c in v], 1).astype(np.float32)      # [n][3]

    class MicrofacetDistribution:
        def __init__(self, type, alpha_u, alpha_v=None, sample_visible=True):
            if isinstance(alpha_v, bool):
                sample_visible = alpha_v; alpha_v = None
            self.t = type; self.au = float(alpha_u); self.av = float(alpha_u if alpha_v is None else alpha_v); self.sv = int(bool(sample_visible))

        def alpha_u(self): return max(self.au, 1e-4)
        def alpha_v(self): return max(self.av, 1e-4)
        def is_isotropic(self): return self.au == self.av
        def is_anisotropic(self): return self.au != self.av

        def _eval3(self, wi, m):
            n = max(max(np.asarray(c).size for c in m), max(np.asarray(c).size for c in wi))
            M = _vec(m, n); W = _vec(wi, n); out = np.empty((n, 3), np.float32)
            for i in range(M.shape[0]):
                o = np.empty(3, np.float32)
                if backend == "oracle":
                    O.lib().orc_microfacet_eval(self.t, C.c_float(self.au), C.c_float(self.av), self.sv, O.fp(np.ascontiguousarray(W[i])), O.fp(np.ascontiguousarray(M[i])), O.fp(o))
                else:
                    H.hh_microfacet_eval(self.t, C.c_float(self.au), C.c_float(self.av), self.sv, O.fp(np.ascontiguousarray(W[i])), O.fp(np.ascontiguousarray(M[i])), O.fp(o))
                out[i] = o
            return out

        def eval(self, m): return self._eval3([0, 0, 1], m)[:, 0]
        def pdf(self, wi, m): return self._eval3(wi, m)[:, 1]
        def smith_g1(self, v, m): return self._eval3(v, m)[:, 2]          # smith_g1(v, m)

        def sample(self, wi, u):
            U = np.stack([np.asarray(u[0], np.float32).ravel(), np.asarray(u[1], np.float32).ravel()], 1)
            W = _vec(wi, U.shape[0]); ms = np.empty((U.shape[0], 3), np.float32); pdfs = np.empty(U.shape[0], np.float32)
            for i in range(U.shape[0]):
                m = np.empty(3, np.float32); p = C.c_float()
                fn = O.lib().orc_microfacet_sample if backend == "oracle" else H.hh_microfacet_sample
                fn(self.t, C.c_float(self.au), C.c_float(self.av), self.sv, O.fp(np.ascontiguousarray(W[i])), O.fp(np.ascontiguousarray(U[i])), O.fp(m), C.byref(p))
                ms[i] = m; pdfs[i] = p.value
            return [ms[:, 0], ms[:, 1], ms[:, 2]], pdfs
    mi.MicrofacetDistribution = MicrofacetDistribution

    # fresnel(cos_theta_i, eta) -> (F, cos_theta_t, eta_it, eta_ti) and fresnel_conductor(cos_theta_i, eta [complex]) (include/mitsuba/render/fresnel.h:38-116)
    def fresnel(cos_theta_i, eta):
        c = np.asarray(cos_theta_i, np.float32).reshape(-1); out = np.empty((c.size, 4), np.float32)
        for i in range(c.size):
            o = np.empty(4, np.float32)
            if backend == "oracle":
                O.lib().orc_fresnel.argtypes = [C.c_float, C.c_float, O.c_f32p]; O.lib().orc_fresnel(C.c_float(c[i]), C.c_float(eta), O.fp(o))
            else:
                H.hh_fresnel.argtypes = [C.c_float, C.c_float, O.c_f32p]; H.hh_fresnel(C.c_float(c[i]), C.c_float(eta), O.fp(o))
            out[i] = o
        if np.ndim(cos_theta_i) == 0:
            return tuple(float(v) for v in out[0])
        return out[:, 0], out[:, 1], out[:, 2], out[:, 3]
    mi.fresnel = fresnel

    def fresnel_conductor(cos_theta_i, eta):
        c = np.asarray(cos_theta_i, np.float32).reshape(-1); e = complex(eta); out = np.empty(c.size, np.float32)
        fn = O.lib().orc_fresnel_conductor if backend == "oracle" else H.hh_fresnel_conductor
        fn.restype = C.c_float; fn.argtypes = [C.c_float, C.c_float, C.c_float]
        for i in range(c.size):
            out[i] = fn(C.c_float(c[i]), C.c_float(e.real), C.c_float(e.imag))
        return float(out[0]) if np.ndim(cos_theta_i) == 0 else out
    mi.fresnel_conductor = fresnel_conductor

    # DiscreteDistribution (include/mitsuba/core/distr_1d.h:27-215) as the mesh area lights use it (face choice with sample re-use); oracle only
    class DiscreteDistribution:
        def __init__(self, pmf):
            self.pmf = np.asarray(pmf, np.float32)

        def _run(self, values):
            v = np.ascontiguousarray(np.asarray(values, np.float32).reshape(-1)); n = v.size
            idx = np.zeros(n, np.uint32); reused = np.zeros(n, np.float32); pmf = np.zeros(n, np.float32)
            L = O.lib()
            L.orc_discrete_sample_reuse.argtypes = [O.c_f32p, C.c_uint32, C.c_uint32, O.c_f32p, C.POINTER(C.c_uint32), O.c_f32p, O.c_f32p]
            L.orc_discrete_sample_reuse(O.fp(np.ascontiguousarray(self.pmf)), self.pmf.size, n, O.fp(v), idx.ctypes.data_as(C.POINTER(C.c_uint32)), O.fp(reused), O.fp(pmf))
            return idx.astype(np.int64), reused, pmf

        def sample(self, values): return self._run(values)[0]
        def sample_pmf(self, values): i, _, p = self._run(values); return i, p
        def sample_reuse(self, values): i, r, _ = self._run(values); return i, r
        def sample_reuse_pmf(self, values): return self._run(values)
    mi.DiscreteDistribution = DiscreteDistribution
    return mi, dr


def run_reference_tests(path, names, mi, dr, extra=None):
    """exec the reference test file with the shim modules and call the named test functions (fixtures get None)"""
    import inspect
    src = open(path).read()
    saved = {k: sys.modules.get(k) for k in ("mitsuba", "drjit", "pytest")}
    sys.modules["mitsuba"] = mi; sys.modules["drjit"] = dr
    try:
        ns = {"__name__": "reference_test"}
        if extra:
            ns.update(extra)
        exec(compile(src, path, "exec"), ns)
        ran = []
        for n in names:
            fn = ns[n]
            fn(*[None] * len(inspect.signature(fn).parameters))
            ran.append(n)
        return ran
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
