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
    dr.linspace = lambda t, a, b, n, endpoint=True: np.linspace(a, b, n, endpoint=endpoint, dtype=np.float64).astype(np.float32)

    def meshgrid(a, b):
        x, y = np.meshgrid(a, b)
        return x.ravel(), y.ravel()
    dr.meshgrid = meshgrid

    def allclose(a, b, rtol=1e-5, atol=1e-8):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        return bool(np.allclose(a, b, rtol=rtol, atol=atol))
    dr.allclose = allclose
    dr.abs = np.abs

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
