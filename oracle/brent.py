"""Oracle (test infrastructure): bounded scalar minimisation ("fminbnd"), restated.

The reference calls ``scipy.optimize.minimize_scalar(method='Bounded')``
(observer.py:445,460,472,485); scipy is a third-party dependency that is not in
the reference tree (unpinned there; 1.15.3 installed in this image).  Its
bounded method is the classic Forsythe/Malcolm/Moler ``fmin`` (Brent's
golden-section + successive parabolic interpolation restricted to [a, b]) with
``xatol=1e-5`` and ``maxiter=500``.  This file restates that published
algorithm as a resumable state machine -- the same shape the device code in
``outlier_suppression_amd/csrc/msefast.hip`` uses -- so the device
implementation can be checked step for step.  ``tests/test_oracle_pinning.py``
pins it against scipy itself (iterates and evaluation count).
"""
import math

_GOLD = 0.5 * (3.0 - math.sqrt(5.0))
_SQRT_EPS = math.sqrt(2.2e-16)


class BoundedBrent:
    """Ask/tell form: ``x = b.start(); while not b.done: x = b.tell(f(x))``."""

    def __init__(self, lo, hi, xatol=1e-5, maxiter=500, f32_values=False):
        """f32_values: the objective returns np.float32 (the reference's loss_fx does, observer.py:431-432).  scipy then
        forms ``fx - ffulc`` / ``fx - fnfc`` as float32 subtractions (numpy scalar arithmetic) before multiplying
        by the float64 abscissa differences; everything else is float64 either way."""
        self.a, self.b = float(lo), float(hi)
        self.xatol, self.maxfun = xatol, maxiter
        self.f32_values = f32_values
        self.done = False
        self.nfev = 0

    def start(self):
        a, b = self.a, self.b
        self.v = self.w = self.xf = a + _GOLD * (b - a)   # fulc, nfc, xf
        self.d = self.e = 0.0                             # rat, e
        self.fx = self.fv = self.fw = None
        self._pending = self.xf
        return self.xf

    def _tolerances(self):
        self.xm = 0.5 * (self.a + self.b)
        self.tol1 = _SQRT_EPS * abs(self.xf) + self.xatol / 3.0
        self.tol2 = 2.0 * self.tol1

    def _converged(self):
        return not (abs(self.xf - self.xm) > (self.tol2 - 0.5 * (self.b - self.a)))

    def _propose(self):
        a, b, xf, xm, tol1, tol2 = self.a, self.b, self.xf, self.xm, self.tol1, self.tol2
        use_golden = True
        if abs(self.e) > tol1:
            use_golden = False
            if self.f32_values:
                import numpy as np
                r = (xf - self.w) * float(np.float32(self.fx) - np.float32(self.fv))
                q = (xf - self.v) * float(np.float32(self.fx) - np.float32(self.fw))
            else:
                r = (xf - self.w) * (self.fx - self.fv)
                q = (xf - self.v) * (self.fx - self.fw)
            p = (xf - self.v) * q - (xf - self.w) * r
            q = 2.0 * (q - r)
            if q > 0.0:
                p = -p
            q = abs(q)
            r = self.e
            self.e = self.d
            if abs(p) < abs(0.5 * q * r) and p > q * (a - xf) and p < q * (b - xf):
                self.d = p / q
                x = xf + self.d
                if (x - a) < tol2 or (b - x) < tol2:
                    sgn = (1.0 if xm > xf else -1.0 if xm < xf else 0.0) + (1.0 if xm == xf else 0.0)
                    self.d = tol1 * sgn
            else:
                use_golden = True
        if use_golden:
            self.e = (a - xf) if xf >= xm else (b - xf)
            self.d = _GOLD * self.e
        sgn = (1.0 if self.d > 0 else -1.0 if self.d < 0 else 0.0) + (1.0 if self.d == 0 else 0.0)
        return xf + sgn * max(abs(self.d), tol1)

    def tell(self, fu):
        fu = float(fu)
        self.nfev += 1
        x = self._pending
        if self.fx is None:                      # first evaluation
            self.fx = self.fv = self.fw = fu
        else:
            if fu <= self.fx:
                if x >= self.xf:
                    self.a = self.xf
                else:
                    self.b = self.xf
                self.v, self.fv = self.w, self.fw
                self.w, self.fw = self.xf, self.fx
                self.xf, self.fx = x, fu
            else:
                if x < self.xf:
                    self.a = x
                else:
                    self.b = x
                if fu <= self.fw or self.w == self.xf:
                    self.v, self.fv = self.w, self.fw
                    self.w, self.fw = x, fu
                elif fu <= self.fv or self.v == self.xf or self.v == self.w:
                    self.v, self.fv = x, fu
            if self.nfev >= self.maxfun:
                self.done = True
                return None
        self._tolerances()
        if self._converged():
            self.done = True
            return None
        self._pending = self._propose()
        return self._pending

    @property
    def x(self):
        return self.xf

    @property
    def fun(self):
        return self.fx


def minimize_bounded(func, lo, hi, xatol=1e-5, maxiter=500, f32_values=False):
    """Convenience driver.  Returns (x, f(x), nfev)."""
    st = BoundedBrent(lo, hi, xatol, maxiter, f32_values)
    x = st.start()
    while x is not None:
        x = st.tell(func(x))
    return st.x, st.fun, st.nfev
