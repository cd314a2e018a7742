"""Headless-safe stand-in for src/libplot.py (matplotlib sugar, not on the hot path): ``from libplot import lp``."""


class _NoPlot:
    def __getattr__(self, name):
        def _noop(*a, **k):
            return None

        return _noop


try:  # pragma: no cover
    import matplotlib

    matplotlib.use("Agg")
    from matplotlib import pyplot as lp

    def plotm(m_data):
        lp.figure()
        ret = lp.imshow(m_data.T, aspect="auto", origin="lower", interpolation="nearest")
        lp.colorbar(ret)

    lp.plotm = plotm
except Exception:  # pragma: no cover
    lp = _NoPlot()
