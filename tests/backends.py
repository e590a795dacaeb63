"""Two interchangeable backends for tests/pipeline.py: the CPU oracle and the HIP library."""
import numpy as np

from oracle import cpu


class OracleBackend:
    """oracle.cpu -- the checker (CPU restatement of the reference)."""
    name = "oracle"

    def load(self, v):
        return np.array(v, np.float64)

    def store(self, x):
        return np.asarray(x)

    def length(self, x):
        return x.size

    smooth = staticmethod(cpu.smooth)
    sliding_sum = staticmethod(cpu.sliding_sum)
    window_sum = staticmethod(cpu.window_sum)
    cumulative_sum = staticmethod(cpu.cumulative_sum)
    local_extrema = staticmethod(cpu.local_extrema)
    best_extrema = staticmethod(cpu.best_extrema)
    dilate = staticmethod(cpu.dilate)
    erode = staticmethod(cpu.erode)
    close = staticmethod(cpu.close)
    open_ = staticmethod(cpu.open_)
    binarize = staticmethod(cpu.binarize)
    clip = staticmethod(cpu.clip)
    erase = staticmethod(cpu.erase)
    add_constant = staticmethod(cpu.add_constant)
    abs_ = staticmethod(cpu.abs_)
    invert = staticmethod(cpu.invert)
    genome_minmax = staticmethod(cpu.genome_minmax)
    map_values = staticmethod(cpu.map_values)
    clump = staticmethod(cpu.clump)

    def percentile(self, vecs, pts, window, lo, hi):
        return cpu.percentile(vecs, pts, window, lo, hi)


class GpuBackend:
    """genodsp_amd -- the product path, through the C ABI of libgenodsp_hip.so."""
    name = "hip"

    def __init__(self, fir_mode=0):
        import genodsp_amd as gd
        self.gd = gd
        self.fir_mode = fir_mode

    def load(self, v):
        return self.gd.DeviceVector.from_numpy(v)

    def store(self, x):
        return x.numpy()

    def length(self, x):
        return x.n

    def smooth(self, x, W):
        return self.gd.smooth(x, W, mode=self.fir_mode)

    def sliding_sum(self, x, W, denom):
        return self.gd.sliding_sum(x, W, denom)

    def window_sum(self, x, W, denom, actual, zero):
        return self.gd.window_sum(x, W, denom, actual, zero)

    def cumulative_sum(self, x):
        return self.gd.cumulative_sum(x)

    def local_extrema(self, x, N, want_max, fill):
        return self.gd.local_extrema(x, N, want_max, fill)

    def best_extrema(self, x, W, want_max):
        return self.gd.best_extrema(x, W, want_max)

    def dilate(self, x, left, right, T, one, zero):
        return self.gd.dilate(x, left, right, T, one, zero)

    def erode(self, x, left, right, T, one, zero):
        return self.gd.erode(x, left, right, T, one, zero)

    def close(self, x, length, T, one, zero):
        return self.gd.close(x, length, T, one, zero)

    def open_(self, x, length, T, one, zero):
        return self.gd.open_(x, length, T, one, zero)

    def binarize(self, x, T, above, one, zero):
        return self.gd.binarize(x, T, above, one, zero)

    def clip(self, x, lo, hi):
        return self.gd.clip(x, lo, hi)

    def erase(self, x, lo, hi, inside, zero):
        return self.gd.erase(x, lo, hi, inside, zero)

    def add_constant(self, x, c):
        return self.gd.add_constant(x, c)

    def abs_(self, x):
        return self.gd.abs_(x)

    def invert(self, x, mid):
        return self.gd.invert(x, mid)

    def map_values(self, x, kin, kout):
        return self.gd.map_values(x, kin, kout)

    def clump(self, x, avg, L, above, one, zero):
        return self.gd.clump(x, avg, L, above, one, zero)

    def genome_minmax(self, vecs):
        lo, hi, _ = self.gd.genome_minmax(vecs)
        return lo, hi

    def percentile(self, vecs, pts, window, lo, hi):
        return self.gd.percentile(vecs, pts, window, lo, hi)
