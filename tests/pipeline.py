"""Test-side interpreter of genodsp pipelines ("= op args = op args") over a backend.

The golden fixtures are recorded as the reference's own command lines; this maps
them onto either the CPU oracle (oracle.cpu) or the HIP library (genodsp_amd) so
both can be held against the same recorded outputs.  Only the flags the fixtures
use are understood; anything else raises.  Argument rules follow the reference's
parsers (cited inline).
"""
import shlex

import numpy as np

DBL_MAX = float(np.finfo(np.float64).max)
DBL_MIN = float(np.finfo(np.float64).tiny)


def to_value(s):
    """utilities.c:334-355 string_to_double: inf -> DBL_MAX, 1/inf -> DBL_MIN."""
    table = {"inf": DBL_MAX, "+inf": DBL_MAX, "-inf": -DBL_MAX, "1/inf": DBL_MIN, "+1/inf": DBL_MIN,
             "-1/inf": -DBL_MIN}
    return table[s] if s in table else float(s)


def to_int(s):
    """utilities.c:236-309 string_to_unitized_int, thousands: K/M/G = 10^3/6/9."""
    mult = 1
    if s[-1] in "KMG":
        mult = {"K": 10 ** 3, "M": 10 ** 6, "G": 10 ** 9}[s[-1]]
        s = s[:-1]
    return int(float(s) * mult) if "." in s else int(s) * mult


def split_ops(pipeline):
    toks = shlex.split(pipeline) if isinstance(pipeline, str) else list(pipeline)
    ops, cur = [], None
    for t in toks:
        if t.startswith("="):
            if cur is not None:
                ops.append(cur)
            cur = []
            if len(t) > 1:
                cur.append(t[1:].strip())
        else:
            cur.append(t)
    if cur is not None:
        ops.append(cur)
    return [(o[0], o[1:]) for o in ops]


def _kv(arg):
    k, _, v = arg.partition("=")
    return k, v


def _window_arg(args, default, odd=False, name="--window"):
    W = default
    for a in args:
        k, v = _kv(a)
        if k in (name, name[2].upper(), "--" + name[2].upper()) and v != "chromosome":
            W = to_int(v)
            if W < 3:
                W = 3                                   # e.g. sum.c:559-564
            if odd and W % 2 == 0:
                W += 1                                  # sum.c:565-570, minmax.c:1127-1132
    if odd and W % 2 == 0:
        W += 1                                          # sum.c:589-594
    return W


class Runner:
    """Runs ops on `backend` over a genome {name: vector}; processing order as the reference."""

    def __init__(self, backend, chroms, vectors, files=None):
        self.files = files or {}
        self.b = backend
        self.chroms = list(chroms)                      # [(name, length)] in file order
        self.v = {c: backend.load(vectors[c]) for c, _ in chroms}
        # longest first (genodsp.c:1113-1145); ties keep file order here, the reference's
        # qsort order for ties is unspecified and only matters for percentile's destroyed state
        self.sorted = [c for c, _ in sorted(chroms, key=lambda x: -x[1])]
        self.globals = {}

    def result(self, c):
        return self.b.store(self.v[c])

    def run(self, pipeline):
        for name, args in split_ops(pipeline):
            getattr(self, "op_" + name)(args)
        return self

    def _each(self, fn):
        for c in self.sorted:
            self.v[c] = fn(self.v[c])

    # ---- sum.c
    def op_smooth(self, args):
        W = _window_arg(args, 101, odd=True)
        self._each(lambda x: self.b.smooth(x, W))

    def op_slidingsum(self, args):
        W = _window_arg(args, 100)
        denom = 1.0
        for a in args:
            k, v = _kv(a)
            if k in ("--denom", "--denominator", "D", "--D"):
                denom = float(W) if v in ("window", "W") else to_value(v)
        self._each(lambda x: self.b.sliding_sum(x, W, denom))

    def op_sum(self, args):
        W = _window_arg(args, 100)
        whole = "--window=chromosome" in args
        denom, actual, zero, denom_is_w = 1.0, False, 0.0, False
        for a in args:
            k, v = _kv(a)
            if k in ("--denom", "--denominator", "D", "--D"):
                denom, actual, denom_is_w = 1.0, False, False
                if v == "actual":
                    actual = True
                elif v in ("window", "W"):
                    denom_is_w = True
                else:
                    denom = to_value(v)
            if k in ("--zero", "Z", "--Z"):
                zero = to_value(v)

        def f(x):
            w = self.b.length(x) if whole else W        # sum.c:225-226
            d = float(w) if denom_is_w else denom
            return self.b.window_sum(x, w, d, actual, zero)
        self._each(f)

    def op_cumulativesum(self, args):
        self._each(self.b.cumulative_sum)

    # ---- minmax.c
    def _local(self, args, want_max):
        N = _window_arg(args, 3, odd=True, name="--neighborhood")
        fill = 0.0 if want_max else DBL_MAX             # minmax.c:1098, :901
        for a in args:
            k, v = _kv(a)
            if (want_max and k in ("--zero", "Z", "--Z")) or ((not want_max) and k == "--infinity"):
                fill = to_value(v)
        self._each(lambda x: self.b.local_extrema(x, N, want_max, fill))

    def op_localmax(self, args):
        self._local(args, True)

    def op_localmin(self, args):
        self._local(args, False)

    def op_bestmax(self, args):
        W = _window_arg(args, 100)
        self._each(lambda x: self.b.best_extrema(x, W, True))

    def op_bestmin(self, args):
        W = _window_arg(args, 100)
        self._each(lambda x: self.b.best_extrema(x, W, False))

    # ---- morphology.c
    def _morph_common(self, args):
        T, one, zero, length, left, right = 0.0, 1.0, 0.0, None, 0, 0
        for a in args:
            k, v = _kv(a)
            if k in ("--threshold", "T", "--T"):
                T = self.globals[v] if v in self.globals else to_value(v)
            elif k in ("--one", "O", "--O"):
                one = to_value(v)
            elif k in ("--zero", "Z", "--Z"):
                zero = to_value(v)
            elif k == "--left":
                left = to_int(v)
            elif k == "--right":
                right = to_int(v)
            elif not a.startswith("--"):
                length = to_int(a)
        return T, one, zero, length, left, right

    def _widen(self, args, fn):
        T, one, zero, length, left, right = self._morph_common(args)
        if left == 0 and right == 0:
            left = int(float(length) / 2)               # morphology.c:919-920
            right = length - left
        self._each(lambda x: fn(x, left, right, T, one, zero))

    def op_dilate(self, args):
        self._widen(args, self.b.dilate)

    def op_erode(self, args):
        self._widen(args, self.b.erode)

    def op_close(self, args):
        T, one, zero, length, _, _ = self._morph_common(args)
        self._each(lambda x: self.b.close(x, float(length), T, one, zero))

    def op_open(self, args):
        T, one, zero, length, _, _ = self._morph_common(args)
        self._each(lambda x: self.b.open_(x, float(length), T, one, zero))

    # ---- logical.c / mask.c / add.c
    def op_binarize(self, args):
        T, above, one, zero = 0.0, False, 1.0, 0.0
        for a in args:
            k, v = _kv(a)
            if k in ("--threshold", "T", "--T"):
                T = self.globals[v]                     # variable name only, logical.c:116-124
            elif a in ("--ties:above", "--ties=above"):
                above = True
            elif a in ("--ties:below", "--ties=below"):
                above = False
            elif k in ("--one", "O", "--O"):
                one = to_value(v)
            elif k in ("--zero", "Z", "--Z"):
                zero = to_value(v)
            else:
                T = to_value(a)
        self._each(lambda x: self.b.binarize(x, T, above, one, zero))

    def _limits(self, args):
        lo = hi = None
        inside, zero = False, 0.0
        for a in args:
            k, v = _kv(a)
            if k == "--min":
                lo = self.globals[v] if v in self.globals else to_value(v)
            elif k == "--max":
                hi = self.globals[v] if v in self.globals else to_value(v)
            elif a == "--keep:inside":
                inside = True
            elif a == "--keep:outside":
                inside = False
            elif k in ("--zero", "Z", "--Z"):
                zero = to_value(v)
        return lo, hi, inside, zero

    def op_clip(self, args):
        lo, hi, _, _ = self._limits(args)
        self._each(lambda x: self.b.clip(x, lo, hi))

    def op_erase(self, args):
        lo, hi, inside, zero = self._limits(args)
        self._each(lambda x: self.b.erase(x, lo, hi, inside, zero))

    def op_addconst(self, args):
        c = to_value(args[0])
        self._each(lambda x: self.b.add_constant(x, c))

    def op_abs(self, args):
        self._each(self.b.abs_)

    def op_invert(self, args):
        if args:
            mid = {"zero": 0.0, "negate": 0.0, "one": 1.0, "1/2": 0.5, "binary": 0.5}.get(args[0])
            if mid is None:
                mid = to_value(args[0])
        else:
            lo, hi = self.b.genome_minmax([self.v[c] for c in self.sorted])
            mid = (lo + hi) / 2.0                       # add.c:925
        self._each(lambda x: self.b.invert(x, mid))

    # ---- clump.c
    def _clump(self, args, above):
        avg, length, rel, one, zero = 0.0, 100, 0.0, 1.0, 0.0
        for a in args:
            k, v = _kv(a)
            if k in ("--average", "T", "--T"):
                avg = self.globals[v]
            elif k in ("--length", "L", "--L"):
                length, rel = parse_clump_length(v)
            elif k in ("--one", "O", "--O"):
                one = to_value(v)
            elif k in ("--zero", "Z", "--Z"):
                zero = to_value(v)
            elif not a.startswith("--"):
                avg = to_value(a)

        def run(x):
            n = self.b.length(x)
            L = max(length, int(np.uint32(rel * n))) if rel > 0 else length      # clump.c:512-518
            return self.b.clump(x, avg, L, above, one, zero)
        self._each(run)

    def op_clump(self, args):
        self._clump(args, True)

    def op_anticlump(self, args):
        self._clump(args, False)

    op_anti_clump = op_anticlump
    op_skimp = op_anticlump

    # ---- map.c
    def op_map(self, args):
        text = self.files[args[0].strip("@")]
        pairs = []
        for line in text.splitlines():
            line = line.strip()
            if line and not line.startswith("#"):
                a, b = line.split()[:2]
                pairs.append((to_value(a), to_value(b)))
        pairs.sort(key=lambda p: p[0])                  # the reference qsorts by the input value (map.c:452)
        kin = np.array([p[0] for p in pairs])
        kout = np.array([p[1] for p in pairs])
        self._each(lambda x: self.b.map_values(x, kin, kout))

    # ---- percentile.c
    def op_percentile(self, args):
        window, lo, hi = 1, -DBL_MAX, DBL_MAX
        spec = None
        for a in args:
            k, v = _kv(a)
            if k in ("--window", "W", "--W"):
                window = to_int(v)
            elif k == "--min":
                lo = to_value(v)
            elif k == "--max":
                hi = to_value(v)
            elif a in ("--quiet",) or k == "--preserve":
                pass
            elif not a.startswith("--"):
                spec = a
        pts = parse_percentile_spec(spec)
        count, vals = self.b.percentile([self.v[c] for c in self.sorted], pts, window, lo, hi)
        if count:
            for pt, val in zip(pts, vals):
                self.globals[percentile_name(pt)] = val


def parse_clump_length(text, max_ok=True):
    """clump.c:353-485: <n> | CL | CL*f | f*CL | CL/k | max(relative, n) -> (minLength, relativeLength)."""
    if max_ok and text.startswith("max(") and text.endswith(")"):
        a, b = text[4:-1].split(",", 1)
        la, ra = parse_clump_length(a, False)
        lb, rb = parse_clump_length(b, False)
        assert (ra > 0) != (rb > 0)
        return (lb, ra) if ra > 0 else (la, rb)
    if text == "CL":
        return 0, 1.0
    if text.startswith("CL*"):
        return 0, float(text[3:])
    if text.endswith("*CL"):
        return 0, float(text[:-3])
    if text.startswith("CL/"):
        return 0, 1.0 / float(text[3:])
    return to_int(text), 0.0


def parse_percentile_spec(spec):
    """percentile.c:131-375: <lo>[..<hi>][by<step>] or <lo>,<hi>; units of 0.001 %."""
    def th(s):
        return int(round(float(s) * 1000))
    if "," in spec:
        a, b = spec.split(",")
        return [th(a), th(b)]
    step = None
    if "by" in spec:
        spec, s = spec.split("by")
        step = th(s)
    if ".." in spec:
        a, b = spec.split("..")
        lo, hi = th(a), th(b)
    else:
        lo = hi = th(spec)
    if step is None:
        step = 1000
    return list(range(lo, hi + 1, step)) if hi > lo else [lo]


def percentile_name(pt):
    """percentile.c:756-780: percentile<p> with trailing zeros of the fraction removed."""
    whole, frac = divmod(pt, 1000)
    if frac == 0:
        return "percentile%d" % whole
    return ("percentile%d.%03d" % (whole, frac)).rstrip("0")
