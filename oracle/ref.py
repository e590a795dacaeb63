"""ctypes binding of oracle/_ref/libgenodsp_ref.so (TEST INFRASTRUCTURE ONLY).

The shared object is the *unmodified* reference compiled from /root/reference by
`make -C oracle ref` behind oracle/ref_harness.c.  It is available in the build
container and -- as a prebuilt, git-ignored binary that travels with the gpurun
snapshot -- on the GPU box; nothing here reads /root/reference at run time.

Usage:
    g = ref.Genome([("chr1", 100), ("chr2", 50)])
    g.set("chr1", vec)                     # full-precision f64 in
    g.run("= smooth W=5 = localmax N=11")  # the reference's own parse + apply
    out = g.get("chr1")                    # full-precision f64 out
The reference keeps its state in C globals, so only one Genome is live at a time.
"""
import ctypes as C
import os
import shlex
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libgenodsp_ref.so")
CLI = os.path.join(_HERE, "_ref", "genodsp")
REF_SRC = "/root/reference"

_lib = None


def build():
    """Build oracle/_ref from the reference sources (build container only)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def available():
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libgenodsp_ref.so not built (make -C oracle ref)")
        L = C.CDLL(_SO)
        L.refh_reset.restype = None
        L.refh_add_chrom.restype = C.c_int
        L.refh_add_chrom.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32]
        L.refh_begin.restype = None
        L.refh_vector.restype = C.POINTER(C.c_double)
        L.refh_vector.argtypes = [C.c_char_p]
        L.refh_length.restype = C.c_uint32
        L.refh_length.argtypes = [C.c_char_p]
        L.refh_sorted_name.restype = C.c_char_p
        L.refh_sorted_name.argtypes = [C.c_int]
        L.refh_run.restype = None
        L.refh_run.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        L.refh_get_global.restype = C.c_int
        L.refh_get_global.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
        L.refh_set_global.restype = None
        L.refh_set_global.argtypes = [C.c_char_p, C.c_double]
        L.refh_read_intervals_file.restype = C.c_int
        L.refh_read_intervals_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        L.refh_report_file.restype = C.c_int
        L.refh_report_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


class Genome:
    def __init__(self, chroms):
        """chroms: [(name, length)] or [(name, start, length)] in chromosome-file order."""
        L = lib()
        L.refh_reset()
        self.names = []
        for c in chroms:
            name, start, length = (c[0], 0, c[1]) if len(c) == 2 else c
            if not L.refh_add_chrom(name.encode(), start, length):
                raise ValueError("duplicate chromosome " + name)
            self.names.append(name)
        L.refh_begin()

    def _view(self, name):
        L = lib()
        n = L.refh_length(name.encode())
        if n == 0:
            raise KeyError(name)
        p = L.refh_vector(name.encode())
        return np.ctypeslib.as_array(p, shape=(n,))

    def set(self, name, vec):
        self._view(name)[:] = np.asarray(vec, np.float64)

    def get(self, name):
        return self._view(name).copy()

    def sorted_names(self):
        """The reference's processing order (longest first, genodsp.c:1113-1145)."""
        out, i = [], 0
        while True:
            s = lib().refh_sorted_name(i)
            if s is None:
                return out
            out.append(s.decode())
            i += 1

    def run(self, pipeline):
        """pipeline: '= op args = op args' (string or token list)."""
        toks = shlex.split(pipeline) if isinstance(pipeline, str) else list(pipeline)
        arr = (C.c_char_p * len(toks))(*[t.encode() for t in toks])
        lib().refh_run(len(toks), arr)

    def get_global(self, name):
        v = C.c_double()
        ok = lib().refh_get_global(name.encode(), C.byref(v))
        return v.value if ok else None

    def set_global(self, name, val):
        lib().refh_set_global(name.encode(), val)

    def read_intervals(self, path, val_col=3, origin_one=False, overlap=0, clear=False, missing=0.0):
        """val_col is 0-based as in the reference (-1 = no value column, each interval counts 1)."""
        if not lib().refh_read_intervals_file(path.encode(), val_col, int(origin_one), overlap,
                                              int(clear), missing):
            raise IOError(path)

    def report(self, path, precision=0, no_values=False, collapse=True, uncovered=0, origin_one=False):
        if not lib().refh_report_file(path.encode(), precision, int(no_values), int(collapse),
                                      uncovered, int(origin_one)):
            raise IOError(path)

    def close(self):
        lib().refh_reset()


def run_cli(args, stdin_text=""):
    """Run the reference CLI binary; returns (returncode, stdout, stderr)."""
    p = subprocess.run([CLI] + list(args), input=stdin_text, capture_output=True, text=True)
    return p.returncode, p.stdout, p.stderr
