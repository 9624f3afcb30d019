"""ctypes binding of oracle/libgtg_oracle.so (the C restatement in mst_oracle.c).
TEST INFRASTRUCTURE ONLY; never imported by the product path."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libgtg_oracle.so")
    src = os.path.join(_HERE, "mst_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgtg_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libgtg_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        for fn in (_LIB.gtg_oracle_mst_full, _LIB.gtg_oracle_mst_triu):
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return _LIB


def _run(fn, arr: np.ndarray, g: int, n: int):
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    edges = np.empty((g, max(n - 1, 0), 2), dtype=np.int32)
    count = np.empty((g,), dtype=np.int32)
    rc = fn(arr.ctypes.data, g, n, edges.ctypes.data, count.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle mst failed rc={rc}")
    return edges, count


def mst_full(mats: np.ndarray):
    """mats (G,N,N) float32 -> (edges (G,N-1,2) int32 padded with -1, count (G,))."""
    g, n, _ = mats.shape
    return _run(lib().gtg_oracle_mst_full, mats, g, n)


def mst_triu(rows: np.ndarray, n: int):
    """rows (G, n(n+1)/2) float32 -> (edges, count)."""
    g = rows.shape[0]
    assert rows.shape[1] == n * (n + 1) // 2
    return _run(lib().gtg_oracle_mst_triu, rows, g, n)
