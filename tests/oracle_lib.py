"""ctypes access to the CHECKERS (test infrastructure only):
   oracle/liboracle.so        from-scratch C restatement of WFA2's gap-affine path
   oracle/_ref/libwfa2ref.so  the reference's own WFA2 sources compiled in place (optional)
   oracle/_ref/libwfacpuref.so  the reference's utils/wfa_cpu.c + utils/cigar.c (+ WFA2) compiled in place (optional)
Nothing under wfa-gpu_amd/ imports this module."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libwfa2ref.so")
REFCPU_SO = os.path.join(ORACLE_DIR, "_ref", "libwfacpuref.so")


class OracleStats(C.Structure):
    _fields_ = [("cells", C.c_int64), ("steps", C.c_int64), ("max_abs_k", C.c_int32), ("num_ops", C.c_int32)]


_oracle = None
_ref = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build()
        o = C.CDLL(ORACLE_SO)
        o.oracle_aligner_new.argtypes = [C.c_int, C.c_int, C.c_int]
        o.oracle_aligner_new.restype = C.c_void_p
        o.oracle_aligner_delete.argtypes = [C.c_void_p]
        o.oracle_score.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(OracleStats)]
        o.oracle_score.restype = C.c_int
        o.oracle_align.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t,
                                   C.POINTER(OracleStats)]
        o.oracle_align.restype = C.c_int
        o.oracle_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.POINTER(C.c_int64), C.c_int]
        o.oracle_batch.restype = C.c_int64
        o.oracle_pack2.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        o.oracle_pack2.restype = C.c_int
        o.oracle_check_cigar.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                         C.c_int, C.POINTER(C.c_int)]
        o.oracle_check_cigar.restype = C.c_int
        _oracle = o
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        r = C.CDLL(REF_SO)
        r.ref_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        r.ref_new.restype = C.c_void_p
        r.ref_delete.argtypes = [C.c_void_p]
        r.ref_run.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
        r.ref_run.restype = C.c_int
        r.ref_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                C.c_void_p, C.c_size_t, C.c_int]
        r.ref_batch.restype = C.c_int64
        _ref = r
    return _ref


def _offsets(meta):
    off = np.empty((len(meta), 4), dtype=np.int64)
    off[:, 0] = meta["pattern_offset"]
    off[:, 1] = meta["pattern_len"]
    off[:, 2] = meta["text_offset"]
    off[:, 3] = meta["text_len"]
    return np.ascontiguousarray(off)


def _cigar_stride(meta):
    return int(2 * (meta["pattern_len"].astype(np.int64) + meta["text_len"].astype(np.int64)).max(initial=0) + 16)


def _split(cbuf, n, stride):
    raw = cbuf.tobytes()
    return [raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode() for i in range(n)]


def oracle_batch(buf, meta, pen, cigar=True, nthreads=1):
    """-> (scores int32[n], cigars or None, total wavefront cells)"""
    o = oracle()
    n = len(meta)
    off = _offsets(meta)
    scores = np.zeros(n, dtype=np.int32)
    cells = C.c_int64(0)
    stride = _cigar_stride(meta) if cigar else 0
    cbuf = np.zeros(n * stride, dtype=np.uint8) if cigar else None
    buf = np.ascontiguousarray(buf)
    o.oracle_batch(buf.ctypes.data, off.ctypes.data, n, pen[0], pen[1], pen[2], scores.ctypes.data,
                   cbuf.ctypes.data if cigar else None, stride, C.byref(cells), nthreads)
    return scores, (_split(cbuf, n, stride) if cigar else None), cells.value


def band_ref_batch(buf, meta, pen, beta, lam, max_steps, nthreads=1):
    """The reference's adaptive-band distance kernel restated (oracle/band_oracle.c): -> scores int32[n], -1 where the pair
    did not finish inside the band within max_steps."""
    o = oracle()
    o.oracle_band_ref_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int]
    o.oracle_band_ref_batch.restype = C.c_int64
    n = len(meta)
    off = _offsets(meta)
    scores = np.zeros(n, dtype=np.int32)
    buf = np.ascontiguousarray(buf)
    o.oracle_band_ref_batch(buf.ctypes.data, off.ctypes.data, n, pen[0], pen[1], pen[2], beta, lam, max_steps, scores.ctypes.data, nthreads)
    return scores


def ref_batch(buf, meta, pen, cigar=True, memory_mode=0, nthreads=1):
    r = ref()
    n = len(meta)
    off = _offsets(meta)
    scores = np.zeros(n, dtype=np.int32)
    stride = _cigar_stride(meta) if cigar else 0
    cbuf = np.zeros(n * stride, dtype=np.uint8) if cigar else None
    buf = np.ascontiguousarray(buf)
    r.ref_batch(buf.ctypes.data, off.ctypes.data, n, pen[0], pen[1], pen[2], memory_mode, scores.ctypes.data,
                cbuf.ctypes.data if cigar else None, stride, nthreads)
    return scores, (_split(cbuf, n, stride) if cigar else None)


_refcpu = None


def have_refcpu():
    return os.path.exists(REFCPU_SO)


def refcpu_batch(buf, meta, pen, cigar=True, nthreads=1):
    """The reference's OWN CPU path as it calls it -- utils/wfa_cpu.c: compute_alignments_cpu_threaded /
    compute_distance_cpu_threaded, one WFA2 aligner per OpenMP thread, memory mode low -- over a batch in the reference
    layout (oracle/ref_cpu_shim.c).  -> (scores int32[n], cigars or None)"""
    global _refcpu
    if _refcpu is None:
        r = C.CDLL(REFCPU_SO)
        r.refcpu_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        r.refcpu_batch.restype = C.c_int64
        _refcpu = r
    n = len(meta)
    meta = np.ascontiguousarray(meta)
    assert meta.dtype.itemsize == 48, "records must be in the reference layout (sequence_pair_t)"
    scores = np.zeros(n, dtype=np.int32)
    stride = _cigar_stride(meta) if cigar else 0
    cbuf = np.zeros(n * stride, dtype=np.uint8) if cigar else None
    buf = np.ascontiguousarray(buf)
    done = _refcpu.refcpu_batch(buf.ctypes.data, meta.ctypes.data, n, pen[0], pen[1], pen[2], scores.ctypes.data,
                                cbuf.ctypes.data if cigar else None, stride, nthreads)
    if done != n:
        raise RuntimeError(f"refcpu_batch: {done} of {n} alignments computed")
    return scores, (_split(cbuf, n, stride) if cigar else None)


def oracle_pair(p, t, pen, cigar=True):
    o = oracle()
    al = o.oracle_aligner_new(*pen)
    st = OracleStats()
    try:
        if cigar:
            cap = 2 * (len(p) + len(t)) + 16
            out = C.create_string_buffer(cap)
            s = o.oracle_align(al, p, len(p), t, len(t), out, cap, C.byref(st))
            return s, out.value.decode(), st
        s = o.oracle_score(al, p, len(p), t, len(t), 0, C.byref(st))
        return s, None, st
    finally:
        o.oracle_aligner_delete(al)


def check_cigar(p, t, cigar, pen):
    cost = C.c_int(0)
    ok = oracle().oracle_check_cigar(p, len(p), t, len(t), cigar.encode(), pen[0], pen[1], pen[2], C.byref(cost))
    return bool(ok), cost.value


def read_alg(path):
    """WFA2 align_benchmark output: 'score<TAB>CIGAR' or just 'score' per line (scores negative)."""
    scores, cigars = [], []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            parts = line.split()
            scores.append(-int(parts[0]))
            cigars.append(parts[1] if len(parts) > 1 else None)
    return np.array(scores, dtype=np.int32), cigars


def read_alg_skip_comments(path):
    scores, cigars = [], []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            parts = line.split()
            scores.append(-int(parts[0]))
            cigars.append(parts[1] if len(parts) > 1 else None)
    return np.array(scores, dtype=np.int32), cigars
