"""ctypes bindings of libwfagpu.so (the C-ABI in include/*.h) for tests and bench.py.

PyTorch is plumbing only: it owns device buffers and the stream; every compute
call goes through the C-ABI.  There is no Python/CPU fallback: importing the
library object fails loudly when the shared object has not been built.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG_DIR, "libwfagpu.so")
GEN_PATH = os.path.join(PKG_DIR, "libwfagen.so")


class SeqPair(C.Structure):  # sequence_pair_t, 48 bytes
    _fields_ = [("text_offset", C.c_size_t), ("pattern_offset", C.c_size_t),
                ("text_offset_packed", C.c_size_t), ("pattern_offset_packed", C.c_size_t),
                ("text_len", C.c_uint), ("pattern_len", C.c_uint), ("has_N", C.c_bool)]


class Penalties(C.Structure):  # affine_penalties_t
    _fields_ = [("x", C.c_int), ("o", C.c_int), ("e", C.c_int)]


class Cigar(C.Structure):  # wfa_cigar_t
    _fields_ = [("buffer", C.c_void_p), ("buffer_size", C.c_size_t), ("last_free_position", C.c_size_t)]


class AlignmentResult(C.Structure):  # wfa_alignment_result_t
    _fields_ = [("error", C.c_uint), ("cigar", Cigar)]


class Options(C.Structure):  # wfa_alignment_options_t
    _fields_ = [("max_error", C.c_int), ("threads_per_block", C.c_int), ("num_workers", C.c_int),
                ("band", C.c_int), ("batch_size", C.c_size_t), ("num_alignments", C.c_size_t),
                ("penalties", Penalties), ("compute_cigar", C.c_bool)]


class Aligner(C.Structure):  # wfagpu_aligner_t
    _fields_ = [("sequences_buffer", C.c_void_p), ("sequences_buffer_len", C.c_size_t),
                ("sequences_metadata", C.POINTER(SeqPair)), ("sequences_metadata_len", C.c_size_t),
                ("num_sequence_pairs", C.c_size_t), ("results", C.POINTER(AlignmentResult)),
                ("last_sequence_pair_idx", C.c_int64), ("alignment_options", Options)]


class Tuning(C.Structure):  # wfagpu_amd_tuning_t: all zero = the defaults
    _fields_ = [("min_tier", C.c_int), ("careful_only", C.c_int), ("force_band", C.c_int), ("no_auto_budget", C.c_int),
                ("max_blocks_per_cu", C.c_int), ("t0_min_blocks", C.c_int), ("waves_per_simd", C.c_int), ("trace_mode", C.c_int),
                ("timed_barriers", C.c_int), ("no_short_cigar", C.c_int), ("no_fused_pack", C.c_int), ("kernel_walk", C.c_int), ("exact_two_waves", C.c_int), ("band_tier", C.c_int), ("no_host_parts", C.c_int), ("short_iterations", C.c_int), ("arena_chunk_cap", C.c_int), ("emit_pairs", C.c_int), ("verify_counters", C.c_int)]


class Config(C.Structure):  # wfagpu_amd_config_t
    _fields_ = [("device", C.c_int), ("stream", C.c_void_p), ("arena_bytes", C.c_size_t), ("text_bytes", C.c_size_t),
                ("arena_limit_bytes", C.c_size_t), ("arena_limit_max_bytes", C.c_size_t), ("tuning", Tuning),
                ("null_stream", C.c_int)]


class LaunchConfig(C.Structure):  # wfagpu_amd_launch_config_t: all zero = automatic
    _fields_ = [("num_devices", C.c_int), ("virtual_devices", C.c_int), ("lanes_per_device", C.c_int),
                ("batches_per_device", C.c_int), ("arena_limit_bytes", C.c_size_t), ("input_pool_bytes", C.c_size_t),
                ("numa_pin", C.c_int), ("timing", C.c_int), ("tuning", Tuning), ("host_pack", C.c_int),
                ("host_pack_threads", C.c_int), ("ascii_every", C.c_int), ("bring_up", C.c_int)]


class LaunchStats(C.Structure):  # wfagpu_amd_launch_stats_t
    _fields_ = [(k, C.c_double) for k in ("total_ms", "plan_ms", "acquire_ms", "prep_ms", "upload_ms", "upload_wait_ms",
                                          "device_ms", "device_wait_ms", "d2h_ms", "scatter_ms", "check_ms")] + \
               [("devices", C.c_int), ("lanes", C.c_int), ("batches", C.c_int), ("host_threads", C.c_uint),
                ("host_pack_ms", C.c_double), ("host_packed_batches", C.c_int), ("host_pack_threads", C.c_int)]


class Batch(C.Structure):  # wfagpu_amd_batch_t
    _fields_ = [("d_sequences", C.c_void_p), ("sequences_bytes", C.c_size_t), ("d_metadata", C.c_void_p),
                ("num_pairs", C.c_size_t), ("packed_bytes", C.c_size_t), ("max_seq_len", C.c_uint),
                ("d_packed", C.c_void_p)]


class Stats(C.Structure):  # wfagpu_amd_stats_t
    _fields_ = [("pack_ms", C.c_float), ("align_ms", C.c_float), ("trace_ms", C.c_float), ("total_ms", C.c_float),
                ("align_launches", C.c_int), ("cells", C.c_ulonglong), ("arena_units", C.c_ulonglong),
                ("text_bytes", C.c_ulonglong), ("pairs_tier", C.c_uint * 6), ("pairs_retried", C.c_uint), ("pairs_raw", C.c_uint), ("pairs_banded", C.c_uint), ("pairs_budget_missed", C.c_uint), ("auto_budget", C.c_int),
                ("sub_batches", C.c_uint), ("lds_bytes_tier0", C.c_size_t), ("blocks_per_cu_tier0", C.c_int),
                ("main_launch_ms", C.c_float), ("main_launch_tier", C.c_int), ("main_launch_pairs", C.c_uint),
                ("main_launch_cells", C.c_ulonglong), ("main_launch_seq_bytes", C.c_ulonglong),
                ("sample_cells", C.c_ulonglong), ("sample_launches", C.c_int), ("sample_passes", C.c_uint),
                ("waves_per_simd_tier0", C.c_int), ("pairs_walked_in_kernel", C.c_uint), ("pairs_trace_split", C.c_uint)]


ABI_SYMBOLS = [
    # include/wfa_gpu_abi.h
    "get_num_cuda_devices", "get_cuda_dev_name", "get_cuda_SM_count", "get_cuda_capability",
    "initialize_wfa_results", "destroy_wfa_results", "launch_alignments", "launch_alignments_distance",
    "wfagpu_initialize_aligner", "wfagpu_add_sequences", "wfagpu_initialize_parameters",
    "wfagpu_set_batch_size", "wfagpu_align", "wfagpu_destroy_aligner",
    # include/wfa_gpu_device.h
    "wfagpu_amd_create", "wfagpu_amd_destroy", "wfagpu_amd_fill_packed_offsets", "wfagpu_amd_pack_device",
    "wfagpu_amd_align_device", "wfagpu_amd_last_stats", "wfagpu_amd_set_num_devices", "wfagpu_amd_release_cache",
    "wfagpu_amd_check_failures", "wfagpu_amd_hint_same_stream", "wfagpu_amd_configure_launch",
    "wfagpu_amd_last_launch_stats", "wfagpu_amd_set_tuning", "wfagpu_amd_stream", "wfagpu_amd_trim", "wfagpu_amd_prime",
    "wfagpu_amd_warmup", "wfagpu_amd_warmup_wait", "wfagpu_amd_last_launch_stats_device", "wfagpu_amd_debug_times",
    "wfagpu_host_pack_sequence", "wfagpu_host_pack_sequence_scalar", "wfagpu_host_pack_strip",
]

_lib = None
_gen = None


def load():
    """Load libwfagpu.so; raises if it is missing (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() or `make -C wfa-gpu_amd`")
    try:
        # PyTorch-ROCm bundles its own HIP runtime; it must be the first (and only) one in the process,
        # otherwise device discovery fails.  Standalone C programs simply use /opt/rocm's.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.wfagpu_amd_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Config)]
    lib.wfagpu_amd_create.restype = C.c_int
    lib.wfagpu_amd_destroy.argtypes = [C.c_void_p]
    lib.wfagpu_amd_destroy.restype = None
    lib.wfagpu_amd_fill_packed_offsets.argtypes = [C.c_void_p, C.c_size_t]
    lib.wfagpu_amd_fill_packed_offsets.restype = C.c_size_t
    lib.wfagpu_amd_pack_device.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p]
    lib.wfagpu_amd_pack_device.restype = C.c_int
    lib.wfagpu_amd_align_device.argtypes = [C.c_void_p, C.POINTER(Batch), Penalties, C.c_int, C.c_int, C.c_int, C.c_bool,
                                            C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p)]
    lib.wfagpu_amd_align_device.restype = C.c_int
    lib.wfagpu_amd_last_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    lib.wfagpu_amd_last_stats.restype = None
    lib.wfagpu_amd_set_num_devices.argtypes = [C.c_int]
    lib.wfagpu_amd_set_tuning.argtypes = [C.c_void_p, C.POINTER(Tuning)]
    lib.wfagpu_amd_set_tuning.restype = None
    lib.wfagpu_amd_configure_launch.argtypes = [C.POINTER(LaunchConfig)]
    lib.wfagpu_amd_configure_launch.restype = None
    lib.wfagpu_amd_last_launch_stats.argtypes = [C.POINTER(LaunchStats)]
    lib.wfagpu_amd_last_launch_stats.restype = None
    lib.wfagpu_amd_check_failures.argtypes = []
    lib.wfagpu_amd_check_failures.restype = C.c_long
    lib.wfagpu_amd_release_cache.argtypes = []
    lib.wfagpu_amd_release_cache.restype = None
    lib.launch_alignments.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, Options, C.c_bool]
    lib.launch_alignments.restype = None
    lib.launch_alignments_distance.argtypes = lib.launch_alignments.argtypes
    lib.launch_alignments_distance.restype = None
    lib.initialize_wfa_results.argtypes = [C.POINTER(C.POINTER(AlignmentResult)), C.c_size_t, C.c_size_t]
    lib.initialize_wfa_results.restype = C.c_bool
    lib.destroy_wfa_results.argtypes = [C.POINTER(AlignmentResult), C.c_size_t]
    lib.destroy_wfa_results.restype = C.c_bool
    lib.wfagpu_initialize_aligner.argtypes = [C.POINTER(Aligner)]
    lib.wfagpu_initialize_aligner.restype = C.c_bool
    lib.wfagpu_add_sequences.argtypes = [C.POINTER(Aligner), C.c_char_p, C.c_char_p]
    lib.wfagpu_add_sequences.restype = C.c_bool
    lib.wfagpu_initialize_parameters.argtypes = [C.POINTER(Aligner), Penalties]
    lib.wfagpu_initialize_parameters.restype = C.c_bool
    lib.wfagpu_set_batch_size.argtypes = [C.POINTER(Aligner), C.c_size_t]
    lib.wfagpu_set_batch_size.restype = C.c_bool
    lib.wfagpu_align.argtypes = [C.POINTER(Aligner)]
    lib.wfagpu_align.restype = C.c_bool
    lib.wfagpu_destroy_aligner.argtypes = [C.POINTER(Aligner)]
    lib.wfagpu_destroy_aligner.restype = None
    lib.get_cuda_SM_count.argtypes = [C.c_int]
    lib.get_cuda_SM_count.restype = C.c_int
    lib.get_num_cuda_devices.argtypes = [C.POINTER(C.c_int)]
    lib.get_cuda_dev_name.argtypes = [C.c_int]
    lib.get_cuda_dev_name.restype = C.c_void_p
    _lib = lib
    return lib


def configure_launch(**kw):
    """wfagpu_amd_configure_launch; keyword arguments are fields of wfagpu_amd_launch_config_t, `tuning` a dict of
    wfagpu_amd_tuning_t fields.  No arguments: back to the defaults."""
    lib = load()
    if not kw:
        lib.wfagpu_amd_configure_launch(None)
        return
    tuning = Tuning(**kw.pop("tuning", {}))
    cfg = LaunchConfig(tuning=tuning, **kw)
    lib.wfagpu_amd_configure_launch(C.byref(cfg))


def last_launch_stats():
    st = LaunchStats()
    load().wfagpu_amd_last_launch_stats(C.byref(st))
    return {k: getattr(st, k) for k, _ in LaunchStats._fields_}


def last_launch_stats_per_device():
    """Stage times of every device slot of the last launch_alignments* call."""
    out = []
    for shard in range(64):
        st = LaunchStats()
        if load().wfagpu_amd_last_launch_stats_device(shard, C.byref(st)) != 0:
            break
        out.append({k: getattr(st, k) for k, _ in LaunchStats._fields_})
    return out


def load_gen():
    global _gen
    if _gen is None:
        if not os.path.exists(GEN_PATH):
            raise RuntimeError(f"{GEN_PATH} is missing: build it with `make -C wfa-gpu_amd libwfagen.so`")
        g = C.CDLL(GEN_PATH)
        g.wfagen_pair_stride.argtypes = [C.c_int, C.c_double]
        g.wfagen_pair_stride.restype = C.c_size_t
        g.wfagen_generate.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_double,
                                      C.c_uint64, C.c_int]
        g.wfagen_generate.restype = C.c_size_t
        _gen = g
    return _gen


META_DTYPE = np.dtype([("text_offset", "<u8"), ("pattern_offset", "<u8"), ("text_offset_packed", "<u8"),
                       ("pattern_offset_packed", "<u8"), ("text_len", "<u4"), ("pattern_len", "<u4"),
                       ("has_N", "u1"), ("_pad", "u1", 7)])
assert META_DTYPE.itemsize == 48


def pad4(x):
    return x + (4 - (x % 4))


def layout_pairs(pairs):
    """[(pattern, text)] of bytes/str -> (uint8 buffer, META_DTYPE records) in the reference layout
    (lib/aligner.c:127-166: 4-byte aligned starts, >=1 NUL after each sequence)."""
    n = len(pairs)
    meta = np.zeros(n, dtype=META_DTYPE)
    off = 0
    chunks = []
    for i, (p, t) in enumerate(pairs):
        p = p.encode() if isinstance(p, str) else bytes(p)
        t = t.encode() if isinstance(t, str) else bytes(t)
        po = off
        to = pad4(po + len(p) + 1)
        off = pad4(to + len(t) + 1)
        meta[i]["pattern_offset"] = po
        meta[i]["pattern_len"] = len(p)
        meta[i]["text_offset"] = to
        meta[i]["text_len"] = len(t)
        chunks.append((po, p))
        chunks.append((to, t))
    buf = np.zeros(off + 16, dtype=np.uint8)
    for o, s in chunks:
        buf[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    return buf, meta


def generate_pairs(n, length, error, seed, nthreads=8):
    """Synthetic pairs straight into the batch layout (tools/generate_dataset.c)."""
    g = load_gen()
    stride = g.wfagen_pair_stride(length, float(error))
    cap = stride * n + 64
    buf = np.zeros(cap, dtype=np.uint8)
    meta = np.zeros(n, dtype=META_DTYPE)
    used = g.wfagen_generate(buf.ctypes.data, cap, meta.ctypes.data, n, length, float(error), seed, nthreads)
    if used == 0:
        raise RuntimeError("wfagen_generate failed")
    return buf[:used], meta


class GenModel(C.Structure):  # wfagen_model_t (tools/generate_dataset.c)
    _fields_ = [("error", C.c_double), ("indel_frac", C.c_double), ("indel_mean", C.c_double), ("long_frac", C.c_double),
                ("long_min", C.c_int), ("long_max", C.c_int), ("cluster", C.c_double)]


def generate_pairs_model(n, length, seed, error=0.1, indel_frac=0.6, indel_mean=2.5, long_frac=0.01, long_min=30, long_max=120,
                         cluster=0.3, nthreads=8):
    """Long-read shaped pairs: multi-base indels (geometric + a few long ones) and clustered errors
    (wfagen_generate_model in tools/generate_dataset.c) -- the hard input for the adaptive band."""
    g = load_gen()
    g.wfagen_model_stride.argtypes = [C.c_int, C.POINTER(GenModel)]
    g.wfagen_model_stride.restype = C.c_size_t
    g.wfagen_generate_model.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(GenModel), C.c_uint64, C.c_int]
    g.wfagen_generate_model.restype = C.c_size_t
    m = GenModel(error, indel_frac, indel_mean, long_frac, long_min, long_max, cluster)
    stride = g.wfagen_model_stride(length, C.byref(m))
    cap = stride * n + 64
    buf = np.zeros(cap, dtype=np.uint8)
    meta = np.zeros(n, dtype=META_DTYPE)
    used = g.wfagen_generate_model(buf.ctypes.data, cap, meta.ctypes.data, n, length, C.byref(m), seed, nthreads)
    if used == 0:
        raise RuntimeError("wfagen_generate_model failed")
    return buf[:used], meta


def pairs_from_layout(buf, meta):
    out = []
    b = buf.tobytes() if isinstance(buf, np.ndarray) else bytes(buf)
    for m in meta:
        po, pl, to, tl = int(m["pattern_offset"]), int(m["pattern_len"]), int(m["text_offset"]), int(m["text_len"])
        out.append((b[po:po + pl], b[to:to + tl]))
    return out


def read_seq_file(path, limit=None):
    """.seq format: alternating '>PATTERN' / '<TEXT' lines (utils/sequence_reader.c:193-227)."""
    pairs = []
    with open(path, "rb") as f:
        pat = None
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                pat = line[1:]
            elif line.startswith(b"<"):
                pairs.append((pat, line[1:]))
                if limit and len(pairs) >= limit:
                    break
    return pairs


def host_pack(buf, meta, packed_bytes, scalar=False):
    """wfagpu_host_pack_sequence over a batch (meta with its packed offsets filled): (uint32 words incl. four spare ones,
    uint8 flags[2 n]) -- what DeviceAligner.pack returns from the device."""
    lib = load()
    fn = lib.wfagpu_host_pack_sequence_scalar if scalar else lib.wfagpu_host_pack_sequence
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    fn.restype = C.c_int
    buf = np.ascontiguousarray(buf)
    words = np.zeros(packed_bytes // 4 + 4, dtype=np.uint32)
    flags = np.zeros(2 * len(meta), dtype=np.uint8)
    base, out = buf.ctypes.data, words.ctypes.data
    for i in range(len(meta)):
        m = meta[i]
        flags[2 * i] = fn(base + int(m["pattern_offset"]), int(m["pattern_len"]), out + int(m["pattern_offset_packed"]))
        flags[2 * i + 1] = fn(base + int(m["text_offset"]), int(m["text_len"]), out + int(m["text_offset_packed"]))
    return words, flags


class DeviceAligner:
    """Owns a wfagpu_amd context on one GPU and runs resident batches through the C-ABI."""

    def __init__(self, device=0, arena_bytes=0, text_bytes=0, use_torch_stream=True, arena_limit_bytes=0, null_stream=False, **tuning):
        """tuning: fields of wfagpu_amd_tuning_t (min_tier=2, force_band=1, ...).

        Streams (include/wfa_gpu_device.h, "Stream ordering"): with a torch.cuda.Stream current, the context runs on it and is
        ordered with the binding's own tensor work by construction.  torch's DEFAULT stream has the handle 0, which the C
        interface reads as "create my own" (a non-blocking stream, NOT ordered behind the null stream): the binding then
        drains torch's current stream before every call that hands the context a tensor (`_inputs_ready`) -- a zero-fill still
        queued on the null stream would otherwise race with the context's kernels (EXPERIMENTS R5.8).  null_stream=True runs the
        context on the null stream itself (wfagpu_amd_config_t::null_stream)."""
        import torch
        self.torch = torch
        self.lib = load()
        self.device = device
        torch.cuda.set_device(device)
        cfg = Config(device=device, stream=None, arena_bytes=arena_bytes, text_bytes=text_bytes,
                     arena_limit_bytes=arena_limit_bytes, tuning=Tuning(**tuning))
        self._ctx_stream = None     # handle of the context's stream when torch knows it (0: the null stream); None: the context's own
        if null_stream:
            cfg.null_stream = 1
            self._ctx_stream = 0
        elif use_torch_stream:
            h = torch.cuda.current_stream(device).cuda_stream
            if h:
                cfg.stream = C.c_void_p(h)
                self._ctx_stream = h
        self.ctx = C.c_void_p()
        if self.lib.wfagpu_amd_create(C.byref(self.ctx), C.byref(cfg)) != 0:
            raise RuntimeError("wfagpu_amd_create failed")

    def _inputs_ready(self):
        """Everything queued on torch's current stream (fills, copies of the tensors about to be handed over) is complete
        before the context's own stream touches them.  Nothing to do when both are the same stream."""
        cur = self.torch.cuda.current_stream(self.device)
        if self._ctx_stream is not None and cur.cuda_stream == self._ctx_stream:
            return
        if not cur.query():
            cur.synchronize()

    def hint_same_stream(self, on=True):
        """The following batches come from the same stream of reads as the last one: score budgets learnt from a sample are
        tried again without sampling (results stay exact; see wfagpu_amd_hint_same_stream)."""
        self.lib.wfagpu_amd_hint_same_stream.argtypes = [C.c_void_p, C.c_int]
        self.lib.wfagpu_amd_hint_same_stream.restype = None
        self.lib.wfagpu_amd_hint_same_stream(self.ctx, 1 if on else 0)

    def set_tuning(self, **tuning):
        """Replace the context's tuning switches (no arguments: the defaults)."""
        t = Tuning(**tuning)
        self.lib.wfagpu_amd_set_tuning(self.ctx, C.byref(t))

    def close(self):
        if self.ctx:
            self.lib.wfagpu_amd_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, buf, meta):
        """Host layout -> resident batch (fills packed offsets on the host copy first)."""
        torch = self.torch
        meta = meta.copy()
        packed_bytes = self.lib.wfagpu_amd_fill_packed_offsets(meta.ctypes.data, len(meta))
        dev = torch.device("cuda", self.device)
        d_seq = torch.from_numpy(np.ascontiguousarray(buf)).to(dev)
        d_meta = torch.from_numpy(meta.view(np.uint8).reshape(-1)).to(dev)
        max_len = int(max(meta["pattern_len"].max(initial=0), meta["text_len"].max(initial=0))) if len(meta) else 0
        batch = Batch(d_sequences=d_seq.data_ptr(), sequences_bytes=d_seq.numel(), d_metadata=d_meta.data_ptr(),
                      num_pairs=len(meta), packed_bytes=packed_bytes, max_seq_len=max_len)
        batch._keep = (d_seq, d_meta)
        batch._meta_host = meta
        return batch

    def upload_packed(self, buf, meta):
        """Host layout -> resident batch that carries 2-bit words packed on the host (wfagpu_amd_batch_t::d_packed) and no
        ASCII at all.  Every byte must be one of ACGT."""
        torch = self.torch
        meta = meta.copy()
        packed_bytes = self.lib.wfagpu_amd_fill_packed_offsets(meta.ctypes.data, len(meta))
        words, flags = host_pack(buf, meta, packed_bytes)
        if flags.any():
            raise ValueError("a sequence holds a byte outside ACGT: such batches go up as ASCII")
        dev = torch.device("cuda", self.device)
        d_words = torch.from_numpy(words.view(np.int32)).to(dev)
        d_meta = torch.from_numpy(meta.view(np.uint8).reshape(-1)).to(dev)
        max_len = int(max(meta["pattern_len"].max(initial=0), meta["text_len"].max(initial=0))) if len(meta) else 0
        batch = Batch(d_sequences=None, sequences_bytes=0, d_metadata=d_meta.data_ptr(), num_pairs=len(meta),
                      packed_bytes=packed_bytes, max_seq_len=max_len, d_packed=d_words.data_ptr())
        batch._keep = (d_words, d_meta)
        batch._meta_host = meta
        return batch

    def pack(self, batch):
        torch = self.torch
        dev = torch.device("cuda", self.device)
        d_packed = torch.zeros(batch.packed_bytes // 4 + 4, dtype=torch.int32, device=dev)
        d_flags = torch.zeros(2 * batch.num_pairs, dtype=torch.uint8, device=dev)
        self._inputs_ready()      # (the two fills above run on torch's stream, the pack kernel on the context's)
        rc = self.lib.wfagpu_amd_pack_device(self.ctx, C.byref(batch), d_packed.data_ptr(), d_flags.data_ptr())
        if rc != 0:
            raise RuntimeError(f"wfagpu_amd_pack_device failed ({rc})")
        return d_packed.cpu().numpy().view(np.uint32), d_flags.cpu().numpy()

    def align(self, batch, penalties, max_error, compute_cigar, band=-1, band_width=0, fetch=True, d_scores=None):
        """Returns (scores ndarray, cigars list or None).  With fetch=False results stay on the device
        and (d_scores tensor, (text_ptr, off_ptr, len_ptr)) is returned.  d_scores: the caller's int32 device tensor for the scores
        (the C interface takes the caller's buffer: a loop of calls need not allocate one per call)."""
        torch = self.torch
        dev = torch.device("cuda", self.device)
        n = batch.num_pairs
        if d_scores is None:
            d_scores = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        assert d_scores.dtype == torch.int32 and d_scores.numel() >= max(n, 1) and d_scores.is_cuda
        t, o, l = C.c_void_p(), C.c_void_p(), C.c_void_p()
        pen = Penalties(*penalties)
        self._inputs_ready()
        rc = self.lib.wfagpu_amd_align_device(self.ctx, C.byref(batch), pen, int(max_error), int(band), int(band_width),
                                              bool(compute_cigar), d_scores.data_ptr(), C.byref(t), C.byref(o),
                                              C.byref(l))
        if rc != 0:
            raise RuntimeError(f"wfagpu_amd_align_device failed ({rc})")
        if not fetch:
            return d_scores[:n], (t.value, o.value, l.value)
        scores = d_scores[:n].cpu().numpy()
        cigars = None
        if compute_cigar and n:
            st = self.stats()
            cigars = fetch_cigars(t.value, o.value, l.value, n, st.text_bytes)
        return scores, cigars

    def stats(self):
        st = Stats()
        self.lib.wfagpu_amd_last_stats(self.ctx, C.byref(st))
        return st


_hip = None


def _hiprt():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("libamdhip64.so not found")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpy.restype = C.c_int
    return _hip


def _d2h(ptr, nbytes):
    out = np.empty(nbytes, dtype=np.uint8)
    if nbytes:
        rc = _hiprt().hipMemcpy(out.ctypes.data, ptr, nbytes, 2)  # hipMemcpyDeviceToHost
        if rc != 0:
            raise RuntimeError(f"hipMemcpy D2H failed ({rc})")
    return out


def fetch_cigars(text_ptr, off_ptr, len_ptr, n, text_bytes):
    off = _d2h(off_ptr, 8 * n).view(np.uint64)
    ln = _d2h(len_ptr, 4 * n).view(np.uint32)
    text = _d2h(text_ptr, int(text_bytes)).tobytes()
    out = []
    for i in range(n):
        if ln[i] == 0xFFFFFFFF:
            out.append(None)
        else:
            o = int(off[i])
            out.append(text[o:o + int(ln[i])].decode())
    return out
