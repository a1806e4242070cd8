"""ctypes binding of libzen_hip.so, mirroring the reference's GPU-side classes by name:
FFTC2CWrapperGPU (libzen/fftw.h:20-49), MedianFilterGPU (libzen/mfilt.h:33-268), BoxFilterGPU
(libzen/box.h:30-215), IOGPU (libzen/libzen/io.h:16-81), HPR<GPU> (libzen/hps.h:152-322),
HPRRealtime<GPU> / HPRIOffline<GPU> (libzen/libzen/hps.h:29-118).

There is no CPU fallback here: a missing library or GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ZEN_HIP_SO") or os.path.join(_HERE, "libzen_hip.so")  # override: A/B builds

TIME_CAUSAL, TIME_ANTICAUSAL, FREQUENCY = 0, 1, 2
OUTPUT_HARMONIC, OUTPUT_PERCUSSIVE, OUTPUT_RESIDUAL = 1, 2, 4
E_FILTER_TOO_BIG, E_BAD_ARG, E_HOPS_NOT_DIVISIBLE, E_HIP, E_UNSUPPORTED = 1, 2, 3, 4, 5


class ZenHipError(RuntimeError):
    def __init__(self, code, msg):
        self.code = code
        super().__init__("zen_hip error %d: %s" % (code, msg))


class ZgException(ZenHipError):
    """Raised where the reference throws zen::ZgException (libzen/libzen/zen.h:8-12)."""


class _Params(C.Structure):
    _fields_ = [("hop", C.c_size_t), ("nwin", C.c_size_t), ("nfft", C.c_size_t),
                ("stft_width", C.c_size_t), ("l_harm", C.c_int), ("l_perc", C.c_int), ("lag", C.c_int),
                ("time_len", C.c_int), ("freq_len", C.c_int), ("cola_factor", C.c_float),
                ("n_streams", C.c_size_t), ("max_hops_per_chunk", C.c_size_t)]


class _MemcheckReport(C.Structure):
    _fields_ = [("redzone_bytes", C.c_ulonglong), ("allocations", C.c_ulonglong), ("live_allocations", C.c_ulonglong),
                ("corrupt_words", C.c_ulonglong), ("corrupt_allocations", C.c_ulonglong),
                ("bounds_violations", C.c_ulonglong), ("bounds_build", C.c_int), ("first_message", C.c_char * 256),
                ("first_violation", C.c_char * 160)]


class _HostStats(C.Structure):
    _fields_ = [("n_ranges", C.c_size_t), ("range_samples", C.c_size_t), ("input_pinned", C.c_int),
                ("outputs_pinned", C.c_int), ("setup_ms", C.c_double), ("enqueue_ms", C.c_double),
                ("total_ms", C.c_double)]


# every symbol include/zen_hip.h declares: (name, restype, argtypes)
_vp, _sz, _i, _u, _f = C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_float
_pvp = C.POINTER(C.c_void_p)
SYMBOLS = [
    ("zen_hip_init", _i, [_i]),
    ("zen_hip_device_count", _i, [C.POINTER(C.c_int)]),
    ("zen_hip_last_error", C.c_char_p, []),
    ("zen_hip_version", C.c_char_p, []),
    ("zen_hip_device_name", _i, [C.c_char_p, _sz]),
    ("zen_hip_synchronize", _i, [_vp]),
    ("zen_hip_stream_create", _i, [_pvp]),
    ("zen_hip_stream_destroy", _i, [_vp]),
    ("zen_hip_event_create", _i, [_pvp]),
    ("zen_hip_event_record", _i, [_vp, _vp]),
    ("zen_hip_event_elapsed_ms", _i, [_vp, _vp, C.POINTER(C.c_float)]),
    ("zen_hip_event_destroy", _i, [_vp]),
    ("zen_hip_set_option", _i, [C.c_char_p, _i]),
    ("zen_hip_memcheck", _i, [C.POINTER(_MemcheckReport)]),
    ("zen_hip_debug_poke", _i, [_vp, C.c_longlong, _u]),
    ("zen_hip_malloc", _i, [_pvp, _sz]),
    ("zen_hip_free", _i, [_vp]),
    ("zen_hip_memset", _i, [_vp, _i, _sz, _vp]),
    ("zen_hip_memcpy_h2d", _i, [_vp, _vp, _sz]),
    ("zen_hip_memcpy_d2h", _i, [_vp, _vp, _sz]),
    ("zen_hip_memcpy_d2d", _i, [_vp, _vp, _sz, _vp]),
    ("zen_hip_memcpy_h2d_async", _i, [_vp, _vp, _sz, _vp]),
    ("zen_hip_memcpy_d2h_async", _i, [_vp, _vp, _sz, _vp]),
    ("zen_hip_host_alloc_mapped", _i, [_sz, _i, _pvp, _pvp]),
    ("zen_hip_host_free", _i, [_vp]),
    ("zen_hip_fft_create", _i, [_sz, _pvp]),
    ("zen_hip_fft_exec", _i, [_vp, _vp, _i, _vp]),
    ("zen_hip_fft_exec_batched", _i, [_vp, _vp, _sz, _i, _vp]),
    ("zen_hip_fft_destroy", _i, [_vp]),
    ("zen_hip_mfilt_create", _i, [_i, _i, _i, _i, _i, _pvp]),
    ("zen_hip_mfilt_run", _i, [_vp, _vp, _vp, _vp]),
    ("zen_hip_mfilt_assume_nonneg", _i, [_vp, _i]),
    ("zen_hip_mfilt_destroy", _i, [_vp]),
    ("zen_hip_box_create", _i, [_i, _i, _i, _i, _pvp]),
    ("zen_hip_box_run", _i, [_vp, _vp, _vp, _vp]),
    ("zen_hip_box_destroy", _i, [_vp]),
    ("zen_hip_hpr_create", _i, [_f, _sz, _f, _u, _i, _i, _sz, _sz, _pvp]),
    ("zen_hip_hpr_destroy", _i, [_vp]),
    ("zen_hip_hpr_get_params", _i, [_vp, C.POINTER(_Params)]),
    ("zen_hip_hpr_set_stream", _i, [_vp, _vp]),
    ("zen_hip_hpr_use_sse_filter", _i, [_vp]),
    ("zen_hip_hpr_use_soft_mask", _i, [_vp]),
    ("zen_hip_hpr_reset_buffers", _i, [_vp]),
    ("zen_hip_hpr_set_resident", _i, [_vp, _i]),
    ("zen_hip_hpr_resident_stats", _i, [_vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]),
    ("zen_hip_hpr_process_next_hop", _i, [_vp, _vp]),
    ("zen_hip_hpr_copy_output", _i, [_vp, _u, _vp]),
    ("zen_hip_hpr_copy_output_async", _i, [_vp, _u, _vp]),
    ("zen_hip_hpr_process", _i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _sz]),
    ("zen_hip_hpr_process_host", _i, [_vp, _vp, _sz, _vp, _vp, _vp]),
    ("zen_hip_hpr_debug_stamps", _i, [_vp, C.POINTER(C.POINTER(C.c_ulonglong))]),
    ("zen_hip_hpr_profile", _i, [_vp, _i]),
    ("zen_hip_hpr_profile_get", _i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong),
                                     C.POINTER(C.c_ulonglong)]),
    ("zen_hip_hpr_profile_get_all", _i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong)]),
    ("zen_hip_hpri_create", _i, [_f, _sz, _sz, _f, _f, _i, _sz, _pvp]),
    ("zen_hip_hpri_destroy", _i, [_vp]),
    ("zen_hip_hpri_set_stream", _i, [_vp, _vp]),
    ("zen_hip_hpri_use_sse_filter", _i, [_vp]),
    ("zen_hip_hpri_use_soft_mask", _i, [_vp]),
    ("zen_hip_hpri_process", _i, [_vp, _vp, _sz, _vp, _vp, _vp]),
    ("zen_hip_hpri_process_sink", _i, [_vp, _vp, _sz, _i, _i, _vp, _vp]),
    ("zen_hip_hpri_host_stats_get", _i, [_vp, C.POINTER(_HostStats)]),
    ("zen_hip_hpri_process_device", _i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _sz]),
    ("zen_hip_hpri_range_halo", _i, [_vp, _sz, _sz, _sz, C.POINTER(_sz), C.POINTER(_sz)]),
    ("zen_hip_hpri_process_range", _i, [_vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    ("zen_hip_hpri_hop_counts", _i, [_vp, _sz, C.POINTER(_sz), C.POINTER(_sz)]),
    ("zen_hip_run_plan", _i, [_sz, _sz, _sz, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    ("zen_hip_hpri_profile", _i, [_vp, _i]),
    ("zen_hip_hpri_profile_get_all", _i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong)]),
]

_lib = None


def load():
    """Load libzen_hip.so (built by zen_amd/build.py or __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError("%s not built: run `python zen_amd/build.py` (needs hipcc)" % _SO)
        L = C.CDLL(_SO)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _ck(rc):
    if rc:
        msg = load().zen_hip_last_error().decode()
        if rc in (E_FILTER_TOO_BIG, E_HOPS_NOT_DIVISIBLE):
            raise ZgException(rc, msg)
        raise ZenHipError(rc, msg)


def init(device=0):
    _ck(load().zen_hip_init(device))
    # A/B timing hook: ZEN_HIP_OPTIONS="no_block_fused=1,no_median47_neighbour=1" -> zen_hip_set_option
    for item in filter(None, os.environ.get("ZEN_HIP_OPTIONS", "").split(",")):
        name, _, val = item.partition("=")
        set_option(name.strip(), int(val or 1))


class Event:
    """hipEvent_t on a stream (zen_hip_event_*): the harness's clock around a launch."""

    def __init__(self):
        e = C.c_void_p()
        _ck(load().zen_hip_event_create(C.byref(e)))
        self._e = e.value

    def record(self, stream=None):
        _ck(load().zen_hip_event_record(self._e, stream))

    def elapsed_ms(self, stop):
        ms = C.c_float()
        _ck(load().zen_hip_event_elapsed_ms(self._e, stop._e, C.byref(ms)))
        return ms.value

    def __del__(self):
        if getattr(self, "_e", None):
            load().zen_hip_event_destroy(self._e)
            self._e = None


def synchronize(stream=None):
    _ck(load().zen_hip_synchronize(stream))


def run_plan(frames, streams, nfft, group_outputs=(2, 1)):
    """zen_hip_run_plan: (frames per run or 0, simulated busy share) for a pass synthesised in runs; host arithmetic only."""
    g = (C.c_int * len(group_outputs))(*group_outputs)
    run, busy = C.c_int(), C.c_double()
    _ck(load().zen_hip_run_plan(frames, streams, nfft, g, len(group_outputs), C.byref(run), C.byref(busy)))
    return run.value, busy.value


def memcheck():
    """zen_hip_memcheck: red zones of every live allocation verified (ZEN_HIP_REDZONE=<bytes> in the environment before the
    library's first allocation), plus what a -DZEN_HIP_BOUNDS build recorded.  Cumulative counters."""
    r = _MemcheckReport()
    _ck(load().zen_hip_memcheck(C.byref(r)))
    d = {k: getattr(r, k) for k, _ in _MemcheckReport._fields_}
    d["first_message"] = d["first_message"].decode(errors="replace")
    d["first_violation"] = d["first_violation"].decode(errors="replace")
    return d


def debug_poke(dev_ptr, byte_offset, value):
    _ck(load().zen_hip_debug_poke(dev_ptr, int(byte_offset), int(value) & 0xFFFFFFFF))


def set_option(name, value):
    _ck(load().zen_hip_set_option(name.encode(), int(value)))


def device_name():
    buf = C.create_string_buffer(256)
    _ck(load().zen_hip_device_name(buf, 256))
    return buf.value.decode()


class DeviceBuffer:
    """Plain device allocation (what thrust::device_vector<float> is to the reference)."""

    def __init__(self, n, dtype=np.float32):
        self.n, self.dtype = int(n), np.dtype(dtype)
        p = C.c_void_p()
        _ck(load().zen_hip_malloc(C.byref(p), self.n * self.dtype.itemsize))
        self.ptr = p.value

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.size, a.dtype)
        b.upload(a)
        return b

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.size <= self.n
        _ck(load().zen_hip_memcpy_h2d(self.ptr, a.ctypes.data, a.size * self.dtype.itemsize))

    def download(self, n=None):
        n = self.n if n is None else n
        out = np.empty(n, self.dtype)
        _ck(load().zen_hip_memcpy_d2h(out.ctypes.data, self.ptr, n * self.dtype.itemsize))
        return out

    def zero(self):
        _ck(load().zen_hip_memset(self.ptr, 0, self.n * self.dtype.itemsize, None))

    def offset(self, n_elems):
        return self.ptr + n_elems * self.dtype.itemsize

    def free(self):
        if getattr(self, "ptr", None):
            load().zen_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        self.free()


class IOGPU:
    """zen::io::IOGPU (libzen/libzen/io.h:16-81): mapped pinned in/out buffers + device aliases."""

    def __init__(self, size):
        self.size = size
        hi, di, ho, do = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _ck(load().zen_hip_host_alloc_mapped(size * 4, 1, C.byref(hi), C.byref(di)))
        _ck(load().zen_hip_host_alloc_mapped(size * 4, 0, C.byref(ho), C.byref(do)))
        self._hi, self._ho = hi.value, ho.value
        self.device_in, self.device_out = di.value, do.value
        self.host_in = np.ctypeslib.as_array(C.cast(hi, C.POINTER(C.c_float)), shape=(size,))
        self.host_out = np.ctypeslib.as_array(C.cast(ho, C.POINTER(C.c_float)), shape=(size,))

    def __del__(self):
        if getattr(self, "_hi", None):
            load().zen_hip_host_free(self._hi)
            load().zen_hip_host_free(self._ho)
            self._hi = None


class PinnedHost:
    """n float32 of pinned host memory (zen_hip_host_alloc_mapped, not write-combined) as a numpy array: what asynchronous
    copies want on the host side (zen_hip_hpr_process_host, zen_hip_hpri_process)."""

    def __init__(self, n):
        h, d = C.c_void_p(), C.c_void_p()
        _ck(load().zen_hip_host_alloc_mapped(n * 4, 0, C.byref(h), C.byref(d)))
        self._h = h.value
        self.array = np.ctypeslib.as_array(C.cast(h, C.POINTER(C.c_float)), shape=(n,))

    def free(self):
        if getattr(self, "_h", None):
            self.array = None
            load().zen_hip_host_free(self._h)
            self._h = None

    def __del__(self):
        self.free()


class FFTC2CWrapperGPU:
    """libzen/fftw.h:20-49: public nfft, public fft_vec (device), forward(), backward()."""

    def __init__(self, nfft):
        self.nfft = nfft
        h = C.c_void_p()
        _ck(load().zen_hip_fft_create(nfft, C.byref(h)))
        self._h = h.value
        self.fft_vec = DeviceBuffer(nfft, np.complex64)
        self.fft_vec.zero()

    def forward(self):
        _ck(load().zen_hip_fft_exec(self._h, self.fft_vec.ptr, 0, None))

    def backward(self):
        _ck(load().zen_hip_fft_exec(self._h, self.fft_vec.ptr, 1, None))

    def exec_batched(self, dev_ptr, batch, inverse=False):
        _ck(load().zen_hip_fft_exec_batched(self._h, dev_ptr, batch, int(inverse), None))

    def __del__(self):
        if getattr(self, "_h", None):
            load().zen_hip_fft_destroy(self._h)
            self._h = None


class _Filter:
    _kind = None

    def __init__(self, time, frequency, filter_len, direction, copy_bord=False):
        self.time, self.frequency, self.filter_len, self.direction = time, frequency, filter_len, direction
        h = C.c_void_p()
        L = load()
        if self._kind == "mfilt":
            _ck(L.zen_hip_mfilt_create(time, frequency, filter_len, direction, int(copy_bord), C.byref(h)))
        else:
            _ck(L.zen_hip_box_create(time, frequency, filter_len, direction, C.byref(h)))
        self._h = h.value

    def assume_nonneg(self, on=True):
        """Per-handle promise that every input sample is >= +0 (zen_hip_mfilt_assume_nonneg); median handles only."""
        _ck(load().zen_hip_mfilt_assume_nonneg(self._h, int(bool(on))))

    def filter(self, src, dst):
        """src, dst: DeviceBuffer or raw device pointers (time x frequency floats)."""
        s = src.ptr if isinstance(src, DeviceBuffer) else src
        d = dst.ptr if isinstance(dst, DeviceBuffer) else dst
        L = load()
        _ck((L.zen_hip_mfilt_run if self._kind == "mfilt" else L.zen_hip_box_run)(self._h, s, d, None))

    def filter_host(self, a):
        """Convenience for tests: host matrix in, host matrix out (synchronous)."""
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.shape == (self.time, self.frequency)
        src, dst = DeviceBuffer.from_host(a), DeviceBuffer(a.size)
        dst.zero()
        self.filter(src, dst)
        synchronize()
        return dst.download().reshape(a.shape)

    def __del__(self):
        if getattr(self, "_h", None):
            L = load()
            (L.zen_hip_mfilt_destroy if self._kind == "mfilt" else L.zen_hip_box_destroy)(self._h)
            self._h = None


class MedianFilterGPU(_Filter):
    """libzen/mfilt.h:33-268 (ctor :61-66, filter :227-267)."""
    _kind = "mfilt"


class BoxFilterGPU(_Filter):
    """libzen/box.h:30-215 (ctor :55-58, filter :182-214)."""
    _kind = "box"

    def __init__(self, time, frequency, filter_len, direction):
        super().__init__(time, frequency, filter_len, direction)


class HPR:
    """zen::internal::hps::HPR<Backend::GPU> (libzen/hps.h:152-322) as a chunked streaming engine."""

    def __init__(self, fs, hop, beta, output_flags, causality, copy_bord=True, n_streams=1,
                 max_hops_per_chunk=0):
        h = C.c_void_p()
        _ck(load().zen_hip_hpr_create(fs, hop, beta, output_flags, causality, int(copy_bord), n_streams,
                                      max_hops_per_chunk, C.byref(h)))
        self._h = h.value
        p = _Params()
        _ck(load().zen_hip_hpr_get_params(self._h, C.byref(p)))
        for name, _ in _Params._fields_:
            setattr(self, name, getattr(p, name))
        self.output_flags = output_flags

    def __del__(self):
        if getattr(self, "_h", None):
            load().zen_hip_hpr_destroy(self._h)
            self._h = None

    def use_sse_filter(self):
        _ck(load().zen_hip_hpr_use_sse_filter(self._h))

    def use_soft_mask(self):
        _ck(load().zen_hip_hpr_use_soft_mask(self._h))

    def reset_buffers(self):
        _ck(load().zen_hip_hpr_reset_buffers(self._h))

    def set_resident(self, idle_ms):
        """Per-hop calls through a resident kernel (zen_hip_hpr_set_resident); 0 switches it off."""
        _ck(load().zen_hip_hpr_set_resident(self._h, int(idle_ms)))

    def resident_stats(self):
        a, b, c = C.c_ulonglong(), C.c_ulonglong(), C.c_int()
        _ck(load().zen_hip_hpr_resident_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"launches": a.value, "hops_of_ended_launches": b.value, "active": bool(c.value)}

    def set_stream(self, stream):
        _ck(load().zen_hip_hpr_set_stream(self._h, stream))

    def process_next_hop(self, in_dev):
        _ck(load().zen_hip_hpr_process_next_hop(self._h, in_dev))

    def copy_output(self, which, out_dev, sync=True):
        f = load().zen_hip_hpr_copy_output if sync else load().zen_hip_hpr_copy_output_async
        _ck(f(self._h, which, out_dev))

    def process(self, in_dev, n_hops, in_stride=None, harm=None, perc=None, resid=None, out_stride=None):
        in_stride = n_hops * self.hop if in_stride is None else in_stride
        out_stride = n_hops * self.hop if out_stride is None else out_stride
        _ck(load().zen_hip_hpr_process(self._h, in_dev, n_hops, in_stride, harm, perc, resid, out_stride))

    def process_host(self, x, harm=None, perc=None, resid=None):
        """zen_hip_hpr_process_host: x and the wanted outputs are HOST float32 arrays of n_hops * hop samples (numpy arrays,
        pageable or views of pinned memory); the others None.  Synchronous."""
        n_hops = x.size // self.hop
        assert x.dtype == np.float32 and x.flags["C_CONTIGUOUS"] and x.size == n_hops * self.hop

        def ptr(a):
            if a is None:
                return None
            assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and a.size == x.size
            return a.ctypes.data_as(C.c_void_p)
        _ck(load().zen_hip_hpr_process_host(self._h, x.ctypes.data_as(C.c_void_p), n_hops, ptr(harm), ptr(perc), ptr(resid)))

    def profile(self, enable=True):
        _ck(load().zen_hip_hpr_profile(self._h, int(enable)))

    def profile_get(self):
        ms, n, el = C.c_double(), C.c_ulonglong(), C.c_ulonglong()
        _ck(load().zen_hip_hpr_profile_get(self._h, C.byref(ms), C.byref(n), C.byref(el)))
        return ms.value, n.value, el.value

    def profile_get_all(self):
        ms, n = (C.c_double * 6)(), (C.c_ulonglong * 6)()
        _ck(load().zen_hip_hpr_profile_get_all(self._h, ms, n))
        names = ("stft", "freq_filter", "time_filter", "istft", "finalize", "rt_fused")
        return {k: {"ms": ms[i], "launches": n[i]} for i, k in enumerate(names)}

    # ---- host-side convenience for tests ---------------------------------------------------------
    def process_stream_host(self, x, block=None):
        """x: (n_streams, n) or (n,) host floats, n a multiple of hop.  Returns dict P/H/R of the same
        shape, computed through zen_hip_hpr_process in blocks of `block` hops (default: all at once)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        one_d = x.ndim == 1
        x2 = x.reshape(1, -1) if one_d else x
        S, n = x2.shape
        assert S == self.n_streams and n % self.hop == 0
        n_hops = n // self.hop
        block = n_hops if block is None else block
        din = DeviceBuffer.from_host(x2)
        outs = {k: DeviceBuffer(S * n) for k in "PHR"}
        for off in range(0, n_hops, block):
            m = min(block, n_hops - off)
            _ck(load().zen_hip_hpr_process(
                self._h, din.offset(off * self.hop), m, n, outs["H"].offset(off * self.hop),
                outs["P"].offset(off * self.hop), outs["R"].offset(off * self.hop), n))
        synchronize()
        res = {k: v.download().reshape(S, n) for k, v in outs.items()}
        return {k: (v[0] if one_d else v) for k, v in res.items()}


class HPRRealtime:
    """zen::hps::HPRRealtime<Backend::GPU> (libzen/libzen/hps.h:74-118, libzen/hps.cu:282-427)."""

    def __init__(self, fs, hop=256, beta=2.0, output_flags=OUTPUT_PERCUSSIVE, nocopybord=False,
                 max_hops_per_chunk=64):
        self.p_impl = HPR(fs, hop, beta, output_flags, TIME_CAUSAL, not nocopybord, 1, max_hops_per_chunk)
        self.hop = hop

    def process_next_hop(self, in_dev):
        self.p_impl.process_next_hop(in_dev)

    def copy_harmonic(self, out_dev):
        self.p_impl.copy_output(OUTPUT_HARMONIC, out_dev)

    def copy_percussive(self, out_dev):
        self.p_impl.copy_output(OUTPUT_PERCUSSIVE, out_dev)

    def copy_residual(self, out_dev):
        self.p_impl.copy_output(OUTPUT_RESIDUAL, out_dev)

    def use_sse_filter(self):
        self.p_impl.use_sse_filter()

    def use_soft_mask(self):
        self.p_impl.use_soft_mask()

    def warmup(self, io):
        """hps.cu:392-408: 1000 hops of iota data through the mapped buffers, then reset_buffers."""
        hop = self.hop
        data = np.arange(1000 * hop, dtype=np.float32)
        for i in range(1000):
            io.host_in[:hop] = data[i * hop:(i + 1) * hop]
            self.p_impl.process_next_hop(io.device_in)
            synchronize()
        self.p_impl.reset_buffers()
        synchronize()


class HPRIOffline:
    """zen::hps::HPRIOffline<Backend::GPU> (libzen/libzen/hps.h:29-72, libzen/hps.cu:21-221)."""

    def __init__(self, fs, hop_h=4096, hop_p=256, beta_h=2.0, beta_p=2.0, nocopybord=False, n_clips=1):
        h = C.c_void_p()
        _ck(load().zen_hip_hpri_create(fs, hop_h, hop_p, beta_h, beta_p, int(nocopybord), n_clips,
                                       C.byref(h)))
        self._h = h.value
        self.n_clips = n_clips

    def __del__(self):
        if getattr(self, "_h", None):
            load().zen_hip_hpri_destroy(self._h)
            self._h = None

    def use_sse_filter(self):
        _ck(load().zen_hip_hpri_use_sse_filter(self._h))

    def use_soft_mask(self):
        _ck(load().zen_hip_hpri_use_soft_mask(self._h))

    def set_stream(self, stream):
        _ck(load().zen_hip_hpri_set_stream(self._h, stream))

    def profile(self, enable):
        _ck(load().zen_hip_hpri_profile(self._h, int(bool(enable))))

    def profile_get_all(self):
        """{"pass1": {kernel class: {"ms", "launches"}}, "pass2": {...}} since profile(True)."""
        names = ("stft", "freq_filter", "time_filter", "istft", "finalize", "rt_fused")
        out = {}
        for ps in (1, 2):
            ms, n = (C.c_double * 6)(), (C.c_ulonglong * 6)()
            _ck(load().zen_hip_hpri_profile_get_all(self._h, ps, ms, n))
            out["pass%d" % ps] = {k: {"ms": ms[i], "launches": n[i]} for i, k in enumerate(names)}
        return out

    def hop_counts(self, n):
        a, b = C.c_size_t(), C.c_size_t()
        _ck(load().zen_hip_hpri_hop_counts(self._h, n, C.byref(a), C.byref(b)))
        return a.value, b.value

    def process_sink(self, audio, want=(True, True)):
        """zen_hip_hpri_process_sink: the ranges of the clip handed to a callback (two library threads, one per output).
        Returns (harm, perc, ranges): the assembled outputs (None where not wanted) and, per output, the list of (begin, count)
        in arrival order."""
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        n = audio.size
        outs = [np.full(n, np.nan, np.float32) if w else None for w in want]
        ranges = ([], [])
        FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_size_t, C.POINTER(C.c_float), C.c_size_t)

        def sink(user, which, begin, samples, count):
            outs[which][begin:begin + count] = np.ctypeslib.as_array(samples, shape=(count,))
            ranges[which].append((begin, count))
        cb = FN(sink)
        _ck(load().zen_hip_hpri_process_sink(self._h, audio.ctypes.data_as(C.c_void_p), n, int(want[0]), int(want[1]),
                                             C.cast(cb, C.c_void_p), None))
        return outs[0], outs[1], ranges

    def process(self, audio, out=None):
        """std::array<std::vector<float>,3> process(std::vector<float>) : (harm, perc, resid).
        out: optional (harm, perc, resid) float32 arrays of the clip's length to fill (any may be None)."""
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        n = audio.size
        h, p, r = out if out is not None else (np.empty(n, np.float32) for _ in range(3))
        for a in (h, p, r):
            assert a is None or (a.dtype == np.float32 and a.size == n and a.flags.c_contiguous)
        _ck(load().zen_hip_hpri_process(self._h, audio.ctypes.data, n, *(a.ctypes.data if a is not None else None
                                                                         for a in (h, p, r))))
        return h, p, r

    def host_stats(self):
        """What the last process() call did: ranges, pinned or not, setup / enqueue / total ms."""
        st = _HostStats()
        _ck(load().zen_hip_hpri_host_stats_get(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in _HostStats._fields_}

    def range_halo(self, n, begin, end):
        a, b = C.c_size_t(), C.c_size_t()
        _ck(load().zen_hip_hpri_range_halo(self._h, n, begin, end, C.byref(a), C.byref(b)))
        return a.value, b.value

    def process_range(self, audio_dev, n, begin, end, harm=None, perc=None):
        """Output samples [begin, end) of the clip (time-sharding)."""
        _ck(load().zen_hip_hpri_process_range(self._h, audio_dev, n, begin, end, harm, perc))

    def process_device(self, audio_dev, n, stride, harm=None, perc=None, resid=None, out_stride=None):
        out_stride = n if out_stride is None else out_stride
        _ck(load().zen_hip_hpri_process_device(self._h, audio_dev, n, stride, harm, perc, resid, out_stride))
