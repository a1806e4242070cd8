"""ctypes loader for the CPU oracle (oracle/zen_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
Nothing under zen_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzen_oracle.so")

TIME_CAUSAL, TIME_ANTICAUSAL, FREQUENCY = 0, 1, 2
OUTPUT_HARMONIC, OUTPUT_PERCUSSIVE, OUTPUT_RESIDUAL = 1, 2, 4
OK, E_FILTER_TOO_BIG, E_BAD_ARG, E_HOPS_NOT_DIVISIBLE = 0, 1, 2, 3


class OracleError(RuntimeError):
    """Stands where the reference throws zen::ZgException."""

    def __init__(self, code):
        self.code = code
        msg = {
            E_FILTER_TOO_BIG: "median filter bigger than matrix dimension",
            E_BAD_ARG: "bad argument",
            E_HOPS_NOT_DIVISIBLE: "hop_h and hop_p should be evenly divisible",
        }.get(code, "oracle error %d" % code)
        super().__init__(msg)


def build(force=False):
    src = os.path.join(_HERE, "zen_oracle.c")
    hdr = os.path.join(_HERE, "zen_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(f) > os.path.getmtime(_SO) for f in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libzen_oracle.so"])
    return _SO


_lib = None


class _Params(C.Structure):
    _fields_ = [("hop", C.c_size_t), ("nwin", C.c_size_t), ("nfft", C.c_size_t),
                ("stft_width", C.c_size_t), ("l_harm", C.c_int), ("l_perc", C.c_int),
                ("lag", C.c_int), ("cola_factor", C.c_float)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    fp = C.POINTER(C.c_float)
    L.zo_window_sqrt_hann.argtypes = [fp, C.c_size_t]
    L.zo_window_hann.argtypes = [fp, C.c_size_t]
    L.zo_twiddles.argtypes = [fp, C.c_size_t]
    L.zo_fft_c2c.argtypes = [fp, C.c_size_t, C.c_int]
    L.zo_fft_c2c.restype = C.c_int
    L.zo_cabs.argtypes = [C.c_float, C.c_float]
    L.zo_cabs.restype = C.c_float
    L.zo_cabs_array.argtypes = [fp, fp, C.c_size_t]
    L.zo_cabs_array.restype = None
    for name in ("zo_median_filter", "zo_median_filter_bruteforce", "zo_box_filter"):
        f = getattr(L, name)
        f.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]
        f.restype = C.c_int
    L.zo_hpr_create.argtypes = [C.c_float, C.c_size_t, C.c_float, C.c_uint, C.c_int, C.c_int,
                                C.POINTER(C.c_int)]
    L.zo_hpr_create.restype = C.c_void_p
    for name in ("zo_hpr_destroy", "zo_hpr_use_sse_filter", "zo_hpr_use_soft_mask",
                 "zo_hpr_reset_buffers", "zo_hpr_warmup"):
        f = getattr(L, name)
        f.argtypes = [C.c_void_p]
        f.restype = None
    L.zo_hpr_process_next_hop.argtypes = [C.c_void_p, fp]
    L.zo_hpr_process_next_hop.restype = None
    for name in ("percussive_out", "harmonic_out", "residual_out", "window", "sliding_stft", "s_mag",
                 "harmonic_matrix", "percussive_matrix", "percussive_mask", "harmonic_mask",
                 "residual_mask"):
        f = getattr(L, "zo_hpr_" + name)
        f.argtypes = [C.c_void_p]
        f.restype = fp
    L.zo_hpr_get_params.argtypes = [C.c_void_p, C.POINTER(_Params)]
    L.zo_hpr_get_params.restype = None
    L.zo_hpri_create.argtypes = [C.c_float, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_int,
                                 C.POINTER(C.c_int)]
    L.zo_hpri_create.restype = C.c_void_p
    for name in ("zo_hpri_destroy", "zo_hpri_use_sse_filter", "zo_hpri_use_soft_mask"):
        f = getattr(L, name)
        f.argtypes = [C.c_void_p]
        f.restype = None
    L.zo_hpss_chunk_padder.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zo_hpss_chunk_padder.restype = C.c_int
    L.zo_hpri_process.argtypes = [C.c_void_p, fp, C.c_size_t, fp, fp, fp]
    L.zo_hpri_process.restype = C.c_int
    L.zo_hpri_process_cpu.argtypes = [C.c_void_p, fp, C.c_size_t, fp]
    L.zo_hpri_process_cpu.restype = C.c_int
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def window_sqrt_hann(n):
    w = np.empty(n, np.float32)
    lib().zo_window_sqrt_hann(_fp(w), n)
    return w


def twiddles(nfft):
    tw = np.empty(max(nfft, 2), np.float32)
    lib().zo_twiddles(_fp(tw), nfft)
    return tw[:nfft].view(np.complex64) if nfft >= 2 else tw


def fft_c2c(x, inverse=False):
    """x: complex64 array of power-of-two length; returns the unnormalised DFT (new array)."""
    d = np.array(x, dtype=np.complex64, copy=True)
    rc = lib().zo_fft_c2c(_fp(d.view(np.float32)), d.size, 1 if inverse else 0)
    if rc:
        raise OracleError(rc)
    return d


def cabs(z):
    """|z| for a complex64 array, element-wise, with the oracle's hypotf formula."""
    z = np.ascontiguousarray(z, dtype=np.complex64)
    out = np.empty(z.shape, np.float32)
    lib().zo_cabs_array(_fp(z.view(np.float32)), _fp(out), z.size)
    return out


def _filter(fn, src, filter_len, direction):
    src = _f32(src)
    assert src.ndim == 2
    dst = np.zeros_like(src)
    rc = fn(_fp(src), _fp(dst), src.shape[0], src.shape[1], int(filter_len), int(direction))
    if rc:
        raise OracleError(rc)
    return dst


def median_filter(src, filter_len, direction):
    """src: (time, frequency) float32. MedianFilterCPU semantics (mfilt.h:270-342)."""
    return _filter(lib().zo_median_filter, src, filter_len, direction)


def median_filter_bruteforce(src, filter_len, direction):
    return _filter(lib().zo_median_filter_bruteforce, src, filter_len, direction)


def box_filter(src, filter_len, direction):
    return _filter(lib().zo_box_filter, src, filter_len, direction)


class HPR:
    """zen::internal::hps::HPR<Backend::CPU> (libzen/hps.h:152-322)."""

    def __init__(self, fs, hop, beta, output_flags, causality, copy_bord=True):
        err = C.c_int(0)
        self._h = lib().zo_hpr_create(fs, hop, beta, output_flags, causality, int(copy_bord),
                                      C.byref(err))
        if not self._h:
            raise OracleError(err.value)
        p = _Params()
        lib().zo_hpr_get_params(self._h, C.byref(p))
        self.hop, self.nwin, self.nfft, self.stft_width = p.hop, p.nwin, p.nfft, p.stft_width
        self.l_harm, self.l_perc, self.lag, self.cola_factor = p.l_harm, p.l_perc, p.lag, p.cola_factor

    def __del__(self):
        if getattr(self, "_h", None):
            lib().zo_hpr_destroy(self._h)
            self._h = None

    def use_sse_filter(self):
        lib().zo_hpr_use_sse_filter(self._h)

    def use_soft_mask(self):
        lib().zo_hpr_use_soft_mask(self._h)

    def reset_buffers(self):
        lib().zo_hpr_reset_buffers(self._h)

    def warmup(self):
        lib().zo_hpr_warmup(self._h)

    def process_next_hop(self, in_hop):
        in_hop = _f32(in_hop)
        assert in_hop.size == self.hop
        lib().zo_hpr_process_next_hop(self._h, _fp(in_hop))

    def _view(self, name, n):
        p = getattr(lib(), "zo_hpr_" + name)(self._h)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    @property
    def percussive_out(self):
        return self._view("percussive_out", self.nwin)

    @property
    def harmonic_out(self):
        return self._view("harmonic_out", self.nwin)

    @property
    def residual_out(self):
        return self._view("residual_out", self.nwin)

    @property
    def window(self):
        return self._view("window", self.nwin)

    def matrix(self, name):
        n = self.stft_width * self.nfft
        if name == "sliding_stft":
            return self._view(name, 2 * n).view(np.complex64).reshape(self.stft_width, self.nfft)
        return self._view(name, n).reshape(self.stft_width, self.nfft)

    def process_stream(self, x):
        """Feed len(x)//hop hops; returns dict of (n_hops*hop,) arrays of what copy_* would hand out."""
        x = _f32(x)
        n = x.size // self.hop
        outs = {k: np.zeros(n * self.hop, np.float32) for k in ("P", "H", "R")}
        for i in range(n):
            self.process_next_hop(x[i * self.hop:(i + 1) * self.hop])
            s = slice(i * self.hop, (i + 1) * self.hop)
            outs["P"][s] = self.percussive_out[:self.hop]
            outs["H"][s] = self.harmonic_out[:self.hop]
            outs["R"][s] = self.residual_out[:self.hop]
        return outs


class HPRIOffline:
    """zen::hps::HPRIOffline (libzen/hps.cu:21-280) with CPU filter semantics."""

    def __init__(self, fs, hop_h=4096, hop_p=256, beta_h=2.0, beta_p=2.0, nocopybord=False):
        err = C.c_int(0)
        self._h = lib().zo_hpri_create(fs, hop_h, hop_p, beta_h, beta_p, int(nocopybord), C.byref(err))
        if not self._h:
            raise OracleError(err.value)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().zo_hpri_destroy(self._h)
            self._h = None

    def use_sse_filter(self):
        lib().zo_hpri_use_sse_filter(self._h)

    def use_soft_mask(self):
        lib().zo_hpri_use_soft_mask(self._h)

    def process(self, audio):
        """Returns (harm, perc, resid) as HPRIOffline<GPU>::process captures them (resid == 0)."""
        audio = _f32(audio)
        n = audio.size
        h, p, r = (np.empty(n, np.float32) for _ in range(3))
        rc = lib().zo_hpri_process(self._h, _fp(audio), n, _fp(h), _fp(p), _fp(r))
        if rc:
            raise OracleError(rc)
        return h, p, r

    def process_cpu(self, audio):
        """HPRIOffline<CPU>::process literally: (perc, perc, perc)."""
        audio = _f32(audio)
        p = np.empty(audio.size, np.float32)
        rc = lib().zo_hpri_process_cpu(self._h, _fp(audio), audio.size, _fp(p))
        if rc:
            raise OracleError(rc)
        return p, p.copy(), p.copy()


def chunk_padder(audio_size, hop, lag):
    ps = C.c_size_t(0)
    n = lib().zo_hpss_chunk_padder(audio_size, hop, lag, C.byref(ps))
    return n, ps.value
