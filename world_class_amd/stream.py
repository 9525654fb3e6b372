"""Chunked Harvest + CheapTrick for many concurrent streams: Python mirror of include/world_class_stream.h (the semantics
are stated there; the reference itself has no streaming mode, reference src/harvest.cpp:431-440, :676-703 are non-causal)."""
import ctypes as C

import numpy as np

from . import DeviceArray, _check, _handle, _ints, _ptr, lib

_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

STREAM_SIGNATURES = {
    "wc_stream_create": (_vp, [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]),
    "wc_stream_destroy": (None, [_vp]),
    "wc_stream_set_incremental": (C.c_int, [_vp, C.c_int]),
    "wc_stream_get_fft_size": (C.c_int, [_vp]),
    "wc_stream_chunk_samples": (C.c_int, [_vp]),
    "wc_stream_max_frames_per_push": (C.c_int, [_vp]),
    "wc_stream_reset": (C.c_int, [_vp, C.c_int]),
    "wc_stream_push_device": (C.c_int, [_vp, _vp, _ip, _ip, _vp, _vp, _vp, _ip]),
    "wc_stream_push_device_fmt": (C.c_int, [_vp, _vp, C.c_int, _ip, _ip, _vp, _vp, _vp, _ip]),
    "wc_stream_rng_position": (C.c_ulonglong, [_vp, C.c_int]),
    "wc_stream_set_rng_position": (C.c_int, [_vp, C.c_int, C.c_ulonglong]),
    "wc_stream_frames_committed": (C.c_longlong, [_vp, C.c_int]),
    "wc_stream_samples_received": (C.c_longlong, [_vp, C.c_int]),
}
_bound = False


def _lib():
    global _bound
    L = lib()
    if not _bound:
        for name, (res, args) in STREAM_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return L


class StreamAnalyzer:
    """n_streams concurrent streams; every push appends `chunk_ms` of samples per stream and returns the frames it commits
    (absolute times, F0, spectrogram rows), `lookahead_ms` behind the newest sample."""

    def __init__(self, fs, n_streams, frame_period=1.0, chunk_ms=200, lookback_ms=400, lookahead_ms=400, harvest_f0_floor=71.0,
                 harvest_f0_ceil=800.0, q1=-0.15, cheaptrick_f0_floor=71.0, fft_size=0, context_ms=0):
        L = _lib()
        self.fs, self.n_streams, self.frame_period = fs, n_streams, float(frame_period)
        self._h = _handle(L.wc_stream_create(fs, n_streams, float(frame_period), chunk_ms, lookback_ms, lookahead_ms, harvest_f0_floor,
                                             harvest_f0_ceil, q1, cheaptrick_f0_floor, fft_size))
        if context_ms:  # incremental mode: Harvest's front on the newest chunk + 2 context only (see the header)
            _check(L.wc_stream_set_incremental(self._h, context_ms))
        self.fft_size = L.wc_stream_get_fft_size(self._h)
        self.bins = self.fft_size // 2 + 1
        self.chunk_samples = L.wc_stream_chunk_samples(self._h)
        self.max_frames = L.wc_stream_max_frames_per_push(self._h)
        self.latency_ms = lookahead_ms + chunk_ms
        cap = n_streams * self.max_frames
        self._d_t, self._d_f, self._d_sp = DeviceArray(cap), DeviceArray(cap), DeviceArray(cap * self.bins)

    def push_device(self, d_chunk, n_new=None, flush=None, d_tpos=None, d_f0=None, d_sp=None, chunk_format=0):
        """device pointers in and out (packed layouts of the header); chunk_format 0 = float64, 1 = int16 PCM, 2 = float32;
        returns frames committed per stream"""
        n = self.n_streams
        out = (C.c_int * n)()
        _check(_lib().wc_stream_push_device_fmt(self._h, _ptr(d_chunk), chunk_format, _ints(n_new) if n_new is not None else None,
                                            _ints(flush) if flush is not None else None,
                                            _ptr(d_tpos if d_tpos is not None else self._d_t), _ptr(d_f0 if d_f0 is not None else self._d_f),
                                            _ptr(d_sp if d_sp is not None else self._d_sp), out))
        return list(out)

    def push(self, chunks, flush=None):
        """chunks: list of n_streams float64 arrays (length chunk_samples; empty = idle; shorter only with flush[u]).
        Returns a list of dicts (tpos, f0, sp) with the frames committed for every stream."""
        filled = [np.asarray(c) for c in chunks if len(c)]  # (idle streams pass empty chunks of any type)
        dt = filled[0].dtype if filled and filled[0].dtype in (np.int16, np.float32) else np.dtype(np.float64)
        assert all(c.dtype == dt or dt == np.float64 for c in filled), "chunks of one push share a sample format"
        fmt = {np.dtype(np.int16): 1, np.dtype(np.float32): 2}.get(np.dtype(dt), 0)
        chunks = [np.ascontiguousarray(c, dtype=dt) for c in chunks]
        n_new = [len(c) for c in chunks]
        flat = np.concatenate(chunks) if sum(n_new) else np.zeros(1, dtype=dt)
        d = DeviceArray.from_host(flat, dtype=dt)
        counts = self.push_device(d, n_new, flush, chunk_format=fmt)
        d.free()
        tot = sum(counts)
        t = self._d_t.to_host()[:tot]
        f = self._d_f.to_host()[:tot]
        sp = self._d_sp.to_host()[:tot * self.bins].reshape(tot, self.bins)
        res, o = [], 0
        for c in counts:
            res.append(dict(tpos=t[o:o + c].copy(), f0=f[o:o + c].copy(), sp=sp[o:o + c].copy()))
            o += c
        return res

    def reset(self, stream):
        _check(_lib().wc_stream_reset(self._h, stream))

    def rng_position(self, stream):
        return int(_lib().wc_stream_rng_position(self._h, stream))

    def set_rng_position(self, stream, position):
        _check(_lib().wc_stream_set_rng_position(self._h, stream, int(position)))

    def frames_committed(self, stream):
        return int(_lib().wc_stream_frames_committed(self._h, stream))

    def run_whole(self, xs):
        """convenience for tests: stream whole signals chunk by chunk (all streams in lockstep, ragged ends flushed) and return the
        concatenated per-stream results"""
        assert len(xs) == self.n_streams
        cs = self.chunk_samples
        acc = [dict(tpos=[], f0=[], sp=[]) for _ in xs]
        done = [False] * len(xs)
        pos = 0
        while not all(done):
            chunks, flush = [], []
            for u, x in enumerate(xs):
                if done[u]:
                    chunks.append(np.zeros(0))
                    flush.append(0)
                    continue
                last = pos + cs >= len(x)
                chunks.append(x[pos:pos + cs])
                flush.append(1 if last else 0)
                done[u] = last
            for u, r in enumerate(self.push(chunks, flush)):
                for k in acc[u]:
                    acc[u][k].append(r[k])
            pos += cs
        return [dict(tpos=np.concatenate(a["tpos"]), f0=np.concatenate(a["f0"]), sp=np.concatenate(a["sp"])) for a in acc]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib().wc_stream_destroy(self._h)
                self._h = None
        except Exception:
            pass
