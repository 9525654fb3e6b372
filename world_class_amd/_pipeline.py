"""Fused device pipeline (extension over the reference's four classes): one call = Harvest -> CheapTrick -> D4C ->
Synthesis for a packed batch, stages overlapped on HIP streams, noise-stream positions chained on the device."""
import numpy as np

from . import (DeviceArray, _c, _check, _handle, _ints, _ptr, _rng_arg, get_samples, lib, synthesis_out_length)


class Pipeline:
    def __init__(self, fs, frame_period=5.0, harvest_f0_floor=71.0, harvest_f0_ceil=800.0, q1=-0.15,
                 cheaptrick_f0_floor=71.0, fft_size=0, d4c_threshold=0.85):
        self.fs, self.frame_period = fs, frame_period
        self._h = _handle(lib().wc_pipeline_create(fs, frame_period, harvest_f0_floor, harvest_f0_ceil, q1,
                                                   cheaptrick_f0_floor, fft_size, d4c_threshold))
        self.fft_size = lib().wc_pipeline_get_fft_size(self._h)
        self.bins = self.fft_size // 2 + 1

    def set_option(self, name, value=None):
        """wc_pipeline_set_option: a schedule knob of this handle ("unchain_below", "schedule", "host_splits", "force_tie", ...;
        None = the default).  The WC_PIPELINE_* environment variables are read once, when the handle is created."""
        _check(lib().wc_pipeline_set_option(self._h, name.encode(), None if value is None else str(value).encode()))

    def lengths(self, x_lengths):
        fl = [get_samples(self.fs, n, self.frame_period) for n in x_lengths]
        yl = [synthesis_out_length(n, self.frame_period, self.fs) for n in fl]
        return fl, yl

    def run_device(self, d_x, x_lengths, d_tpos, d_f0, d_sp, d_ap, d_y, rng_pos=None):
        n = len(x_lengths)
        arr, arg = _rng_arg(rng_pos, n)
        _check(lib().wc_pipeline_run_device(self._h, n, _ptr(d_x), _ints(x_lengths), _ptr(d_tpos), _ptr(d_f0), _ptr(d_sp),
                                            _ptr(d_ap), _ptr(d_y), arg))
        return list(arr) if arr is not None else None

    def run_batch(self, xs, rng_pos=None):
        """Host lists in, list of dicts (tpos, f0, sp, ap, y) out."""
        xl = [len(x) for x in xs]
        fl, yl = self.lengths(xl)
        d_x = DeviceArray.from_host(np.concatenate([_c(v) for v in xs]))
        d_t, d_f = DeviceArray(sum(fl)), DeviceArray(sum(fl))
        d_sp, d_ap = DeviceArray(sum(fl) * self.bins), DeviceArray(sum(fl) * self.bins)
        d_y = DeviceArray(sum(yl))
        pos = self.run_device(d_x, xl, d_t, d_f, d_sp, d_ap, d_y, rng_pos)
        t, f, y = d_t.to_host(), d_f.to_host(), d_y.to_host()
        sp, ap = d_sp.to_host((sum(fl), self.bins)), d_ap.to_host((sum(fl), self.bins))
        out, fo, yo = [], 0, 0
        for nf, ny in zip(fl, yl):
            out.append(dict(tpos=t[fo:fo + nf], f0=f[fo:fo + nf], sp=sp[fo:fo + nf], ap=ap[fo:fo + nf], y=y[yo:yo + ny]))
            fo += nf
            yo += ny
        return (out, pos) if rng_pos is not None else out

    def run_batch_host(self, xs, want=("tpos", "f0", "sp", "ap", "y"), y_pcm16=False, rng_pos=None, out=None):
        """Host front-end of the C-ABI (wc_pipeline_run_batch_host): xs = list of float64 or int16 (WAV PCM) arrays;
        pinned staging, one copy each way, int16 expanded / quantised on the device.  Returns a list of dicts.
        out: a result of an earlier call with the same lengths and `want` whose arrays are written again (a caller
        that reuses its buffers spares the page faults of 2 GB of fresh memory per batch)."""
        import ctypes as C
        fmt = {np.dtype(np.int16): 1, np.dtype(np.float32): 2}.get(xs[0].dtype, 0)  # 0 float64, 1 int16 PCM, 2 float32
        xs = [np.ascontiguousarray(v, dtype=(np.float64, np.int16, np.float32)[fmt]) for v in xs]
        n = len(xs)
        xl = [len(v) for v in xs]
        fl, yl = self.lengths(xl)
        VP = C.c_void_p * n
        outs = {k: None for k in ("tpos", "f0", "sp", "ap", "y")}
        tabs = {}
        for k in outs:
            if k not in want:
                tabs[k] = None
                continue
            if out is not None:
                outs[k] = [o[k] for o in out]
            elif k in ("tpos", "f0"):
                outs[k] = [np.empty(f) for f in fl]
            elif k in ("sp", "ap"):
                outs[k] = [np.empty((f, self.bins)) for f in fl]
            else:
                outs[k] = [np.empty(m, dtype=np.int16 if y_pcm16 else np.float64) for m in yl]
            tabs[k] = VP(*[a.ctypes.data for a in outs[k]])
        arr, arg = _rng_arg(rng_pos, n)
        _check(lib().wc_pipeline_run_batch_host(self._h, n, VP(*[v.ctypes.data for v in xs]), fmt, _ints(xl), tabs["tpos"],
                                                tabs["f0"], tabs["sp"], tabs["ap"], tabs["y"], 1 if y_pcm16 else 0, arg))
        res = [{k: outs[k][u] for k in outs if outs[k] is not None} for u in range(n)]
        return (res, list(arr)) if rng_pos is not None else res

    def host_buffers(self, x_lengths, want=("tpos", "f0", "sp", "ap", "y"), y_pcm16=False, pinned=True):
        """Result buffers for run_batch_host(..., out=...), one dict per utterance.  pinned: page-locked (torch pinned
        tensors seen as numpy arrays) -- the spectrogram / aperiodicity rows are then written by the copy engine directly,
        without the staging copy and the host-side scatter."""
        import torch
        fl, yl = self.lengths(list(x_lengths))

        def arr(shape, dtype):
            if pinned:
                return torch.empty(shape, dtype={np.float64: torch.float64, np.int16: torch.int16}[dtype], pin_memory=True).numpy()
            return np.empty(shape, dtype=dtype)
        res = []
        for f, m in zip(fl, yl):
            r = {}
            for k in want:
                r[k] = arr(f, np.float64) if k in ("tpos", "f0") else arr((f, self.bins), np.float64) if k in ("sp", "ap") else \
                    arr(m, np.int16 if y_pcm16 else np.float64)
            res.append(r)
        return res

    def host_inputs(self, xs):
        """Page-locked copies of the utterances (float64): run_batch_host then lets the copy engine read them where they lie,
        half batch by half batch, instead of gathering them into its own pinned staging first."""
        import torch
        out = []
        for v in xs:
            t = torch.empty(len(v), dtype=torch.float64, pin_memory=True)
            a = t.numpy()
            a[:] = v
            out.append(a)
        return out

    def coded_host_buffers(self, x_lengths, number_of_dimensions=60, want=("f0", "csp", "cap", "y"), y_pcm16=False, pinned=True):
        """Result buffers for run_batch_host_coded(..., out=...), one dict per utterance (page-locked when `pinned`)."""
        import torch
        from .codec import number_of_aperiodicities
        fl, yl = self.lengths(list(x_lengths))
        n_ap = number_of_aperiodicities(self.fs)

        def arr(shape, dtype):
            if pinned:
                return torch.empty(shape, dtype={np.float64: torch.float64, np.int16: torch.int16}[dtype], pin_memory=True).numpy()
            return np.empty(shape, dtype=dtype)
        shapes = {"tpos": lambda f, m: (f,), "f0": lambda f, m: (f,), "csp": lambda f, m: (f, number_of_dimensions),
                  "cap": lambda f, m: (f, n_ap), "y": lambda f, m: (m,)}
        return [{k: arr(shapes[k](f, m), np.int16 if (k == "y" and y_pcm16) else np.float64) for k in want} for f, m in zip(fl, yl)]

    def run_batch_host_coded(self, xs, number_of_dimensions=60, want=("f0", "csp", "cap", "y"), y_pcm16=False, rng_pos=None, out=None):
        """wc_pipeline_run_batch_host_coded: the host front-end with the reference's feature codec as the epilogue -- per frame
        `number_of_dimensions` mel-cepstral coefficients ("csp") and the band aperiodicities ("cap") instead of the rows."""
        import ctypes as C
        from .codec import number_of_aperiodicities
        fmt = {np.dtype(np.int16): 1, np.dtype(np.float32): 2}.get(xs[0].dtype, 0)
        xs = [np.ascontiguousarray(v, dtype=(np.float64, np.int16, np.float32)[fmt]) for v in xs]
        n = len(xs)
        xl = [len(v) for v in xs]
        fl, yl = self.lengths(xl)
        n_ap = number_of_aperiodicities(self.fs)
        VP = C.c_void_p * n
        shapes = {"tpos": lambda f, m: (f,), "f0": lambda f, m: (f,), "csp": lambda f, m: (f, number_of_dimensions),
                  "cap": lambda f, m: (f, n_ap), "y": lambda f, m: (m,)}
        outs, tabs = {}, {}
        for k in shapes:
            if k in want:
                if out is not None:  # the caller's buffers of an earlier call, written again
                    outs[k] = [o[k] for o in out]
                else:
                    outs[k] = [np.empty(shapes[k](f, m), dtype=np.int16 if (k == "y" and y_pcm16) else np.float64) for f, m in zip(fl, yl)]
                tabs[k] = VP(*[a.ctypes.data for a in outs[k]])
            else:
                tabs[k] = None
        arr, arg = _rng_arg(rng_pos, n)
        _check(lib().wc_pipeline_run_batch_host_coded(self._h, n, VP(*[v.ctypes.data for v in xs]), fmt, _ints(xl), tabs["tpos"], tabs["f0"],
                                                      tabs["csp"], number_of_dimensions, tabs["cap"], tabs["y"], 1 if y_pcm16 else 0, arg))
        res = [{k: outs[k][u] for k in outs} for u in range(n)]
        return (res, list(arr)) if rng_pos is not None else res

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().wc_pipeline_destroy(self._h)
                self._h = None
        except Exception:
            pass
