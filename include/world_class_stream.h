/* world_class_stream.h -- chunked ("streaming") Harvest + CheapTrick for many concurrent streams (extension).
 *
 * The reference has no streaming mode: Harvest (reference src/harvest.cpp) is non-causal -- zero-phase decimation
 * (:1400-1410), a +-100-frame section extension (:431-440), a forward-backward smoothing filter run over the contour
 * padded by 300 frames either side (:676-703) -- and CheapTrick draws its noise from a position that depends on every
 * earlier frame.  The semantics defined here (BASELINE config 5):
 *
 *   Every stream accumulates its samples in a device-resident history of at most  W = lookback + chunk + lookahead  ms.
 *   A push appends `chunk` ms to every stream and runs the WHOLE-UTTERANCE Harvest of this library on each stream's
 *   history window -- the same kernels, the window being an utterance that starts at an absolute time which is a
 *   multiple of lcm(8 ms, frame period), so that frame grid, decimation phase and the smoothing filter's start phase
 *   fall where they fall in the whole signal.  Of the window's contour only the frames at least `lookahead` ms before the
 *   newest sample (and, once the history is full, at least `lookback` ms after the oldest) are committed: each absolute
 *   frame k (time k * frame_period) is committed exactly once, in order, `chunk / frame_period` frames per push in the
 *   steady state.  CheapTrick then runs on the committed frames only, reading the samples from the history and taking
 *   its noise draws from the stream's own position in the reference's xorshift128 sequence, which is carried from push to
 *   push -- exactly the draws the frames would have got in one whole-utterance call.
 *   A stream is closed by a push with flush[u] != 0 (its last chunk may be shorter): all remaining frames are committed,
 *   the window ending where the signal ends.
 *   One property of the reference has to be pinned for this to be well defined: its decimator aligns the sampling phase to
 *   the END of the signal (reference src/world_matlabfunctions.cpp:201-206, nbeg = length mod ratio, MATLAB's decimate), so
 *   the contour of a whole-utterance call changes by tenths of a Hz with (total length mod decimation ratio) -- one or two
 *   trailing samples -- which no stream can know in advance.  Harvest therefore always sees windows that are a multiple of
 *   the ratio long: full chunks are, and of a short final chunk the last (length mod ratio) samples are used by CheapTrick
 *   only.  The stream equals the whole-utterance call exactly for totals that are multiples of the ratio; for others it
 *   equals Harvest on the signal without those trailing samples (frame count wc_get_samples of that length) followed by
 *   CheapTrick on the complete signal.
 *
 *   Result: the committed (tpos, f0, spectrogram rows) equal those of ONE whole-utterance Harvest + CheapTrick call on
 *   the complete signal wherever the influence of the window edges has died out: lookahead and lookback of >= 400 ms
 *   (300 padded + 100 extension frames at Harvest's internal 1 ms grid) make that every frame on ordinary speech;
 *   tests/test_gpu_stream.py compares whole streams (voicing decisions identical, F0 within 1e-9 Hz -- last-bit
 *   differences come from the limit cycle of the smoothing filter's backward pass, whose phase depends on where the
 *   window ends -- spectrogram within 1e-7 relative).  Algorithmic latency: lookahead + chunk (+ the push's run time).
 *
 * Layout: d_chunk holds the new samples of the streams back to back (stream u's at sum(n_new[<u])); the outputs are
 * packed the same way by the number of frames committed for each stream (frames_out, host array).  Capacity needed:
 * wc_stream_max_frames_per_push() rows per stream.
 */
#ifndef WORLD_CLASS_STREAM_H
#define WORLD_CLASS_STREAM_H

#include "world_class_c.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wc_stream wc_stream;

/* fs must be a multiple of 1000 with fs/1000 a multiple of Harvest's decimation ratio (8, 16, 24, 32, 48, 96 kHz ...);
 * frame_period_ms a whole number of ms; chunk_ms, lookback_ms, lookahead_ms multiples of lcm(8, frame_period_ms).
 * Harvest / CheapTrick options as in wc_harvest_create / wc_cheaptrick_create (fft_size 0 = automatic). */
wc_stream *wc_stream_create(int fs, int n_streams, double frame_period_ms, int chunk_ms, int lookback_ms, int lookahead_ms,
                            double harvest_f0_floor, double harvest_f0_ceil, double q1, double cheaptrick_f0_floor, int fft_size);
void wc_stream_destroy(wc_stream *s);
/* Incremental mode (call before the first push; 0 switches back).  Harvest's front -- decimation, band-pass, zero crossings, raw
 * candidates, refinement -- is local: what it yields for a 1 ms frame depends on +-`context_ms` of signal (160 ms covers the
 * longest band-pass, the lowest band's periods, the refinement window and the +-3-frame overlap with room to spare).  In this mode
 * a push runs the front on the newest chunk + 2 context only and appends the refined candidate / score rows of the frames that
 * have their full context to a per-stream ring; only Harvest's tail (unreliable-candidate test, contour logic, smoothing) runs over
 * the window, on rows from the ring.  The context is part of the lookahead (rows exist up to `context` behind the newest sample, the
 * tail looks `lookahead - context` ahead of the newest committed frame): lookahead 560 ms with context 160 ms gives the tail the
 * same 400 ms it has with whole windows.  Same committed frames, same parity bar; about half the work per push. */
int wc_stream_set_incremental(wc_stream *s, int context_ms);
int wc_stream_get_fft_size(const wc_stream *s);
int wc_stream_chunk_samples(const wc_stream *s);        /* samples per stream of a full chunk */
int wc_stream_max_frames_per_push(const wc_stream *s);  /* most frames one push can commit for one stream (a flush) */
/* Forget stream u's history and position (a new signal starts on it). */
int wc_stream_reset(wc_stream *s, int stream);
/* n_new: host array, samples appended per stream: wc_stream_chunk_samples() (NULL = that for all), 0 (stream idle this
 * push) or, only together with flush[u], anything in between.  flush: host array of flags or NULL.
 * d_tpos / d_f0: committed frames (absolute times in seconds), d_sp: their spectrogram rows [fft_size/2+1];
 * frames_out: host array [n_streams], frames committed by this push. */
int wc_stream_push_device(wc_stream *s, const double *d_chunk, const int *n_new, const int *flush, double *d_tpos, double *d_f0,
                          double *d_sp, int *frames_out);
/* The same with the new samples as int16 PCM (chunk_format 1: sample / 32768, the reference's wavread scaling) or float32 (2),
 * widened on the device; 0 = float64. */
int wc_stream_push_device_fmt(wc_stream *s, const void *d_chunk, int chunk_format, const int *n_new, const int *flush, double *d_tpos,
                              double *d_f0, double *d_sp, int *frames_out);
/* Position of stream u in the reference's noise sequence (reference src/world_matlabfunctions.cpp:243-264): where CheapTrick's next
 * committed frame takes its draws.  0 after creation and after wc_stream_reset -- the position a fresh reference process starts
 * from; set it to continue the numbering of an earlier analysis (e.g. the value wc_rng_get_position() reports after one). */
unsigned long long wc_stream_rng_position(const wc_stream *s, int stream);
int wc_stream_set_rng_position(wc_stream *s, int stream, unsigned long long position);
/* frames committed so far / samples received so far for stream u */
long long wc_stream_frames_committed(const wc_stream *s, int stream);
long long wc_stream_samples_received(const wc_stream *s, int stream);

#ifdef __cplusplus
}
#endif
#endif /* WORLD_CLASS_STREAM_H */
