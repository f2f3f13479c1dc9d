"""Single-pass, double-buffered frame source -> device.

The reference decodes every video twice per stage (`Video.get_robust_reader`, pipeline.py:47-87) and then hands
frames to the models one `cap.read()` at a time (wrappers/mmtrack.py:38-45, wrappers/mmpose.py:60-75), each with its
own blocking host->device copy.  Here a video is read ONCE, a chunk at a time:

  reader thread   fills page-locked staging buffers (pp_host_alloc) from the frame source
  copy stream     uploads chunk k+1 (pp_upload_begin) while
  compute stream  runs the cascade on chunk k (pp_upload_wait orders it after its own upload)

so the PCIe transfer (6.2 MB per 1080p frame) and the file read never sit on the critical path.

NV12 sources (video.Nv12Video: a decoder's native output, 3.1 MB per 1080p frame) are staged and uploaded as NV12 and converted
to the BGR frames the stages read by a kernel on the copy stream, right behind the transfer (pp_upload_begin_nv12): half the
host copy and half the PCIe bytes, and no conversion on the host.
"""
from __future__ import annotations

import ctypes as C
import queue
import threading

import numpy as np

from . import _lib as L


class FrameStreamer:
    """Iterate over a frame source in chunks that are already resident on the device.

    for dev_ptr, n, first in FrameStreamer(ctx, video, chunk):   # n frames [first, first + n) at dev_ptr (u8 BGR)
        cascade.step(None, frames_dev=(dev_ptr, n)); streamer.release()
    """

    N_DEV = 3     # device buffers: one computing, one resident ahead (`ahead`: the detector may already run on it), one uploading
    N_PIN = 3     # staging buffers: one uploading, one being filled, one ready

    def __init__(self, ctx: L.Context, video, chunk: int, max_frames: int | None = None):
        self.ctx, self.video, self.chunk = ctx, video, int(chunk)
        self.h, self.w = int(video.height), int(video.width)
        self.frame_bytes = self.h * self.w * 3                      # of a BGR frame on the device
        self.nv12 = getattr(video, "pixfmt", "bgr24") == "nv12"
        self.src_shape = (self.h * 3 // 2, self.w) if self.nv12 else (self.h, self.w, 3)
        self.src_bytes = int(np.prod(self.src_shape))                # of a frame as the source delivers it
        self.max_frames = max_frames
        self.dev = [ctx.malloc(self.chunk * self.frame_bytes) for _ in range(self.N_DEV)]
        self.dev_nv12 = ctx.malloc(self.chunk * self.src_bytes) if self.nv12 else None     # one: the copy stream is in order
        nbytes = self.chunk * self.src_bytes
        self.pin_ptr, self.pin = [], []
        for _ in range(self.N_PIN):
            p = C.c_void_p()
            L.check(ctx.lib.pp_host_alloc(ctx.handle, nbytes, C.byref(p)), "pp_host_alloc")
            self.pin_ptr.append(p)
            buf = (C.c_uint8 * nbytes).from_address(p.value)
            self.pin.append(np.frombuffer(buf, np.uint8).reshape((self.chunk,) + self.src_shape))
        self.free_q: queue.Queue = queue.Queue()
        self.ready_q: queue.Queue = queue.Queue()
        for i in range(self.N_PIN):
            self.free_q.put(i)
        self.error = None
        self.ahead = None
        self._stop = False
        self.thread = threading.Thread(target=self._read_loop, name="posepipe-frame-reader", daemon=True)
        self.thread.start()

    def _read_loop(self):
        try:
            done = 0
            while True:
                want = self.chunk if self.max_frames is None else min(self.chunk, self.max_frames - done)
                if want <= 0:
                    break
                i = self.free_q.get()
                if i is None or self._stop:
                    return
                a = self.video.read_batch(want)
                n = int(a.shape[0])
                if n == 0:
                    self.free_q.put(i)
                    break
                np.copyto(self.pin[i][:n], a)     # the one host-side copy: frame source -> page-locked memory
                self.ready_q.put((i, n, done))
                done += n
                if n < want:
                    break
        except Exception as e:  # surfaced on the consumer side
            self.error = e
        self.ready_q.put(None)

    def _begin(self, item, d):
        i, n, first = item
        if self.nv12:
            L.check(self.ctx.lib.pp_upload_begin_nv12(self.ctx.handle, C.c_void_p(self.dev[d]), C.c_void_p(self.dev_nv12),
                                                      self.pin_ptr[i], n, self.h, self.w), "pp_upload_begin_nv12")
        else:
            L.check(self.ctx.lib.pp_upload_begin(self.ctx.handle, C.c_void_p(self.dev[d]), self.pin_ptr[i],
                                                 n * self.frame_bytes), "pp_upload_begin")
        return (i, n, first, d)

    def _wait(self, flight):
        """order the compute stream after the (last begun) upload, wait for it on the host, hand the staging buffer back"""
        i, n, first, d = flight
        L.check(self.ctx.lib.pp_upload_wait(self.ctx.handle, 1), "pp_upload_wait")
        self.free_q.put(i)
        return self.dev[d], n, first

    def __iter__(self):
        """yields (dev_ptr, n, first) of chunk k; meanwhile `self.ahead` = the same triple of chunk k + 1, already resident
        (None at the end of the clip), and chunk k + 2 is uploading behind the caller's compute"""
        d = 0

        def begin():
            nonlocal d
            item = self.ready_q.get()
            if item is None:
                return None
            fl = self._begin(item, d)
            d = (d + 1) % self.N_DEV
            return fl
        fl = begin()
        cur = self._wait(fl) if fl is not None else None
        fl = begin() if cur is not None else None
        self.ahead = self._wait(fl) if fl is not None else None
        while cur is not None:
            flight = begin() if self.ahead is not None else None       # overlaps the caller's compute on `cur`
            yield cur
            cur = self.ahead
            self.ahead = self._wait(flight) if flight is not None else None
        if self.error is not None:
            raise self.error

    def release(self):
        """Call after the work that reads the yielded device buffer has been enqueued."""
        L.check(self.ctx.lib.pp_upload_release(self.ctx.handle), "pp_upload_release")

    def close(self):
        self._stop = True
        self.free_q.put(None)
        self.thread.join(timeout=30)
        if self.thread.is_alive():      # never free staging memory under a reader that is still copying into it
            raise RuntimeError("frame reader thread did not stop")
        if getattr(self.ctx, "handle", None):
            self.ctx.synchronize()
            for p in self.pin_ptr:
                self.ctx.lib.pp_host_free(self.ctx.handle, p)
            for dptr in self.dev + ([self.dev_nv12] if self.dev_nv12 else []):
                self.ctx.free(dptr)
        self.pin, self.pin_ptr, self.dev = [], [], []
