"""
Device plumbing for the HIP hot path: PyTorch-ROCm tensors are the allocator/stream provider, every
computation is a libmagphase_hip.so call (ctypes, include/magphase_hip.h).  One Engine per GPU/process.

Data layout in HBM (all float32, row-major):
  sig      [sum_u n_u]          PCM of the batch's utterances, concatenated
  pos/left/right [F_tot]        per-frame epoch index into sig (int64) and Hann half lengths (int32)
  mag/real/imag  [F_tot x H]    lossless features, H = N/2+1 (same layout as the reference's arrays)
  frames   [F_tot x N]          epoch-centred time-domain frames (scratch between IFFT and PSOLA)
  pcm_out  [sum_u len_u]        resynthesised PCM, concatenated
"""
import ctypes
import os

import numpy as np

from . import _lib, hostmath as hm, hostplan


def _torch():
    import torch

    return torch


class HostTicket:
    """A batch result that is still on its way to the host: views of a page-locked ring slot filled by a non-blocking
    D2H copy.  wait() blocks until the copy has landed (releases the GIL: meant for the writer thread of iobatch),
    release() hands the slot back to the ring once the views have been consumed."""

    def __init__(self, ring, slot, event, keep):
        self._ring, self._slot, self._event, self._keep = ring, slot, event, keep

    def wait(self):
        if self._event is not None:
            self._event.synchronize()
            self._event = self._keep = None

    def release(self):
        if self._ring is not None:
            self.wait()
            self._ring.release(self._slot)
            self._ring = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class _PinnedRing:
    """A few page-locked host buffers (grown on demand, never shrunk).  A slot is reusable after its ticket's release();
    acquire() takes ANY free slot (a queue of free slots: two concurrent acquirers can never get the same one) and waits
    for one when the writer is behind -- the natural back-pressure of the pipeline.  If none comes free within
    MAGPHASE_RING_WAIT_S (default 20 s: e.g. a batch that needs more tickets at once than there are slots) it returns
    (None, None) and the caller falls back to a synchronous copy instead of stalling."""

    def __init__(self, slots=4):
        import queue
        import threading

        self._bufs = [None] * slots
        # SimpleQueue: release() runs from HostTicket.__del__, i.e. possibly from the cyclic GC while this very thread
        # is inside get() / put() -- queue.Queue's mutex is not reentrant (deadlock), SimpleQueue.put() is documented
        # safe from destructors and weakref callbacks
        self._freeq = queue.SimpleQueue()
        for k in range(slots):
            self._freeq.put(k)
        self._lock = threading.Lock()

    def acquire(self, nbytes, timeout=None):
        import queue

        torch = _torch()
        if timeout is None:
            timeout = float(os.environ.get("MAGPHASE_RING_WAIT_S", "20"))
        try:
            slot = self._freeq.get(timeout=timeout)
        except queue.Empty:
            return None, None
        try:
            with self._lock:
                buf = self._bufs[slot]
                if buf is None or buf.numel() < nbytes:
                    # Page-locking is slow (about 60 ms per 24 MB here): every slot is sized by the largest request so far
                    # with headroom, and a request that outgrows the slots re-sizes ALL the free ones at once -- the first
                    # (warm-up) launch of a job with larger launches pays, once, instead of each of the next launches paying
                    # for its own slot inside the job (round 5: bench.py's corpus shard ran at 2/3 of its warm rate because
                    # the slots sized by the earlier blocks re-grew one per launch).
                    size = max([int(nbytes * 1.25), 1 << 20] + [b.numel() for b in self._bufs if b is not None])
                    self._bufs[slot] = buf = torch.empty(size, dtype=torch.uint8).pin_memory()
                    idle = []
                    while True:   # the slots nobody holds right now
                        try:
                            idle.append(self._freeq.get_nowait())
                        except queue.Empty:
                            break
                    try:
                        for k in idle:
                            if self._bufs[k] is None or self._bufs[k].numel() < size:
                                self._bufs[k] = torch.empty(size, dtype=torch.uint8).pin_memory()
                    finally:
                        for k in idle:
                            self._freeq.put(k)
        except BaseException:
            self._freeq.put(slot)
            raise
        return slot, buf

    def release(self, slot):
        self._freeq.put(slot)


class Engine:
    def __init__(self, device=None):
        torch = _torch()
        self.lib = _lib.load()  # raises if the HIP library is missing: no fallback
        if not torch.cuda.is_available():
            raise _lib.MagphaseHipError("magphase_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                                        "there is no CPU fallback")
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self._tables = {}

    # ------------------------------------------------------------------ helpers
    def stream_ptr(self):
        return _torch().cuda.current_stream(self.device).cuda_stream

    def background(self, fn, *args):
        """fn(*args) on the engine's helper thread -> Future (result() re-raises).  For the native, GIL-free host passes of a
        plan build (staging copies into page-locked memory): they run while the constructor goes on with its index
        arithmetic, and are joined before the upload.  MAGPHASE_HOST_THREAD=0: inline."""
        import concurrent.futures as cf

        if os.environ.get("MAGPHASE_HOST_THREAD", "1") == "0":
            f = cf.Future()
            try:
                f.set_result(fn(*args))
            except BaseException as exc:   # noqa: B902 -- delivered by result(), as the threaded form does
                f.set_exception(exc)
            return f
        ex = getattr(self, "_helper", None)
        if ex is None:
            ex = self._helper = cf.ThreadPoolExecutor(max_workers=1, thread_name_prefix="mpx-host")
        return ex.submit(fn, *args)

    def copy_stream(self, kind):
        """The engine's H2D ('up') / D2H ('down') stream, or None (MAGPHASE_COPY_STREAMS=0: copies in the compute stream).
        A corpus job's launches are device-bound, and a third of a launch's device time was its own PCIe traffic queued
        in front of / behind its kernels (30 MB of PCM or 17 MB of coefficients up, 15 MB of PCM down): on their own
        streams the next launch's upload and the previous launch's download run beside this launch's kernels -- the
        host builds plans ahead of the device, so the uploads are there to be overlapped."""
        if os.environ.get("MAGPHASE_COPY_STREAMS", "1") == "0":
            return None
        cs = getattr(self, "_copy_streams", None)
        if cs is None:
            torch = _torch()
            cs = self._copy_streams = {"up": torch.cuda.Stream(self.device), "down": torch.cuda.Stream(self.device)}
        if kind == "rng" and os.environ.get("MAGPHASE_RNG_STREAM", "1") == "0":
            return None          # the noise generator's kernels in the compute stream (measured: generation -10 %)
        if kind not in cs:   # "rng": the noise stream's generator kernels (a few workgroups each: numpy_global_uniform)
            cs[kind] = _torch().cuda.Stream(self.device)
        return cs[kind]

    def _download(self, pairs):
        """pairs: [(pinned host tensor view, device tensor)]: non-blocking D2H copies on the download stream, behind
        what the compute stream has queued so far.  Returns the event that marks their completion."""
        torch = _torch()
        cur = torch.cuda.current_stream(self.device)
        down = self.copy_stream("down")
        if down is None:
            for dst, src in pairs:
                dst.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cur)
            return ev
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(down):
            down.wait_event(ready)
            for dst, src in pairs:
                dst.copy_(src, non_blocking=True)
                src.record_stream(down)
            ev = torch.cuda.Event()
            ev.record(down)
        return ev

    def empty(self, shape, dtype=None):
        torch = _torch()
        return torch.empty(shape, dtype=dtype or torch.float32, device=self.device)

    def empty_feats(self, n_frames, n_bins, ld=None):
        """
        One lossless feature matrix [n_frames x n_bins] float32 on the device, as a VIEW of a buffer whose rows are
        `ld` floats apart (default mpx_feat_ld(): the dense layout, measured fastest; MAGPHASE_FEAT_LD overrides it
        for experiments).  .stride(0) is the `ld` the C entry points take.
        """
        ld = int(ld or os.environ.get("MAGPHASE_FEAT_LD", 0) or self.lib.mpx_feat_ld(2 * (int(n_bins) - 1)) or n_bins)
        return self.empty((int(n_frames), ld))[:, :int(n_bins)]

    def feats_cat_to_device(self, parts, n_bins):
        """List of host [F_u x n_bins] arrays (float64 or float32) -> ONE dense device float32 matrix [sum F_u x n_bins]:
        narrowed / copied into the page-locked staging buffer by a few native threads (numpy's float64 -> float32 cast
        is one thread), then one DMA.  Returns None when the engine's feature pitch is not the dense one."""
        H = int(n_bins)
        if int(os.environ.get("MAGPHASE_FEAT_LD", 0) or self.lib.mpx_feat_ld(2 * (H - 1)) or H) != H:
            return None
        rows = [int(np.shape(p)[0]) for p in parts]
        total = int(sum(rows)) * H
        if total == 0 or total > (128 << 20):   # more than 512 MB per stream: not worth page-locking, the plain path does it
            return None
        stage = self.host_staging(total)
        n_thr = self.host_threads(4 * total, big=16)
        off = 0
        for p_, r in zip(parts, rows):
            a = np.asarray(p_)
            if a.ndim != 2 or a.shape[1] != H:
                raise ValueError("feature matrices must be [frames x %d]" % H)
            if a.dtype == np.float64 and a.flags.c_contiguous:
                if self.lib.mpx_host_narrow_f64(a.ctypes.data, stage[off:off + r * H].ctypes.data, r * H, n_thr) != 0:
                    raise _lib.MagphaseHipError("mpx_host_narrow_f64 failed")
            else:
                stage[off:off + r * H] = a.reshape(-1)
            off += r * H
        return self.upload_staged(total).view(int(sum(rows)), H)

    def feats_to_device(self, arr):
        """Host [F x H] array -> device float32 matrix (see empty_feats)."""
        torch = _torch()
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        out = self.empty_feats(arr.shape[0], arr.shape[1])
        out.copy_(torch.from_numpy(arr))
        return out

    # ------------------------------------------------------------------ pinned staging
    def _pinned_ring(self, n_floats, depth):
        """`depth` page-locked float32 staging buffers of at least n_floats elements (grown on demand, kept)."""
        torch = _torch()
        cur = getattr(self, "_pinned", None)
        if cur is None or len(cur) < depth or cur[0].numel() < n_floats:   # grown with headroom: page-locking is slow
            n_alloc = max(int(n_floats * 1.5), 1 << 20)
            self._pinned = tuple(torch.empty(n_alloc, dtype=torch.float32).pin_memory() for _ in range(depth))
        return self._pinned

    def to_host_f64_many(self, tensors, chunk_bytes=None, depth=3):
        """
        Device float32 tensors (1-D or 2-D, rows may be pitched) -> fresh float64 numpy arrays through a ring of pinned
        staging buffers: the chunks of ALL tensors form one pipeline -- chunks i+1, i+2 cross PCIe (async copies on the
        current stream) while the host widens chunk i to float64 (native threads, streaming stores).  One matrix at a
        time with two 64 MB chunks each, the three feature matrices of a batch paid the pipeline's fill and drain
        three times (12 of a call's 16 ms); as one stream of 16 MB chunks the copies hide behind the widening.
        Bounded pinned memory (depth x chunk_bytes) whatever the sizes.
        """
        torch = _torch()
        # the widening writes 2 bytes for every byte that crosses PCIe: eight threads sustain ~75 GB/s of streaming stores on
        # this host, half of what the link delivers (round 6: 16 utterances' features 14 ms for 0.35 GB); 32 threads from
        # 16 MB up (Engine.host_threads: capped by the cores this rank may use)
        # (tools/array_api_probe.py, 16 utterances: 8 / 16 / 32 / 64 threads = 0.54 / 0.73 / 0.64 / 0.63 M frames/s)
        n_thr = self.host_threads(sum(int(t.numel()) * 4 for t in tensors), big=16)
        if chunk_bytes is None:
            chunk_bytes = int(os.environ.get("MAGPHASE_D2H_CHUNK_MB", "32")) << 20
        outs, views, work = [], [], []
        for k, t in enumerate(tensors):
            v = t.view(1, -1) if t.dim() == 1 else t
            rows, cols = int(v.shape[0]), int(v.shape[1])
            out = np.empty((rows, cols), dtype=np.float64)
            outs.append(out.reshape(-1) if t.dim() == 1 else out)
            views.append((v, out, cols))
            if rows and cols:
                rows_per = max(1, int(chunk_bytes) // (4 * cols))
                work.extend((k, r0, min(rows, r0 + rows_per)) for r0 in range(0, rows, rows_per))
        if not work:
            return outs
        biggest = max((r1 - r0) * views[k][2] for k, r0, r1 in work)
        bufs = self._pinned_ring(biggest, depth)
        events = [torch.cuda.Event() for _ in range(depth)]

        def drain(i):
            k, r0, r1 = work[i]
            _v, out, cols = views[k]
            events[i % depth].synchronize()
            if self.lib.mpx_host_widen_f32(bufs[i % depth].data_ptr(), out[r0:r1].ctypes.data, (r1 - r0) * cols, n_thr) != 0:
                raise _lib.MagphaseHipError("mpx_host_widen_f32 failed")

        with torch.cuda.device(self.device):
            for i, (k, r0, r1) in enumerate(work):
                if i >= depth:
                    drain(i - depth)                  # the buffer about to be overwritten
                v, _out, cols = views[k]
                bufs[i % depth][:(r1 - r0) * cols].view(r1 - r0, cols).copy_(v[r0:r1], non_blocking=True)
                events[i % depth].record()
            for i in range(max(0, len(work) - depth), len(work)):
                drain(i)
        return outs

    def to_host_f64(self, t, chunk_bytes=None):
        """One tensor through to_host_f64_many."""
        return self.to_host_f64_many([t], chunk_bytes=chunk_bytes)[0]

    def to_host_f32(self, t):
        """Device float32 tensor (rows may be pitched) -> fresh float32 numpy array (one D2H copy into a new pageable
        array: for the megabyte-sized compressed features that beats staging + a second host copy)."""
        torch = _torch()
        dst = torch.empty(tuple(int(x) for x in t.shape), dtype=torch.float32)
        if dst.numel():
            with torch.cuda.device(self.device):
                dst.copy_(t)
        return dst.numpy()

    def out_ring(self):
        r = getattr(self, "_out_ring", None)
        if r is None:
            r = self._out_ring = _PinnedRing(slots=max(2, int(os.environ.get("MAGPHASE_OUT_RING_SLOTS", "4"))))
        return r

    def to_host_f32_async(self, tensors):
        """Device float32 tensors -> float32 numpy VIEWS of one page-locked ring slot, copied without blocking; returns
        (views, HostTicket).  The views are valid after ticket.wait() and until ticket.release()."""
        torch = _torch()
        sizes = [int(t.numel()) for t in tensors]
        offs = np.concatenate(([0], np.cumsum([(n + 63) // 64 * 64 for n in sizes]))).astype(np.int64)
        slot, buf = self.out_ring().acquire(4 * int(offs[-1]) + 256)
        if slot is None:   # ring exhausted: plain synchronous copies, a ticket with nothing to wait for
            return [t.detach().to("cpu").numpy() for t in tensors], HostTicket(None, None, None, None)
        try:
            host = buf[:4 * int(offs[-1])].view(torch.float32)
            views, pairs = [], []
            with torch.cuda.device(self.device):
                for t, n, o in zip(tensors, sizes, offs[:-1]):
                    dst = host[int(o):int(o) + n].view(tuple(int(x) for x in t.shape))
                    if n:
                        pairs.append((dst, t))
                    views.append(dst.numpy())
                ev = self._download(pairs)
        except BaseException:
            self.out_ring().release(slot)   # a failed copy must not leak the slot
            raise
        return views, HostTicket(self.out_ring(), slot, ev, list(tensors))

    def output_pcm16(self, y, out_off_host, norm=0.98, async_out=False):
        """
        libaudio.py:352-365 on the device (mpx_pcm16): y float64 or float32 [total] (utterances at out_off_host) ->
        int16 numpy [total], each utterance peak-normalised to `norm` (None: no normalisation) and rounded like
        libsndfile's PCM_16 conversion -- bit-identical to la.write_audio_file's samples.
        async_out: returns (view of a page-locked ring slot, HostTicket) without waiting for the copy.
        """
        torch = _torch()
        out_off_host = np.asarray(out_off_host, dtype=np.int64)
        lens = np.diff(out_off_host)
        total = int(out_off_host[-1])
        d_off = self.to_device(out_off_host, np.int64)
        peaks = torch.empty(max(lens.size, 1), dtype=torch.float64, device=self.device)
        out = torch.empty(max(total, 1), dtype=torch.int16, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mpx_pcm16(self.stream_ptr(), y.data_ptr(), 1 if y.dtype == torch.float64 else 0,
                                          d_off.data_ptr(), int(lens.size), int(lens.max()) if lens.size else 0,
                                          float(norm) if norm is not None else 0.0, peaks.data_ptr(), out.data_ptr()),
                       "mpx_pcm16")
            if async_out:   # non-blocking copy into a page-locked ring slot; the consumer waits on the ticket
                slot, buf = self.out_ring().acquire(2 * max(total, 1))
                if slot is not None:
                    try:
                        host = buf[:2 * max(total, 1)].view(torch.int16)
                        ev = self._download([(host, out)])
                    except BaseException:
                        self.out_ring().release(slot)
                        raise
                    return host[:total].numpy(), HostTicket(self.out_ring(), slot, ev, [out, peaks, d_off])
                host = torch.empty(max(total, 1), dtype=torch.int16)   # ring exhausted: synchronous copy
                host.copy_(out)
                return host[:total].numpy(), HostTicket(None, None, None, None)
            # one D2H copy into a fresh pageable array (pinning a new 15-30 MB buffer per batch cost 7 ms, more than the copy)
            host = torch.empty(max(total, 1), dtype=torch.int16)
            host.copy_(out)
        return host[:total].numpy()

    @staticmethod
    def _mt_next_pos(pos, n_words):
        """Position of numpy's MT19937 word cursor (0..624) after n_words more 32-bit draws (randomkit: a draw at 624
        regenerates the state and restarts at 0)."""
        q = int(pos) + int(n_words)
        return q if (n_words == 0 or q <= 624) else ((q - 1) % 624) + 1

    def _mt_generate(self, key, pos, n, key_from_compute):
        """n uniforms continuing numpy's MT19937 stream from (key [624 words on the device], word cursor pos), on the
        generator's own stream: (samples float32 [n], state int32 [625] = key + cursor after them, event).
        key_from_compute: the key was just uploaded in the compute stream (the generator's stream waits for it; a state
        that comes from the generator's own stream needs no wait -- and must not get one: waiting for the compute stream here
        is waiting for the previous launch)."""
        torch = _torch()
        # The generator's kernels are a ladder of launches of 1 .. 128 workgroups: they run on their own stream, beside
        # whatever the compute stream has queued (the previous launch's synthesis), one generation after the other; the
        # compute stream waits for the samples' event where it uses them.
        rng = self.copy_stream("rng")
        done = None
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            if rng is not None:
                ready = None
                if key_from_compute:
                    ready = torch.cuda.Event()
                    ready.record(cur)
                ctx = torch.cuda.stream(rng)
            else:
                import contextlib
                ctx = contextlib.nullcontext()
            with ctx:
                if rng is not None and ready is not None:
                    rng.wait_event(ready)
                out = self.empty((max(int(n), 1),))
                raw = torch.empty(max(2 * int(n), 1), dtype=torch.int32, device=self.device)
                state = torch.empty(625, dtype=torch.int32, device=self.device)
                work = getattr(self, "_mt_work", None)
                if work is None:   # segment windows + jump polynomials of the many-workgroup form
                    work = self._mt_work = torch.empty(int(self.lib.mpx_noise_numpy_mt19937_work_words()),
                                                       dtype=torch.int32, device=self.device)
                _lib.check(self.lib.mpx_noise_numpy_mt19937(self.stream_ptr(), key.data_ptr(), int(pos), int(n),
                                                            raw.data_ptr(), out.data_ptr(), state.data_ptr(),
                                                            state.data_ptr() + 4 * 624, work.data_ptr()),
                           "mpx_noise_numpy_mt19937")
                if rng is not None:
                    done = torch.cuda.Event()
                    done.record(rng)
                    for t_ in (key,):
                        t_.record_stream(rng)
        return out, state, done

    def numpy_global_uniform(self, n, defer=False):
        """np.random.uniform(-1, 1, n).astype(float32) drawn from numpy's global generator, on the device
        (mpx_noise_numpy_mt19937): same values, and the global state is left where the host draw would leave it.
        defer=True (the batches of a corpus run, iobatch): the advanced state STAYS on the device and the next deferred
        call continues from it -- no download, no synchronisation per batch; numpy's own state is stale until mt_sync(),
        which the caller owes before anything else draws from it.  Deferred draws can be generated AHEAD
        (MAGPHASE_MT_AHEAD = k: k requests' worth per generation; a call that finds its samples in what an earlier call
        produced launches nothing; mt_sync() puts numpy's state where the samples actually HANDED OUT end).  Measured in
        round 6 on the generation workload: no gain (k = 4 / 8: 118-160 / 108-146 k x real time against 133-170 k at k = 1
        on the same box -- the larger draws' allocations and copies cost what the saved jump ladders gain), so k = 1."""
        torch = _torch()
        n = int(n)
        pend = getattr(self, "_mt_pending", None)
        fresh = pend is None
        if fresh:
            st = np.random.get_state()
            if st[0] != "MT19937":
                raise RuntimeError("numpy's global generator is not MT19937")
            key = self.to_device(np.ascontiguousarray(st[1], dtype=np.uint32).view(np.int32), np.int32)
            pend = {"key": key, "pos": int(st[2]), "meta": (st[0], st[3], st[4]), "buf": None, "gen": 0, "lead": 0, "used": 0,
                    "end_state": None, "end_pos": int(st[2]), "done": None}
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            if pend["buf"] is not None and pend["used"] + n <= pend["gen"]:      # already generated
                out = pend["buf"][pend["used"]:pend["used"] + n]
                pend = dict(pend, used=pend["used"] + n)
            else:
                left = pend["gen"] - pend["used"] if pend["buf"] is not None else 0
                ahead = max(1, int(os.environ.get("MAGPHASE_MT_AHEAD", "1"))) if defer else 1
                n_gen = max(n - left, ahead * n - left, 0)
                if pend["buf"] is not None:      # continue where the generated samples end
                    key, pos, from_compute = pend["end_state"], pend["end_pos"], False
                else:
                    key, pos, from_compute = pend["key"], pend["pos"], fresh
                gen, state, done = self._mt_generate(key, pos, n_gen, from_compute)
                if left:     # the unused tail of the previous generation in front of the new samples: one contiguous draw
                    rng = self.copy_stream("rng")
                    buf = self.empty((left + n_gen,))
                    ctx = torch.cuda.stream(rng) if rng is not None else None
                    if ctx is not None:
                        ctx.__enter__()
                    try:
                        buf[:left].copy_(pend["buf"][pend["used"]:pend["gen"]])
                        buf[left:].copy_(gen[:n_gen])
                        if rng is not None:
                            done = torch.cuda.Event()
                            done.record(rng)
                            buf.record_stream(rng)
                    finally:
                        if ctx is not None:
                            ctx.__exit__(None, None, None)
                else:
                    buf = gen
                pend = {"key": key, "pos": int(pos), "meta": pend["meta"], "buf": buf, "gen": left + n_gen, "lead": left,
                        "used": n, "end_state": state, "end_pos": self._mt_next_pos(pos, 2 * n_gen), "done": done}
                out = buf[:n]
            if pend["done"] is not None:
                cur.wait_event(pend["done"])       # everything the caller enqueues from here on sees the samples
                pend["buf"].record_stream(cur)
        self._mt_pending = pend
        if not defer:
            self.mt_sync()
        return out

    def mt_sync(self):
        """Puts a deferred MT19937 state (numpy_global_uniform(defer=True)) back into numpy's global generator: the state
        after the samples handed out so far (samples generated ahead and not handed out are dropped)."""
        pend = getattr(self, "_mt_pending", None)
        if pend is None:
            return
        self._mt_pending = None
        if pend["buf"] is None:
            return
        used_gen, total_gen = pend["used"] - pend["lead"], pend["gen"] - pend["lead"]
        if used_gen < 0:
            raise RuntimeError("MT19937: cursor inside the carried-over samples")
        if used_gen == total_gen:
            state, pos = pend["end_state"], pend["end_pos"]
        else:      # numpy's state where the handed-out samples end: the generation repeated up to there (once per job)
            _g, state, _d = self._mt_generate(pend["key"], pend["pos"], used_gen, False)
            pos = self._mt_next_pos(pend["pos"], 2 * used_gen)
        torch = _torch()
        with torch.cuda.device(self.device):
            rng = self.copy_stream("rng")
            if rng is not None:
                rng.synchronize()
            h = state.cpu().numpy()          # synchronises
        if int(h[624]) != int(pos):
            raise RuntimeError("MT19937 cursor: host %d, device %d" % (pos, int(h[624])))
        meta = pend["meta"]
        np.random.set_state((meta[0], h[:624].view(np.uint32).copy(), int(pos), meta[1], meta[2]))

    def mt_snapshot(self):
        """Opaque copy of the generator's current state (deferred device state or numpy's), for mt_restore.  (The device
        tensors of a deferred state are never written again once generated: the snapshot shares them.)"""
        pend = getattr(self, "_mt_pending", None)
        if pend is not None:
            return ("dev", dict(pend))
        return ("host", np.random.get_state())

    def mt_restore(self, snap):
        kind, val = snap
        if kind == "dev":
            self._mt_pending = dict(val)
        else:
            self._mt_pending = None
            np.random.set_state(val)

    def host_staging(self, n_floats):
        """float32 numpy view [n_floats] of a page-locked staging buffer (grown on demand, reused by every plan).  TWO
        buffers alternate: the DMA out of one (upload_staged, not waited for) runs while the host fills the other for the
        next plan; a buffer is waited for only when its turn comes again.  Always paired with upload_staged, one thread."""
        torch = _torch()
        st = getattr(self, "_stage", None)
        if st is None:
            st = self._stage = {"bufs": [None, None], "events": [None, None], "cur": 1}
        k = st["cur"] = 1 - st["cur"]
        if st["events"][k] is not None:
            st["events"][k].synchronize()
            st["events"][k] = None
        cur = st["bufs"][k]
        if cur is None or cur.numel() < n_floats:   # grown with headroom (batches of a corpus differ a little in length:
            # re-pinning 30 MB for every slightly longer batch cost 40 ms each); the OTHER buffer grows with it when it is
            # idle, so that the second launch of a job does not pay for it inside the job
            size = max(int(n_floats * 1.5), 1 << 22)
            st["bufs"][k] = cur = torch.empty(size, dtype=torch.float32).pin_memory()
            o = 1 - k
            if (st["bufs"][o] is None or st["bufs"][o].numel() < size) and (st["events"][o] is None or st["events"][o].query()):
                st["events"][o] = None
                st["bufs"][o] = torch.empty(size, dtype=torch.float32).pin_memory()
        self._stage_up = cur
        return cur.numpy()[:int(n_floats)]

    def stage_rows(self, arrays, out):
        """np.concatenate(arrays, axis=0, out=out, casting='same_kind') for a float32 ``out`` (a slice of the staging
        buffer): float32 C-contiguous blocks are copied, float64 ones narrowed (round to nearest even, as astype), both on
        a few native threads (mpx_host_copy_many / mpx_host_narrow_f64) -- numpy's concatenate is one thread at ~10 GB/s and
        was a quarter of a synthesis plan's build time.  Anything else goes through numpy."""
        # (launches of 100+ utterances stage tens of MB per matrix: 8 / 16 / 32 threads = 116 / 131 / 144 k x real time)
        n_thr = self.host_threads(out.nbytes)
        k = len(arrays)
        if k > 1 and all(a.dtype == np.float32 and a.flags.c_contiguous for a in arrays):
            src = (ctypes.c_void_p * k)(*[a.ctypes.data for a in arrays])
            nb = np.fromiter((a.nbytes for a in arrays), dtype=np.int64, count=k)
            doff = np.zeros(k, dtype=np.int64)
            np.cumsum(nb[:-1], out=doff[1:])
            if int(nb.sum()) != out.nbytes:
                raise ValueError("stage_rows: blocks do not fill the destination")
            if self.lib.mpx_host_copy_many(k, src, nb.ctypes.data, doff.ctypes.data, out.ctypes.data, n_thr) != 0:
                raise _lib.MagphaseHipError("mpx_host_copy_many failed")
            return
        if k > 0 and all(a.dtype == np.float64 and a.flags.c_contiguous for a in arrays):
            flat, o = out.reshape(-1), 0
            if int(sum(a.size for a in arrays)) != flat.size:
                raise ValueError("stage_rows: blocks do not fill the destination")
            for a in arrays:
                if a.size and self.lib.mpx_host_narrow_f64(a.ctypes.data, flat[o:].ctypes.data, a.size, n_thr) != 0:
                    raise _lib.MagphaseHipError("mpx_host_narrow_f64 failed")
                o += a.size
            return
        np.concatenate(arrays, axis=0, out=out, casting="same_kind")

    def upload_staged(self, n_floats):
        """The first n_floats of the current staging buffer -> a fresh device tensor (one DMA from pinned memory, in
        stream order; the buffer is protected by an event until host_staging hands it out again)."""
        torch = _torch()
        st = self._stage
        up = self.copy_stream("up")
        with torch.cuda.device(self.device):
            if up is None:
                t = self._stage_up[:int(n_floats)].to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
            else:   # on the upload stream; the compute stream waits for it, the tensor is the compute stream's from then on
                cur = torch.cuda.current_stream(self.device)
                with torch.cuda.stream(up):
                    t = self._stage_up[:int(n_floats)].to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(up)
                cur.wait_event(ev)
                t.record_stream(cur)
            st["events"][st["cur"]] = ev
        return t

    # ------------------------------------------------------------------ prepared launches (native planners, planner thread)
    def host_threads(self, nbytes=0, big=32):
        """Native threads one staging pass of `nbytes` may use: MAGPHASE_IO_NATIVE_THREADS, or 8 (32 from 16 MB up: launches of
        100+ utterances), never more than the cores this process may run on (sharding.bind_rank_to_cores gives every rank of
        a node its own share -- the reference's model is one worker per core with nothing shared, libutils.py:61-62)."""
        env = os.environ.get("MAGPHASE_IO_NATIVE_THREADS")
        want = int(env) if env else (int(big) if nbytes >= (16 << 20) else 8)
        try:
            cores = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            cores = os.cpu_count() or 1
        return max(1, min(want, cores))

    # Staging slots.  A slot is held from prepare_* until the upload out of it has completed; a generation batch is TWO launches
    # (one per sample rate) and the planner works one batch ahead of the thread that enqueues, so four are in use at once: with
    # three (rounds 5-6) the planner thread waited for a slot in every batch, the enqueuing thread for the planner, and the
    # device for the upload -- 13-19 ms of a 45 ms generation pass (tools/corpus_marks_probe.py)
    _N_SLOTS = max(2, int(os.environ.get("MAGPHASE_STAGE_SLOTS", "6")))

    def _slot_acquire(self, stage_bytes, desc_bytes, wait=True):
        """One of the engine's sets of page-locked buffers (sample / coefficient staging + table image) for a prepared launch;
        blocks while all are in use (wait=False: returns None instead -- a plan constructor that prepares its own launch must
        not wait for slots that launches prepared AHEAD of it hold: they are committed after it).  A slot is handed out again
        only after the H2D copies out of it have completed."""
        import queue

        torch = _torch()
        pool = getattr(self, "_slots", None)
        if pool is None:
            import threading

            with self.__dict__.setdefault("_slots_lock", threading.Lock()):
                pool = getattr(self, "_slots", None)
                if pool is None:
                    # tokens in a SimpleQueue (blocking; safe to put from a destructor running inside the cyclic collector,
                    # unlike queue.Queue's mutex), the slots themselves on a stack: the most recently returned slot -- page-
                    # locked, sized, its event long since passed -- goes out first, so one launch at a time keeps reusing ONE
                    # warm slot instead of walking through all of them (each first use pins tens of MB: 40 ms)
                    import collections
                    pool = queue.SimpleQueue()
                    self._slot_stack = collections.deque()
                    for _ in range(self._N_SLOTS):
                        self._slot_stack.append({"stage": None, "desc": None, "event": None})
                        pool.put(None)
                    self._slots = pool
        try:
            pool.get(block=bool(wait))
        except queue.Empty:
            return None
        # the most recently returned slot whose upload has completed (warm, and free NOW); none: the oldest one
        slot = None
        with self._slots_lock:   # (two acquirers -- the planner thread and a caller on the generic path -- must not pick the same entry)
            for k in range(len(self._slot_stack) - 1, -1, -1):
                ev_ = self._slot_stack[k]["event"]
                if ev_ is None or ev_.query():
                    slot = self._slot_stack[k]
                    del self._slot_stack[k]
                    break
            if slot is None:
                slot = self._slot_stack.popleft()
        try:
            if slot["event"] is not None:
                slot["event"].synchronize()
                slot["event"] = None
            def grow(sl, sizes):
                for key, need in sizes:
                    cur = sl[key]
                    if cur is None or cur.numel() < need:
                        sl[key] = None
                        sl[key] = torch.empty(need, dtype=torch.uint8).pin_memory()
                        sl[key + "_np"] = sl[key].numpy()

            with torch.cuda.device(self.device):
                sizes = []
                for key, need in (("stage", int(stage_bytes)), ("desc", int(desc_bytes))):
                    cur = slot[key]
                    if cur is None or cur.numel() < need:   # grown with headroom: page-locking is slow (40 ms per 30 MB)
                        sizes.append((key, max(int(need * 1.5), 1 << 20)))
                if sizes:
                    grow(slot, sizes)
                    # ... and the slots nobody holds grow with it, NOW: a job's first (warm-up) launch pays for all of them
                    # instead of the next launches paying one by one inside the job (as the output ring does)
                    idle = []
                    with self._slots_lock:
                        while True:
                            try:
                                pool.get_nowait()
                            except queue.Empty:
                                break
                            idle.append(self._slot_stack.pop())
                    try:
                        for sl in idle:
                            if sl["event"] is None or sl["event"].query():
                                sl["event"] = None
                                grow(sl, sizes)
                    finally:
                        for sl in idle:
                            self._slot_stack.append(sl)
                            pool.put(None)
        except BaseException:
            self._slot_stack.append(slot)
            pool.put(None)
            raise
        return slot

    def _slot_release(self, slot, event=None):
        slot["event"] = event
        self._slot_stack.append(slot)
        self._slots.put(None)

    def _slot_upload(self, slot, stage_bytes, desc_bytes):
        """The first stage_bytes / desc_bytes of the slot's two buffers -> device uint8 tensors (two DMAs on the upload stream);
        the slot goes back to the pool guarded by the copies' event.  Returns (stage_dev, desc_dev, ready event)."""
        torch = _torch()
        up = self.copy_stream("up")
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            ctx = torch.cuda.stream(up) if up is not None else None
            if ctx is not None:
                ctx.__enter__()
            try:
                sd = slot["stage"][:max(int(stage_bytes), 1)].to(self.device, non_blocking=True)
                dd = slot["desc"][:max(int(desc_bytes), 1)].to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(up if up is not None else cur)
            finally:
                if ctx is not None:
                    ctx.__exit__(None, None, None)
            if up is not None:
                cur.wait_event(ev)
                sd.record_stream(cur), dd.record_stream(cur)
        self._slot_release(slot, ev)
        return sd, dd, ev

    def planner(self):
        """The engine's planner thread (one): prepare_* calls for launch i + 1 run here while the calling thread enqueues
        launch i -- the native planners release the interpreter lock.  MAGPHASE_PLANNER_THREAD=0: inline."""
        import concurrent.futures as cf

        ex = getattr(self, "_planner", None)
        if ex is None:
            # (one worker: two -- the two launches of a generation batch prepared side by side -- measured 207-214 k x real time
            # against 219-221 k on the same box: their staging copies compete for the same memory bandwidth)
            ex = self._planner = cf.ThreadPoolExecutor(max_workers=1, thread_name_prefix="mpx-plan")
        return ex

    def prepare_async(self, kind, *args, **kw):
        """Future of prepare_analysis / prepare_synthesis (kind 'analysis' / 'synthesis') on the planner thread."""
        import concurrent.futures as cf

        fn = self.prepare_analysis if kind == "analysis" else self.prepare_synthesis
        if os.environ.get("MAGPHASE_PLANNER_THREAD", "1") == "0":
            f = cf.Future()
            try:
                f.set_result(fn(*args, **kw))
            except BaseException as exc:   # noqa: B902 -- delivered by result()
                f.set_exception(exc)
            return f
        return self.planner().submit(fn, *args, **kw)

    def prepare_analysis(self, utts, fft_len=None, wait=True):
        """The host side of an analysis launch (LosslessAnalysisPlan / CompressedAnalysisPlan) without touching a stream:
        utterance list walked in native code, samples staged into a page-locked slot, frame tables written in their device
        types (mpx_host_plan_analysis_batch), f0 and its median-3 on the host.  Returns a PreparedAnalysis, or None when
        the batch is not in the plain shape the native path handles (the plan constructor then takes the generic path,
        which converts -- or raises what the reference's arithmetic raises)."""
        ph = hostplan.pyhost()
        if ph is None or not utts:
            return None
        m = ph.analysis_marshal(utts)
        if m is None:
            return None
        U, total, E, all_i16 = ph.analysis_info(m)
        fs_list = [u[1] for u in utts]
        N = None
        for fs in set(fs_list):
            n_ = fft_len if fft_len is not None else hm.define_fft_len(fs)
            if N is not None and n_ != N:
                return None    # the generic path raises "all utterances of a plan must share fft_len"
            N = n_
        if E <= 0 or total <= 0:
            return None
        stage_bytes = (2 * total + 8) if all_i16 else 4 * total
        a256 = lambda n: (int(n) + 255) // 256 * 256   # noqa: E731
        o_pos, o_left, o_right, o_voi = 0, a256(8 * E), a256(8 * E) + a256(4 * E), a256(8 * E) + 2 * a256(4 * E)
        desc_bytes = o_voi + a256(4 * E)
        slot = self._slot_acquire(stage_bytes, desc_bytes, wait=wait)
        if slot is None:
            return None
        try:
            d = slot["desc_np"]
            pm, left64, f0, f0_med = (np.empty(E, dtype=np.int64), np.empty(E, dtype=np.int64), np.empty(E), np.empty(E))
            frame_off = np.empty(U + 1, dtype=np.int64)
            long_f, long_l = np.empty(512, dtype=np.int64), np.empty(512, dtype=np.int64)
            F, n_long = ph.analysis_run(m, slot["stage"].data_ptr(), 0 if all_i16 else 1, d[o_pos:o_pos + 8 * E],
                                        d[o_left:o_left + 4 * E], d[o_right:o_right + 4 * E], d[o_voi:o_voi + 4 * E], pm,
                                        left64, f0, f0_med, frame_off, int(N), long_f, long_l,
                                        self.host_threads(stage_bytes))
            if F < 0 or n_long > 512:
                self._slot_release(slot)
                return None
        except BaseException:
            self._slot_release(slot)
            raise
        p = PreparedAnalysis()
        p.engine, p.slot, p.n_utts, p.fs, p.fft_len = self, slot, U, fs_list, int(N)
        p.total_smpls, p.total_frames, p.all_i16 = int(total), int(F), bool(all_i16)
        p.stage_bytes, p.desc_bytes, p.cap = stage_bytes, desc_bytes, int(E)
        p.offs = (o_pos, o_left, o_right, o_voi)
        p.frame_off, p.pm, p.left64, p.f0, p.f0_med = frame_off, pm[:F], left64[:F], f0[:F], f0_med[:F]
        p.long = [(int(long_f[k]), int(long_l[k])) for k in range(int(n_long))]
        return p

    def _comp_slot_shares(self):
        """(slots of the compressed synthesis kernel, np.concatenate(([0], cumsum(w))), w.sum()) of their float64 weights --
        what mpx_host_plan_synthesis_batch deals the frames by (hostmath.slot_cuts' operands, evaluated by numpy once)."""
        w = self.synth_ola_slot_weights(comp=True)
        n = self.host_constant("comp_slots_n", self.synth_comp_slots)
        if w is None:
            return n, None, 0.0
        key = ("comp_shares", id(w))
        if key not in self._tables:
            w64 = np.asarray(w, dtype=np.float64)[:n]
            self._tables[key] = (np.ascontiguousarray(np.concatenate(([0.0], np.cumsum(w64)))), float(w64.sum()))
        return (n,) + self._tables[key]

    def prepare_synthesis(self, utts, fs, fft_len=None, b_voi_ap_win=True, b_const_rate=False, wait=True):
        """The host side of a compressed-feature synthesis launch (CompressedSynthesisPlan) without touching a stream:
        coefficient rows staged into a page-locked slot, every device table written in its final type
        (mpx_host_plan_synthesis_batch).  Returns a PreparedSynthesis, or None when the batch is not in the plain shape the
        native path handles or the numpy form would raise (the plan constructor then takes the generic path)."""
        ph = hostplan.pyhost()
        if ph is None or not utts:
            return None
        m = ph.synthesis_marshal(utts)
        if m is None:
            return None
        U, R, mag_dim, phase_dim = ph.synthesis_info(m)
        if R <= 0 or R >= (1 << 30):
            return None
        N = int(fft_len) if fft_len else hm.define_fft_len(fs)
        f0 = np.empty(R)
        ph.synthesis_lf0(m, f0)
        np.exp(f0, out=f0)                                               # magphase.py:846 (numpy's exp, as the array API)
        n_slots, wcum, wsum = self._comp_slot_shares()
        unwarp_rows = bool(b_const_rate) or os.environ.get("MAGPHASE_UNWARP_ROWS_VAR", "1") != "0"
        stage_bytes = 4 * R * (mag_dim + 2 * phase_dim)
        desc_cap = hostplan.synth_desc_bytes(R, U, n_slots, True)
        cap = 2 * R + 2 * U
        slot = self._slot_acquire(stage_bytes, desc_cap, wait=wait)
        if slot is None:
            return None
        try:
            v_shift, v_pm = np.empty(cap, dtype=np.int64), np.empty(cap, dtype=np.int64)
            voiced_host = np.empty(cap, dtype=np.int32)
            frame_off = np.empty(U + 1, dtype=np.int64)
            ns_len, out_start, out_len = (np.empty(U, dtype=np.int64) for _ in range(3))
            runs_host = np.zeros(U + n_slots + 1, dtype=hm.OLA_RUN_DTYPE)
            counts, desc_off = np.zeros(8, dtype=np.int64), np.zeros(18, dtype=np.int64)
            def run(n_sl, wc, ws, stage_ptr):
                return ph.synthesis_run(m, stage_ptr, f0, float(fs), N, int(bool(b_const_rate)), int(bool(b_voi_ap_win)),
                                        int(n_sl), wc, float(ws), int(unwarp_rows), slot["desc_np"][:desc_cap], desc_off,
                                        v_shift, v_pm, voiced_host, frame_off, ns_len, out_start, out_len, runs_host, counts,
                                        self.host_threads(stage_bytes))

            F = run(n_slots, wcum, wsum, slot["stage"].data_ptr())
            if F == -4000000 and 0 < counts[0] < n_slots:   # fewer frames than slots: shares by the first F weights
                nf = int(counts[0])                        # (hostmath.slot_cuts: w[:ns], their sum numpy's)
                w64 = np.asarray(self.synth_ola_slot_weights(comp=True), dtype=np.float64)[:nf]
                F = run(nf, np.ascontiguousarray(np.concatenate(([0.0], np.cumsum(w64)))), float(w64.sum()), 0)
            if F < 0:
                self._slot_release(slot)
                return None
        except BaseException:
            self._slot_release(slot)
            raise
        p = PreparedSynthesis()
        p.engine, p.slot = self, slot
        p.key = (int(fs), N, bool(b_const_rate), bool(b_voi_ap_win), unwarp_rows)
        p.n_utts, p.n_rows, p.mag_dim, p.phase_dim = int(U), int(R), int(mag_dim), int(phase_dim)
        p.total_frames, p.n_runs, p.n_slots = int(F), int(counts[1]), int(counts[2])
        p.stage_bytes, p.desc_bytes, p.n_tiles1 = stage_bytes, int(counts[3]), int(counts[6])
        p.desc_off = desc_off
        p.v_shift, p.v_pm, p.voiced_host = v_shift[:F], v_pm[:F], voiced_host[:F]
        p.frame_off, p.ns_len, p.out_start, p.out_len = frame_off, ns_len, out_start, out_len
        p.runs_host = runs_host[:p.n_runs]
        return p

    def to_device_pinned(self, arr, dtype):
        """Host array -> device tensor through a page-locked copy (async H2D on the current stream)."""
        torch = _torch()
        h = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype)).pin_memory()
        return h.to(self.device, non_blocking=True)

    @staticmethod
    def feat_ld(mag, real, imag):
        """Common row pitch of three feature views (unit column stride, equal row stride) for the C ABI."""
        ld = int(mag.stride(0)) if mag.shape[0] > 1 else max(int(mag.stride(0)), int(mag.shape[1]))
        for t in (mag, real, imag):
            if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
                raise ValueError("feature matrices must be 2-D with unit column stride")
            if t.shape[0] > 1 and int(t.stride(0)) != ld:
                raise ValueError("mag/real/imag must share one row pitch")
        return ld

    def host_constant(self, key, build):
        """A host value computed once per key."""
        if key not in self._tables:
            self._tables[key] = build()
        return self._tables[key]

    def constant(self, key, build, dtype=np.float32):
        """Device-resident constant table, built (float64 on the host) and uploaded once per key."""
        if key not in self._tables:
            self._tables[key] = self.to_device(build(), dtype)
        return self._tables[key]

    def to_device_packed(self, items):
        """
        items: list of (name, array, numpy dtype).  ONE host-to-device copy for all of them (each ~25 us on its own: a
        plan has 15-25 small index tables); returns {name: tensor}, the tensors being 256-byte aligned views of one
        device buffer.  The arrays are cast straight into the page-locked arena (one pass: no intermediate host image) and
        the copy never blocks: round 5 found the tables of a launch of more than ~20 k frames (> 1 MB) going up as a PAGEABLE
        synchronous copy, which waits for everything queued on the stream -- the host and the device took turns.
        """
        torch = _torch()
        tmap = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
                np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64, np.dtype(np.uint8): torch.uint8}
        shapes, offs, sizes, total = [], [], [], 0
        for _name, arr, dt in items:
            shp = np.shape(arr)
            nb = int(np.prod(shp, dtype=np.int64)) * np.dtype(dt).itemsize
            total = (total + 255) // 256 * 256
            offs.append(total)
            shapes.append(shp)
            sizes.append(nb)
            total += nb
        total = max(total, 1)

        def fill(host):   # host: uint8 view of `total` bytes (page-locked arena, or a fresh array for oversized sets)
            for (_name, arr, dt), off, shp, nb in zip(items, offs, shapes, sizes):
                if nb:
                    np.copyto(host[off:off + nb].view(dt).reshape(shp), arr, casting="unsafe")

        if total <= self._ARENA_MAX_ITEM:
            dev = self._arena_upload(None, nbytes=total, fill=fill)
        else:
            host = np.zeros(total, dtype=np.uint8)
            fill(host)
            dev = torch.from_numpy(host).to(self.device, non_blocking=False)
        out = {}
        for (name, _arr, dt), off, shp, nb in zip(items, offs, shapes, sizes):
            if nb == 0:
                out[name] = torch.empty(shp, dtype=tmap[np.dtype(dt)], device=self.device)
            else:
                out[name] = dev[off:off + nb].view(tmap[np.dtype(dt)]).view(shp)
        return out

    _ARENA_BYTES = 64 << 20
    _ARENA_PARTS = 4
    _ARENA_MAX_ITEM = (64 << 20) // 4

    def _arena_upload(self, a, nbytes=None, fill=None):
        """A host array (or `nbytes` written by fill(view)) -> device tensor through a page-locked bump arena, WITHOUT
        blocking: a pageable `tensor.to(device)` waits for everything queued on the stream before it -- after a batch's
        kernels have been launched that is the whole batch, which serialised the host with the device once per table.
        The arena is four parts used in turn; a part is waited for (the event of its last copy) only when the bump pointer
        comes round to it again, three parts of uploads later -- in practice never a wait."""
        import threading

        torch = _torch()
        ar = getattr(self, "_arena", None)
        if ar is None:
            buf = torch.empty(self._ARENA_BYTES, dtype=torch.uint8).pin_memory()
            ar = self._arena = {"t": buf, "np": buf.numpy(), "off": 0, "lock": threading.Lock(),
                                "ev": [None] * self._ARENA_PARTS}
        n = int(a.nbytes) if a is not None else int(nbytes)
        part = self._ARENA_BYTES // self._ARENA_PARTS
        with ar["lock"]:
            off = (ar["off"] + 255) // 256 * 256
            q = off // part
            if q >= self._ARENA_PARTS or off + n > (q + 1) * part:   # does not fit the part the pointer is in: on to the next
                q = (q + 1) % self._ARENA_PARTS if q < self._ARENA_PARTS else 0
                off = q * part
            if q != ar.get("cur"):   # entering a part (by overflow or because the pointer walked into it): its old copies first
                if ar["ev"][q] is not None:
                    ar["ev"][q].synchronize()
                    ar["ev"][q] = None
                ar["cur"] = q
            ar["off"] = off + n
            view = ar["np"][off:off + n]
            if a is not None:
                view[:] = a.reshape(-1).view(np.uint8)
            else:
                fill(view)
            with torch.cuda.device(self.device):
                dev = ar["t"][off:off + n].to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                ar["ev"][q] = ev
        if a is None:
            return dev
        tmap = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64, np.dtype(np.int32): torch.int32,
                np.dtype(np.int64): torch.int64, np.dtype(np.uint8): torch.uint8, np.dtype(np.int16): torch.int16}
        return dev.view(tmap[a.dtype]).view(a.shape)

    def to_device(self, arr, dtype):
        torch = _torch()
        a = np.ascontiguousarray(arr, dtype=dtype)
        if 0 < a.nbytes <= self._ARENA_MAX_ITEM and a.dtype in (np.float32, np.float64, np.int32, np.int64, np.uint8,
                                                                 np.int16) and a.ndim >= 1:
            return self._arena_upload(a)
        return torch.from_numpy(a).to(self.device, non_blocking=False)

    def tables(self, fft_len):
        if fft_len not in self._tables:
            torch = _torch()
            nbytes = self.lib.mpx_tables_bytes(int(fft_len))
            if nbytes == 0:
                raise ValueError("fft_len %r not supported by the HIP path (1024, 2048 or 4096)" % (fft_len,))
            t = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.mpx_tables_init(self.stream_ptr(), int(fft_len), t.data_ptr()), "mpx_tables_init")
            self._tables[fft_len] = t
        return self._tables[fft_len]

    def tables_f64(self, fft_len):
        key = ("f64", fft_len)
        if key not in self._tables:
            torch = _torch()
            nbytes = self.lib.mpx_tables_f64_bytes(int(fft_len))
            if nbytes == 0:
                raise ValueError("fft_len %r not supported by the HIP path (1024, 2048 or 4096)" % (fft_len,))
            t = torch.empty(nbytes // 8, dtype=torch.float64, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.mpx_tables_f64_init(self.stream_ptr(), int(fft_len), t.data_ptr()), "mpx_tables_f64_init")
            self._tables[key] = t
        return self._tables[key]

    # ------------------------------------------------------------------ kernels
    def hann_table(self):
        """hostmath.hann_half_table on the device (float64, built once per engine: ~30 ms of numpy, 16.8 MB)."""
        t = getattr(self, "_hann_table", None)
        if t is None:
            t = self._hann_table = self.to_device(hm.hann_half_table(), np.float64)
        return t

    def analysis_frames(self, fft_len, sig, pos, left, right, out=None, precise=False, rows_in_use=None):
        """sig f32[n], pos i64[F], left/right i32[F] (device) -> (mag, real, imag) f32[F x H] (device).
        precise: window / transform / epilogue in float64 (mpx_analysis_frames_f64): the compressed analysis' choice;
        rows_in_use (precise only): f32[F], 0 = this frame's phase rows are never read, write the magnitudes only."""
        torch = _torch()
        nfr = int(pos.numel())
        H = fft_len // 2 + 1
        if out is None:
            out = tuple(self.empty_feats(nfr, H) for _ in range(3))
        ld = self.feat_ld(*out)
        tab = self.tables_f64(fft_len) if precise else self.tables(fft_len)
        fn = self.lib.mpx_analysis_frames_f64w if precise else self.lib.mpx_analysis_frames
        if precise:   # window weights from numpy's own np.hanning (MAGPHASE_F64_WINDOW=analytic: evaluated on the device)
            wt = self.hann_table() if os.environ.get("MAGPHASE_F64_WINDOW", "table") != "analytic" else None
            extra = ((rows_in_use.data_ptr() if rows_in_use is not None else None),
                     (wt.data_ptr() if wt is not None else None), (hm.HANN_TABLE_CAP if wt is not None else 0))
        else:
            extra = ()
        with torch.cuda.device(self.device):
            _lib.check(fn(self.stream_ptr(), int(fft_len), tab.data_ptr(), sig.data_ptr(), pos.data_ptr(), left.data_ptr(),
                          right.data_ptr(), nfr, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), ld, *extra),
                       "mpx_analysis_frames_f64" if precise else "mpx_analysis_frames")
        return out

    def synthesis_lossless_frames(self, fft_len, mag, real, imag, out=None):
        torch = _torch()
        nfr = int(mag.shape[0])
        if out is None:
            out = self.empty((nfr, fft_len))
        tab = self.tables(fft_len)
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib.mpx_synthesis_lossless_frames(self.stream_ptr(), int(fft_len), tab.data_ptr(), mag.data_ptr(),
                                                       real.data_ptr(), imag.data_ptr(), nfr, out.data_ptr(),
                                                       self.feat_ld(mag, real, imag)),
                "mpx_synthesis_lossless_frames")
        return out

    def ola_gather(self, fft_len, frames, utt_frame_off, pm_rel, out_start, out_off, max_out_len, total_out, out=None):
        torch = _torch()
        if out is None:
            out = self.empty((int(total_out),))
        n_utts = int(out_start.numel())
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib.mpx_ola_gather(self.stream_ptr(), int(fft_len), frames.data_ptr(), n_utts,
                                        utt_frame_off.data_ptr(), pm_rel.data_ptr(), out_start.data_ptr(),
                                        out_off.data_ptr(), int(max_out_len), out.data_ptr()),
                "mpx_ola_gather")
        return out


    def synth_ola_slots(self):
        torch = _torch()
        with torch.cuda.device(self.device):
            n = int(self.lib.mpx_synth_ola_slots())
        cus = os.environ.get("MAGPHASE_SYN_CUS")   # experiment: the synthesis launch on fewer CUs (slots = 6 per workgroup)
        return n if not cus else max(6, min(n, 6 * int(cus)))

    def synth_ola_slot_weights(self, comp=False):
        """Relative speeds of the slots of the lossless (comp=True: the compressed, comp="roundtrip": the one-launch copy
        synthesis) kernel (mpx_synth_ola_slot_weights / mpx_synth_comp_slot_weights / mpx_roundtrip_slot_weights), cached;
        MAGPHASE_OLA_WEIGHTS=0 -> None (equal shares)."""
        if os.environ.get("MAGPHASE_OLA_WEIGHTS", "1") == "0":
            return None
        key = ("rt_w" if comp == "roundtrip" else "comp_w") if comp else "ola_w"
        if key not in self._tables:
            n = self.synth_comp_slots() if comp else self.synth_ola_slots()
            w = np.zeros(n, dtype=np.float32)
            fn = (self.lib.mpx_roundtrip_slot_weights if comp == "roundtrip" else
                  self.lib.mpx_synth_comp_slot_weights) if comp else self.lib.mpx_synth_ola_slot_weights
            _lib.check(fn(w.ctypes.data, n), "mpx_synth_*_slot_weights")
            self._tables[key] = w
        return self._tables[key]

    def post_filter(self, mag_mel_log, fs, **kw):
        """Device MagPhase post-filter (mpx_post_filter) of a float32 [F x D] tensor; kw as magphase.post_filter."""
        torch = _torch()
        F, D = int(mag_mel_log.shape[0]), int(mag_mel_log.shape[1])
        key = ("post_filter", D, int(fs)) + tuple(sorted(kw.items()))
        if key not in self._tables:   # device-resident per configuration (two small uploads per call otherwise)
            nx0, nx1, half, tilt = hm.post_filter_tables(D, fs, **kw)
            self._tables[key] = (nx0, nx1, self.to_device(half, np.int32), self.to_device(tilt, np.float32))
        nx0, nx1, d_half, d_tilt = self._tables[key]
        out = self.empty((F, D))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mpx_post_filter(self.stream_ptr(), mag_mel_log.data_ptr(), F, D, d_half.data_ptr(),
                                                nx0, nx1, d_tilt.data_ptr(), out.data_ptr()), "mpx_post_filter")
        return out

    def post_filter_merlin(self, mag_mel_log, fs, pf_coef=1.4):
        """Device Merlin-style post-filter (mpx_post_filter_merlin, magphase.py:3375-3465) of a float32 [F x D] tensor
        (3 <= D <= 64) -> float32 [F x D].  Tables: hostmath.merlin_tables, resident on the device per configuration."""
        torch = _torch()
        from . import libaudio as la

        F, D = int(mag_mel_log.shape[0]), int(mag_mel_log.shape[1])
        key = ("merlin", D, int(fs), float("%1.2f" % pf_coef))
        if key not in self._tables:
            t = hm.merlin_tables(D, fs, pf_coef)
            self._tables[key] = (t["alpha"], int(t["g"].shape[1])) + tuple(
                self.to_device(t[k], np.float32) for k in ("c1", "lifter", "g", "wk", "cf"))
        alpha, nb, c1, lifter, g, wk, cf = self._tables[key]
        mcep, mcep_w, out = (self.empty((max(F, 1), D)) for _ in range(3))
        r0, p_r0 = self.empty((max(F, 1),)), self.empty((max(F, 1),))
        x = mag_mel_log.contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mpx_post_filter_merlin(self.stream_ptr(), x.data_ptr(), F, D, c1.data_ptr(),
                                                       lifter.data_ptr(), g.data_ptr(), wk.data_ptr(), nb, float(alpha),
                                                       cf.data_ptr(), float(la.MAGIC), mcep.data_ptr(), mcep_w.data_ptr(),
                                                       r0.data_ptr(), p_r0.data_ptr(), out.data_ptr()),
                       "mpx_post_filter_merlin")
        return out[:F]

    def output_hpf(self, pcm, out_off_host, fs):
        """
        magphase.py:981-995 on the device: float32 pcm [total] (utterances concatenated at out_off_host) ->
        float64 tensor, every utterance filtered from a zero state (mpx_output_hpf: cascade of biquads, blocked scan).
        """
        torch = _torch()
        block = int(self.lib.mpx_hpf_block())
        key = ("hpf", int(fs))
        if key not in self._tables:
            sos, pm, g = hm.hpf_tables(fs, block)
            self._tables[key] = (sos, self.to_device(pm, np.float64), self.to_device(g, np.float64))
        sos, d_pm, d_g = self._tables[key]
        out_off_host = np.asarray(out_off_host, dtype=np.int64)
        lens = np.diff(out_off_host)
        nblk = (lens + block - 1) // block
        blk_off = np.concatenate(([0], np.cumsum(nblk))).astype(np.int32)
        d_off = self.to_device(out_off_host, np.int64)
        d_blk = self.to_device(blk_off, np.int32)
        total, tb = int(out_off_host[-1]), int(blk_off[-1])
        zend = torch.empty(2 * max(tb, 1), dtype=torch.float64, device=self.device)
        zstart = torch.empty(2 * max(tb, 1), dtype=torch.float64, device=self.device)
        y_tmp = torch.empty(max(total, 1), dtype=torch.float64, device=self.device)
        y = torch.empty(max(total, 1), dtype=torch.float64, device=self.device)
        import ctypes
        sos_c = np.ascontiguousarray(sos, dtype=np.float64)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mpx_output_hpf(self.stream_ptr(), pcm.data_ptr(), d_off.data_ptr(), d_blk.data_ptr(),
                                               int(lens.size), int(lens.max()) if lens.size else 0,
                                               sos_c.ctypes.data_as(ctypes.c_void_p), d_pm.data_ptr(), d_g.data_ptr(),
                                               zend.data_ptr(), zstart.data_ptr(), y_tmp.data_ptr(), y.data_ptr()),
                       "mpx_output_hpf")
        return y[:total]

    def mel_unwarp_single(self, m_x, n_bins, alpha, exp_out=False):
        """la.sp_mel_unwarp for one [F x n] host matrix through mpx_mel_unwarp (the phase jobs run on a 1-frame dummy)."""
        torch = _torch()
        m_x = np.atleast_2d(np.asarray(m_x, dtype=np.float64))
        F, n = m_x.shape
        u = self.constant(("u_mag", int(n), int(n_bins), float(alpha)), lambda: hm.unwarp_matrix(n, n_bins, alpha))
        ld = int(self.lib.mpx_spec_ld(int(n_bins)))
        a = self.to_device(m_x, np.float32)
        o_exp, o_lin, o_dummy = (self.empty((F, ld)) for _ in range(3))
        with torch.cuda.device(self.device):   # magnitude job: exp(x U); "real" job: x U; "imag" job: scratch
            _lib.check(self.lib.mpx_mel_unwarp(self.stream_ptr(), F, int(n_bins), a.data_ptr(), n, u.data_ptr(),
                                               o_exp.data_ptr(), a.data_ptr(), a.data_ptr(), n, u.data_ptr(),
                                               o_lin.data_ptr(), o_dummy.data_ptr(), ld), "mpx_mel_unwarp")
        return self.to_host_f64((o_exp if exp_out else o_lin)[:, :int(n_bins)])

    def mel_warp_single(self, m_abs, nbins_out, alpha):
        """la.sp_mel_warp's linear map for one host matrix of MAGNITUDES [F x H] through mpx_mel_warp (the magnitude job:
        W ln(x^2 + 1e-8) = the warped ln|f| (the one-sided cepstral sum carries the 1/2); the phase jobs run on a 1-row dummy):
        float64 [F x nbins_out]."""
        torch = _torch()
        m_abs = np.atleast_2d(np.asarray(m_abs, dtype=np.float64))
        F, H = m_abs.shape
        w = self.constant(("w_mag", int(nbins_out), H, float(alpha)), lambda: hm.warp_matrix(nbins_out, H, alpha))
        w1 = self.constant(("w_dummy", H), lambda: np.zeros((1, H)))
        mag = self.feats_to_device(m_abs)
        voi = torch.zeros(F, dtype=torch.float32, device=self.device)
        out, d0, d1 = self.empty((F, int(nbins_out))), self.empty((F, 1)), self.empty((F, 1))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mpx_mel_warp(self.stream_ptr(), F, H, mag.data_ptr(), mag.data_ptr(), mag.data_ptr(),
                                             None, None, None, w.data_ptr(), int(nbins_out), w1.data_ptr(), 1,
                                             voi.data_ptr(), out.data_ptr(), d0.data_ptr(), d1.data_ptr(),
                                             self.feat_ld(mag, mag, mag)), "mpx_mel_warp")
        return self.to_host_f64(out)

    def min_phase_single(self, m_mag):
        """la.build_min_phase_from_mag_spec for one [F x H] host matrix through mpx_min_phase."""
        torch = _torch()
        m_mag = np.atleast_2d(m_mag)
        F, H = m_mag.shape
        N = 2 * (H - 1)
        tab = self.tables(N)
        ld = int(self.lib.mpx_spec_ld(H))
        mag = self.empty((F, ld))
        mag[:, :H].copy_(torch.from_numpy(np.ascontiguousarray(m_mag, dtype=np.float32)))
        ident = torch.arange(F, dtype=torch.int32, device=self.device)
        zeros_t = torch.zeros(F, dtype=torch.float32, device=self.device)
        o_m, o_r, o_i = (self.empty((F, ld)) for _ in range(3))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mpx_min_phase(self.stream_ptr(), N, tab.data_ptr(), mag.data_ptr(), ident.data_ptr(),
                                              ident.data_ptr(), zeros_t.data_ptr(), F, o_m.data_ptr(), o_r.data_ptr(),
                                              o_i.data_ptr(), ld), "mpx_min_phase")
        m = self.to_host_f64(o_m[:, :H])
        return m * (self.to_host_f64(o_r[:, :H]) + 1j * self.to_host_f64(o_i[:, :H]))

    def warp_mag_matrix(self, mag_dim, H, alpha, b_mag_fbank_mel=False):
        """Device-resident [mag_dim x H] matrix of the magnitude compression and the C entry point that goes with it:
        the cepstral mel warp (la.sp_mel_warp, mpx_mel_warp) or the mel filter bank (la.sp_mel_warp_fbank,
        mpx_mel_warp_fbank)."""
        if b_mag_fbank_mel:
            return (self.constant(("w_fbank", int(mag_dim), H, float(alpha)),
                                  lambda: hm.warp_fbank_matrix(mag_dim, H, alpha)), self.lib.mpx_mel_warp_fbank,
                    "mpx_mel_warp_fbank")
        return (self.constant(("w_mag", int(mag_dim), H, float(alpha)), lambda: hm.warp_matrix(mag_dim, H, alpha)),
                self.lib.mpx_mel_warp, "mpx_mel_warp")

    def mel_warp_feats(self, mag, real, imag, voi_host, fs, mag_dim, phase_dim, alpha_phase=None, b_mag_fbank_mel=False):
        """format_for_modelling's two warps (magphase.py:2504-2529) on device feature matrices [F x H] -> three device
        matrices [F x mag_dim], [F x phase_dim], [F x phase_dim]."""
        torch = _torch()
        F, H = int(mag.shape[0]), int(mag.shape[1])
        alpha = hm.define_alpha(fs)
        a_ph = alpha if alpha_phase is None else alpha_phase
        cf, _ = hm.define_crossfade_params(fs)
        k_full = hm.get_num_full_mel_coeffs_from_num_phase_coeffs(cf, phase_dim, a_ph, fs)
        w_mag, warp_fn, warp_name = self.warp_mag_matrix(mag_dim, H, alpha, b_mag_fbank_mel)
        w_ph = self.constant(("w_ph", int(k_full), H, float(a_ph), int(phase_dim)),
                             lambda: hm.warp_matrix(k_full, H, a_ph, nrows=phase_dim))
        voi = self.to_device(np.asarray(voi_host, dtype=np.float64), np.float32)
        out = (self.empty((F, int(mag_dim))), self.empty((F, int(phase_dim))), self.empty((F, int(phase_dim))))
        with torch.cuda.device(self.device):
            _lib.check(warp_fn(self.stream_ptr(), F, H, mag.data_ptr(), real.data_ptr(), imag.data_ptr(),
                               None, None, None, w_mag.data_ptr(), int(mag_dim), w_ph.data_ptr(),
                               int(phase_dim), voi.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                               out[2].data_ptr(), self.feat_ld(mag, real, imag)), warp_name)
        return out

    def synth_comp_slots(self):
        torch = _torch()
        with torch.cuda.device(self.device):
            return int(self.lib.mpx_synth_comp_slots())

    def synthesis_lossless_ola(self, fft_len, mag, real, imag, plan, strips, pcm_out):
        """plan: LosslessSynthesisPlan (run + slot tables resident on this device).  Writes every output sample of
        pcm_out once and the runs' head strips; ola_fixup(plan, strips, pcm_out) completes the run boundaries."""
        torch = _torch()
        tab = self.tables(fft_len)
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib.mpx_synthesis_lossless_ola(self.stream_ptr(), int(fft_len), tab.data_ptr(), mag.data_ptr(),
                                                    real.data_ptr(), imag.data_ptr(), plan.runs.data_ptr(),
                                                    int(plan.n_runs), plan.slot_off.data_ptr(),
                                                    plan.slot_runs.data_ptr(), int(plan.n_slots),
                                                    plan.pm_rel.data_ptr(), strips.data_ptr(), pcm_out.data_ptr(),
                                                    self.feat_ld(mag, real, imag)),
                "mpx_synthesis_lossless_ola")
        return pcm_out

    def roundtrip_lossless_ola(self, fft_len, plan_a, plan_s, feats, strips, pcm_out):
        """Copy synthesis in one launch (mpx_roundtrip_lossless_ola): plan_a's frames are analysed, their feature rows
        written to feats = (mag, real, imag) and overlap-added by plan_s' runs (a LosslessSynthesisPlan built for this
        kernel's slots from plan_a's v_f0); ola_fixup(plan_s, strips, pcm_out) completes the run boundaries."""
        torch = _torch()
        tab = self.tables(fft_len)
        mag, real, imag = feats
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib.mpx_roundtrip_lossless_ola(self.stream_ptr(), int(fft_len), tab.data_ptr(), plan_a.sig.data_ptr(),
                                                    plan_a.pos.data_ptr(), plan_a.left.data_ptr(), plan_a.right.data_ptr(),
                                                    int(plan_a.total_frames), plan_s.runs.data_ptr(), int(plan_s.n_runs),
                                                    plan_s.slot_off.data_ptr(), plan_s.slot_runs.data_ptr(),
                                                    int(plan_s.n_slots), plan_s.pm_rel.data_ptr(), mag.data_ptr(),
                                                    real.data_ptr(), imag.data_ptr(), strips.data_ptr(),
                                                    pcm_out.data_ptr(), self.feat_ld(mag, real, imag)),
                "mpx_roundtrip_lossless_ola")
        return pcm_out

    def ola_fixup(self, fft_len, plan, strips, pcm_out):
        torch = _torch()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mpx_ola_fixup(self.stream_ptr(), int(fft_len), plan.runs.data_ptr(), int(plan.n_runs),
                                              strips.data_ptr(), pcm_out.data_ptr()), "mpx_ola_fixup")
        return pcm_out


class _Prepared:
    """Host side of one launch, built by Engine.prepare_* (possibly on the planner thread): page-locked slot filled, tables
    ready; the plan constructor uploads it.  release(): hands the slot back when the launch is not going to happen."""
    slot = None
    engine = None

    def release(self):
        if self.slot is not None:
            slot, self.slot = self.slot, None
            self.engine._slot_release(slot)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class PreparedAnalysis(_Prepared):
    pass


class PreparedSynthesis(_Prepared):
    pass


class _FlatRows:
    """A list of per-utterance rows kept as ONE array + offsets; iterating / indexing cuts the views."""

    def __init__(self, flat, off):
        self.flat, self.off = flat, np.asarray(off, dtype=np.int64)
        self._o = self.off.tolist()

    def __len__(self):
        return len(self._o) - 1

    def __getitem__(self, u):
        if u < 0:
            u += len(self._o) - 1
        return self.flat[self._o[u]:self._o[u + 1]]

    def __iter__(self):
        return (self[u] for u in range(len(self)))


def _plan_ola_runs(plan, pm_rel_list, starts, out_lens, out_off_host, fft_len, n_slots, frames_per_run, up, weights=None):
    """Shared by the two synthesis plans: runs + slot work lists (hostmath.ola_runs / balance_chunks) -> upload list.
    weights: the slots' relative speeds (Engine.synth_ola_slot_weights) or None for equal shares."""
    fpr = frames_per_run or int(os.environ.get("MAGPHASE_OLA_FRAMES_PER_RUN", 0)) or None
    try:
        if fpr:
            raise hostplan.PlanFallback()      # per-utterance run lengths (tests, tuning): the numpy planner only
        if isinstance(pm_rel_list, _FlatRows):   # already one array + offsets (CompressedSynthesisPlan)
            rel_cat, f_off = np.asarray(pm_rel_list.flat, dtype=np.int64), pm_rel_list.off
            sizes = np.diff(f_off)
        else:
            sizes = [int(np.size(r)) for r in pm_rel_list]
            rel_cat = np.concatenate([np.asarray(r, dtype=np.int64) for r in pm_rel_list]) if pm_rel_list else np.zeros(0, np.int64)
            f_off = np.concatenate(([0], np.cumsum(sizes)))
        runs, slot_off, slot_runs = hostplan.ola_runs(rel_cat, f_off, starts, out_lens,
                                                      np.asarray(out_off_host)[:len(sizes)], fft_len, n_slots,
                                                      weights=weights)
    except hostplan.PlanFallback:
        runs, slot_off, slot_runs = hm.ola_runs(pm_rel_list, starts, out_lens, out_off_host, fft_len, n_slots,
                                                frames_per_run=fpr, weights=None if fpr else weights)
    plan.n_runs = int(runs.size)
    plan.runs_host = runs
    plan.strip_floats = plan.n_runs * (int(fft_len) + 64)
    plan.n_slots = int(slot_off.size - 1)
    up.append(("runs", runs.view(np.uint8), np.uint8))
    up.append(("slot_off", slot_off, np.int32))
    up.append(("slot_runs", slot_runs, np.int32))


_ENGINES = {}


def get_engine(device=None):
    torch = _torch()
    if device is None:
        if not torch.cuda.is_available():
            return Engine()  # raises the loud error
        device = torch.device("cuda", torch.cuda.current_device())
    key = str(device)
    if key not in _ENGINES:
        _ENGINES[key] = Engine(device)
    return _ENGINES[key]


# ======================================================================================================
# Batch plans: host fp64 index math -> descriptor tensors resident in HBM
# ======================================================================================================
class LosslessAnalysisPlan:
    """
    Frame descriptors of a batch of utterances for mpx_analysis_frames.
    utts: list of (v_sig float array in [-1,1) or int16 PCM, fs, v_pm_sec, v_voi).  All must share fft_len.
    Host math follows magphase.py:2877-2879 (pm_sec*fs), libaudio.py:435-447, magphase.py:77-98, :2198-2199.
    """

    def __init__(self, engine, utts, fft_len=None, prepared=None):
        # prepared: a PreparedAnalysis of these utterances (Engine.prepare_analysis, e.g. from the planner thread); None:
        # prepared here when the batch is in the plain shape the native path takes, else the generic path below
        self.engine = engine
        if prepared is None and hasattr(engine, "prepare_analysis") and os.environ.get("MAGPHASE_NATIVE_PREPARE", "1") != "0":
            prepared = engine.prepare_analysis(utts, fft_len, wait=False)
        if prepared is not None:
            self._from_prepared(prepared, utts)
            return
        pos, left, right = [], [], []
        self.v_shift, self.v_f0, self.fs, self.n_frames, self.n_smpls, self.v_pm = [], [], [], [], [], []
        # the samples of all utterances go straight into ONE float32 buffer (page-locked when the engine has one):
        # int16 PCM * 2^-15 and float64 -> float32 are each a single pass, no per-utterance temporaries, no concatenate
        total = int(sum(np.shape(u[0])[0] for u in utts))
        staged = hasattr(engine, "host_staging")
        # a batch of 16-bit wavs (what the batch scripts read) is staged and uploaded as int16 and widened on the
        # device (mpx_pcm16_to_f32): half the PCIe bytes and no host pass over the samples
        all_i16 = staged and len(utts) > 0 and all(np.asarray(u[0]).dtype == np.int16 for u in utts)
        if all_i16:
            buf = engine.host_staging((total + 1) // 2 + 2).view(np.int16)
        else:
            buf = engine.host_staging(total) if staged else np.empty(total, dtype=np.float32)
        off = 0
        if all_i16 and len(utts) > 1:   # 16-bit PCM of the whole batch into the staging buffer on a few native threads
            import ctypes
            arrs = [np.ascontiguousarray(u[0]) for u in utts]
            k = len(arrs)
            src = (ctypes.c_void_p * k)(*[a.ctypes.data for a in arrs])
            nb = np.asarray([a.nbytes for a in arrs], dtype=np.int64)
            doff = np.concatenate(([0], np.cumsum(nb)[:-1])).astype(np.int64)
            n_thr = engine.host_threads(int(nb.sum())) if hasattr(engine, "host_threads") else 8

            def _copy(arrs=arrs, src=src, nb=nb, doff=doff):   # (keeps the arrays alive until the copy is done)
                if engine.lib.mpx_host_copy_many(len(arrs), src, nb.ctypes.data, doff.ctypes.data, buf.ctypes.data, n_thr) != 0:
                    raise _lib.MagphaseHipError("mpx_host_copy_many failed")

            copy_done = engine.background(_copy) if hasattr(engine, "background") else None
            if copy_done is None:
                _copy()
            copied = True
        else:
            copied, copy_done = False, None
        try:
            self._build_generic(engine, utts, fft_len, buf, off, copied, all_i16, staged, total, pos, left, right)
        except BaseException:
            # Whatever goes wrong between the submit and the upload (a malformed utterance, an fft_len mismatch): the native
            # copy must have stopped writing into the page-locked staging buffer before this constructor is left --
            # iobatch retries a failed batch one utterance at a time straight away, and host_staging would hand the same
            # buffer out again while the copy still runs (silent corruption of the retry's samples)
            if copy_done is not None:
                try:
                    copy_done.result()
                except BaseException:
                    pass
            raise
        if copy_done is not None:
            copy_done.result()   # the staged samples are in place (the copy ran beside the index arithmetic)
        self._upload_generic(engine, buf, all_i16, staged, total, pos, left, right)

    def _build_generic(self, engine, utts, fft_len, buf, off, copied, all_i16, staged, total, pos, left, right):
        for (v_sig, fs, v_pm_sec, v_voi) in utts:
            v_sig = np.asarray(v_sig)
            n = v_sig.shape[0]
            if copied:
                pass
            elif all_i16:
                buf[off:off + n] = v_sig
            elif v_sig.dtype == np.int16:
                np.multiply(v_sig, np.float32(1.0 / 32768.0), out=buf[off:off + n])   # exact: == astype(f32) / 32768
            else:
                buf[off:off + n] = v_sig
            N = fft_len if fft_len is not None else hm.define_fft_len(fs)
            if not hasattr(self, "fft_len"):
                self.fft_len = N
            elif N != self.fft_len:
                raise ValueError("all utterances of a plan must share fft_len (bucket by sample rate)")
            self.fs.append(fs)
            self.n_smpls.append(n)
            off += n
        sig_off = np.concatenate(([0], np.cumsum(self.n_smpls)))[:-1] if utts else np.zeros(0)
        try:     # the index arithmetic of the whole batch in one native call (hostplan / csrc/magphase_plan.cpp) ...
            r = hostplan.plan_analysis([u[2] for u in utts], [u[3] for u in utts], self.n_smpls, self.fs, sig_off)
            fo = r["frame_off"]
            for u in range(len(utts)):
                a, b = int(fo[u]), int(fo[u + 1])
                self.v_shift.append(r["left"][a:b]), self.v_pm.append(r["pm"][a:b]), self.v_f0.append(r["f0"][a:b])
                self.n_frames.append(b - a)
            pos[:], left[:], right[:] = [r["pos"]], [r["left"]], [r["right"]]
        except hostplan.PlanFallback:   # ... or utterance by utterance in numpy (same arithmetic; raises what it raises)
            for (v_sig, fs, v_pm_sec, v_voi), n, o in zip(utts, self.n_smpls, sig_off):
                pm_sec, voi = hm.clean_epochs(v_pm_sec, v_voi, check_len_smpls=n, fs=fs)
                pm, lft, rgt = hm.frame_bounds(pm_sec * fs, n)
                pos.append(pm + int(o))
                left.append(lft)
                right.append(rgt)
                self.v_shift.append(lft)
                self.v_pm.append(pm)
                self.v_f0.append(hm.shift_to_f0(lft, voi, fs))
                self.n_frames.append(pm.size)
        self.total_frames = int(sum(self.n_frames))
        self.frame_off = np.concatenate(([0], np.cumsum(self.n_frames))).astype(np.int64)
        right_cat = np.concatenate(right) if right else np.zeros(0, dtype=np.int64)
        # frames longer than fft_len (the reference warns once per such frame): rare -- one pass over the batch, the
        # per-utterance lists only where there is something to list
        self.long_frame_lens = [[] for _ in self.v_shift]
        if right_cat.size:
            left_cat = np.concatenate(left) if len(left) > 1 else np.asarray(left[0])
            tot = left_cat + right_cat + 1
            hit = np.flatnonzero(tot > self.fft_len)
            if hit.size:
                utt_of = np.searchsorted(self.frame_off, hit, side="right") - 1
                for i, u in zip(hit.tolist(), utt_of.tolist()):
                    self.long_frame_lens[u].append(int(tot[i]))
        self.total_smpls = int(off)

    def _upload_generic(self, engine, buf, all_i16, staged, total, pos, left, right):
        e = engine
        if all_i16:
            raw = e.upload_staged((total + 1) // 2 + 2)
            self.sig = e.empty((max(total, 1),))
            with _torch().cuda.device(e.device):
                _lib.check(e.lib.mpx_pcm16_to_f32(e.stream_ptr(), raw.data_ptr(), total, self.sig.data_ptr()),
                           "mpx_pcm16_to_f32")
            self.sig = self.sig[:total]
        else:
            self.sig = e.upload_staged(total) if staged else e.to_device(buf, np.float32)
        desc = e.to_device_packed([("pos", np.concatenate(pos) if pos else np.zeros(0), np.int64),     # one H2D copy
                                   ("left", np.concatenate(left) if left else np.zeros(0), np.int32),
                                   ("right", np.concatenate(right) if right else np.zeros(0), np.int32)])
        self.pos, self.left, self.right = desc["pos"], desc["left"], desc["right"]

    def _from_prepared(self, p, utts):
        """Takes over a PreparedAnalysis: two DMAs (samples, tables) and, for 16-bit input, the widening kernel."""
        e, torch = self.engine, _torch()
        if p.n_utts != len(utts) or p.engine is not e:
            p.release()
            raise ValueError("prepared: not the host side of this batch on this engine")
        self.fft_len, self.fs = p.fft_len, p.fs
        self.n_smpls = [int(u[0].shape[0]) for u in utts]
        fo = self.frame_off = p.frame_off
        self.total_frames, self.total_smpls = p.total_frames, p.total_smpls
        self.n_frames = np.diff(fo).tolist()
        self.v_shift, self.v_pm, self.v_f0 = _FlatRows(p.left64, fo), _FlatRows(p.pm, fo), _FlatRows(p.f0, fo)
        self.f0_med_flat = p.f0_med
        self.long_frame_lens = [[] for _ in range(p.n_utts)]
        if p.long:
            utt_of = np.searchsorted(fo, [i for i, _n in p.long], side="right") - 1
            for (i, n), u in zip(p.long, utt_of.tolist()):
                self.long_frame_lens[u].append(n)
        F, total = p.total_frames, p.total_smpls
        o_pos, o_left, o_right, o_voi = p.offs
        slot, p.slot = p.slot, None          # from here on the upload's event guards the slot
        sd, dd, ev = e._slot_upload(slot, p.stage_bytes, p.desc_bytes)
        self._ready = ev
        if p.all_i16:
            self.sig = e.empty((max(total, 1),))
            with torch.cuda.device(e.device):
                _lib.check(e.lib.mpx_pcm16_to_f32(e.stream_ptr(), sd.data_ptr(), total, self.sig.data_ptr()),
                           "mpx_pcm16_to_f32")
            self.sig = self.sig[:total]
        else:
            self.sig = sd[:4 * total].view(torch.float32)
        self.pos = dd[o_pos:o_pos + 8 * F].view(torch.int64)
        self.left = dd[o_left:o_left + 4 * F].view(torch.int32)
        self.right = dd[o_right:o_right + 4 * F].view(torch.int32)
        self.voi_dev = dd[o_voi:o_voi + 4 * F].view(torch.float32)    # (f0 > 0): CompressedAnalysisPlan's voicing row

    def _wait_ready(self):
        """A plan may be run on another stream than the one it was built on (bench.py alternates streams): that stream
        waits for the plan's uploads too (the build stream already does)."""
        ev = getattr(self, "_ready", None)
        if ev is not None:
            e = self.engine
            _torch().cuda.current_stream(e.device).wait_event(ev)

    def run(self, out=None, precise=False, rows_in_use=None):
        self._wait_ready()
        return self.engine.analysis_frames(self.fft_len, self.sig, self.pos, self.left, self.right, out=out,
                                           precise=precise, rows_in_use=rows_in_use if precise else None)


class LosslessSynthesisPlan:
    """
    PSOLA bookkeeping for a batch: per utterance v_f0 (float64) -> shift -> pm (magphase.py:1771-1772, Q2/Q3)
    -> ola() offsets and trimming (magphase.py:34-62) -> runs of frames for the fused overlap-add (hostmath.ola_runs).
    All float64/int host math; device gets int tables.
    """

    def __init__(self, engine, f0_list, fs_list, fft_len, frames_per_run=None, comp_slots=False):
        # comp_slots: the slot count and weights of the compressed / round-trip pair kernels (mpx_synth_comp_slots)
        self.engine = engine
        self.fft_len = fft_len
        pm_rel, starts, lens, nfr = [], [], [], []
        self.v_pm = []
        try:
            r = hostplan.plan_lossless_synthesis(f0_list, fs_list, fft_len)
            fo = r["frame_off"]
            for u in range(len(f0_list)):
                a, b = int(fo[u]), int(fo[u + 1])
                self.v_pm.append(r["v_pm"][a:b]), pm_rel.append(r["pm_rel"][a:b])
                starts.append(int(r["out_start"][u])), lens.append(int(r["out_len"][u])), nfr.append(b - a)
        except hostplan.PlanFallback:
            for v_f0, fs in zip(f0_list, fs_list):
                v_pm = np.cumsum(hm.f0_to_shift(np.asarray(v_f0, dtype=np.float64), fs)).astype(int)
                rel, start, out_len = hm.ola_plan(v_pm, fft_len)
                self.v_pm.append(v_pm)
                pm_rel.append(rel)
                starts.append(start)
                lens.append(out_len)
                nfr.append(v_pm.size)
        self.out_len = [int(x) for x in lens]
        self.out_off_host = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        self.total_out = int(self.out_off_host[-1])
        self.max_out_len = int(max(lens)) if lens else 0
        self.total_frames = int(sum(nfr))
        e = engine
        _up = []   # (attribute, host array, dtype): uploaded together (Engine.to_device_packed)
        _up.append(("utt_frame_off", np.concatenate(([0], np.cumsum(nfr))), np.int32))
        _up.append(("pm_rel", np.concatenate(pm_rel) if pm_rel else np.zeros(0), np.int32))
        _up.append(("out_start", np.asarray(starts), np.int32))
        _up.append(("out_off", self.out_off_host, np.int64))
        if comp_slots:   # True: the compressed synthesis kernel's slots and shares; "roundtrip": k_roundtrip_pair's
            n_slots = e.synth_comp_slots()
            weights = e.synth_ola_slot_weights(comp=comp_slots)
            if os.environ.get("MAGPHASE_RT_WEIGHTS") and weights is not None:   # experiment: "w0,w1,w2" by age rank of the pair
                w3 = [float(x) for x in os.environ["MAGPHASE_RT_WEIGHTS"].split(",")]
                # (6 wave pairs per 12-wave workgroup; pairs 0-1 / 2-3 / 4-5 hold the oldest / middle / youngest waves)
                weights = np.asarray([w3[((i % 6) * 2) // 4] for i in range(n_slots)], dtype=np.float32)
        else:
            n_slots = e.synth_ola_slots() if hasattr(e, "synth_ola_slots") else 1024
            weights = e.synth_ola_slot_weights() if hasattr(e, "synth_ola_slot_weights") else None
        _plan_ola_runs(self, pm_rel, starts, lens, self.out_off_host, fft_len, n_slots, frames_per_run, _up, weights=weights)
        for _k, _t in e.to_device_packed(_up).items():
            setattr(self, _k, _t)

    def run(self, mag, real, imag, strips=None, out=None):
        """Fused path: k_synth_ola_pair (per-run LDS overlap-add, output written in place) + k_ola_fixup (run boundaries)."""
        e = self.engine
        if strips is None:
            strips = e.empty((max(self.strip_floats, 1),))
        if out is None:
            out = e.empty((self.total_out,))
        e.synthesis_lossless_ola(self.fft_len, mag, real, imag, self, strips, out)
        return e.ola_fixup(self.fft_len, self, strips, out)

    def run_unfused(self, mag, real, imag, frames=None, out=None):
        """Two-kernel form: frames to HBM, then the ascending-order gather (bit-for-bit the reference's sum order)."""
        e = self.engine
        frames = e.synthesis_lossless_frames(self.fft_len, mag, real, imag, out=frames)
        return e.ola_gather(self.fft_len, frames, self.utt_frame_off, self.pm_rel, self.out_start, self.out_off,
                            self.max_out_len, self.total_out, out=out)


class LosslessRoundTripPlan:
    """
    Copy synthesis of a batch (analysis_lossless followed by synthesis_from_lossless on the same frames,
    demos/demo_copy_synthesis_lossless.py:44-50) as ONE launch: the analysis plan's frame tables plus a synthesis plan
    built from the f0 values the analysis derives on the host (magphase.py:2198-2207 -> :1771-1772), cut into runs for
    the round-trip kernel's slots.  run() returns ((mag, real, imag), pcm): the feature rows analysis_lossless returns
    and the waveform synthesis_from_lossless builds from them.
    """

    def __init__(self, engine, utts, fft_len=None, frames_per_run=None):
        self.engine = engine
        if not utts:   # an empty batch: nothing to plan, run() returns empty tensors
            self.analysis = self.synthesis = None
            self.fft_len = fft_len or 4096
            self.total_frames = self.total_out = 0
            self.out_off_host = np.zeros(1, dtype=np.int64)
            return
        self.analysis = LosslessAnalysisPlan(engine, utts, fft_len=fft_len)
        self.fft_len = self.analysis.fft_len
        self.synthesis = LosslessSynthesisPlan(engine, self.analysis.v_f0, self.analysis.fs, self.fft_len,
                                               frames_per_run=frames_per_run, comp_slots="roundtrip")
        if self.synthesis.total_frames != self.analysis.total_frames:
            raise ValueError("round trip: the synthesis plan must cover exactly the analysed frames")
        self.total_frames = self.analysis.total_frames
        self.total_out = self.synthesis.total_out
        self.out_off_host = self.synthesis.out_off_host

    def run(self, feats=None, strips=None, out=None):
        e, a, s = self.engine, self.analysis, self.synthesis
        if feats is None:
            feats = tuple(e.empty_feats(self.total_frames, self.fft_len // 2 + 1) for _ in range(3))
        if out is None:
            out = e.empty((self.total_out,))
        if self.total_frames == 0:
            return feats, out
        if strips is None:
            strips = e.empty((max(s.strip_floats, 1),))
        e.roundtrip_lossless_ola(self.fft_len, a, s, feats, strips, out)
        e.ola_fixup(self.fft_len, s, strips, out)
        return feats, out


# ======================================================================================================
# compressed-feature synthesis (magphase.py:825-997)
# ======================================================================================================
class CompressedSynthesisPlan:
    """
    Host fp64 bookkeeping + device tables for a batch of utterances synthesised from compressed features.
    utts: list of (m_mag_mel_log [F x mag_dim], m_real_mel [F x phase_dim], m_imag_mel, v_lf0 [F]) float arrays.
    Follows magphase.py:836-897 (constants, f0/voicing/shift, constant->variable rate scan, epochs, noise length,
    noise windows) and :969-976 (anti-ringing lengths, ola) -- all index math in float64/int on the host.
    """

    def __init__(self, engine, utts, fs, fft_len=None, b_voi_ap_win=True, b_const_rate=False, alpha_phase=None,
                 noise=None, frames_per_run=None, per_phase_type="magphase", post_filter=False, b_fbank_mel=False,
                 noise_mode="reference", noise_seeds=None, defer_rng=False, noise_spectra=None, prepared=None):
        # prepared: a PreparedSynthesis of these utterances (Engine.prepare_synthesis, e.g. from the planner thread)
        # noise_spectra: None = MAGPHASE_NOISE_SPECTRA ("recompute", the default / "store"); True: every noise frame is
        #            transformed once, its spectrum kept in HBM between the statistics and the synthesis launch (N = 4096)
        # defer_rng: the reference noise stream's advanced state stays on the device (Engine.numpy_global_uniform(defer=True));
        #            the caller owes Engine.mt_sync() before numpy's global generator is used again
        # post_filter: False / True ('magphase': mp.post_filter on the device) / 'merlin' (mp.post_filter_merlin on the device)
        # (the reference's pf_type vocabulary: 'no' means no filtering, magphase.py:3229-3262 -- anything else is an error,
        #  not silently "on")
        if isinstance(post_filter, str):
            if post_filter not in ("no", "magphase", "merlin"):
                raise ValueError("post_filter must be False / None / 'no', True / 'magphase' or 'merlin', not %r" % (post_filter,))
            self.apply_post_filter = {"no": False, "magphase": "magphase", "merlin": "merlin"}[post_filter]
        elif post_filter is None or isinstance(post_filter, (bool, np.bool_, int, np.integer)):
            # truthy non-bool callers (b_post_filter=1, a numpy comparison's np.bool_) mean what bool() says
            if post_filter is not None and not isinstance(post_filter, (bool, np.bool_)) and int(post_filter) not in (0, 1):
                raise ValueError("post_filter must be False / None / 'no', True / 'magphase' or 'merlin', not %r" % (post_filter,))
            self.apply_post_filter = bool(post_filter)
        else:
            raise ValueError("post_filter must be False / None / 'no', True / 'magphase' or 'merlin', not %r" % (post_filter,))
        self.b_const_rate = bool(b_const_rate)
        if noise_mode not in ("reference", "device"):
            raise ValueError("noise_mode must be 'reference' (numpy global RNG, magphase.py:883) or 'device' (Philox on the GPU)")
        self.noise_mode = noise_mode
        if noise_mode == "device" and noise is not None:
            raise ValueError("noise_mode='device' generates the source itself: do not pass noise")

        if per_phase_type not in ("magphase", "min_phase", "linear"):
            raise ValueError("per_phase_type must be 'magphase', 'min_phase' or 'linear'")
        self.per_phase_type = per_phase_type

        self.engine = e = engine
        self.fs = fs
        N = self.fft_len = int(fft_len) if fft_len else hm.define_fft_len(fs)
        alpha = hm.define_alpha(fs)
        self.alpha_phase = alpha if alpha_phase is None else alpha_phase
        # Variable-rate features (rows == frames: identity tables, weight 0) take the same unwarp launch as constant-rate ones
        # (mpx_mel_unwarp_rows): the interpolation is then exact (fmaf(0, 0, m) = m: the same values as mpx_mel_unwarp), and the
        # phase rows are produced only where the synthesis reads them -- voiced frames, bins below the crossfade's end: a
        # quarter of the work of the plain form, which unwarped all 2 049 bins of both phase streams for every frame (round 5:
        # 0.81 -> ... ms per 128-utterance generation launch).  MAGPHASE_UNWARP_ROWS_VAR=0: the plain form.
        self.unwarp_rows = self.b_const_rate or os.environ.get("MAGPHASE_UNWARP_ROWS_VAR", "1") != "0"
        # the native whole-launch planner (Engine.prepare_synthesis; `prepared`: built ahead, e.g. on the planner thread) takes
        # the plain case: ndarray coefficient matrices, the default run planner
        if (prepared is None and frames_per_run is None and hasattr(e, "prepare_synthesis")
                and not os.environ.get("MAGPHASE_OLA_FRAMES_PER_RUN") and os.environ.get("MAGPHASE_NATIVE_PREPARE", "1") != "0"):
            prepared = e.prepare_synthesis(utts, fs, fft_len=fft_len, b_voi_ap_win=b_voi_ap_win, b_const_rate=b_const_rate,
                                           wait=False)
        if prepared is not None and (prepared.n_utts != len(utts) or prepared.engine is not e):
            prepared.release()
            raise ValueError("prepared: not the host side of this batch on this engine")
        if prepared is not None and (frames_per_run is not None or prepared.key != (
                int(fs), N, bool(b_const_rate), bool(b_voi_ap_win), bool(self.unwarp_rows))):
            prepared.release()
            prepared = None
        if prepared is not None:
            mt_device = self._tables_prepared(prepared, noise, noise_mode, noise_seeds)
        else:
            mt_device = self._tables_generic(utts, b_voi_ap_win, noise, noise_mode, noise_seeds, frames_per_run)
        H = N // 2 + 1
        # constants: unwarp matrices and per-bin curves (float64 -> float32)
        # (resident on the device per configuration: rebuilding them costs 7 ms on the host, as much as the rest of a
        # single-utterance call -- tools/archive/latency_probe.py)
        if b_fbank_mel:   # magphase.py:851-852: filter-bank unwarp = a different [mag_dim x H] matrix, same kernel
            self.u_mag = e.constant(("u_mag_fbank", self.mag_dim, H, float(alpha)),
                                    lambda: hm.unwarp_fbank_matrix(self.mag_dim, H, alpha))
        else:
            self.u_mag = e.constant(("u_mag", self.mag_dim, H, float(alpha)),
                                    lambda: hm.unwarp_matrix(self.mag_dim, H, alpha))
        self.u_phase = e.constant(("u_phase", self.phase_dim, N, int(fs), float(self.alpha_phase)),
                                  lambda: hm.phase_unwarp_matrix(self.phase_dim, N, fs, self.alpha_phase))
        self.per_v, self.ap_v, self.ap_u = (
            e.constant(("bin_curve", k, int(fs), N), lambda k=k: hm.synthesis_bin_curves(fs, N)[k]) for k in range(3))
        self._gains_dev = None
        # "noise spectra once" (opt-in): see run()
        self.noise_spectra = ((os.environ.get("MAGPHASE_NOISE_SPECTRA", "recompute") == "store")
                              if noise_spectra is None else bool(noise_spectra))
        if noise_mode == "device":
            torch = _torch()
            self.noise = e.empty((max(int(self.noise_off_host[-1]), 1),))
            with torch.cuda.device(e.device):
                _lib.check(e.lib.mpx_noise_uniform(e.stream_ptr(), self.n_utts, self.noise_seeds_dev.data_ptr(),
                                                   self.noise_off_dev.data_ptr(), int(max(self.ns_len)),
                                                   self.noise.data_ptr()), "mpx_noise_uniform")
        elif mt_device:
            self.noise = e.numpy_global_uniform(int(sum(self.ns_len)), defer=bool(defer_rng))

    def _mt_device(self, noise, noise_mode, mt_total):
        """Reference noise (np.random.uniform from numpy's GLOBAL generator, magphase.py:883) for more than a few utterances
        is continued on the device from numpy's own MT19937 state (mpx_noise_numpy_mt19937: the same samples, the state put
        back advanced) -- the host draw is 4 ns per sample, 0.13 s per 128 utterances."""
        return (noise_mode == "reference" and noise is None and mt_total >= (1 << 18)
                and os.environ.get("MAGPHASE_MT_DEVICE", "1") != "0" and np.random.get_state()[0] == "MT19937")

    def _host_noise(self, noise, ui, ns_len):
        e = self.engine
        if noise is not None:
            v_ns = np.asarray(noise[ui], dtype=np.float64)
            if v_ns.size != ns_len:
                raise ValueError("noise length %d != ns_len %d" % (v_ns.size, ns_len))
            return v_ns
        if hasattr(e, "mt_sync"):
            e.mt_sync()                                            # a deferred device state goes back first
        return np.random.uniform(-1, 1, ns_len)                    # :883 (global numpy RNG, as the reference)

    def _tables_prepared(self, p, noise, noise_mode, noise_seeds):
        """Takes over a PreparedSynthesis (Engine.prepare_synthesis): two DMAs (coefficient rows, every table)."""
        e, torch = self.engine, _torch()
        self.mag_dim, self.phase_dim = p.mag_dim, p.phase_dim
        F, U = p.total_frames, p.n_utts
        fo = p.frame_off
        self._tabs = {"v_shift": p.v_shift, "v_pm": p.v_pm, "voiced": p.voiced_host}
        self._fo = fo
        self.ns_len = p.ns_len.tolist()
        self.n_rows, self.total_frames, self.frame_off, self.n_utts = p.n_rows, F, fo, U
        self.out_len = p.out_len.tolist()
        self.out_off_host = np.concatenate(([0], np.cumsum(p.out_len))).astype(np.int64)
        self.total_out = int(self.out_off_host[-1])
        self.max_out_len = int(p.out_len.max())
        self.voiced_host = p.voiced_host.astype(bool)
        self.n_runs, self.n_slots = p.n_runs, p.n_slots
        self.runs_host = p.runs_host
        self.strip_floats = self.n_runs * (self.fft_len + 64)
        self.n_per = e.host_constant(("n_per", int(self.fs), self.fft_len), lambda: _first_all_zero_from(
            np.asarray(hm.synthesis_bin_curves(self.fs, self.fft_len)[0], dtype=np.float32)))
        mt_device = self._mt_device(noise, noise_mode, int(p.ns_len.sum()))
        slot, p.slot = p.slot, None
        sd, dd, ev = e._slot_upload(slot, p.stage_bytes, p.desc_bytes)
        self._ready = ev
        n_m, n_p = self.n_rows * self.mag_dim, self.n_rows * self.phase_dim
        coef = sd.view(torch.float32)
        self.a_mag = coef[:n_m].view(self.n_rows, self.mag_dim)
        self.a_real = coef[n_m:n_m + n_p].view(self.n_rows, self.phase_dim)
        self.a_imag = coef[n_m + n_p:n_m + 2 * n_p].view(self.n_rows, self.phase_dim)
        sizes = {"utt_frame_off": U + 1, "tile_first": p.n_tiles1, "out_start": U, "out_off": U + 1,
                 "runs": 56 * p.n_runs, "slot_off": p.n_slots + 1, "slot_runs": p.n_runs}
        tmap = {np.int32: torch.int32, np.int64: torch.int64, np.float32: torch.float32, np.uint8: torch.uint8}
        for (name, dt), off in zip(hostplan.SYNTH_TABLES, p.desc_off.tolist()):
            n = sizes.get(name, F)
            if name == "tile_first" and not self.unwarp_rows:
                continue
            setattr(self, name, dd[off:off + n * np.dtype(dt).itemsize].view(tmap[dt]))
        if noise_mode == "device":
            self._noise_seed_tables(noise_seeds)
        elif not mt_device:
            self.noise = e.to_device(np.concatenate([self._host_noise(noise, ui, n) for ui, n in enumerate(self.ns_len)]),
                                     np.float32)
        return mt_device

    def _noise_seed_tables(self, noise_seeds):
        e = self.engine
        seeds = np.arange(self.n_utts, dtype=np.uint64) if noise_seeds is None else np.asarray(noise_seeds).astype(np.uint64)
        if seeds.size != self.n_utts:
            raise ValueError("noise_seeds: one per utterance")
        self.noise_seeds = seeds
        self.noise_off_host = np.concatenate(([0], np.cumsum(self.ns_len))).astype(np.int64)
        d = e.to_device_packed([("s", seeds.view(np.int64), np.int64), ("o", self.noise_off_host, np.int64)])
        self.noise_seeds_dev, self.noise_off_dev = d["s"], d["o"]

    def _tables_generic(self, utts, b_voi_ap_win, noise, noise_mode, noise_seeds, frames_per_run):
        """The generic path: any array-like input, utterance by utterance in Python where the native planner declines."""
        e = self.engine
        fs, N, b_const_rate, per_phase_type = self.fs, self.fft_len, self.b_const_rate, self.per_phase_type
        H = N // 2 + 1
        alpha = hm.define_alpha(fs)
        _up = []   # (attribute, host array, dtype): uploaded together at the end (Engine.to_device_packed)
        self.mag_dim = int(np.shape(utts[0][0])[1])
        self.phase_dim = int(np.shape(utts[0][1])[1])

        a_mag, a_real, a_imag = [], [], []
        npos, nleft, nright, wtype, voiced, row0, row1, rowt, win_l, win_r = ([] for _ in range(10))
        noises, lf0s = [], []
        row_base = 0
        nd = np.ndarray
        for ui, (mml, rm, im, lf0) in enumerate(utts):
            # the coefficient matrices go to the device as float32 whatever they arrive as: no float64 round trip here
            # (a plan is built per launch of a corpus job: the usual case -- 2-D ndarrays -- skips the generic conversions)
            if not (type(mml) is nd and type(rm) is nd and type(im) is nd and mml.ndim == 2 and rm.ndim == 2 and im.ndim == 2):
                mml, rm, im = np.atleast_2d(np.asarray(mml)), np.atleast_2d(np.asarray(rm)), np.atleast_2d(np.asarray(im))
            lf0 = np.atleast_1d(np.asarray(lf0, dtype=np.float64))
            n_rows = mml.shape[0]
            if rm.shape[0] != n_rows or im.shape[0] != n_rows or lf0.shape[0] != n_rows:
                raise ValueError("utterance %d: mag / real / imag / lf0 have %d / %d / %d / %d frames"
                                 % (ui, n_rows, rm.shape[0], im.shape[0], lf0.shape[0]))
            if rm.shape[1] != im.shape[1]:
                raise ValueError("utterance %d: real and imag have different dimensions" % ui)
            if mml.shape[1] != self.mag_dim or rm.shape[1] != self.phase_dim:
                # (stage_rows checks totals only: rows of another width whose totals happen to match would be copied flat,
                # silently scrambled -- np.concatenate(axis=0, out=[rows x dim]) used to raise here)
                raise ValueError("utterance %d: mag / phase dimensions %d / %d differ from the batch's %d / %d"
                                 % (ui, mml.shape[1], rm.shape[1], self.mag_dim, self.phase_dim))
            a_mag.append(mml), a_real.append(rm), a_imag.append(im), lf0s.append(lf0)
            row_base += n_rows

        try:    # index arithmetic of the whole batch in one native call (hostplan / csrc/magphase_plan.cpp) ...
            r = hostplan.plan_synthesis([np.exp(l) for l in lf0s], fs, N, b_const_rate, b_voi_ap_win)   # :846
        except hostplan.PlanFallback:   # ... or utterance by utterance in numpy: the same arithmetic, spelled out
            r = plan_synthesis_numpy(lf0s, fs, N, b_const_rate, b_voi_ap_win)
        fo = r["frame_off"]
        mt_total = int(np.sum(r["ns_len"]))
        mt_device = self._mt_device(noise, noise_mode, mt_total)
        # per-utterance views (v_shift / v_pm / v_voi: properties below) are cut from the batch tables on demand
        self._tabs, self._fo = r, fo
        self.ns_len = [int(x) for x in np.asarray(r["ns_len"]).tolist()]
        starts = [int(x) for x in np.asarray(r["out_start"]).tolist()]
        lens = [int(x) for x in np.asarray(r["out_len"]).tolist()]
        nfr = np.diff(np.asarray(fo, dtype=np.int64)).tolist()
        pm_rel = _FlatRows(np.asarray(r["pm_rel"]), fo)
        if noise is not None or not (noise_mode == "device" or mt_device):
            noises = [self._host_noise(noise, ui, self.ns_len[ui]) for ui in range(len(utts))]
        npos, nleft, nright, wtype, voiced = [r["npos"]], [r["nleft"]], [r["nright"]], [r["wtype"]], [r["voiced"]]
        row0, row1, rowt, win_l, win_r = [r["row0"]], [r["row1"]], [r["rowt"]], [r["win_l"]], [r["win_r"]]

        cat = np.concatenate
        self.n_rows = row_base
        self.total_frames = int(sum(nfr))
        self.frame_off = cat(([0], np.cumsum(nfr))).astype(np.int64)
        self.out_len = [int(x) for x in lens]
        self.out_off_host = cat(([0], np.cumsum(lens))).astype(np.int64)
        self.total_out = int(self.out_off_host[-1])
        self.max_out_len = int(max(lens))
        self.voiced_host = cat(voiced).astype(bool)
        _up.append(("utt_frame_off", self.frame_off, np.int32))
        self.n_utts = len(nfr)
        # coefficient matrices: concatenated straight into the page-locked staging buffer, one DMA
        n_m, n_p = self.n_rows * self.mag_dim, self.n_rows * self.phase_dim
        stage = e.host_staging(n_m + 2 * n_p)
        # (inline: on the helper thread -- Engine.background -- the launch loop of a generation job got 5 % SLOWER, the three
        # calls' Python glue fights the constructor for the GIL; the analysis plan's single native copy gains 7 % there)
        e.stage_rows(a_mag, stage[:n_m].reshape(self.n_rows, self.mag_dim))
        e.stage_rows(a_real, stage[n_m:n_m + n_p].reshape(self.n_rows, self.phase_dim))
        e.stage_rows(a_imag, stage[n_m + n_p:].reshape(self.n_rows, self.phase_dim))
        if noise_mode == "device":
            seeds = np.arange(len(nfr), dtype=np.uint64) if noise_seeds is None else np.asarray(noise_seeds).astype(np.uint64)
            if seeds.size != len(nfr):
                raise ValueError("noise_seeds: one per utterance")
            self.noise_seeds = seeds
            self.noise_off_host = cat(([0], np.cumsum(self.ns_len))).astype(np.int64)
            _up.append(("noise_seeds_dev", seeds.view(np.int64), np.int64))
            _up.append(("noise_off_dev", self.noise_off_host, np.int64))
        elif not mt_device:
            _up.append(("noise", cat(noises), np.float32))
        _up.append(("npos", cat(npos), np.int64))
        _up.append(("nleft", cat(nleft), np.int32))
        _up.append(("nright", cat(nright), np.int32))
        _up.append(("wtype", cat(wtype), np.int32))
        _up.append(("voiced", cat(voiced), np.int32))
        if self.unwarp_rows:   # frames of every 31-row tile of the coefficient matrix (mpx_mel_unwarp_rows)
            r0c = cat(row0)
            self._check_rows_for_tiles(r0c, cat(row1))
            _up.append(("tile_first", np.searchsorted(r0c, 31 * np.arange((self.n_rows + 30) // 31 + 1), side="left"),
                        np.int32))
        _up.append(("row0", cat(row0), np.int32))
        _up.append(("row1", cat(row1), np.int32))
        _up.append(("rowt", cat(rowt), np.float32))
        _up.append(("win_l", cat(win_l), np.int32))
        _up.append(("win_r", cat(win_r), np.int32))
        _up.append(("pm_rel", pm_rel.flat, np.int32))
        _up.append(("out_start", np.asarray(starts), np.int32))
        _up.append(("out_off", self.out_off_host, np.int64))
        # bins from n_per on have no periodic component (the crossfade mask is exactly zero there): their phase rows are
        # neither unwarped nor read
        self.n_per = e.host_constant(("n_per", int(fs), N), lambda: _first_all_zero_from(
            np.asarray(hm.synthesis_bin_curves(fs, N)[0], dtype=np.float32)))
        # OLA runs
        n_slots = e.synth_comp_slots() if hasattr(e, "synth_comp_slots") else 1024
        _plan_ola_runs(self, pm_rel, starts, self.out_len, self.out_off_host, N, n_slots, frames_per_run, _up,
                       weights=e.synth_ola_slot_weights(comp=True) if hasattr(e, "synth_ola_slot_weights") else None)
        coef = e.upload_staged(n_m + 2 * n_p)
        self.a_mag = coef[:n_m].view(self.n_rows, self.mag_dim)
        self.a_real = coef[n_m:n_m + n_p].view(self.n_rows, self.phase_dim)
        self.a_imag = coef[n_m + n_p:].view(self.n_rows, self.phase_dim)
        for _k, _t in e.to_device_packed(_up).items():
            setattr(self, _k, _t)
        return mt_device

    def _per_utt(self, key, cast=None):
        r, fo = self._tabs, self._fo
        out = [r[key][int(fo[u]):int(fo[u + 1])] for u in range(len(self.ns_len))]
        return [cast(x) for x in out] if cast else out

    @property
    def v_shift(self):
        """Per utterance: the frames' shifts in samples (magphase.py:862-868 / :2210-2215)."""
        return self._per_utt("v_shift")

    @property
    def v_pm(self):
        """Per utterance: the frames' epochs in samples (la.shift_to_pm, magphase.py:880)."""
        return self._per_utt("v_pm")

    @property
    def v_voi(self):
        """Per utterance: the frames' voicing decisions (magphase.py:847, :866)."""
        return self._per_utt("voiced", lambda x: x.astype(bool))

    @staticmethod
    def _check_rows_for_tiles(r0, r1):
        """What the tiled unwarp relies on: row0 ascending over the batch, row1 - row0 in {0, 1}."""
        if r0.size and (np.any(np.diff(r0) < 0) or np.any((r1 - r0) < 0) or np.any((r1 - r0) > 1)):
            raise ValueError("constant -> variable rate tables out of order")

    @property
    def gains(self):
        """[(g_voiced, g_unvoiced)] per utterance (float64), fetched from the device on demand."""
        if self._gains_dev is None:
            return None
        g = self._gains_dev.cpu().numpy()
        return [(float(a), float(b)) for a, b in g]

    def noise_gains(self, sums_host):
        """magphase.py:902-906 (Q10) from the per-frame sums of (ln|Ns|)^2: two gains per utterance, float64."""
        H = self.fft_len // 2 + 1
        inv = np.ones(self.total_frames)
        gains = []
        for u in range(len(self.out_len)):
            a, b = int(self.frame_off[u]), int(self.frame_off[u + 1])
            s = np.asarray(sums_host[a:b], dtype=np.float64)
            v = self.voiced_host[a:b]
            g = []
            for cls in (v, ~v):
                ncls = int(np.sum(cls))
                g.append(np.sqrt(np.exp(np.sum(s[cls]) / (ncls * (H - 2)))) if ncls else np.nan)
                if ncls:
                    inv[a:b][cls] = 1.0 / g[-1]
            gains.append(tuple(g))
        return inv

    def _buffers(self):
        """Work buffers of run(), allocated once per plan (the caching allocator makes a re-allocation per call cheap
        but not free: ~1.6 GB of spectra + strips + per-frame scalars)."""
        b = getattr(self, "_buf", None)
        if b is None:
            e, torch = self.engine, _torch()
            H = self.fft_len // 2 + 1
            ld = int(e.lib.mpx_spec_ld(H))
            b = self._buf = dict(
                ld=ld,
                # unwarped spectra at the VARIABLE rate: one row per synthesis frame (mpx_mel_unwarp_rows interpolates)
                spec=tuple(e.empty((self.total_frames, ld))[:, :H] for _ in range(3)),
                sums=e.empty((self.total_frames,)),
                inv_gain=e.empty((self.total_frames,)),
                gains=torch.empty((self.n_utts, 2), dtype=torch.float64, device=e.device),
                strips=e.empty((max(self.strip_floats, 1),)),
            )
            if self.per_phase_type != "magphase":
                F = self.total_frames
                b["ident"] = torch.arange(F, dtype=torch.int32, device=e.device)
                b["zeros_t"] = torch.zeros(F, dtype=torch.float32, device=e.device)
                b["spec_v"] = tuple(e.empty((F, ld))[:, :H] for _ in range(3))
        return b

    def run(self, out=None, keep=False, mark=None):
        """mark: optional callable(name), called after every kernel launch has been enqueued (bench.py: HIP events)."""
        e, lib, N = self.engine, self.engine.lib, self.fft_len
        torch = _torch()
        H = N // 2 + 1
        tab = e.tables(N)
        mark = mark or (lambda name: None)
        ev = getattr(self, "_ready", None)
        if ev is not None:   # (a plan run on another stream than the one it was built on: that stream waits for the uploads too)
            torch.cuda.current_stream(e.device).wait_event(ev)
        buf = self._buffers()
        # unwarped spectra: internal matrices, rows 128-byte aligned (mpx_spec_ld: full-line stores of the MFMA unwarp)
        ld = buf["ld"]
        mag, real, imag = buf["spec"]
        sums, strips, inv_gain = buf["sums"], buf["strips"], buf["inv_gain"]
        pcm = out if out is not None else e.empty((self.total_out,))
        with torch.cuda.device(e.device):
            st = e.stream_ptr()
            mark("start")
            a_mag = self.a_mag
            if self.apply_post_filter == "merlin":   # magphase.py:3262-3264
                a_mag = e.post_filter_merlin(self.a_mag, self.fs)
                mark("k_post_filter_merlin")
            elif self.apply_post_filter:   # magphase.py:3259-3261
                a_mag = e.post_filter(self.a_mag, self.fs)
                mark("k_post_filter")
            if self.unwarp_rows:   # constant -> variable rate inside the unwarp: one spectrum row per synthesis frame
                _lib.check(lib.mpx_mel_unwarp_rows(
                    st, self.total_frames, H, a_mag.data_ptr(), self.mag_dim, self.u_mag.data_ptr(), mag.data_ptr(),
                    self.a_real.data_ptr(), self.a_imag.data_ptr(), self.phase_dim, self.u_phase.data_ptr(),
                    real.data_ptr(), imag.data_ptr(), ld, self.row0.data_ptr(), self.row1.data_ptr(),
                    self.rowt.data_ptr(), self.n_rows, self.tile_first.data_ptr(),
                    self.voiced.data_ptr() if self.per_phase_type == "magphase" else None,
                    self.n_per if self.per_phase_type == "magphase" else 0), "mpx_mel_unwarp_rows")
            else:                   # variable-rate features: rows == frames
                _lib.check(lib.mpx_mel_unwarp(st, self.n_rows, H, a_mag.data_ptr(), self.mag_dim, self.u_mag.data_ptr(),
                                              mag.data_ptr(), self.a_real.data_ptr(), self.a_imag.data_ptr(),
                                              self.phase_dim, self.u_phase.data_ptr(), real.data_ptr(), imag.data_ptr(),
                                              ld), "mpx_mel_unwarp")
            mark("k_mel_unwarp_mfma")
            # (the noise chain is independent of the unwarp, but a second HIP stream does not help: measured 3.13 vs
            # 3.18 ms per step with 12-wave and 3.15 vs 3.16 with 8-wave noise workgroups -- the two grids do not co-run)
            # "noise spectra once" (opt-in, MAGPHASE_NOISE_SPECTRA=store; N = 4096): the statistics launch
            # stores every frame's noise spectrum and the synthesis launch loads it instead of a second transform --
            # 17.4 KB per frame each way for the arithmetic of one forward FFT (measured: docs/LAB_NOTES.md, round 5)
            nspec = None
            if N == 4096 and self.total_frames > 0 and self.noise_spectra:
                nspec = buf.get("nspec")
                if nspec is None:
                    nspec = buf["nspec"] = e.empty((int(lib.mpx_noise_spectra_floats(N, self.total_frames)),))
            if nspec is not None:
                _lib.check(lib.mpx_noise_stats_spectra(st, N, tab.data_ptr(), self.noise.data_ptr(), self.npos.data_ptr(),
                                                       self.nleft.data_ptr(), self.nright.data_ptr(),
                                                       self.wtype.data_ptr(), self.total_frames, sums.data_ptr(),
                                                       nspec.data_ptr()), "mpx_noise_stats_spectra")
            else:
                _lib.check(lib.mpx_noise_stats(st, N, tab.data_ptr(), self.noise.data_ptr(), self.npos.data_ptr(),
                                               self.nleft.data_ptr(), self.nright.data_ptr(), self.wtype.data_ptr(),
                                               self.total_frames, sums.data_ptr()), "mpx_noise_stats")
            mark("k_noise_stats")
            # two gains per utterance (Q10): float64 reduction on the device, no host round trip
            self._gains_dev = buf["gains"]
            _lib.check(lib.mpx_noise_gains(st, sums.data_ptr(), self.voiced.data_ptr(), self.utt_frame_off.data_ptr(),
                                           self.n_utts, H - 2, inv_gain.data_ptr(), self._gains_dev.data_ptr()),
                       "mpx_noise_gains")
            mark("k_noise_gains")
            if self.per_phase_type != "magphase":
                # periodic component's phase is not the transmitted one (magphase.py:933-938):
                #   'min_phase': complex-cepstrum minimum phase of the magnitude, per frame
                #   'linear'   : zero phase
                F = self.total_frames
                ident, zeros_t = buf["ident"], buf["zeros_t"]
                if self.per_phase_type == "min_phase":
                    mag_v, real_v, imag_v = buf["spec_v"]
                    _lib.check(lib.mpx_min_phase(st, N, tab.data_ptr(), mag.data_ptr(), ident.data_ptr(),
                                                 ident.data_ptr(), zeros_t.data_ptr(), F, mag_v.data_ptr(),
                                                 real_v.data_ptr(), imag_v.data_ptr(), ld), "mpx_min_phase")
                    mark("k_min_phase")
                    mag, real, imag = mag_v, real_v, imag_v
                else:
                    real.fill_(1.0)
                    imag.fill_(0.0)
            ola_args = (
                st, N, tab.data_ptr(), mag.data_ptr(), real.data_ptr(), imag.data_ptr(), self.noise.data_ptr(),
                self.npos.data_ptr(), self.nleft.data_ptr(), self.nright.data_ptr(), self.wtype.data_ptr(),
                self.voiced.data_ptr(), inv_gain.data_ptr(), None, None, None,
                self.win_l.data_ptr(), self.win_r.data_ptr(), self.pm_rel.data_ptr(),
                self.per_v.data_ptr(), self.ap_v.data_ptr(), self.ap_u.data_ptr(), self.runs.data_ptr(),
                self.n_runs, self.slot_off.data_ptr(), self.slot_runs.data_ptr(), self.n_slots,
                strips.data_ptr(), pcm.data_ptr(), ld, self.n_per if self.per_phase_type == "magphase" else 0)
            if nspec is not None:
                _lib.check(lib.mpx_synthesis_compressed_ola_spectra(*ola_args, nspec.data_ptr()),
                           "mpx_synthesis_compressed_ola_spectra")
            else:
                _lib.check(lib.mpx_synthesis_compressed_ola(*ola_args), "mpx_synthesis_compressed_ola")
            mark("k_synth_comp_pair")
        e.ola_fixup(N, self, strips, pcm)
        mark("k_ola_fixup")
        if keep:
            self.debug = dict(mag=mag, real=real, imag=imag, sums=sums)
        return pcm


def _first_all_zero_from(v):
    """Smallest n with v[k] == 0 for every k >= n."""
    nz = np.flatnonzero(np.asarray(v) != 0)
    return int(nz[-1]) + 1 if nz.size else 0


def plan_synthesis_numpy(lf0s, fs, N, b_const_rate, b_voi_ap_win):
    """
    The per-utterance index arithmetic of synthesis_from_compressed in numpy, reference line by reference line; returns
    the batch's tables in hostplan.plan_synthesis' layout.  The native planner (csrc/magphase_plan.cpp) is this, for the
    whole batch in one call; this form raises what the reference's arithmetic raises and is what the tests compare the
    native one with.
    """
    from scipy import interpolate

    keys = ("v_shift", "v_pm", "npos", "nleft", "nright", "wtype", "voiced", "row0", "row1", "rowt", "win_l", "win_r",
            "pm_rel")
    acc = {k: [] for k in keys}
    ns_lens, starts, lens, nfr = [], [], [], []
    row_base, noise_base = 0, 0
    for lf0 in lf0s:
        lf0 = np.atleast_1d(np.asarray(lf0, dtype=np.float64))
        n_rows = lf0.shape[0]
        v_f0 = np.exp(lf0)                                         # magphase.py:846
        v_voi = v_f0 > 1.0                                         # :847
        v_shift = hm.f0_to_shift(v_f0, fs)                         # :848
        if b_const_rate:                                           # :861-870
            v_shift, v_locs = _const_to_variable_scan(v_shift, 5.0, fs)
            step = fs * 5.0 / 1000
            centres = step * np.arange(1, n_rows + 1)
            v_voi = interpolate.interp1d(centres, v_voi, axis=0, kind="linear")(v_locs) > 0.5
            idx = np.clip(np.searchsorted(centres, v_locs), 1, n_rows - 1)   # scipy's _call_linear bracketing
            lo, hi = idx - 1, idx
            t = (v_locs - centres[lo]) / (centres[hi] - centres[lo])
        else:
            lo = hi = np.arange(n_rows)
            t = np.zeros(n_rows)
        v_shift = v_shift.astype(int)                              # :879
        v_pm = np.cumsum(v_shift)                                  # :880
        n = v_pm.size
        ns_len = int(v_pm[-1] + (v_pm[-1] - v_pm[-2]))             # :882
        _, lft, rgt = hm.frame_bounds(v_pm, ns_len)                # windowing(v_ns, v_pm): magphase.py:77-98
        if np.any(lft > N // 2) or np.any(rgt + 1 > N // 2):
            raise ValueError("negative dimensions are not allowed")   # np.zeros(<0) in la.frm_list_to_matrix
        se = np.r_[v_shift[0], v_shift, v_shift[-1], v_shift[-1]]   # :969
        wl, wr = se[:n] + se[1:n + 1], se[2:n + 2] + se[3:n + 3]
        if np.any(wl > N // 2) or np.any(wr + 1 > N // 2):
            raise ValueError("could not broadcast input array (anti-ringing window longer than the frame)")
        rel, start, out_len = hm.ola_plan(v_pm, N)
        for k, v in (("v_shift", v_shift), ("v_pm", v_pm), ("npos", v_pm + noise_base), ("nleft", lft), ("nright", rgt),
                     ("wtype", (v_voi & bool(b_voi_ap_win)).astype(np.int32)), ("voiced", v_voi.astype(np.int32)),
                     ("row0", lo + row_base), ("row1", hi + row_base), ("rowt", t), ("win_l", wl), ("win_r", wr),
                     ("pm_rel", rel)):
            acc[k].append(v)
        ns_lens.append(ns_len), starts.append(start), lens.append(out_len), nfr.append(n)
        row_base += n_rows
        noise_base += ns_len
    i32 = ("nleft", "nright", "wtype", "voiced", "row0", "row1", "win_l", "win_r")
    out = {k: (np.concatenate(v).astype(np.int32 if k in i32 else (np.float64 if k == "rowt" else np.int64))
               if v else np.zeros(0)) for k, v in acc.items()}
    out.update(frame_off=np.concatenate(([0], np.cumsum(nfr))).astype(np.int64), ns_len=np.asarray(ns_lens, dtype=np.int64),
               out_start=np.asarray(starts, dtype=np.int64), out_len=np.asarray(lens, dtype=np.int64))
    return out


def _const_to_variable_scan(v_shift_c_rate, frm_rate_ms, fs):
    """
    magphase.py:1426-1449 (Q16): serial backward scan pos_{k-1} = pos_k - lerp(shift)(pos_k) from the last
    constant-rate centre until the position leaves the grid.  Runs in the library's host function
    mpx_host_const_to_var_scan (scipy interp1d's float64 operation sequence without the per-step Python / scipy call:
    bit-identical results, golden G7; _const_to_variable_scan_scipy is the literal form the tests compare it with).
    """
    v = np.ascontiguousarray(v_shift_c_rate, dtype=np.float64)
    n = int(v.shape[0])
    step = fs * frm_rate_ms / 1000
    centres = np.ascontiguousarray(step * np.arange(1, n + 1), dtype=np.float64)
    shifts, locs = np.empty(2 * n), np.empty(2 * n)
    start = int(_lib.load().mpx_host_const_to_var_scan(centres.ctypes.data, v.ctypes.data, n, shifts.ctypes.data,
                                                       locs.ctypes.data))
    if start < 0:
        return _const_to_variable_scan_scipy(v_shift_c_rate, frm_rate_ms, fs)
    return shifts[start:], locs[start:]


def _const_to_variable_scan_scipy(v_shift_c_rate, frm_rate_ms, fs):
    """The same scan written like the reference: one scipy interp1d call per step (8 us each)."""
    from scipy import interpolate

    n = np.size(v_shift_c_rate, 0)
    step = fs * frm_rate_ms / 1000
    centres = step * np.arange(1, n + 1)
    f = interpolate.interp1d(centres, v_shift_c_rate, axis=0, kind="linear")
    shifts, locs = np.zeros(n * 2), np.zeros(n * 2)
    pos = centres[-1]
    for i in range(2 * n - 1, 0, -1):
        locs[i] = pos
        try:
            shifts[i] = f(pos)
        except ValueError:
            locs, shifts = locs[i + 1:], shifts[i + 1:]
            break
        pos = pos - shifts[i]
    return shifts, locs



# ======================================================================================================
# compressed-feature analysis (magphase.py:2947-2988, 2490-2544)
# ======================================================================================================
class CompressedAnalysisPlan:
    """
    Lossless analysis plan + host tables for the mel warp of a batch (one sample rate).  run() = k_analysis ->
    k_mel_warp, everything resident on the device; host fp64 does f0 / lf0 / constant-rate tables only.
    """

    def __init__(self, engine, utts, fft_len=None, mag_dim=60, phase_dim=10, b_const_rate=False, alpha_phase=None,
                 b_mag_fbank_mel=False, prepared=None):
        # prepared: see LosslessAnalysisPlan (Engine.prepare_analysis, e.g. from the planner thread)
        self.engine = e = engine
        self.lossless = plan = LosslessAnalysisPlan(engine, utts, fft_len=fft_len, prepared=prepared)
        fs = self.fs = plan.fs[0]
        if plan.fs.count(fs) != len(plan.fs):
            raise ValueError("one sample rate per batch")
        N = self.fft_len = plan.fft_len
        H = N // 2 + 1
        self.mag_dim, self.phase_dim, self.b_const_rate = int(mag_dim), int(phase_dim), bool(b_const_rate)
        alpha = hm.define_alpha(fs)
        a_ph = alpha if alpha_phase is None else alpha_phase
        cf, _ = hm.define_crossfade_params(fs)
        k_full = hm.get_num_full_mel_coeffs_from_num_phase_coeffs(cf, phase_dim, a_ph, fs)
        self.w_mag, self._warp_fn, self._warp_name = e.warp_mag_matrix(mag_dim, H, alpha, b_mag_fbank_mel)
        self.w_ph = e.constant(("w_ph", int(k_full), H, float(a_ph), int(phase_dim)),
                               lambda: hm.warp_matrix(k_full, H, a_ph, nrows=phase_dim))
        row0, row1, rowt, self.f0_out = [], [], [], []
        if b_const_rate:
            for u in range(len(utts)):
                v_f0 = plan.v_f0[u]
                base = int(plan.frame_off[u])
                v_pm = np.cumsum(plan.v_shift[u])
                lo, hi, t = hm.var_to_const_rate_table(v_pm, 5.0, fs)
                v_f0 = _const_rate_f0_voi(v_f0, v_pm, fs)
                row0.append(lo + base), row1.append(hi + base), rowt.append(t), self.f0_out.append(v_f0)
            self.out_off = np.concatenate(([0], np.cumsum([len(f) for f in self.f0_out]))).astype(np.int64)
        else:   # variable rate: output rows == frames (no row tables go to the device)
            self.f0_out = plan.v_f0 if isinstance(plan.v_f0, _FlatRows) else list(plan.v_f0)
            self.out_off = np.asarray(plan.frame_off, dtype=np.int64)
        self.total_out_frames = int(self.out_off[-1])
        if isinstance(self.f0_out, _FlatRows):
            f0_cat = self.f0_out.flat
        else:
            f0_cat = np.concatenate(self.f0_out) if self.f0_out else np.zeros(0)
        voi_dev = None if b_const_rate else getattr(plan, "voi_dev", None)   # already in the prepared tables' upload
        items = [] if voi_dev is not None else [("voi", (f0_cat > 0).astype(np.float32), np.float32)]
        if b_const_rate:
            items += [("row0", np.concatenate(row0), np.int32), ("row1", np.concatenate(row1), np.int32),
                      ("rowt", np.concatenate(rowt), np.float32)]
        # phase streams warped on the variable-rate rows, their 45 outputs interpolated afterwards (mpx_mel_warp_rows):
        # the rows a voiced constant-rate frame interpolates from
        self.phase_on_rows = bool(b_const_rate) and os.environ.get("MAGPHASE_WARP_PHASE_ROWS", "1") != "0"
        if self.phase_on_rows:
            voiced = f0_cat > 0
            r0, r1 = np.concatenate(row0), np.concatenate(row1)
            need = np.zeros(plan.total_frames, dtype=np.float32)
            need[r0[voiced]] = 1.0
            need[r1[voiced]] = 1.0
            items.append(("rows_in_use", need, np.float32))
        desc = e.to_device_packed(items) if items else {}   # one H2D copy
        self.voi = desc["voi"] if voi_dev is None else voi_dev
        self.row0, self.row1, self.rowt = (desc.get(k) for k in ("row0", "row1", "rowt"))
        self.rows_in_use = desc.get("rows_in_use")
        self._phase_tmp = None
        # Variable frame rate: ONE fused kernel, the lossless features never reach HBM (mpx_analysis_compressed_fused;
        # MAGPHASE_COMP_FUSED=0 keeps the staged pair k_analysis_f64 -> k_mel_warp_mfma).  The constant-rate path
        # interpolates staged lossless rows, as the reference does (SURVEY.md 8d allows that staging).
        fusable = (N in (2048, 4096) and self.mag_dim <= 64 and self.phase_dim <= 48
                   and os.environ.get("MAGPHASE_COMP_FUSED", "1") != "0"
                   and os.environ.get("MAGPHASE_COMP_ANALYSIS", "f64") != "f32")
        self.fused = fusable and not b_const_rate
        # Constant rate: the staged pair k_analysis_f64 -> k_mel_warp_mfma, which interpolates staged lossless rows as the
        # reference does (SURVEY.md 8d allows that staging), or -- MAGPHASE_COMP_FUSED_CR=1 -- the same ONE kernel with the
        # row interpolation inside (mpx_analysis_compressed_fused_cr: the magnitudes' operand rows are built per
        # constant-rate frame, the phase streams are warped at the variable rate and finished by mpx_warp_phase_rows).
        # Measured on configs[2] (round 6, bench.py configs2.analysis_one_kernel): HBM traffic of the analysis side 2.87 ->
        # 0.37 GB, no 1.4 GB of staged rows -- and 1.38 ms instead of 0.98: opt-in.  (The filter-bank magnitudes take the
        # logarithm AFTER the product: staged only.)
        self.fused_cr = (fusable and b_const_rate and self.phase_on_rows and self._warp_name != "mpx_mel_warp_fbank"
                         and self.total_out_frames > 0 and int(e.lib.mpx_analysis_compressed_fused_waves()) == 8
                         and int(e.lib.mpx_analysis_compressed_fused_layout()) == 1
                         and os.environ.get("MAGPHASE_COMP_FUSED_CR", "0") == "1")
        self._cr_work = None
        if self.fused or self.fused_cr:
            nw = int(e.lib.mpx_analysis_compressed_fused_waves())
            layout = int(e.lib.mpx_analysis_compressed_fused_layout())   # fragment order this build of the kernel reads
            key = ("wpack", self._warp_name, int(mag_dim), int(k_full), int(phase_dim), H, float(alpha), float(a_ph), nw,
                   layout)
            if key not in e._tables:
                wm = (hm.warp_fbank_matrix(mag_dim, H, alpha) if self._warp_name == "mpx_mel_warp_fbank"
                      else hm.warp_matrix(mag_dim, H, alpha))
                wph = hm.warp_matrix(k_full, H, a_ph, nrows=phase_dim)
                wpack, whalf = hm.pack_warp_fused(wm, wph, N, n_waves=nw, layout=layout)
                e._tables[key] = (e.to_device(wpack, np.float32), e.to_device(whalf, np.float32))
            self.wpack, self.whalf = e._tables[key]

    def run(self, feats=None, out=None, mark=None):
        e, torch = self.engine, _torch()
        H = self.fft_len // 2 + 1
        mark = mark or (lambda name: None)
        mark("start")
        # float64 transform: the warp's log / division amplify an fp32 FFT's noise on weak bins (magphase_f64.hip)
        precise = os.environ.get("MAGPHASE_COMP_ANALYSIS", "f64") != "f32"
        self.lossless._wait_ready()
        if self.fused:   # (feats, the staged path's lossless feature buffers, are not used)
            if out is None:
                out = (e.empty((self.total_out_frames, self.mag_dim)), e.empty((self.total_out_frames, self.phase_dim)),
                       e.empty((self.total_out_frames, self.phase_dim)))
            pl = self.lossless
            wt = e.hann_table() if os.environ.get("MAGPHASE_F64_WINDOW", "table") != "analytic" else None
            with torch.cuda.device(e.device):
                _lib.check(e.lib.mpx_analysis_compressed_fused(
                    e.stream_ptr(), int(self.fft_len), e.tables_f64(self.fft_len).data_ptr(), pl.sig.data_ptr(),
                    pl.pos.data_ptr(), pl.left.data_ptr(), pl.right.data_ptr(), int(pl.total_frames),
                    (wt.data_ptr() if wt is not None else None), (hm.HANN_TABLE_CAP if wt is not None else 0),
                    self.wpack.data_ptr(), self.whalf.data_ptr(), self.mag_dim, self.phase_dim, self.voi.data_ptr(),
                    1 if self._warp_name == "mpx_mel_warp_fbank" else 0, out[0].data_ptr(), out[1].data_ptr(),
                    out[2].data_ptr()), "mpx_analysis_compressed_fused")
            mark("k_analysis_warp_fused")
            return out
        if self.fused_cr:
            if out is None:
                out = (e.empty((self.total_out_frames, self.mag_dim)), e.empty((self.total_out_frames, self.phase_dim)),
                       e.empty((self.total_out_frames, self.phase_dim)))
            pl = self.lossless
            n_var = int(pl.total_frames)
            if self._phase_tmp is None:
                self._phase_tmp = (e.empty((n_var, self.phase_dim)), e.empty((n_var, self.phase_dim)))
                nbytes = int(e.lib.mpx_analysis_compressed_fused_cr_work_bytes(int(self.fft_len), n_var))
                self._cr_work = e.empty(((nbytes + 3) // 4,))
            wt = e.hann_table() if os.environ.get("MAGPHASE_F64_WINDOW", "table") != "analytic" else None
            with torch.cuda.device(e.device):
                _lib.check(e.lib.mpx_analysis_compressed_fused_cr(
                    e.stream_ptr(), int(self.fft_len), e.tables_f64(self.fft_len).data_ptr(), pl.sig.data_ptr(),
                    pl.pos.data_ptr(), pl.left.data_ptr(), pl.right.data_ptr(), n_var,
                    (wt.data_ptr() if wt is not None else None), (hm.HANN_TABLE_CAP if wt is not None else 0),
                    self.wpack.data_ptr(), self.whalf.data_ptr(), self.mag_dim, self.phase_dim,
                    self.rows_in_use.data_ptr(), self.row0.data_ptr(), self.row1.data_ptr(), self.rowt.data_ptr(),
                    int(self.total_out_frames), out[0].data_ptr(), self._phase_tmp[0].data_ptr(),
                    self._phase_tmp[1].data_ptr(), self._cr_work.data_ptr()), "mpx_analysis_compressed_fused_cr")
                mark("k_analysis_warp_fused_cr")
                _lib.check(e.lib.mpx_warp_phase_rows(
                    e.stream_ptr(), int(self.total_out_frames), self.phase_dim, self._phase_tmp[0].data_ptr(),
                    self._phase_tmp[1].data_ptr(), self.row0.data_ptr(), self.row1.data_ptr(), self.rowt.data_ptr(),
                    self.voi.data_ptr(), out[1].data_ptr(), out[2].data_ptr()), "mpx_warp_phase_rows")
            mark("k_warp_phase_rows")
            return out
        # (the phase rows nobody reads -- rows_in_use == 0 -- are not written either)
        mag, real, imag = self.lossless.run(out=feats, precise=precise,
                                            rows_in_use=self.rows_in_use if self.phase_on_rows else None)
        mark("k_analysis_f64" if precise else "k_analysis")
        if out is None:
            out = (e.empty((self.total_out_frames, self.mag_dim)), e.empty((self.total_out_frames, self.phase_dim)),
                   e.empty((self.total_out_frames, self.phase_dim)))
        ptr = (lambda t: t.data_ptr() if t is not None else None)
        with torch.cuda.device(e.device):
            if self.phase_on_rows:
                n_var = self.lossless.total_frames
                if self._phase_tmp is None:
                    self._phase_tmp = (e.empty((n_var, self.phase_dim)), e.empty((n_var, self.phase_dim)))
                _lib.check(e.lib.mpx_mel_warp_rows(
                    e.stream_ptr(), self.total_out_frames, H, mag.data_ptr(), real.data_ptr(), imag.data_ptr(),
                    ptr(self.row0), ptr(self.row1), ptr(self.rowt), self.w_mag.data_ptr(), self.mag_dim,
                    self.w_ph.data_ptr(), self.phase_dim, self.voi.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                    out[2].data_ptr(), e.feat_ld(mag, real, imag), 1 if self._warp_name == "mpx_mel_warp_fbank" else 0,
                    n_var, self.rows_in_use.data_ptr(), self._phase_tmp[0].data_ptr(), self._phase_tmp[1].data_ptr()),
                    "mpx_mel_warp_rows")
            else:
                _lib.check(self._warp_fn(e.stream_ptr(), self.total_out_frames, H, mag.data_ptr(), real.data_ptr(),
                                         imag.data_ptr(), ptr(self.row0), ptr(self.row1), ptr(self.rowt),
                                         self.w_mag.data_ptr(), self.mag_dim, self.w_ph.data_ptr(), self.phase_dim,
                                         self.voi.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                                         out[2].data_ptr(), e.feat_ld(mag, real, imag)), self._warp_name)
        mark("k_mel_warp_mfma")
        return out


def _const_rate_f0_voi(v_f0, v_pm_smpls, fs, const_rate_ms=5.0):
    """magphase.py:2975-2980: f0 interpolated through the voiced points only, voicing by interpolation > 0.5."""
    from scipy import interpolate

    step = fs * const_rate_ms / 1000

    def interp1(y, x):
        centres = np.arange(step, x[-1], step)
        if x[0] > 0:
            f = interpolate.interp1d(np.r_[0, x], np.r_[y[0], y], axis=0, kind='linear')
        else:
            f = interpolate.interp1d(x, y, axis=0, kind='linear')
        return f(centres)

    v_voi = v_f0 > 1.0
    v_f0_c = interp1(np.r_[v_f0[v_voi][0], v_f0[v_voi], v_f0[v_voi][-1]], np.r_[0, v_pm_smpls[v_voi], v_pm_smpls[-1]])
    v_voi_c = interp1(v_voi.astype(np.float64), v_pm_smpls) > 0.5
    return v_f0_c * v_voi_c
