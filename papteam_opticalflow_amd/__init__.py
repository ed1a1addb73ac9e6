"""MI355X-native coarse-to-fine variational optical flow: the one hot path of
ElijahHyndman/PAPTeam_OpticalFlow (`pyflow.coarse2fine_flow`) as hand-written HIP kernels for gfx950 behind
a C ABI (include/papof.h).

    from papteam_opticalflow_amd import coarse2fine_flow          # same signature / return as the reference
    timing, vx, vy, warpI2 = coarse2fine_flow(im1, im2, pyramidLevels)

`papteam_opticalflow_amd/dropin/` holds the Cython module named `pyflow` that the reference's
OpticalFlowCalculation.py / TestSuite.py import unchanged (put that directory on PYTHONPATH).
There is no CPU fallback: without libpapof.so or without a gfx950 device every call raises.
"""
from . import capi  # noqa: F401
from .capi import SOR_EXACT, SOR_JACOBI, SOR_REDBLACK, Papof, PapofError, default_params  # noqa: F401

__version__ = "0.1.0"

_default = None


def _handle():
    global _default
    if _default is None:
        import os
        _default = Papof(int(os.environ.get("PAPOF_DEVICE", "0")))
    return _default


def coarse2fine_flow(Im1, Im2, pyramidLevels, nCores=1, **solver):
    """ctypes twin of dropin/pyflow.pyx (reference: Code/Serial/pyflow.pyx:31-70): returns
    (dict[str, str] of ten timers, vx[h,w], vy[h,w], warpI2[h,w,c]); `nCores` is accepted and ignored."""
    import numpy as np
    for name, a in (("Im1", Im1), ("Im2", Im2)):
        if a is None:
            raise TypeError("Argument '%s' must not be None" % name)
        if not isinstance(a, np.ndarray) or a.dtype != np.float64:
            raise ValueError("Buffer dtype mismatch for %s, expected 'double'" % name)
        if a.ndim != 3:
            raise ValueError("Buffer has wrong number of dimensions (expected 3, got %d)" % a.ndim)
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("ndarray is not C-contiguous")
    if int(pyramidLevels) < 1:
        raise ValueError("pyramidLevels must be >= 1")
    params = default_params(**solver) if solver else None
    vx, vy, warp, t = _handle().coarse2fine_flow(Im1, Im2, int(pyramidLevels), params)
    return capi.format_timing(t), vx, vy, warp


def coarse2fine_flow_u8(Im1, Im2, pyramidLevels, nCores=1, **solver):
    """Same as coarse2fine_flow for the uint8 arrays `np.array(Image.open(path))` yields: the caller's
    `astype(float) / 255.` (Code/Serial/OpticalFlowCalculation.py:69-70) is done on the device, bit-identically, so
    1/8 of the bytes cross PCIe on the way in."""
    if int(pyramidLevels) < 1:
        raise ValueError("pyramidLevels must be >= 1")
    params = default_params(**solver) if solver else None
    vx, vy, warp, t = _handle().coarse2fine_flow_u8(Im1, Im2, int(pyramidLevels), params)
    return capi.format_timing(t), vx, vy, warp


class FlowSequence:
    """Flow along a video: `push(frame)` returns None for the first frame, then the reference's 4-tuple for
    (previous frame -> this frame).  One upload and one pyramid per frame instead of two (the reference's TestSuite
    walks a 102-frame collection as 101 overlapping pairs, Code/Serial/TestSuite.py:69-81); results are bit-identical
    to calling coarse2fine_flow on each pair.  Frames may be float64 in [0,1] or uint8."""

    def __init__(self, pyramidLevels, device=None, handle=None, **solver):
        import os
        self.levels = int(pyramidLevels)
        if self.levels < 1:
            raise ValueError("pyramidLevels must be >= 1")
        self.params = default_params(**solver) if solver else None
        self.gpu = handle if handle is not None else Papof(
            int(os.environ.get("PAPOF_DEVICE", "0")) if device is None else device)
        self.gpu.seq_reset()

    def push(self, frame, out=None):
        r = self.gpu.seq_push(frame, self.levels, self.params, out)
        if r is None:
            return None
        vx, vy, warp, t = r
        return capi.format_timing(t), vx, vy, warp

    def reset(self):
        self.gpu.seq_reset()

    def close(self):
        self.gpu.close()


# ---- the reference's 16-bit flow file (OpticalFlow::SaveOpticalFlow / LoadOpticalFlow, src/OpticalFlow.cpp:963-1015) ----
_FLOW16_HEADER = 29  # Image<unsigned short>::saveImage (src/Image.h:825-837): char type[16], int w, int h, int c, bool


def save_flow16(path, vx, vy):
    """Write (vx, vy) as the reference's SaveOpticalFlow does: the quantisation runs on the GPU
    (papof_flow_quantize16), the 29-byte header is Image<unsigned short>::saveImage's (type name `t` =
    typeid(unsigned short).name(); the reference leaves the rest of the 16 bytes uninitialised, written as zeros here)."""
    import struct
    q = _handle().flow_quantize16(vx, vy)
    h, w, _ = q.shape
    with open(path, "wb") as f:
        f.write(b"t".ljust(16, b"\0") + struct.pack("<iii?", w, h, 2, False))
        f.write(q.tobytes())


def load_flow16(path):
    """Read a file written by the reference's SaveOpticalFlow (or save_flow16); returns (vx, vy) float64."""
    import struct
    import numpy as np
    raw = open(path, "rb").read()
    if len(raw) < _FLOW16_HEADER or raw[:2] != b"t\0":
        raise ValueError("not an Image<unsigned short> file")
    w, h, c, _deriv = struct.unpack("<iii?", raw[16:_FLOW16_HEADER])
    if c != 2 or len(raw) != _FLOW16_HEADER + 2 * w * h * c:
        raise ValueError("not a 2-channel flow file")
    q = np.frombuffer(raw, dtype=np.uint16, offset=_FLOW16_HEADER).reshape(h, w, 2)
    return _handle().flow_dequantize16(q)


_collection_handles = {}  # device -> handles kept between flow_collection calls (arena, staging and pinned buffers
                          # of a handle cost tens of milliseconds to allocate)


def collection_in_flight(height, width):
    """Sequences the UNBATCHED flow_collection() keeps in flight by default: small frames leave the chip idle, so many calls run
    side by side (measured on MI355X, reference schedule, tools/collection_probe.py: 240x135 and 480x270 peak at 16, 960x540 is
    flat from 4 to 16, 1920x1080 peaks at 4).  What caps that at ~2 ms per 240x135 pair is NOT the host's launch path -- a call
    spends 1.0 of its 24 ms enqueueing, with 16 threads as with one -- but the device's dispatch of ~84 k small dependent kernels
    per second over 16 queues, each stretched to 50-65 us (tools/collection_trace.py, tools/trace_concurrency.py;
    profiles/r04_collection_trace_240.txt, r04_collection_concurrency_240x16.txt): hence the batched path, collection_batch()."""
    mpix = height * width / 1e6
    return 16 if mpix <= 0.3 else (8 if mpix <= 1.0 else 4)


def collection_batch(height, width):
    """Pairs per launch chain flow_collection() uses by default (0: unbatched): frames too small to fill the chip share every
    launch (csrc/batch.hip) -- 240x135 on the reference schedule: 8.3 ms per single call, 2.2 ms per pair as 16 independent calls in
    flight, 0.72 / 0.58 ms per pair in batches of 16 / 32 (0.48 with two chains in flight) (tools/batch_probe.py, profiles/r04_batch_probe_*.txt)."""
    mpix = height * width / 1e6
    # measured per pair, reference schedule, host uint8 in / float64 out (profiles/r04_batch_probe_*.txt, r04_collection_probe_*):
    # 240x135: 8.3 ms alone, 2.2 as 16 calls in flight, 0.72 / 0.58 in batches of 16 / 32; 480x270: 11.7 / 3.25 / 2.0 in batches of
    # 16; 960x540: 16.5 / 8.9 / 9.2 in batches of 8 -- from there on a pair fills enough of the chip by itself
    return 32 if mpix <= 0.05 else (16 if mpix <= 0.14 else 0)


def flow_collection(frames, pyramidLevels, in_flight=None, device=None, on_pair=None, batch=None, **solver):
    """Flow of every consecutive pair of a frame list -- what the reference's TestSuite does with a collection
    (Code/Serial/TestSuite.py:69-81: frame n -> n+1 for 101 pairs) -- with `in_flight` sequences running concurrently
    on one GPU (default: by frame size, collection_in_flight()): the list is cut into contiguous segments (overlapping by one frame), each pushed through its own
    FlowSequence (own handle, arena and streams) by its own host thread.  One pair alone leaves most of the chip idle
    while the coarse pyramid levels are solved; several in flight fill it (bench.py `concurrent_pairs`).

    on_pair(i, timing, vx, vy, warpI2), if given, is called for every pair (from the worker threads, in no particular
    order) with arrays that are REUSED for the worker's next pair -- copy what you keep; nothing is returned.  This is
    the fast form: 83 MB of fresh result memory per 1080p pair costs more in page faults than the GPU takes to fill
    it.  Without on_pair the list of (timing, vx, vy, warpI2) in pair order is returned (fresh arrays).
    Every result is bit-identical to coarse2fine_flow(frames[i], frames[i + 1], pyramidLevels).

    batch: pairs per launch chain (default by frame size, collection_batch(); 0 / 1: every pair its own call).  Small frames
    cannot fill the chip and are bound by the device's dispatch of their ~210 little dependent kernels per pair; in a batch
    every launch serves all its pairs (include/papof.h: papof_flow_batch*).  With batch > 1 the ten timers of a pair are its
    equal SHARE of its chain's (Total = wall time of the chain / its pairs, Phase5_SOR = its solver kernels / its pairs; the
    other phases are not separated in a batch), so that the reference's per-pair timing file still adds up."""
    import threading
    import numpy as np
    capi.load()  # once, in this thread
    n_pairs = len(frames) - 1
    if n_pairs < 1:
        return None if on_pair else []
    h0, w0 = np.shape(frames[0])[:2]
    if batch is None:
        batch = collection_batch(h0, w0)
    if batch and batch > 1 and n_pairs > 1:
        return _flow_collection_batched(frames, pyramidLevels, int(batch), in_flight, device, on_pair, solver)
    if in_flight is None:
        in_flight = collection_in_flight(h0, w0)
    k = max(1, min(int(in_flight), n_pairs))
    bounds = [n_pairs * s // k for s in range(k + 1)]  # segment s computes pairs bounds[s] .. bounds[s+1]-1
    results, errors = [None] * n_pairs, []
    import os
    dev = int(os.environ.get("PAPOF_DEVICE", "0")) if device is None else int(device)
    pool = _collection_handles.setdefault(dev, [])
    while len(pool) < k:
        pool.append(Papof(dev))
    for hnd in pool[:k]:  # several handles in flight fill each other's idle time: one stream per handle is faster then
        hnd.set_stream_overlap(k == 1 and os.environ.get("PAPOF_OVERLAP", "1") != "0")

    def run(s):
        try:
            seq = FlowSequence(pyramidLevels, handle=pool[s], **solver)
            try:
                out = None
                for i in range(bounds[s], bounds[s + 1] + 1):
                    if on_pair is not None and out is None and i > bounds[s]:
                        h, w, c = np.shape(frames[i])
                        out = (capi.result_array((h, w)), capi.result_array((h, w)), capi.result_array((h, w, c)))
                    r = seq.push(frames[i], out)
                    if i > bounds[s]:
                        if on_pair is not None:
                            on_pair(i - 1, *r)
                        else:
                            results[i - 1] = r
            finally:
                seq.reset()  # the handle stays in the pool
        except Exception as e:  # noqa: BLE001 -- re-raised in the caller's thread
            errors.append(e)
    threads = [threading.Thread(target=run, args=(s,)) for s in range(k)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return None if on_pair else results


def _flow_collection_batched(frames, levels, batch, in_flight, device, on_pair, solver):
    """flow_collection() through papof_flow_batch*: the pairs of a collection in chains of `batch` consecutive pairs per launch
    chain (a video: every frame's pyramid is built once per chain), `in_flight` chains at a time (default 3: one handle's uploads,
    downloads and host work beside the others' kernels).  Bit-identical results, pair for pair."""
    import os
    import threading
    n_pairs = len(frames) - 1
    dev = int(os.environ.get("PAPOF_DEVICE", "0")) if device is None else int(device)
    # chains in flight: 240x135 in batches of 32: 0.58 / 0.43 / 0.40 / 0.39 / 0.49 ms per pair with 1 / 2 / 3 / 4 / 6
    # (tools/collection_chains_probe.py, profiles/r04_collection_chains_probe.txt): one chain's uploads, downloads, host work and
    # one-workgroup solves of its smallest levels beside the others' kernels
    k = max(1, min(int(in_flight) if in_flight else 3, (n_pairs + batch - 1) // batch))
    pool = _collection_handles.setdefault(dev, [])
    while len(pool) < k:
        pool.append(Papof(dev))
    params = default_params(**solver) if solver else None
    chains = [(i0, min(i0 + batch, n_pairs)) for i0 in range(0, n_pairs, batch)]
    results, errors = [None] * n_pairs, []
    lock = threading.Lock()
    nxt = [0]

    def run(s):
        try:
            out = None
            while True:
                with lock:
                    if nxt[0] >= len(chains):
                        return
                    i0, i1 = chains[nxt[0]]
                    nxt[0] += 1
                reuse = out if (on_pair is not None and out is not None and len(out) == i1 - i0) else None
                out, t = pool[s].flow_batch(frames[i0:i1 + 1], levels, params, sequence=True, out=reuse)
                timing = capi.format_timing(t / (i1 - i0))  # a pair's share of its chain: sums over a collection stay meaningful
                for j in range(i1 - i0):
                    if on_pair is not None:
                        on_pair(i0 + j, timing, *out[j])
                    else:
                        results[i0 + j] = (timing,) + tuple(out[j])
                if on_pair is None:
                    out = None  # the caller keeps these arrays
        except Exception as e:  # noqa: BLE001 -- re-raised in the caller's thread
            errors.append(e)
    threads = [threading.Thread(target=run, args=(s,)) for s in range(k)]
    for t_ in threads:
        t_.start()
    for t_ in threads:
        t_.join()
    if errors:
        raise errors[0]
    return None if on_pair else results
