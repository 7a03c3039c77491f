"""hipGraph replay of the per-image extraction function.

The reference runs its network eagerly, one image at a time (``extract_vectors``,
cirtorch/networks/imageretrievalnet.py:277-304).  On an MI355X that loop is bound by the HOST: a
ResNet101 trunk at three scales is ~700-1100 kernel launches per image, each ~13 us of Python /
dispatcher time, against 9-11 ms of GPU work.  The launch sequence of a given input shape never
changes at inference, so it is captured once (after the eager warm-up runs that let MIOpen pick its
algorithms) and replayed with one call.  One graph per input shape, least-recently-used eviction;
the first ``warmup`` (default 1) occurrences of a shape run eagerly
(MIOpen's algorithm search happens there), the next one is captured.
"""
import collections
import ctypes
import os
import warnings

import torch


def graphs_enabled(device):
    return torch.device(device).type == "cuda" and os.environ.get("MDIR_AMD_GRAPHS", "1") != "0"


_side_streams = {}


def parallel_map(fn, items):
    """``[fn(x) for x in items]`` with every item after the first on its own HIP stream, joined
    before returning.  The scales of the image pyramid are independent, and the trunk kernels of the
    smaller ones (a quarter / half of the pixels) leave most CUs idle; as parallel branches of the
    captured graph they fill each other's gaps.  Eager launches are host-bound and gain nothing, so
    this matters inside ``ShapeGraphs`` captures (where the forked streams become graph branches)."""
    if len(items) < 2 or not isinstance(items[0], torch.Tensor) or not items[0].is_cuda \
            or os.environ.get("MDIR_AMD_SCALE_STREAMS", "1") == "0":
        return [fn(x) for x in items]
    dev = items[0].device
    cur = torch.cuda.current_stream(dev)
    side = _side_streams.setdefault(dev.index, [])
    while len(side) < len(items) - 1:
        side.append(torch.cuda.Stream(device=dev))
    outs = [None] * len(items)
    for i in range(1, len(items)):
        side[i - 1].wait_stream(cur)                  # inputs were produced on the current stream
        with torch.cuda.stream(side[i - 1]):
            outs[i] = fn(items[i])
    outs[0] = fn(items[0])
    for i in range(1, len(items)):
        cur.wait_stream(side[i - 1])
        if isinstance(outs[i], torch.Tensor):
            outs[i].record_stream(cur)                # allocated on the side stream, consumed on this one
    return outs


# PyTorch cannot start another capture in a process after one that failed: CUDAGraph.capture_end() leaves through its first
# check, before the caching allocator is told that the pool's capture is over, and the next capture_begin aborts the process
# (measured, torch 2.10 / ROCm 7.0: tests/test_gpu_round5.py).  So a refusal switches captures off for the rest of the process;
# graphs captured before it keep replaying, everything else runs eagerly -- slower, never different.
_captures_off = False
_refusals = 0


def capture_stats():
    """``{"captures_off": bool, "refusals": int}`` of this process: after ONE refused capture every later shape stays eager (PyTorch
    aborts on the next capture_begin after a failed one) -- a throughput cliff that must show in results, not only in a warning
    (ADVICE round 5); bench.py puts it into `descriptors_per_s`."""
    return {"captures_off": bool(_captures_off), "refusals": int(_refusals)}


class _Entry:
    __slots__ = ("graph", "static_in", "static_out")


class ShapeGraphs:
    """``fn``: tensor -> tensor (or list/tuple of tensors) with no host synchronisation inside."""

    # Capturing a graph of a ResNet101 pyramid costs ~0.2 s and a replay saves ~6 ms per image against eager batched
    # launches (tools/bench_extract.py --list), so a capture only pays for itself after ~35 images of that shape.
    PAYOFF_IMAGES = int(os.environ.get("MDIR_AMD_GRAPH_PAYOFF", "32"))

    def __init__(self, fn, warmup=1, max_graphs=None):
        self.fn = fn
        self.warmup = warmup
        self.upcoming = None               # images of the current shape still to come, if the caller knows (else None)
        self.captures = 0
        self.max_graphs = max_graphs or int(os.environ.get("MDIR_AMD_MAX_GRAPHS", "64"))   # with the shared pool a graph costs its input + output tensors
        self.graphs = collections.OrderedDict()
        # ONE memory pool for all captures of this object (MDIR_AMD_GRAPH_SHARED_POOL=0: a private pool per graph).  The
        # graphs are replayed one after the other on one stream, a graph's intermediates are dead when its replay ends and its
        # input / output tensors stay allocated, so a later capture may live in the memory an earlier one used in between:
        # 16 image sizes x 2 batch shapes of a ResNet101 pyramid reserve the largest graph's memory instead of the sum
        # (tools/extract_mem_probe.py)
        self.pool = torch.cuda.graph_pool_handle() if os.environ.get("MDIR_AMD_GRAPH_SHARED_POOL", "1") != "0" and torch.cuda.is_available() else None
        self.seen = collections.Counter()
        self.refused = set()
        self.replays = 0

    def __call__(self, x):
        key = (tuple(x.shape), x.dtype, x.device.index)
        entry = self.graphs.get(key)
        if entry is None:
            self.seen[key] += 1
            if _captures_off or key in self.refused or self.seen[key] <= self.warmup:
                return self.fn(x)
            if self.upcoming is not None and self.upcoming < self.PAYOFF_IMAGES:
                return self.fn(x)                     # too few images of this shape left for a capture to pay off
            entry = self._capture(x, key)
            if entry is None:
                return self.fn(x)
        else:
            self.graphs.move_to_end(key)
        entry.static_in.copy_(x, non_blocking=True)
        entry.graph.replay()
        self.replays += 1
        # INVARIANT of the shared pool: `static_out` is valid only until the next replay of ANY graph of this pool (replay
        # order is LRU, not capture order: an earlier graph's intermediates may occupy the memory a later graph's output
        # lives in).  It holds because static_in is allocated outside the pool, everything runs on ONE stream and the output
        # is cloned here, on that stream, before anything else is replayed -- never return static_out itself or read it
        # from another stream (tests/test_gpu_round5.py replays in reverse capture order against eager).
        out = entry.static_out
        return out.clone() if isinstance(out, torch.Tensor) else type(out)(o.clone() for o in out)

    def _capture(self, x, key):
        while len(self.graphs) >= self.max_graphs:
            self.graphs.popitem(last=False)          # frees that graph (and, without the shared pool, its private memory pool)
        entry = _Entry()
        entry.static_in = x.clone()
        entry.graph = torch.cuda.CUDAGraph()
        # thread_local: the loader's pin-memory thread and RCCL's watchdog thread keep making HIP
        # calls (host allocations, event queries) while this thread captures; only calls made by
        # the capturing thread itself may invalidate the capture
        before = torch.cuda.current_stream()
        ctx = torch.cuda.graph(entry.graph, pool=self.pool, capture_error_mode="thread_local")
        try:
            with ctx:
                entry.static_out = self.fn(entry.static_in)
        except Exception as err:                     # keep extracting eagerly; never silently change results
            # What a failed capture leaves behind (found by tests/test_gpu_round5.py::test_graph_bookkeeping_refusal_and_eviction,
            # round 5 -- until then the eager call that followed a refusal failed with "operation failed due to a previous error
            # during capture"): torch.cuda.graph.__exit__ raises out of capture_end() BEFORE it restores the stream context, so
            # this thread's current stream is still the (invalidated) capture stream; that stream may still be capturing; and
            # HIP's per-thread last-error is set.  Put the stream back, end the capture, clear the error.
            # (`stream_ctx` / `capture_stream` are private attributes of torch.cuda.graph: on a PyTorch without them the recovery
            # falls back to the public calls -- the original capture error must never be replaced by an AttributeError here)
            from . import _lib
            if torch.cuda.current_stream() != before:
                stream_ctx = getattr(ctx, "stream_ctx", None)
                try:
                    if stream_ctx is None:
                        raise AttributeError("stream_ctx")
                    stream_ctx.__exit__(None, None, None)
                except Exception:
                    torch.cuda.set_stream(before)
            streams = {before}
            if getattr(ctx, "capture_stream", None) is not None:
                streams.add(ctx.capture_stream)
            for stream in streams:
                _lib.lib().mdx_capture_recover(ctypes.c_void_p(stream.cuda_stream))
            torch.cuda.synchronize()
            # ... and PyTorch's random generator of the device was told that a capture began and never that it ended (the same
            # early exit of capture_end): the next torch.rand would fail with "Offset increment outside graph capture".  A
            # clone of its state is a fresh state object (same seed and offset) without the flag.
            try:
                gen = torch.cuda.default_generators[before.device_index]
                gen.graphsafe_set_state(gen.clone_state())
            except Exception:                       # older / newer PyTorch without the graph-safe state API: nothing to repair with
                pass
            self.refused.add(key)
            global _captures_off, _refusals
            _captures_off = True
            _refusals += 1
            warnings.warn("hipGraph capture refused for input shape %s (%s); this and every further shape of the process stay eager"
                          % (key[0], err))
            return None
        self.graphs[key] = entry
        self.captures += 1
        return entry
