"""Row-sharded database over the GPUs of one node (one process per GPU).

The reference has no distributed code; this is the
MI355X-native scale-out of ``cirscore.py:69-70``.  Database rows are independent,
so rank g keeps rows ``[row_offset, row_offset + n_local)`` resident as its own
``DescriptorIndex`` and computes ``S_g [Q, n_local]`` with no communication.  Two
exchange patterns sit on top, both through ``torch.distributed`` (backend "nccl"
= RCCL over xGMI on the GPU box, "gloo" in the CPU tests):

* ``rank_queries``  -- exact FULL ranking.  One all-to-all re-partitions the
  partial scores from "all queries x my rows" to "my queries x all rows"
  (each rank sends 1/G of its block to every peer: point-to-point xGMI links, no
  ring), then every rank sorts only its own ceil(Q/G) queries, so the sort is
  G-way parallel too.  Query q's ranking lives on rank ``owner(q)``.
* ``positions``     -- rank positions of labelled ids WITHOUT sorting: labelled
  scores are summed over shards (all-reduce, a few kB), each shard counts how many
  of its rows precede each labelled item, and the counts are all-reduced.

The compute backend is injected so the exchange logic is testable on CPU: the
product default is the HIP library (``HipBackend``) and raises without a GPU;
tests pass an oracle-backed object explicitly.
"""
import os

import torch
import torch.distributed as dist


class HipBackend:
    """libmdx.so through mdir_amd.ops (the only backend the product ships)."""

    def make_index(self, vecs, layout, row_offset, storage="f32"):
        from . import ops
        return ops.DescriptorIndex(vecs, layout, row_offset, storage=storage)

    def rank_full(self, scores, id_offset=0):
        from . import ops
        return ops.rank_full(scores, id_offset)

    def rank_full_segments(self, blocks, id_offset=0):
        from . import ops
        if len(blocks) > ops.MAX_RANK_SEGMENTS:
            return ops.rank_full(torch.cat(blocks, dim=1), id_offset)
        return ops.rank_full_segments(blocks, id_offset)

    def topk(self, scores, k, id_offset=0):
        from . import ops
        return ops.topk(scores, k, id_offset)

    def gather_scores(self, scores, ids, offsets):
        from . import ops
        return ops.gather_scores(scores, ids, offsets)

    def rank_count_(self, cnt, scores, id_offset, ref_scores, ref_ids, offsets):
        from . import ops
        return ops.rank_count_(cnt, scores, id_offset, ref_scores, ref_ids, offsets)


class BlockScores:
    """``[Q_mine, N]`` similarities held as the column blocks the exchange delivered (``.blocks``: block g is
    ``[Q_mine, rows of that peer chunk]``; ``.dense()`` concatenates on demand).  The blocks are the caller's: every
    ``rank_queries`` / ``exchange`` call receives into fresh memory unless the index was told to recycle its receive
    buffers (``ShardedIndex.reuse_buffers = True``: a benchmark loop that keeps only the newest result) -- then they
    are valid until the next exchange of the same index."""

    def __init__(self, blocks):
        self.blocks = blocks

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, item):
        return self.dense()[item]

    def cpu(self):
        return self.dense().cpu()

    @property
    def shape(self):
        return (self.blocks[0].shape[0], sum(int(b.shape[1]) for b in self.blocks))

    def dense(self):
        return self.blocks[0] if len(self.blocks) == 1 else torch.cat(self.blocks, dim=1)


def private_miopen_caches(local_rank, job_id=None):
    """One process per GPU: give every rank its OWN MIOpen user database and kernel cache directory (unless the user has set
    them).  MIOpen keeps both in sqlite files under ~/.config/miopen and ~/.cache/miopen; eight ranks that meet the same new
    convolution shapes at the same moment would all write them at once.  Must run before the process's first convolution.

    The directories live under ``~/.cache/mdir_amd/miopen/<job>/rank<N>`` (``$XDG_CACHE_HOME`` honoured): private to the user
    (mode 0700, ownership checked -- nobody else can plant code objects there), private to the JOB (``job`` = the launcher's
    ``TORCHELASTIC_RUN_ID`` + ``MASTER_PORT``, so two jobs of one user on one node, both with local ranks 0..3, do not share
    sqlite files), and they survive a ``/tmp`` clean-up.  A fresh rank directory is seeded ONCE with a copy of the user's shared
    MIOpen user database, so that a rank does not start from nothing; the shared files themselves are never written."""
    import os
    import shutil
    import stat
    if job_id is None:
        job_id = "%s_%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "norun"), os.environ.get("MASTER_PORT", str(os.getppid())))
    job_id = "".join(c if c.isalnum() or c in "-_." else "_" for c in str(job_id))[:80] or "job"
    cache_home = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    base = os.path.join(cache_home, "mdir_amd", "miopen", job_id, "rank%d" % int(local_rank))
    made = {}
    for var, sub_dir in (("MIOPEN_USER_DB_PATH", "db"), ("MIOPEN_CUSTOM_CACHE_DIR", "cache")):
        if var in os.environ:
            continue
        path = os.path.join(base, sub_dir)
        fresh = not os.path.isdir(path)
        os.makedirs(path, mode=0o700, exist_ok=True)
        # every level this function owns must be a real directory of this user that nobody else can write
        probe = path
        while len(probe) >= len(os.path.join(cache_home, "mdir_amd")):
            st = os.lstat(probe)
            if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid():
                raise PermissionError("%s is not a directory owned by uid %d: refusing to point MIOpen at it" % (probe, os.getuid()))
            if st.st_mode & 0o022:
                os.chmod(probe, st.st_mode & 0o7755 & ~0o022)
            probe = os.path.dirname(probe)
        os.environ[var] = path
        made[var] = (path, fresh)
    if made.get("MIOPEN_USER_DB_PATH", (None, False))[1]:
        shared = os.path.join(os.environ.get("XDG_CONFIG_HOME") or os.path.join(os.path.expanduser("~"), ".config"), "miopen")
        if os.path.isdir(shared):
            for name in os.listdir(shared):
                src = os.path.join(shared, name)
                try:
                    if os.path.isfile(src) and os.path.getsize(src) < (256 << 20):
                        shutil.copy2(src, os.path.join(made["MIOPEN_USER_DB_PATH"][0], name))
                except OSError:
                    pass            # a seed is a convenience: a file another process is writing is simply not taken
    return base


def shard_bounds(n_total, world, rank):
    """Rows ``[lo, hi)`` of shard ``rank``: contiguous, sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def query_bounds(nq, world, rank):
    return shard_bounds(nq, world, rank)


def exchange_chunks(n_total, world):
    """How many row chunks every shard is cut into so that the all-to-all of chunk c runs on
    RCCL's stream while the similarity kernel of chunk c+1 runs on the compute stream.  Small
    chunks cost kernel efficiency (measured: tools/shard_model.py, tools/g8_budget.py), so only big shards are split:
    three chunks at G=2, two at G=4 for 1 M rows (chunk sizes halve, see ``chunk_bounds``), and two chunks at G=8 (125 k-row
    shards: a first launch of exactly one round of the chip's 512 workgroup slots + the rest take less time than the one launch
    of 982 workgroups, and the first chunk's transfer hides behind the second chunk's kernel; profiles/r05_g8_budget.md).  Same value on every rank (derived from the largest shard)."""
    import os
    forced = os.environ.get("MDIR_AMD_EXCHANGE_CHUNKS")
    if forced:
        chunks = max(1, int(forced))
        return chunks if n_total // world >= (1 << chunks) else 1      # no empty chunk on any rank
    if world == 1:
        return 1
    biggest = shard_bounds(n_total, world, 0)[1]
    return 3 if biggest >= 400_000 else (2 if biggest >= 120_000 else 1)


MIN_CHUNK_ROWS = 60_000
# one full round of the exact similarity kernel on an MI355X: 256 CUs x 2 resident workgroups x 128 rows (mdx_scores_kernel.h, R = 2)
FULL_ROUND_ROWS = 512 * 128


def chunk_bounds(lo, hi, chunks):
    """Contiguous chunks of rows ``[lo, hi)`` with sizes halving from one to the next (4/7, 2/7, 1/7 for
    three): a link moves a chunk's scores in about half the time the similarity kernel needs for the
    same rows, so the transfer of chunk c hides behind the kernel of the half-sized chunk c+1, and
    only the LAST, smallest transfer is exposed.

    Shards too small for halving chunks (G = 8 at 1 M rows) are cut in two with the FIRST chunk exactly one full round of
    the chip's 512 workgroup slots (65 536 rows) and the rest second: 0.344 ms against 0.359 for two equal chunks and 0.356
    for the one launch (profiles/r05_g8_budget.md; re-measured in profiles/r06_g8_budget.md) -- the first launch has no
    partial last round, and the smaller second chunk is also the smaller exposed transfer."""
    n = hi - lo
    weights = [1 << (chunks - 1 - c) for c in range(chunks)]
    if n // sum(weights) < MIN_CHUNK_ROWS:
        if chunks == 2 and FULL_ROUND_ROWS + FULL_ROUND_ROWS // 2 <= n <= 2 * FULL_ROUND_ROWS:
            return [(lo, lo + FULL_ROUND_ROWS), (lo + FULL_ROUND_ROWS, hi)]
        # the smallest of the halving chunks would not fill the chip once (a similarity launch has 512 workgroup slots of 128
        # rows): equal chunks instead (at 125 625 rows the halving cut 83 750 + 41 875 takes 0.40 ms against 0.34: tools/g8_budget.py)
        weights = [1] * chunks
    total, edges, acc = sum(weights), [lo], 0
    for w in weights[:-1]:
        acc += w
        edges.append(lo + (n * acc) // total)
    edges.append(hi)
    return [(edges[c], edges[c + 1]) for c in range(chunks)]


class ShardedIndex:
    def __init__(self, local_vecs, layout, n_total, group=None, backend=None, storage="f32", compute="chain"):
        """``storage``: "f32" (exact chain) or "f16" (fp16 shard on the fp16 MFMA, BASELINE.json configs[4]).
        ``compute``: "chain" (default), "split3" or "split2" -- the labelled split-precision similarities on an fp32 shard
        (``DescriptorIndex.scores(compute=...)``, ``include/mdx.h`` ``MDX_F32_SPLIT3``)."""
        self.group = group
        self.storage = storage
        if compute not in ("chain", "exact", "split3", "split2"):
            raise ValueError("compute %r" % (compute,))
        if compute in ("split3", "split2") and storage != "f32":
            raise ValueError("compute=%r multiplies an fp32 shard" % (compute,))
        self._score_kw = {"compute": compute} if compute in ("split3", "split2") else {}
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_total = int(n_total)
        self.lo, self.hi = shard_bounds(self.n_total, self.world, self.rank)
        dim_major = layout in ("DN", "dim_major")
        n_local = local_vecs.shape[1] if dim_major else local_vecs.shape[0]
        if n_local != self.hi - self.lo:
            raise ValueError("rank %d holds %d rows, expected %d" % (self.rank, n_local, self.hi - self.lo))
        self.backend = backend or HipBackend()
        self.device = local_vecs.device
        self.chunks = exchange_chunks(self.n_total, self.world)
        self.parts = []                    # (lo, hi, index) per chunk, ascending rows
        for a, b in chunk_bounds(self.lo, self.hi, self.chunks):
            if self.chunks == 1:
                piece = local_vecs
            else:
                piece = (local_vecs[:, a - self.lo:b - self.lo] if dim_major else local_vecs[a - self.lo:b - self.lo])
                piece = piece.contiguous()
            self.parts.append((a, b, self.backend.make_index(piece, layout, a, storage) if storage != "f32"
                               else self.backend.make_index(piece, layout, a)))
        self.index = self.parts[0][2] if self.chunks == 1 else None
        # the direct-store exchange wants the shard as ONE launch (its transfer is spread over the kernel by construction, and
        # every launch ends by draining its write-through stores: two chunks 0.370 ms, one launch 0.343, profiles/r06_g8_budget.md):
        # a whole-shard index is built on first use from the caller's matrix (kept by reference, not copied)
        self._whole = self.index
        self._local = (local_vecs, layout)
        # RCCL moves device buffers directly; under gloo (CPU tests, or several ranks
        # sharing one GPU for a dry run) collectives are staged through host memory.
        self._host_staged = (self.world > 1 and dist.get_backend(group) == "gloo" and self.device.type == "cuda")
        # False: every exchange receives into fresh memory (results stay valid).  True: one receive buffer per chunk is
        # kept and OVERWRITTEN by the next exchange -- for loops that only keep the newest result (bench.py)
        self.reuse_buffers = False
        self._recv = {}                    # (chunk, elements) -> receive buffer of the exchange (reuse_buffers only)
        self.phases = None                 # events of the last rank_queries(): see phase_ms()
        # MDIR_AMD_COMM=mdx: the exchange goes through libmdx's own RCCL communicator (mdx_exchange_scores /
        # mdx_allgather_scores: the C-ABI form a non-Python host would call) on a side stream, instead of
        # torch.distributed's all_to_all_single; the process group is then only used to hand out the communicator id.
        self._comm = self._comm_stream = None
        if os.environ.get("MDIR_AMD_COMM") == "mdx" and self.device.type == "cuda" and not self._host_staged:
            from . import ops
            self._comm = ops.Comm.from_process_group(self.device, group)
            self._comm_stream = torch.cuda.Stream(device=self.device)
        # MDIR_AMD_COMM=p2p (bench.py --comm p2p): no collective at all -- the similarity kernel writes every query's scores
        # straight into the owner rank's receive buffer (hipIpc mapping, xGMI stores; ops.P2P / mdx_scores_p2p) and one flag per
        # peer closes the step; the owner ranks a dense [Q_mine, N] matrix.  Exact fp32 shards only.  The exchange object is
        # built at the first step (it is sized by the number of queries); the process group only carries its 64-byte handles.
        self._p2p_on = (os.environ.get("MDIR_AMD_COMM") == "p2p" and self.device.type == "cuda" and storage == "f32"
                        and not self._score_kw)
        self._p2p = None
        self._use_a2a = True if self._p2p_on else self._probe_all_to_all()

    def _probe_all_to_all(self):
        """Decide the exchange form ONCE, identically on every rank: a tiny UNEVEN all_to_all_single is run and
        waited for, and the per-rank outcome is all-reduced with MIN.  (RCCL reports most failures at wait() or
        from its watchdog, not at the call, and a rank that switched form alone would deadlock the others.)"""
        if self.world == 1:
            return True
        ok = 0 if os.environ.get("MDIR_AMD_EXCHANGE") == "allgather" else 1
        dev = "cpu" if self._host_staged else self.device
        if ok:
            try:
                send = torch.arange(self.world * (self.world + 1) // 2, dtype=torch.float32, device=dev)
                in_split = [r + 1 for r in range(self.world)]             # rank r receives r+1 elements from everybody
                out_split = [self.rank + 1] * self.world
                recv = torch.empty(sum(out_split), dtype=torch.float32, device=dev)
                work = dist.all_to_all_single(recv, send, out_split, in_split, group=self.group, async_op=True)
                work.wait()
                if recv.is_cuda:
                    torch.cuda.synchronize(recv.device)
                lo = self.rank * (self.rank + 1) // 2
                ok = int(bool((recv.cpu().view(self.world, -1) == torch.arange(lo, lo + self.rank + 1, dtype=torch.float32)).all()))
            except (RuntimeError, NotImplementedError) as err:
                import warnings
                warnings.warn("all_to_all_single unavailable (%s)" % err)
                ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def _all_reduce_sum(self, t):
        if self._host_staged:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    # ------------------------------------------------------------------ scores
    def local_scores(self, queries, qlayout="DN"):
        """``[Q, n_local]`` similarities against this rank's rows (no communication)."""
        if self.chunks == 1:
            return self.index.scores(queries, qlayout, **self._score_kw)
        return torch.cat([ix.scores(queries, qlayout, **self._score_kw) for _, _, ix in self.parts], dim=1)

    # ----------------------------------------------------------- full ranking
    def _peer_widths(self, c):
        """Rows of chunk ``c`` on every rank (what each peer sends me per query)."""
        out = []
        for r in range(self.world):
            a, b = chunk_bounds(*shard_bounds(self.n_total, self.world, r), self.chunks)[c]
            out.append(b - a)
        return out

    def _start_exchange(self, s_part, widths, chunk=0):
        """Enqueue the all-to-all that turns "all queries x my rows" into "my queries x every
        peer's rows" for one chunk.  Returns ``(work, recv, host_recv, keepalive)``; the
        collective runs on the communicator's own stream, so kernels launched next on the
        compute stream overlap with it."""
        nq = s_part.shape[0]
        qlo, qhi = query_bounds(nq, self.world, self.rank)
        if self._comm is not None:
            # libmdx's communicator: enqueue on the side stream once the compute stream has produced s_part
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ready)
                if self._use_a2a:
                    blocks, _ = self._comm.exchange_scores(s_part, widths)
                else:
                    blocks = [b[qlo:qhi] for b in self._comm.allgather_scores(s_part, widths)]
                done = torch.cuda.Event()
                done.record()
            s_part.record_stream(self._comm_stream)
            return "mdx", blocks, done, s_part
        in_split = [(query_bounds(nq, self.world, r)[1] - query_bounds(nq, self.world, r)[0]) * s_part.shape[1]
                    for r in range(self.world)]
        out_split = [(qhi - qlo) * w for w in widths]
        key = (chunk, sum(out_split))
        recv = self._recv.get(key) if self.reuse_buffers else None
        if recv is None or recv.device != s_part.device:
            recv = torch.empty(sum(out_split), dtype=s_part.dtype, device=s_part.device)
            if self.reuse_buffers:
                self._recv[key] = recv
        if self._host_staged:
            send, host_recv = s_part.reshape(-1).cpu(), torch.empty(recv.shape, dtype=recv.dtype)
            if self._use_a2a:
                work = dist.all_to_all_single(host_recv, send, out_split, in_split, group=self.group, async_op=True)
                return work, recv, host_recv, send
            s_part = s_part.cpu()
        send = s_part.reshape(-1)
        if self._use_a2a:
            work = dist.all_to_all_single(recv, send, out_split, in_split, group=self.group, async_op=True)
            return work, recv, None, send
        # the all-gather form (chosen by every rank in __init__): every rank gathers the (padded) blocks and keeps
        # its own queries -- G times the bytes, same result
        wmax = max(widths)
        block = s_part if s_part.shape[1] == wmax else torch.cat(
            [s_part, s_part.new_zeros((nq, wmax - s_part.shape[1]))], dim=1)
        parts = [torch.empty_like(block) for _ in range(self.world)]
        dist.all_gather(parts, block.contiguous(), group=self.group)
        o = 0
        for part, w in zip(parts, widths):
            recv[o:o + (qhi - qlo) * w].copy_(part[qlo:qhi, :w].reshape(-1))
            o += (qhi - qlo) * w
        return None, recv, None, send

    def _finish_exchange(self, pending, widths, nq_mine):
        work, recv, host_recv, _send = pending
        if isinstance(work, str):                     # "mdx": recv = the blocks, host_recv = the side stream's event
            torch.cuda.current_stream().wait_event(host_recv)
            for b in recv:
                b.record_stream(torch.cuda.current_stream())
            return recv
        if work is not None:
            work.wait()                               # compute stream waits for the collective
        if host_recv is not None:
            recv.copy_(host_recv)
        blocks, o = [], 0
        for w in widths:
            blocks.append(recv[o:o + nq_mine * w].view(nq_mine, w))
            o += nq_mine * w
        return blocks

    def exchange(self, s_local):
        """``[Q, n_local]`` on every rank  ->  ``[Q_mine, N]`` on every rank (one all-to-all)."""
        nq = s_local.shape[0]
        if self.world == 1:
            return s_local, (0, nq)
        qlo, qhi = query_bounds(nq, self.world, self.rank)
        widths = [shard_bounds(self.n_total, self.world, r)[1] - shard_bounds(self.n_total, self.world, r)[0]
                  for r in range(self.world)]
        blocks = self._finish_exchange(self._start_exchange(s_local, widths), widths, qhi - qlo)
        return torch.cat(blocks, dim=1), (qlo, qhi)

    def exchanged_scores(self, queries, qlayout="DN"):
        """Similarities of MY queries against ALL rows as one ``[Q_mine, N]`` matrix (``exchanged_blocks`` + one
        concatenation)."""
        blocks, bounds = self.exchanged_blocks(queries, qlayout)
        return (blocks[0] if len(blocks) == 1 else torch.cat(blocks, dim=1)), bounds

    def exchanged_blocks(self, queries, qlayout="DN"):
        """Similarities of MY queries against ALL rows, as the column blocks the exchange delivers (global row order:
        peer-major, chunk-minor; block = ``[Q_mine, rows of that chunk]``): per chunk, similarity kernel then
        all-to-all, with chunk c's transfer overlapping chunk c+1's kernel."""
        if self._p2p_on:
            return self._exchanged_p2p(queries, qlayout)
        if self.world == 1 and self._comm is None:
            s = self.local_scores(queries, qlayout)
            return [s], (0, s.shape[0])
        pending, nq = [], None
        ev = self._events()
        if ev:
            ev[0].record()
        for c, (_, _, ix) in enumerate(self.parts):
            s_part = ix.scores(queries, qlayout, **self._score_kw)
            nq = s_part.shape[0]
            widths = self._peer_widths(c)
            pending.append((self._start_exchange(s_part, widths, c), widths))
        if ev:
            ev[1].record()                                # similarity kernels enqueued up to here
        qlo, qhi = query_bounds(nq, self.world, self.rank)
        per_chunk = [self._finish_exchange(p, widths, qhi - qlo) for p, widths in pending]
        if ev:
            ev[2].record()                                # the compute stream has waited for the last transfer
        # global row order: peer-major, chunk-minor
        return [per_chunk[c][r] for r in range(self.world) for c in range(self.chunks)], (qlo, qhi)

    def use_direct_store(self, on):
        """Switch the exchange of ``rank_queries`` between the direct-store form and the collective one at run time (every rank
        must make the same call between the same steps).  Exact fp32 shards only."""
        if on and (self.storage != "f32" or self._score_kw or self.device.type != "cuda"):
            raise ValueError("the direct-store exchange needs an exact fp32 shard on the GPU")
        self._p2p_on = bool(on)
        self.phases = None

    def prepare_direct_store(self, nq):
        """Everything the direct-store form needs BEFORE its first step, made with the ranks in lockstep: the whole-shard index,
        the exchange object (handles gathered through the process group) and the peer mappings.  Returns True on every rank or
        False on every rank -- a rank whose local part failed (no memory, a peer buffer it cannot map) drags the others out with
        it through two MIN all-reduces, so nobody is left waiting in a collective the failed rank never reaches."""
        from . import ops

        def agree(ok):
            if self.world == 1:
                return bool(ok)
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if self._host_staged else self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return bool(int(flag.item()))
        ok = self.storage == "f32" and not self._score_kw and self.device.type == "cuda"
        if ok and self._whole is None:
            try:
                self._whole = self.backend.make_index(self._local[0], self._local[1], self.lo)
            except Exception:               # noqa: BLE001 -- reported through the agreement below
                ok = False
        if not agree(ok):
            return False
        if self._p2p is not None and self._p2p.nq == nq and self._p2p.connected:
            return True
        if self._p2p is not None:
            self._p2p.close()
            self._p2p = None
        me, ok = None, True
        try:
            me = ops.P2P(self.world, self.rank, nq, self.n_total, self.device)
        except Exception:                   # noqa: BLE001
            ok = False
        handles = [(me.handle if me is not None else b"", bool(me is not None and (me.exportable or self.world == 1)))]
        if self.world > 1:
            handles = [None] * self.world
            dist.all_gather_object(handles, (me.handle if me is not None else b"", bool(me is not None and me.exportable)), group=self.group)
        ok = all(good for _, good in handles)
        if ok:
            try:
                me.connect([h for h, _ in handles])
            except Exception:               # noqa: BLE001
                ok = False
        if not agree(ok):
            if me is not None:
                me.close()
            return False
        self._p2p = me
        return True

    def _exchanged_p2p(self, queries, qlayout):
        """The direct-store form: every chunk's similarity kernel writes to the owners' buffers, one flag per peer closes the
        step, and MY queries' rows of ALL shards are one dense matrix (a view of the receive buffer, valid for two steps;
        cloned unless ``reuse_buffers``)."""
        from . import ops
        nq = queries.shape[1] if qlayout in ("DN", "dim_major") else queries.shape[0]
        if self._p2p is None or self._p2p.nq != nq:
            if not self.prepare_direct_store(nq):
                raise RuntimeError("the direct-store exchange could not be set up on every rank (peer buffers could not be shared or "
                                   "mapped: is HSA_ENABLE_IPC_MODE_LEGACY=0 set?); use the collective form")
        ev = self._events()
        if ev:
            ev[0].record()
        self._whole.scores_p2p(queries, self._p2p, qlayout)
        if ev:
            ev[1].record()
        mine = self._p2p.close_step()
        if ev:
            ev[2].record()
        if not self.reuse_buffers:
            mine = mine.clone()
        return [mine], query_bounds(nq, self.world, self.rank)

    def rank_queries(self, queries, qlayout="DN"):
        """Exact full ranking, query-partitioned: returns ``(ranks [Q_mine, N] int64 with GLOBAL ids,
        scores, (qlo, qhi))``.  The peer blocks of the exchange enter the sort as segments
        (``mdx_rank_full_segments``): no re-blocked ``[Q_mine, N]`` copy is made on the way to the ranking.
        ``scores`` is a :class:`BlockScores`: ``.blocks`` as delivered, ``.dense()`` the ``[Q_mine, N]`` matrix
        (concatenated on demand)."""
        blocks, (qlo, qhi) = self.exchanged_blocks(queries, qlayout)
        if qhi == qlo:
            ranks = torch.empty((0, self.n_total), dtype=torch.int64, device=self.device)
        elif len(blocks) == 1:
            ranks = self.backend.rank_full(blocks[0], 0)
        elif hasattr(self.backend, "rank_full_segments"):
            ranks = self.backend.rank_full_segments(blocks, 0)
        else:
            ranks = self.backend.rank_full(torch.cat(blocks, dim=1), 0)
        if self.phases:
            self.phases[3].record()
        return ranks, BlockScores(blocks), (qlo, qhi)

    def _events(self):
        """Four events on the compute stream around the phases of ``rank_queries`` (GPU only)."""
        if self.device.type != "cuda" or (self.world == 1 and self._comm is None and not self._p2p_on):
            self.phases = None
        elif self.phases is None:
            self.phases = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        return self.phases

    def phase_ms(self):
        """``{"scores_ms", "exchange_exposed_ms", "sort_ms"}`` of the last ``rank_queries`` on this rank (call after a
        synchronize).  ``exchange_exposed_ms`` = what the compute stream waited for transfers AFTER its last
        similarity kernel (the re-block copy included); transfers hidden behind kernels do not show."""
        if not self.phases:
            return None
        e = self.phases
        return {"scores_ms": e[0].elapsed_time(e[1]), "exchange_exposed_ms": e[1].elapsed_time(e[2]),
                "sort_ms": e[2].elapsed_time(e[3])}

    # ------------------------------------------------- all-gather of partial scores
    def all_scores(self, queries, qlayout="DN"):
        """``[Q, N]`` similarities of ALL queries against ALL rows on every rank: the literal
        "all-gather of per-shard partial scores" (G times the bytes of ``rank_queries``' all-to-all
        per rank, and every rank then holds -- and would sort -- everything; use it when one rank
        needs the whole score matrix)."""
        s_local = self.local_scores(queries, qlayout)
        if self.world == 1:
            return s_local
        nq = s_local.shape[0]
        widths = [shard_bounds(self.n_total, self.world, r)[1] - shard_bounds(self.n_total, self.world, r)[0]
                  for r in range(self.world)]
        wmax = max(widths)
        mine = s_local if s_local.shape[1] == wmax else torch.cat(
            [s_local, s_local.new_zeros((nq, wmax - s_local.shape[1]))], dim=1)      # shards differ by <= 1 row
        parts = self._all_gather(mine.contiguous())
        return torch.cat([p[:, :w] for p, w in zip(parts, widths)], dim=1)

    # ------------------------------------------------------------ global top-k
    def topk_queries(self, queries, k, qlayout="DN"):
        """Exact global top-k of every query on every rank: ``(ids int64 [Q,k'], scores [Q,k'])`` with
        ``k' = min(k, N)``.  Each shard selects its own best ``min(k, n_local)`` rows (``mdx_topk``),
        only those candidates travel (one all-gather of ``Q*k`` (id, score) pairs per rank instead of
        the ``Q*n_local`` scores of the full ranking), and the ``G*k`` candidates are ranked again.
        Candidates are laid out shard after shard and each shard's list is already in (score
        descending, id ascending) order, so equal scores keep ascending global ids: the tie rule of
        the full ranking."""
        s_local = self.local_scores(queries, qlayout)
        nq, n_local = s_local.shape
        k_total = min(int(k), self.n_total)
        k_local = min(int(k), n_local)
        ids, vals = self.backend.topk(s_local, k_local, self.lo)
        if self.world == 1:
            return ids, vals
        # shards differ by at most one row, but k may exceed a shard: pad to a common width
        width = min(int(k), shard_bounds(self.n_total, self.world, 0)[1])
        if k_local < width:
            pad = width - k_local
            ids = torch.cat([ids, torch.full((nq, pad), -1, dtype=ids.dtype, device=ids.device)], dim=1)
            vals = torch.cat([vals, torch.full((nq, pad), float("nan"), dtype=vals.dtype, device=vals.device)], dim=1)
        all_ids = self._all_gather(ids.contiguous())            # G x [Q, width]
        all_vals = self._all_gather(vals.contiguous())
        cand_ids = torch.cat(all_ids, dim=1)
        cand_vals = torch.cat(all_vals, dim=1).contiguous()      # NaN padding ranks last
        order = self.backend.rank_full(cand_vals, 0)[:, :k_total]
        return torch.gather(cand_ids, 1, order), torch.gather(cand_vals, 1, order)

    def _all_gather(self, t):
        if self._host_staged:
            h = t.cpu()
            parts = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(parts, h, group=self.group)
            return [p.to(t.device) for p in parts]
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t, group=self.group)
        return parts

    # -------------------------------------------------- positions w/o sorting
    def positions(self, s_local, id_lists):
        """Global zero-based rank positions of labelled GLOBAL ids, per query.

        Returns ``(pos int64 [total], offsets list)`` identical on every rank."""
        nq = s_local.shape[0]
        offsets, flat = [0], []
        for ids in id_lists:
            flat.extend(int(i) for i in ids)
            offsets.append(len(flat))
        ids_t = torch.tensor(flat, dtype=torch.int64, device=self.device)
        off_t = torch.tensor(offsets, dtype=torch.int64, device=self.device)
        total = ids_t.numel()
        cnt = torch.zeros(total, dtype=torch.int64, device=self.device)
        if total == 0:
            return cnt, offsets
        mine = (ids_t >= self.lo) & (ids_t < self.hi)
        local_ids = torch.where(mine, ids_t - self.lo, torch.zeros_like(ids_t))
        ref = self.backend.gather_scores(s_local, local_ids, off_t)
        ref = torch.where(mine, ref, torch.zeros_like(ref))
        if self.world > 1:
            self._all_reduce_sum(ref)   # exactly one owner per id
        self.backend.rank_count_(cnt, s_local, self.lo, ref, ids_t, off_t)
        if self.world > 1:
            self._all_reduce_sum(cnt)
        return cnt, offsets


# ---------------------------------------------------------------------------
# Extraction is embarrassingly parallel over images: rank g extracts the contiguous slice of
# database images that is going to be ITS shard anyway, so descriptors never move between
# GPUs; only the (few) query descriptors are exchanged.
# ---------------------------------------------------------------------------

def extract_shard(net, images, image_size, transform, device, group=None, **kwargs):
    """Descriptors ``[n_local, D]`` (device) of this rank's slice ``images[lo:hi]`` plus ``(lo, hi)``."""
    from .networks import extract_vectors_device
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(len(images), world, rank)
    vecs = extract_vectors_device(net, images[lo:hi], image_size, transform, device=device, **kwargs)
    return vecs, (lo, hi)


def gather_query_vectors(local_q, nq_total, group=None):
    """All ranks' slices of the query descriptors -> the full ``[Q, D]`` on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_q
    staged = dist.get_backend(group) == "gloo" and local_q.is_cuda
    d = local_q.shape[1]
    pieces = []
    for r in range(world):
        lo, hi = shard_bounds(nq_total, world, r)
        pieces.append(torch.empty((hi - lo, d), dtype=local_q.dtype, device="cpu" if staged else local_q.device))
    mine = local_q.cpu() if staged else local_q.contiguous()
    if _even(pieces):
        dist.all_gather(pieces, mine, group=group)
    else:
        _all_gather_uneven(pieces, mine, group)
    return torch.cat(pieces, dim=0).to(local_q.device)


def _even(pieces):
    return len({p.shape[0] for p in pieces}) == 1


def _all_gather_uneven(pieces, mine, group):
    """Slices differ by at most one row: broadcast each rank's piece (tiny tensors)."""
    rank = dist.get_rank(group)
    for r, p in enumerate(pieces):
        if r == rank:
            p.copy_(mine)
        dist.broadcast(p, src=dist.get_global_rank(group, r) if group is not None else r, group=group)


def sharded_retrieval_map(net, images, qimages, bbxs, gnd, dataset, image_size, transform, device,
                          group=None, backend=None, lap=None, storage="f32", compute="chain", **kwargs):
    """Distributed form of ``CirDatasetAp.__call__`` (cirscore.py:49-71): every rank extracts its
    slice of the database (which stays resident as its shard) and its slice of the queries, query
    descriptors are all-gathered, similarities are computed against the local shard, and mAP comes
    from the sort-free position counts.  Returns the same ``(averages, per_query)`` on every rank."""
    from .evaluate import _evaluate, _positions_of_rows
    import numpy as np
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if len(images) < world:
        raise ValueError("%d database images cannot be sharded over %d ranks (every rank needs at least one)"
                         % (len(images), world))
    vecs, _ = extract_shard(net, images, image_size, transform, device, group, **kwargs)
    if images == qimages and set(bbxs) == {None}:
        qlocal = vecs            # the query set IS the database (cirscore.py:56-57): slices coincide
    else:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        qlo, qhi = shard_bounds(len(qimages), world, rank)
        from .networks import extract_vectors_device
        qlocal = extract_vectors_device(net, qimages[qlo:qhi], image_size, transform, device=device,
                                        bbxs=bbxs[qlo:qhi] if bbxs else None, **kwargs) if qhi > qlo else \
            torch.empty((0, vecs.shape[1]), dtype=torch.float32, device=vecs.device)
    qvecs = gather_query_vectors(qlocal, len(qimages), group)
    if lap:
        lap("extract_descriptors")
    index = ShardedIndex(vecs, "ND", len(images), group=group, backend=backend, storage=storage, compute=compute)
    s_local = index.local_scores(qvecs.contiguous(), "ND")

    def positions_of(lists):
        pos, off = index.positions(s_local, lists)              # partial counts of every shard, summed (one all-reduce)
        pos = pos.cpu().numpy()
        return [pos[off[q]:off[q + 1]] for q in range(len(lists))]

    # every labelled id of every protocol level is ranked ONCE (one counting pass per shard + one collective), then looked up
    positions = _positions_of_rows(len(images), gnd, positions_of)
    result = _evaluate(dataset, gnd, [1, 5, 10], lambda g, kappas: positions.map(g, kappas), positions)
    if lap:
        lap("compute_score")
    return result
