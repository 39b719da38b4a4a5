"""Multi-GPU host driver: one process per GPU, families sharded in contiguous blocks.

The reference has no distributed path (its only parallelism is multiprocessing over independent
organism chunks, ppanggolin/ppanggolin.py:1039-1095); this module shards ONE NEM problem the way
SURVEY.md §8(e) lays out, for NCEM (hard labels), where every M-step sum is an integer count and
therefore exact under any reduction order:

  * E-step density (E1): embarrassingly parallel over the shard.
  * E-step sweep (E2): the Gauss-Seidel sweep is solved by relaxation rounds (k_sweep in
    csrc/nem_kernels.hip); a round only needs the other shards' labels of the previous round, so each
    round ends with ONE all-gather of the uint8 label blocks.  Every block carries, in its tail, the
    rank's "a label changed" byte, so the same all-gather tells every rank whether another round is
    needed -- no separate flag collective and no host round trip.  The fixed point is the global
    sequential sweep of the reference, so labels are bit-identical to a single-GPU run.
  * M-step: each rank popcounts its shard -> int32 statistics {N_k, S1[k][d]}, stored behind the flag
    byte in the block's tail; the all-gather that closes the iteration's LAST relaxation round carries
    them, and every rank sums the world partial arrays while deriving mu, epsilon, pi (integer sums:
    exact in any order).  They are counted from the labels of the round before (while the last,
    verifying round runs), which are final exactly when that round changes nothing -- and when it does,
    the pipeline stops anyway and the host redoes the statistics.  So an EM iteration costs TWO
    collectives (one with beta = 0) and no all-reduce.
  * Loop control runs on the device (k_ctrl logic, identical data on every rank => identical
    decisions); the host enqueues a batch of whole iterations, collectives included, and synchronises
    once per batch.

torch is plumbing here: device memory for the collective buffers, streams, torch.distributed.  All
arithmetic happens in the HIP kernels behind the C ABI (``nemgpu_shard_*``).  The driver is written
against a small *stepper* interface so that the protocol can be rehearsed on CPU with the gloo
backend (tests/test_distributed.py plugs the CPU oracle in as the stepper).
"""
import numpy as np

STATUS_OK, STATUS_W_EMPTYCLASS = 0, 2
FLAG_TAIL = 16          # bytes between a rank's label block (rounded up to 16) and its statistics (byte `blk` = "changed" flag)


def shard_bounds(n_total, world, rank):
    """Contiguous blocks of ceil(n/world) families (keeps most contiguity edges local)."""
    blk = (n_total + world - 1) // world
    lo = min(rank * blk, n_total)
    hi = min(lo + blk, n_total)
    return lo, hi, blk


def slot_layout(n_total, world, stats_words=0):
    """Label arrays: rank r owns slots [r*stride, r*stride+blk); behind them the flag byte (at blk) and, from
    stats_offset(blk) on, the rank's int32 statistics; stride is a multiple of 16."""
    blk = (n_total + world - 1) // world
    stride = stats_offset(blk) + (4 * stats_words + 15) // 16 * 16
    return blk, stride


def stats_offset(blk):
    """Byte offset of a rank's statistics inside its block of a label array."""
    return (blk + 15) // 16 * 16 + FLAG_TAIL


def family_to_slot(idx, blk, stride):
    idx = np.asarray(idx, np.int64)
    return ((idx // blk) * stride + idx % blk).astype(np.int32)


def slice_graph(nei, lo, hi, blk=None, stride=None):
    """Rows [lo, hi) of a global CSR graph.  Neighbour indices stay global family indices, or become label
    slots when (blk, stride) is given."""
    ptr, idx, w = nei
    b, e = int(ptr[lo]), int(ptr[hi])
    sub = np.ascontiguousarray(idx[b:e], np.int32)
    if blk is not None:
        sub = family_to_slot(sub, blk, stride)
    return (np.asarray(ptr[lo:hi + 1], np.int64) - b).astype(np.int32), sub, np.ascontiguousarray(w[b:e], np.float32)


class Comm:
    """The collectives of the sharded EM.  With the nccl (= RCCL) backend they run directly on device
    tensors, asynchronously; with gloo, device tensors are staged through host memory (rehearsal only)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)

    def allreduce_sum_int(self, v):
        t = self.torch.tensor([int(v)], dtype=self.torch.int64)
        if self.backend != "gloo":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return int(t.item())

    def allreduce_max_int(self, v):
        t = self.torch.tensor([int(v)], dtype=self.torch.int32)
        if self.backend != "gloo":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def allgather_blocks_(self, buf, stride):
        """In-place all-gather: rank r owns buf[r*stride:(r+1)*stride]; afterwards every rank holds all blocks."""
        if self.world == 1:
            return                    # (a rank alone holds every block already)
        mine = buf[self.rank * stride:(self.rank + 1) * stride]
        if self.backend == "gloo":
            h = mine.cpu() if mine.is_cuda else mine.clone()
            parts = [self.torch.empty_like(h) for _ in range(self.world)]
            self.dist.all_gather(parts, h, group=self.group)
            buf.copy_(self.torch.cat(parts).to(buf.device))
        else:
            self.dist.all_gather_into_tensor(buf, mine, group=self.group)


class GpuStepper:
    """The local compute of one rank: a NemEngine shard driven through the nemgpu_shard_* C ABI."""

    def __init__(self, x_local, nei_slots, k, n_total, world, rank, prop, center, disp, device, cfg):
        import torch
        from .engine import NemEngine
        self.torch = torch
        self.device = torch.device("cuda", device)
        d = x_local.shape[1]
        if shard_bounds(n_total, world, world - 1)[0] >= n_total:      # (the same answer on every rank: all raise together)
            raise ValueError("the last of %d ranks would own no family (n_total=%d): use fewer ranks" % (world, n_total))
        blk, stride = slot_layout(n_total, world, k + k * d)
        self.rank, self.blk, self.stride = rank, blk, stride
        # one dedicated (non-default) stream carries this rank's kernels AND the collectives torch issues for
        # it, so everything is stream-ordered and the whole batch can be captured into a graph
        self.stream = torch.cuda.Stream(self.device)
        self.eng = NemEngine(world * stride, d, k, device=device, stream=self.stream.cuda_stream,
                             site_lo=rank * stride, site_hi=rank * stride + x_local.shape[0])
        self.eng.set_matrix(x_local)
        self.eng.set_graph(nei_slots)
        self.eng.set_params(prop, center, disp)
        self.eng.configure(**cfg)
        self.eng.shard_layout(world, rank, blk, stride, n_total)
        self.libc = cfg.get("tie", "hash") == "libc" and cfg.get("algo", "ncem") == "ncem"

    def alloc(self, n, dtype):
        return self.torch.zeros(n, dtype=getattr(self.torch, dtype), device=self.device)

    # ---- RCCL called directly from the library (one C call enqueues a whole batch, collectives included)
    native = False

    def enable_native(self, comm):
        """Give the engine its own RCCL communicator over the ranks of `comm`.  torch.distributed is the bootstrap:
        it carries rank 0's ncclUniqueId to everybody and lets the ranks agree on whether the attempt worked.
        Collective; returns True when every rank attached."""
        import ctypes
        import os
        torch, dist = self.torch, comm.dist
        ok = 1
        uid = torch.zeros(128, dtype=torch.uint8, device=self.device)
        try:
            lib = self.eng.lib
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            if lib.nemgpu_rccl_open(path.encode() if os.path.isfile(path) else None) != 0:
                ok = 0
            if ok and comm.rank == 0:
                buf = (ctypes.c_uint8 * 128)()
                if lib.nemgpu_rccl_unique_id(buf) != 0:
                    ok = 0
                else:
                    uid.copy_(torch.tensor(list(buf), dtype=torch.uint8))
        except Exception:
            ok = 0
        # every rank goes through the same collectives whatever happened above
        dist.broadcast(uid, src=0, group=comm.group)
        if comm.allreduce_max_int(0 if ok else 1) != 0:
            return False
        host = (ctypes.c_uint8 * 128).from_buffer_copy(uid.cpu().numpy().tobytes())
        try:
            self.eng._chk(self.eng.lib.nemgpu_rccl_attach(self.eng._h, ctypes.addressof(host), comm.world, comm.rank))
        except Exception:
            ok = 0
        if comm.allreduce_max_int(0 if ok else 1) != 0:
            return False
        # Self-test before the EM relies on it: one all-gather of per-rank patterns through the engine's own
        # communicator (waited for with a deadline; a timeout aborts that communicator) against the same gather
        # through torch.distributed.  Any mismatch or failure on any rank keeps every rank on the torch path.
        stride = 256
        try:
            pattern = ((torch.arange(stride, device=self.device, dtype=torch.int32) * 7 + 13 * comm.rank + 1) % 251).to(torch.uint8)
            buf = torch.zeros(stride * comm.world, dtype=torch.uint8, device=self.device)
            buf[comm.rank * stride:(comm.rank + 1) * stride] = pattern
            want = torch.zeros_like(buf)
            dist.all_gather_into_tensor(want, pattern, group=comm.group)
            with torch.cuda.stream(self.stream):
                self.stream.wait_stream(torch.cuda.current_stream(self.device))
            torch.cuda.synchronize(self.device)
            rc = self.eng.lib.nemgpu_rccl_selftest(self.eng._h, ctypes.c_void_p(buf.data_ptr()), stride, 20000)
            ok = 1 if (rc == 0 and bool(torch.equal(buf, want))) else 0
        except Exception:
            ok = 0
        self.native = comm.allreduce_max_int(0 if ok else 1) == 0
        return self.native

    def enqueue_batch_native(self, with_init, n_iters, base, beta, want_stats, labels):
        self.eng.shard_enqueue_batch(with_init, n_iters, base, beta, want_stats, [t.data_ptr() for t in labels],
                                     stats_offset(self.blk))

    def on_stream(self):
        """Context under which the driver issues this rank's work (kernels through the engine, collectives
        through torch): the stepper's stream."""
        return self.torch.cuda.stream(self.stream)

    def stats_words(self):
        return self.eng.stats_words()

    # ---- TIE_LIBC (the reference's tie stream): see nemgpu_shard_set_labels in include/nem_mi355x.h
    def set_labels(self, labels):
        self.eng.shard_set_labels([t.data_ptr() for t in labels])

    def reset_now(self):
        self.eng.reset()

    def density(self):
        self.eng.lib.nemgpu_density(self.eng._h)

    def round_draws(self):
        return self.eng.shard_round_draws()

    def grow_draws(self):
        self.eng.shard_grow_draws()

    def book_draws(self, n):
        self.eng.shard_book_draws(n)

    def reset(self):
        pass        # (a run starts with begin(restart=True): the reset is the head launch of its first batch)

    def begin(self, restart=False):
        self.eng.shard_begin(restart)

    def _own_stats_ptr(self, buf):
        return buf.data_ptr() + self.rank * self.stride + stats_offset(self.blk)

    def mstep_partial(self, labels, dst):
        """partial counts of this rank's families under `labels` -> own statistics tail of label array `dst`"""
        self.eng.shard_mstep_partial(labels.data_ptr(), self._own_stats_ptr(dst))

    def counts(self, dst):
        """the same for the labels the last estep_round0 produced (their class masks already exist)"""
        self.eng.shard_counts(self._own_stats_ptr(dst))

    def estep_round0(self, stats_src, beta, sweep_id, old, out):
        """stats_src: all-gathered label array whose tails hold every rank's partial statistics, or None"""
        ptr = stats_src.data_ptr() + stats_offset(self.blk) if stats_src is not None else None
        self.eng.shard_estep_round0(ptr, beta, sweep_id, old.data_ptr(), out.data_ptr())

    def estep_round1(self, beta, sweep_id, old, guess, out):
        self.eng.shard_estep_round1(beta, sweep_id, old.data_ptr(), guess.data_ptr(), out.data_ptr())

    def estep_round1_counts(self, beta, sweep_id, old, guess, out):
        """round 1 and counts(out) in one launch where the shape has a kernel for it"""
        self.eng.shard_estep_round1_counts(beta, sweep_id, old.data_ptr(), guess.data_ptr(), out.data_ptr(),
                                           self._own_stats_ptr(out))

    def finish_iteration(self, beta, is_init, old, q, r):
        self.eng.shard_finish_iteration(beta, is_init, old.data_ptr(), q.data_ptr(), r.data_ptr())

    def round_sync(self, beta, sweep_id, old, guess, out):
        return self.eng.shard_round_sync(beta, sweep_id, old.data_ptr(), guess.data_ptr(), out.data_ptr())

    def end_enqueue(self):
        self.eng.shard_end_enqueue()

    def end(self):
        return self.eng.shard_end()

    # ---- whole-batch capture (kernels AND collectives) into one graph replay
    can_capture = True

    def capture(self, enqueue):
        """Record everything `enqueue()` launches -- this engine's kernels and the RCCL collectives torch issues --
        into a torch.cuda.CUDAGraph on the stepper's stream.  Returns the graph, or None when something in the
        batch cannot be captured here (the caller then stays eager)."""
        torch = self.torch
        self.stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, stream=self.stream):
                enqueue()
        except Exception:
            self.stream.synchronize()
            return None
        return graph

    def set_sweep_number(self, n):
        self.eng.shard_set_sweep_number(n)

    def set_cvtest(self, name):
        from .engine import CVT
        self.eng.cfg.cvtest = CVT[name]
        self.eng._chk(self.eng.lib.nemgpu_configure(self.eng._h, self.eng.cfg))   # also resets the engine state

    def params(self):
        return self.eng.params()


class ShardedNem:
    """EM driver over a stepper + Comm (mirrors NemAlgo, nem_alg.c:1746-1879, for NCEM)."""

    PIPE_DEPTH = 7       # whole iterations enqueued between host synchronisations (the single engine's kPipeDepth: a run of
                         # PPanGGOLiN's, 5-7 iterations, is ONE batch; NEM_DIST_PIPE_DEPTH overrides)

    def __init__(self, stepper, comm, n_total, beta, cvtest="clas", cvthres=1e-8, param_fix=False):
        if cvtest not in ("none", "clas"):
            raise ValueError("the sharded path implements the convergence tests 'none' and 'clas' only (got %r): the "
                             "criterion is an i-ordered float sum over all families" % (cvtest,))
        self.st, self.comm = stepper, comm
        self.n_total = n_total
        self.blk, self.stride = slot_layout(n_total, comm.world, stepper.stats_words())
        self.lo, self.hi, _ = shard_bounds(n_total, comm.world, comm.rank)
        self.beta, self.cvtest, self.cvthres, self.param_fix = float(beta), cvtest, float(cvthres), param_fix
        self.labels = [stepper.alloc(self.stride * comm.world, "uint8") for _ in range(3)]
        # the reference's tie stream (TIE_LIBC): the sites of a sweep are coupled through the stream's position even
        # without neighbours (two rounds always), and a start is completed from the host, sweep by sweep
        self.libc = bool(getattr(stepper, "libc", False))
        if self.libc:
            stepper.set_labels(self.labels)
        import os
        # Whole batches as graphs.  A rank alone (no collective inside a batch) gets them from the LIBRARY
        # (nemgpu_shard_enqueue_batch captures a batch shape into a hipGraph of its own the second time it is enqueued,
        # as the single engine does).  Capturing kernels AND collectives through torch (torch.cuda.CUDAGraph) is opt-in
        # (NEM_DIST_GRAPHS=1): with more ranks captured RCCL collectives could not be rehearsed on the one-GPU
        # development box and a wedged replay would cost the whole job; with one rank, round 3 saw replays of such a
        # graph block in hipGraphLaunch for good in two runs out of four once nothing else was enqueued between them.
        self.use_graphs = os.environ.get("NEM_DIST_GRAPHS", "0") != "0"
        # RCCL straight from the library: a whole batch (kernels + all-gathers) per C call.  On by default over the
        # nccl backend; NEM_DIST_NATIVE=0 keeps every collective in torch.distributed.
        self.native = False
        if (comm.backend == "nccl" and hasattr(stepper, "enable_native")
                and os.environ.get("NEM_DIST_NATIVE", "1") != "0"):
            self.native = bool(stepper.enable_native(comm))
        self._graphs, self._seen = {}, set()
        self.PIPE_DEPTH = max(1, int(os.environ.get("NEM_DIST_PIPE_DEPTH", self.PIPE_DEPTH)))
        self.n_batches, self.n_host_sweeps = 0, 0             # (what a timed region cost besides its launches)
        self.library_graphs = bool(self.native and comm.world == 1)
        self.first_sweep_is_long, self._last_need_rounds = False, False
        self.reset()

    @property
    def eng(self):
        return self.st.eng

    def reset(self):
        self.st.reset()               # (buffer 0 needs no clearing: the blind sweep of a start never reads a partition)
        self.cur = 0
        self.sweep_id = 0
        self.iters, self.converged, self.status, self.emptyk, self.sweep_rounds = 0, False, STATUS_OK, 0, 0

    def _block_view(self, t):
        return t.reshape(self.comm.world, self.stride)[:, :self.blk] & 0x7F      # (bit 7: the site drew, TIE_LIBC)

    @property
    def two_rounds(self):
        return self.beta != 0.0 or self.libc

    # ---- enqueue helpers (no host synchronisation)
    def _stats_buf(self, cur):
        """Which label array's tails hold the statistics of the partition in buffer `cur`: the array gathered
        last in the iteration that produced it (its R buffer with a graph, the partition's own buffer without)."""
        return (cur + 1) % 3 if self.two_rounds else cur

    def _enqueue_init(self):
        L, use_nei, want_stats = self.labels, self.two_rounds, not self.param_fix
        self.st.estep_round0(None, 0.0, 0, L[0], L[1])                    # blind beta = 0 sweep (nem_alg.c:1972-1976)
        self.comm.allgather_blocks_(L[1], self.stride)
        self.st.estep_round0(None, self.beta, 1, L[1], L[2])              # the sweep with the real beta (:1980)
        if want_stats and not use_nei:
            self.st.counts(L[2])
        self.comm.allgather_blocks_(L[2], self.stride)
        if use_nei:
            self._round1(1, L[1], L[2], L[0], want_stats)                 # (+ the statistics of L[2], riding with L[0])
            self.comm.allgather_blocks_(L[0], self.stride)
        self.st.finish_iteration(self.beta, 1, L[1], L[2], L[0])

    def _enqueue_iteration(self, P):
        L, Q, R = self.labels, (P + 1) % 3, (P + 2) % 3
        use_nei, want_stats = self.two_rounds, not self.param_fix         # (parameters fixed: nem_alg.c:1806)
        self.st.estep_round0(L[self._stats_buf(P)] if want_stats else None, self.beta, -1, L[P], L[Q])
        if want_stats and not use_nei:
            self.st.counts(L[Q])
        self.comm.allgather_blocks_(L[Q], self.stride)
        if use_nei:
            self._round1(-1, L[P], L[Q], L[R], want_stats)                # (+ the statistics of L[Q], riding with L[R])
            self.comm.allgather_blocks_(L[R], self.stride)
        self.st.finish_iteration(self.beta, 0, L[P], L[Q], L[R])

    def _round1(self, sweep_id, old, guess, out, want_stats):
        """the verifying round and, beside it, this rank's partial counts of the labels it verifies"""
        if want_stats and hasattr(self.st, "estep_round1_counts"):
            self.st.estep_round1_counts(self.beta, sweep_id, old, guess, out)
            return
        self.st.estep_round1(self.beta, sweep_id, old, guess, out)
        if want_stats:
            self.st.counts(out)

    def _publish_stats(self, cur):
        """Statistics of the partition in buffer `cur`, recomputed and gathered explicitly (after the host had
        to finish a sweep, i.e. when the speculative ones of the pipeline were counted on non-final labels)."""
        if self.param_fix:
            return
        dst = self.labels[self._stats_buf(cur)]
        with self.st.on_stream():
            self.st.mstep_partial(self.labels[cur], dst)
            self.comm.allgather_blocks_(dst, self.stride)

    def _finish_sweep_on_host(self, P, sweep_id, first_round=2, beta=None):
        """A sweep whose rounds the host checks one by one: from round 2 on when the two enqueued rounds of a pipelined
        sweep did not reach the fixed point, from round 0 on for the sweeps of a start under TIE_LIBC (whose draws come
        one sweep after the other in the stream).  TIE_LIBC: a round one of whose draws fell outside the draw table is
        void on every rank (the tables grow, the round runs again); the stream moves on by the ranks' draws of the
        final round."""
        L, Q, R = self.labels, (P + 1) % 3, (P + 2) % 3
        beta = self.beta if beta is None else beta
        r, draws = first_round, 0
        while True:
            guess, out = (P, Q) if r == 0 else ((R, Q) if r % 2 == 0 else (Q, R))
            with self.st.on_stream():
                changed = self.st.round_sync(beta, sweep_id, L[P], L[guess], L[out])
                if self.libc:
                    draws, short = self.st.round_draws()
                    if self.comm.allreduce_max_int(short) != 0:
                        self.st.grow_draws()
                        continue
                self.comm.allgather_blocks_(L[out], self.stride)
                r += 1
                if self.comm.allreduce_max_int(changed) == 0:
                    break
        if self.libc:
            self.st.book_draws(self.comm.allreduce_sum_int(draws))
        self.sweep_rounds += r - first_round if first_round == 0 else r   # out == guess now, so L[Q] holds the result whichever buffer was last
        return Q

    def _host_start(self):
        """ComputePartitionFromPara(Needinit=1) under TIE_LIBC: the blind sweep and the sweep with the real beta, each
        completed (its rounds verified, its draws booked) before the next begins -- in the reference's stream the blind
        sweep's ties come first (nem_alg.c:1972-1980)."""
        self.st.reset_now()
        with self.st.on_stream():
            self.st.density()
        self._finish_sweep_on_host(0, 0, first_round=0, beta=0.0)         # 0 -> 1
        self._finish_sweep_on_host(1, 1, first_round=0)                   # 1 -> 2
        self.cur, self.sweep_id = 2, 2
        self.st.set_sweep_number(2)
        self._publish_stats(2)

    def _enqueue_batch(self, with_init, g, base):
        if self.native:
            self.st.enqueue_batch_native(with_init, g, base, self.beta, not self.param_fix, self.labels)
            return
        if with_init and isinstance(self.st, GpuStepper):
            self.st.begin(restart=True)
        else:
            self.st.begin()
        if with_init:
            self._enqueue_init()
        for j in range(g):
            self._enqueue_iteration((base + j) % 3)
        self.st.end_enqueue()

    def _run_batch(self, with_init, g):
        """Enqueue [the two initial sweeps +] g whole iterations, synchronise once, account for what ran.
        The second time a batch shape is seen it is captured (kernels + collectives) into one graph."""
        if with_init and self.libc:
            self._host_start()
            self._last_need_rounds = False
            return self._run_batch(False, g) if g > 0 else 0
        base = 2 if with_init else self.cur
        s0 = 2 if with_init else self.sweep_id
        key = (with_init, g, base)
        graph = self._graphs.get(key)
        if graph is None and self.use_graphs and getattr(self.st, "can_capture", False) and key in self._seen:
            graph = self.st.capture(lambda: self._enqueue_batch(with_init, g, base))
            # every rank must take the same route: captured only if the capture worked everywhere
            if self.comm.world > 1 and self.comm.allreduce_max_int(0 if graph is not None else 1) != 0:
                graph = None
            if graph is None:
                self.use_graphs = False
            else:
                self._graphs[key] = graph
        self._seen.add(key)
        self.n_batches += 1
        with self.st.on_stream():
            if graph is not None:
                graph.replay()
            else:
                self._enqueue_batch(with_init, g, base)
        res = self.st.end()
        self._last_need_rounds = res["need_rounds"] == 1
        self.sweep_rounds += res["sweep_rounds"]
        if with_init and res["need_rounds"] == 2:
            # the initial beta sweep was not at its fixed point after two rounds: finish it from the host;
            # the iterations enqueued behind it all returned at the stop word
            self.n_host_sweeps += 1
            self._finish_sweep_on_host(1, 1)
            self.cur, self.sweep_id = 2, 2
            self.st.set_sweep_number(2)
            self._publish_stats(2)
            return 0
        self.iters += res["iters"]
        self.cur = (base + res["commits"]) % 3
        self.sweep_id = s0 + res["iters"]
        if res["status"] == STATUS_W_EMPTYCLASS:                          # nem_alg.c:1831-1838
            self.status, self.emptyk = STATUS_W_EMPTYCLASS, res["emptyk"]
        elif res["converged"]:
            self.converged = True
        elif res["need_rounds"] == 1:
            P = self.cur
            self.n_host_sweeps += 1
            new = self._finish_sweep_on_host(P, s0 + res["iters"] - 1)
            self.cur = new
            self._publish_stats(new)
            if self.cvtest == "clas":                                     # HasConverged, nem_alg.c:2075-2089
                with self.st.on_stream():
                    moved = not bool((self._block_view(self.labels[new]) == self._block_view(self.labels[P])).all().item())
                self.converged = (1.0 < self.cvthres) if moved else (0.0 < self.cvthres)
        return res["iters"]

    def init_partition(self):
        """ComputePartitionFromPara(Needinit=1), nem_alg.c:1967-1981 (one batch, one synchronisation)."""
        self._run_batch(True, 0)
        self.cur, self.sweep_id = 2, 2

    def iterate(self, n_iters, with_init=False):
        first, remaining = with_init, int(n_iters)
        while True:
            # The sweep of the first iteration after a start moves many labels and may need a third relaxation round,
            # which this pipeline does not enqueue: the iterations behind it in the batch then all return at the stop
            # word (four launches each, for nothing) and the host finishes the sweep.  A job whose start did that once
            # sends ONE iteration with its next starts (kept across restarts, like the single engine's deep_iters).
            depth = 1 if (first and self.first_sweep_is_long) else self.PIPE_DEPTH
            before = self.iters
            done = self._run_batch(first, min(remaining, depth))
            if first and depth > 1 and done >= 1 and self.iters == before + 1 and self._last_need_rounds:
                self.first_sweep_is_long = True
            first = False
            remaining -= done
            if self.converged or self.status != STATUS_OK or remaining <= 0:
                break
        return dict(iters=self.iters, converged=self.converged, status=self.status, emptyk=self.emptyk,
                    sweep_rounds=self.sweep_rounds)

    def run(self, it_max=100):
        self.reset()
        return self.iterate(it_max, with_init=True)

    def global_labels(self):
        with self.st.on_stream():
            return self._block_view(self.labels[self.cur]).reshape(-1)[:self.n_total].cpu().numpy()   # (bit 7 masked)

    # ---- bench helpers
    def set_cvtest(self, name):
        if name not in ("none", "clas"):
            raise ValueError("the sharded path implements the convergence tests 'none' and 'clas' only (got %r)" % (name,))
        self.cvtest = name
        self.st.set_cvtest(name)
        self._graphs, self._seen = {}, set()      # kernel arguments are baked into captured batches

    def iters_to_converge(self, it_max=100):
        """EM iterations this job needs from its initial parameters (the reference's clas test, 1e-8); at least 1."""
        keep = self.cvtest
        self.set_cvtest("clas")
        res = self.run(it_max)
        self.set_cvtest(keep)
        return max(1, int(res["iters"]))

    def time_collective(self, reps=100):
        """Seconds per all-gather of one label array (an iteration issues two), on this job's own path; collective."""
        import time
        if self.comm.world == 1:
            return 0.0
        if self.native:
            return self.eng.rccl_time_allgather(self.labels[(self.cur + 2) % 3].data_ptr(), reps) * 1e-3
        buf = self.labels[(self.cur + 2) % 3]
        torch = self.st.torch
        with self.st.on_stream():
            for _ in range(3):
                self.comm.allgather_blocks_(buf, self.stride)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                self.comm.allgather_blocks_(buf, self.stride)
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def run_steps(self, count, cycle):
        done = 0
        while done < count:
            m = min(cycle, count - done)
            self.reset()
            self.iterate(m, with_init=True)
            done += m

    @classmethod
    def synthetic(cls, n_total, d, k, beta, rank, world, device, algo="ncem", spectrum="ushape", seed=2, comm=None):
        """One synthetic n_total x d problem (the generators of pangenomenem_amd/synth.py, the same matrix on every
        rank count) sharded in contiguous family blocks: every rank draws the global matrix from the shared seed
        and keeps its rows; the contiguity graph is the global path + chords."""
        from . import synth
        if algo != "ncem":
            raise ValueError("the sharded path is NCEM-only (fuzzy sums are order-dependent, SURVEY.md §8e)")
        lo, hi, _ = shard_bounds(n_total, world, rank)
        blk, stride = slot_layout(n_total, world, k + k * d)
        if spectrum == "ushape":
            x_local, _ = synth.ushaped_pa_matrix(n_total, d, seed, rows=(lo, hi))    # (the shard of the same matrix)
        else:
            x, _ = synth.bernoulli_pa_matrix(n_total, d, seed)
            x_local = np.ascontiguousarray(x[lo:hi])
            del x
        nei = slice_graph(synth.contiguity_graph(n_total, seed), lo, hi, blk, stride)
        prop, center, disp = synth.default_init(d)
        cfg = dict(algo="ncem", beta=beta, disper="sk_", propor="pk", cvtest="none", it_max=100)
        st = GpuStepper(x_local, nei, k, n_total, world, rank, prop, center, disp, device, cfg)
        return cls(st, comm or Comm(), n_total, beta, cvtest="none")


# ======================================================================================================
# Fuzzy NEM (algo = "nem") on several GPUs: E-step sharded over families, M-step over organisms
# ======================================================================================================
def organism_bounds(d, world, rank):
    """Contiguous blocks of ceil(d/world) organisms: the chains a rank sums in the M-step."""
    blk = (d + world - 1) // world
    lo = min(rank * blk, d)
    return lo, min(lo + blk, d), blk


class FuzzyGpuStepper:
    """One rank's compute of the organism/family-sharded fuzzy EM: a ROW engine (its families x all organisms: E1 and
    the relaxation rounds of the E-step sweep) and a COLUMN engine (all families x its organisms: the M-step's
    i-ordered chains), both behind the nemgpu_shard_fuzzy_* C ABI.  Membership matrices are torch tensors
    float32[world * blk, K] owned by the driver."""

    def __init__(self, x_rows, x_cols, nei_rows, k, n_total, d, world, rank, prop, center, disp, device, cfg):
        import ctypes as C
        import torch
        from .engine import NemEngine
        self.torch, self.C = torch, C
        self.device = torch.device("cuda", device)
        self.k, self.d, self.n_total, self.world, self.rank = k, d, n_total, world, rank
        lo, hi, blk = shard_bounds(n_total, world, rank)
        dlo, dhi, dblk = organism_bounds(d, world, rank)
        # (decided from global quantities, so that EVERY rank raises -- a rank that went on alone would wait for the others
        #  in its first collective for good)
        if shard_bounds(n_total, world, world - 1)[0] >= n_total or organism_bounds(d, world, world - 1)[0] >= d:
            raise ValueError("the last of %d ranks would own no family or no organism (%d x %d): use fewer ranks"
                             % (world, n_total, d))
        assert x_rows.shape == (hi - lo, d) and x_cols.shape == (n_total, dhi - dlo)
        self.blk, self.dblk, self.n_pad, self.dlo, self.dhi = blk, dblk, world * blk, dlo, dhi
        self.stream = torch.cuda.Stream(self.device)
        cfg = dict(cfg, algo="nem")
        center = np.asarray(center, np.float32).reshape(k, d)
        disp = np.asarray(disp, np.float32).reshape(k, d)
        self.row = NemEngine(self.n_pad, d, k, device=device, stream=self.stream.cuda_stream, site_lo=lo, site_hi=hi)
        self.row.set_matrix(x_rows)
        self.row.set_graph(nei_rows)
        self.row.set_params(prop, center, disp)
        self.row.configure(**cfg)
        self.row._chk(self.row.lib.nemgpu_shard_fuzzy_layout(self.row._h, n_total))
        self.col = NemEngine(self.n_pad, dhi - dlo, k, device=device, stream=self.stream.cuda_stream, site_lo=0, site_hi=n_total)
        self.col.set_matrix(x_cols)
        self.col.set_graph(None)
        self.col.set_params(prop, np.ascontiguousarray(center[:, dlo:dhi]), np.ascontiguousarray(disp[:, dlo:dhi]))
        self.col.configure(**cfg)
        self._tmp = [torch.zeros(k * (dhi - dlo), dtype=torch.float32, device=self.device) for _ in range(2)]

    def alloc(self, *shape):
        return self.torch.zeros(*shape, dtype=self.torch.float32, device=self.device)

    def _p(self, t):
        return self.C.c_void_p(t.data_ptr())

    def reset(self):
        self.row.reset()

    def density(self):
        with self.torch.cuda.stream(self.stream):
            self.row._chk(self.row.lib.nemgpu_density(self.row._h))

    def round(self, beta, sweep_id, r, old, guess, out):
        C = self.C
        ch, nz, fz = C.c_int(0), C.c_int(0), C.c_int(-1)
        with self.torch.cuda.stream(self.stream):
            self.row._chk(self.row.lib.nemgpu_shard_fuzzy_round(self.row._h, C.c_float(beta), int(sweep_id), int(r), self._p(old),
                                                                self._p(guess), self._p(out), C.byref(ch), C.byref(nz), C.byref(fz)))
        return ch.value, nz.value, fz.value

    def mstep_cols(self, c, stats_block):
        """stats_block: this rank's float32[K + 2 K dblk] block {N_k | centres [K, dblk] | inertia [K, dblk]}"""
        k, dl, db = self.k, self.dhi - self.dlo, self.dblk
        nb = stats_block[:k]
        with self.torch.cuda.stream(self.stream):
            self.col._chk(self.col.lib.nemgpu_shard_fuzzy_mstep_cols(self.col._h, self._p(c), self._p(nb), self._p(self._tmp[0]),
                                                                     self._p(self._tmp[1])))
            stats_block[k:k + k * db].view(k, db)[:, :dl] = self._tmp[0].view(k, dl)
            stats_block[k + k * db:].view(k, db)[:, :dl] = self._tmp[1].view(k, dl)
        self.stream.synchronize()

    def finish(self, nb, center_full, iner_full):
        ek = self.C.c_int(0)
        with self.torch.cuda.stream(self.stream):
            rc = self.row.lib.nemgpu_shard_fuzzy_finish(self.row._h, self._p(nb), self._p(center_full), self._p(iner_full), self.C.byref(ek))
        self.row._chk(rc, allow=(STATUS_OK, STATUS_W_EMPTYCLASS))
        return ek.value

    def moved(self, new, old):
        mv = self.C.c_int(0)
        with self.torch.cuda.stream(self.stream):
            self.row._chk(self.row.lib.nemgpu_shard_fuzzy_moved(self.row._h, self._p(new), self._p(old), self.C.byref(mv)))
        return mv.value

    def on_stream(self):
        """torch work (collectives, the reshuffling of gathered blocks) goes on the engines' stream: stream order is
        what orders it against the library's launches"""
        return self.torch.cuda.stream(self.stream)

    def sync(self):
        self.stream.synchronize()

    def params(self):
        return self.row.params()


class ShardedFuzzyNem:
    """NemAlgo (nem_alg.c:1746-1879) for algo = "nem" on `world` ranks, bit-identical to one engine (and to the
    reference): SURVEY.md 8e's exact alternative to an all-reduce of float sums.
      * E1 and the E-step sweep run on the rank's FAMILIES; the sweep is the relaxation of pangenomenem_amd's single
        engine (round r evaluates every site against the memberships of round r-1; a round that changes nothing
        anywhere is the sequential sweep's result) with ONE all-gather of the membership rows per round and one
        max-reduction of the "changed" flags;
      * the M-step's i-ordered chains (class sizes, the zeros' weights of the medians, the inertia sums) run on the
        rank's ORGANISMS over the whole gathered membership matrix, in family order; one all-gather carries every
        rank's {N_k, centres, inertia} block; every rank then derives dispersions, proportions and tables for all
        organisms (InerToDisp's d-ordered sums need them all).
    Host-driven (one step per library call): an exactness path, not a fast one."""

    def __init__(self, stepper, comm, n_total, d, k, beta, cvtest="clas", cvthres=1e-8):
        self.st, self.comm = stepper, comm
        self.n_total, self.d, self.k, self.beta = n_total, d, k, beta
        self.cvtest, self.cvthres = cvtest, cvthres
        self.world, self.rank = comm.world, comm.rank
        self.blk = (n_total + self.world - 1) // self.world
        self.dblk = (d + self.world - 1) // self.world
        self.n_pad = self.world * self.blk
        self.c = [stepper.alloc(self.n_pad, k) for _ in range(3)]
        self.sw = k + 2 * k * self.dblk
        self.stats = stepper.alloc(self.world * self.sw)
        self.iters, self.converged, self.status, self.emptyk = 0, False, STATUS_OK, 0
        self.sweep_rounds, self.zero_density, self.first_zero = 0, 0, -1
        self.cur = 0

    def _gather_rows(self, t):
        with self.st.on_stream():
            flat = t.view(self.st.torch.uint8).reshape(-1)
            self.comm.allgather_blocks_(flat, self.blk * self.k * 4)
        self.st.sync()

    def _sweep(self, P, beta, sweep_id):
        """One Gauss-Seidel sweep (ComputePartitionNEM, nem_alg.c:2330-2405) from buffer P by relaxation rounds."""
        Q, R = (P + 1) % 3, (P + 2) % 3
        r = 0
        verify = beta != 0.0
        while True:
            guess, out = (P, Q) if r == 0 else ((R, Q) if r % 2 == 0 else (Q, R))
            changed, nz, fz = self.st.round(beta, sweep_id, r, self.c[P], self.c[guess], self.c[out])
            self._gather_rows(self.c[out])
            r += 1
            if not verify or self.comm.allreduce_max_int(changed) == 0:
                break
        self.sweep_rounds += r
        nz_all = self.comm.allreduce_sum_int(nz)
        if nz_all > 0:
            self.zero_density += nz_all
            first = -self.comm.allreduce_max_int(-fz if fz >= 0 else -(1 << 30))
            if self.first_zero < 0 and first < (1 << 30):
                self.first_zero = first
        return out

    def _mstep(self, cur):
        k, db, sw = self.k, self.dblk, self.sw
        mine = self.stats[self.rank * sw:(self.rank + 1) * sw]
        self.st.mstep_cols(self.c[cur], mine)
        with self.st.on_stream():
            flat = self.stats.view(self.st.torch.uint8).reshape(-1)
            self.comm.allgather_blocks_(flat, sw * 4)
            blocks = self.stats.view(self.world, sw)
            nb = blocks[self.rank, :k].contiguous()             # (every rank summed the same memberships in the same order)
            cen = blocks[:, k:k + k * db].reshape(self.world, k, db).permute(1, 0, 2).reshape(k, self.world * db)[:, :self.d].contiguous()
            ine = blocks[:, k + k * db:].reshape(self.world, k, db).permute(1, 0, 2).reshape(k, self.world * db)[:, :self.d].contiguous()
        self.st.sync()
        return self.st.finish(nb, cen, ine)

    def run(self, it_max=100):
        self.st.reset()
        with self.st.on_stream():
            for t in self.c:
                t.zero_()
        self.st.sync()
        self.iters, self.converged, self.status, self.emptyk = 0, False, STATUS_OK, 0
        self.sweep_rounds, self.zero_density, self.first_zero = 0, 0, -1
        # ComputePartitionFromPara(Needinit = 1): densities of the initial parameters, blind sweep, sweep
        self.st.density()
        cur = self._sweep(0, 0.0, 0)
        cur = self._sweep(cur, self.beta, 1)
        sweep_id = 2
        for _ in range(it_max):
            ek = self._mstep(cur)
            if ek != 0:                                          # nem_alg.c:1831-1838
                self.status, self.emptyk = STATUS_W_EMPTYCLASS, ek
                self.iters += 1
                break
            new = self._sweep(cur, self.beta, sweep_id)
            sweep_id += 1
            self.iters += 1
            old, cur = cur, new
            if self.cvtest == "clas":
                moved = self.comm.allreduce_max_int(self.st.moved(self.c[cur], self.c[old]))
                if not moved:
                    self.converged = True
                    break
        if self.iters == 0:                                      # nem_alg.c:1845-1851: parameters of the partition, their densities
            self._mstep(cur)
        self.cur = cur
        return dict(status=self.status, iters=self.iters, converged=self.converged, emptyk=self.emptyk,
                    sweep_rounds=self.sweep_rounds, zero_density=self.zero_density, first_zero=self.first_zero)

    def memberships(self):
        """float32[n_total, K]: the final partition (every rank holds all of it)."""
        t = self.c[self.cur][:self.n_total]
        return t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)

    def params(self):
        return self.st.params()

    @classmethod
    def from_problem(cls, x, nei, k, prop, center, disp, beta, rank, world, device, comm=None, disper="sk_", propor="pk",
                     cvtest="clas", cvthres=1e-8):
        """Shard a whole problem held by every rank (tests, small runs): rows [lo, hi) for the row engine, the
        organism slice of all rows for the column engine."""
        n, d = x.shape
        lo, hi, _ = shard_bounds(n, world, rank)
        dlo, dhi, _ = organism_bounds(d, world, rank)
        nei_rows = slice_graph(nei, lo, hi) if nei is not None else None
        cfg = dict(beta=beta, disper=disper, propor=propor, cvtest=cvtest, cvthres=cvthres, it_max=100)
        st = FuzzyGpuStepper(np.ascontiguousarray(x[lo:hi]), np.ascontiguousarray(x[:, dlo:dhi]), nei_rows, k, n, d, world, rank,
                             prop, center, disp, device, cfg)
        return cls(st, comm or Comm(), n, d, k, beta, cvtest=cvtest, cvthres=cvthres)
