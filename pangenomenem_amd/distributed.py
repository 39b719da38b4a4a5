"""Multi-GPU host driver: one process per GPU, families sharded in contiguous blocks.

The reference has no distributed path (its only parallelism is multiprocessing over independent
organism chunks, ppanggolin/ppanggolin.py:1039-1095); this module shards ONE NEM problem the way
SURVEY.md §8(e) lays out, for NCEM (hard labels), where every M-step sum is an integer count and
therefore exact under any reduction order:

  * M-step: each rank popcounts its shard -> int32 statistics {N_k, S1[k][d]} ->
    ONE all-reduce(sum) over RCCL/xGMI -> every rank derives mu, epsilon, pi locally.
  * E-step density (E1): embarrassingly parallel over the shard.
  * E-step sweep (E2): the Gauss-Seidel sweep is solved by relaxation rounds (see k_sweep in
    csrc/nem_kernels.hip); a round only needs the other shards' labels of the previous round, so
    each round ends with ONE all-gather of the uint8 label shards (n_total bytes) plus a 4-byte
    all-reduce(max) of the "changed" flag.  The fixed point is the same global sequential sweep
    the reference computes, so labels are bit-identical to a single-GPU run.

torch is plumbing here: device memory for the collective buffers, streams, torch.distributed.
All arithmetic happens in the HIP kernels behind the C ABI (``nemgpu_ext_*``).  The driver is
written against a small *stepper* interface so that the protocol can be rehearsed on CPU with the
gloo backend (tests/test_distributed.py plugs the CPU oracle in as the stepper).
"""
import numpy as np

STATUS_OK, STATUS_W_EMPTYCLASS = 0, 2


def shard_bounds(n_total, world, rank):
    """Contiguous blocks of ceil(n/world) families (keeps most contiguity edges local)."""
    blk = (n_total + world - 1) // world
    lo = min(rank * blk, n_total)
    hi = min(lo + blk, n_total)
    return lo, hi, blk


def slice_graph(nei, lo, hi):
    """Rows [lo, hi) of a global CSR graph; neighbour indices stay global."""
    ptr, idx, w = nei
    b, e = int(ptr[lo]), int(ptr[hi])
    return (np.asarray(ptr[lo:hi + 1], np.int64) - b).astype(np.int32), np.ascontiguousarray(idx[b:e], np.int32), \
        np.ascontiguousarray(w[b:e], np.float32)


class Comm:
    """The three collectives of the sharded EM.  With the nccl (= RCCL) backend they run directly on
    device tensors; with gloo, device tensors are staged through host memory (rehearsal only)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)

    def _staged(self, t):
        return self.backend == "gloo" and t.is_cuda

    def allreduce_sum_(self, t):
        if self._staged(t):
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def allreduce_max_(self, t):
        if self._staged(t):
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)

    def allgather_blocks_(self, buf, blk):
        """In-place all-gather: rank r owns buf[r*blk:(r+1)*blk]; afterwards every rank holds all blocks."""
        mine = buf[self.rank * blk:(self.rank + 1) * blk]
        if self.backend == "gloo":
            h = mine.cpu() if mine.is_cuda else mine.clone()
            parts = [self.torch.empty_like(h) for _ in range(self.world)]
            self.dist.all_gather(parts, h, group=self.group)
            buf.copy_(self.torch.cat(parts).to(buf.device))
        else:
            self.dist.all_gather_into_tensor(buf, mine, group=self.group)


class GpuStepper:
    """The local compute of one rank: a NemEngine shard driven through the nemgpu_ext_* C ABI."""

    def __init__(self, x_local, nei_local, k, n_total, lo, hi, prop, center, disp, device, cfg):
        import torch
        from .engine import NemEngine
        self.torch = torch
        self.device = torch.device("cuda", device)
        d = x_local.shape[1]
        self.eng = NemEngine(n_total, d, k, device=device, stream=torch.cuda.current_stream(self.device).cuda_stream,
                             site_lo=lo, site_hi=hi)
        self.eng.set_matrix(x_local)
        self.eng.set_graph(nei_local)
        self.eng.set_params(prop, center, disp)
        self.eng.configure(**cfg)

    def alloc(self, n, dtype):
        return self.torch.zeros(n, dtype=getattr(self.torch, dtype), device=self.device)

    def stats_words(self):
        return self.eng.stats_words()

    def reset(self):
        self.eng.reset()

    def density(self):
        self.eng.ext_density()

    def sweep_round(self, beta, sweep_id, old, guess, out, flags):
        self.eng.ext_sweep_round(beta, sweep_id, old.data_ptr(), guess.data_ptr(), out.data_ptr(), flags.data_ptr())

    def mstep_partial(self, labels, stats):
        self.eng.ext_mstep_partial(labels.data_ptr(), stats.data_ptr())

    def mstep_finalize(self, stats):
        self.eng.ext_mstep_finalize(stats.data_ptr())

    def emptyk(self):
        return self.eng.ext_emptyk()

    def params(self):
        return self.eng.params()


class ShardedNem:
    """EM driver over a stepper + Comm (mirrors NemAlgo, nem_alg.c:1746-1879, for NCEM)."""

    def __init__(self, stepper, comm, n_total, beta, cvtest="clas", cvthres=1e-8, param_fix=False):
        self.st, self.comm = stepper, comm
        self.n_total = n_total
        self.lo, self.hi, self.blk = shard_bounds(n_total, comm.world, comm.rank)
        self.beta, self.cvtest, self.cvthres, self.param_fix = float(beta), cvtest, float(cvthres), param_fix
        npad = self.blk * comm.world
        self.labels = [stepper.alloc(npad, "uint8") for _ in range(3)]
        self.stats = stepper.alloc(stepper.stats_words(), "int32")
        self.flags = stepper.alloc(4, "int32")
        self.reset()

    @property
    def eng(self):
        return self.st.eng

    def reset(self):
        self.st.reset()
        self.cur = 0
        self.sweep_id = 0
        self.iters, self.converged, self.status, self.emptyk, self.sweep_rounds = 0, False, STATUS_OK, 0, 0

    # one Gauss-Seidel sweep == relaxation rounds until no label differs from its guess on ANY rank
    def _sweep(self, beta, use_nei):
        P = self.cur
        Q, R = (P + 1) % 3, (P + 2) % 3
        sid = self.sweep_id
        self.sweep_id += 1
        r = 0
        while True:
            guess = P if r == 0 else (Q if (r - 1) % 2 == 0 else R)
            out = Q if r % 2 == 0 else R
            self.flags.zero_()
            self.st.sweep_round(beta if use_nei else 0.0, sid, self.labels[P], self.labels[guess], self.labels[out],
                                self.flags)
            self.comm.allgather_blocks_(self.labels[out], self.blk)
            r += 1
            if not use_nei:
                break
            self.comm.allreduce_max_(self.flags)
            if int(self.flags[0].item()) == 0:
                break
        self.sweep_rounds += r
        return out

    def init_partition(self):
        """ComputePartitionFromPara(Needinit=1), nem_alg.c:1967-1981."""
        self.st.density()
        self.cur = self._sweep(0.0, False)
        self.cur = self._sweep(self.beta, self.beta != 0.0)

    def iterate(self, n_iters):
        for _ in range(n_iters):
            if self.converged or self.status != STATUS_OK:
                break
            old = self.cur
            if not self.param_fix:
                self.st.mstep_partial(self.labels[old], self.stats)
                self.comm.allreduce_sum_(self.stats)
                self.st.mstep_finalize(self.stats)
            self.st.density()
            new = self._sweep(self.beta, self.beta != 0.0)
            self.iters += 1
            ek = 0 if self.param_fix else self.st.emptyk()
            if ek:
                self.status, self.emptyk = STATUS_W_EMPTYCLASS, ek       # nem_alg.c:1831-1838
                break
            self.cur = new
            if self.cvtest == "clas":                                    # HasConverged, nem_alg.c:2075-2089
                moved = not bool((self.labels[new][:self.n_total] == self.labels[old][:self.n_total]).all().item())
                self.converged = (1.0 < self.cvthres) if moved else (0.0 < self.cvthres)
        return dict(iters=self.iters, converged=self.converged, status=self.status, emptyk=self.emptyk,
                    sweep_rounds=self.sweep_rounds)

    def run(self, it_max=100):
        self.reset()
        self.init_partition()
        return self.iterate(it_max)

    def global_labels(self):
        return self.labels[self.cur][:self.n_total].cpu().numpy()

    # ---- bench helpers
    def iters_to_converge(self, it_max=100):
        keep = self.cvtest
        self.cvtest = "clas"
        res = self.run(it_max)
        self.cvtest = keep
        return max(5, int(res["iters"]))

    def run_steps(self, count, cycle):
        keep = self.cvtest
        self.cvtest = "none"
        done = 0
        while done < count:
            m = min(cycle, count - done)
            self.reset()
            self.init_partition()
            self.iterate(m)
            done += m
        self.cvtest = keep

    @classmethod
    def synthetic(cls, n_loc, d, k, beta, rank, world, local_rank, algo="ncem"):
        """Weak-scaling job: every rank owns an n_loc x d shard of BASELINE configs[1]'s generator; the
        contiguity graph is the global path + chords over all n_loc*world families."""
        from . import synth
        if algo != "ncem":
            raise ValueError("the sharded path is NCEM-only (fuzzy sums are order-dependent, SURVEY.md §8e)")
        n_total = n_loc * world
        lo, hi, _ = shard_bounds(n_total, world, rank)
        x_local, _ = synth.bernoulli_pa_matrix(hi - lo, d, 2 + 1000 * rank)
        nei = slice_graph(synth.contiguity_graph(n_total, 2), lo, hi)
        prop, center, disp = synth.default_init(d)
        cfg = dict(algo="ncem", beta=beta, disper="sk_", propor="pk", cvtest="none", it_max=100)
        st = GpuStepper(x_local, nei, k, n_total, lo, hi, prop, center, disp, local_rank, cfg)
        return cls(st, Comm(), n_total, beta, cvtest="none")
