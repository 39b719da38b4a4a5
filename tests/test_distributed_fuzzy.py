"""Fuzzy NEM (algo = "nem") over several ranks -- E-step sharded over families, M-step over organisms (SURVEY.md 8e's
exact alternative to an all-reduce of float sums) -- rehearsed on the CPU: the driver of pangenomenem_amd/distributed.py
(ShardedFuzzyNem) over gloo with the oracle as the stepper.  Every rank must end with the single-process oracle's
answer bit for bit: memberships, centres, dispersions, proportions, class sizes, iteration count."""
import os
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleFuzzyStepper:
    """CPU stand-in for FuzzyGpuStepper: same interface, numbers from the oracle (density, relaxation rounds, the
    M-step of an organism slice) and, for InerToDisp on the gathered statistics, the reference's loops restated in
    float32 (nem_mod.c:1021-1077 sk_, 1135-1174 skd; MISSING_IGNORE, no missing data: NbObs_KD = NbObs_K)."""

    def __init__(self, oracle, x, nei, k, world, rank, prop, center, disp, disper, propor, cvthres):
        import torch
        from pangenomenem_amd.distributed import organism_bounds, shard_bounds, slice_graph
        self.torch, self.o = torch, oracle
        self.n, self.d = x.shape
        self.k, self.world, self.rank = k, world, rank
        self.lo, self.hi, self.blk = shard_bounds(self.n, world, rank)
        self.dlo, self.dhi, self.dblk = organism_bounds(self.d, world, rank)
        self.x_rows = np.ascontiguousarray(x[self.lo:self.hi])
        self.x_cols = np.ascontiguousarray(x[:, self.dlo:self.dhi])
        self.nei_rows = slice_graph(nei, self.lo, self.hi)
        self.disper, self.propor, self.cvthres = disper, propor, cvthres
        self.p0 = (np.array(prop, np.float32), np.array(center, np.float32).reshape(k, self.d), np.array(disp, np.float32).reshape(k, self.d))
        self.reset()

    def alloc(self, *shape):
        return self.torch.zeros(*shape, dtype=self.torch.float32)

    def reset(self):
        self.prop, self.center, self.disp = (a.copy() for a in self.p0)
        self.nbobs_k = np.zeros(self.k, np.float32)

    def sync(self):
        pass

    def on_stream(self):
        import contextlib
        return contextlib.nullcontext()

    def density(self):
        self.pk, self.lp, _ = self.o.density(self.x_rows, self.prop, self.center, self.disp)

    def round(self, beta, sweep_id, r, old, guess, out):
        changed = self.o.relax_round(self.lo, self.hi, self.nei_rows, beta, self.pk, False, old.numpy(), guess.numpy(), out.numpy())
        return (1 if changed else 0), 0, -1

    def mstep_cols(self, c, stats_block):
        k, dl, db = self.k, self.dhi - self.dlo, self.dblk
        m = self.o.mstep(self.x_cols, c.numpy()[:self.n], "skd", self.propor, self.prop, self.center[:, self.dlo:self.dhi],
                         self.disp[:, self.dlo:self.dhi])
        blk = stats_block.numpy()
        blk[:k] = m["nbobs_k"]
        blk[k:k + k * db].reshape(k, db)[:, :dl] = m["center"]
        blk[k + k * db:].reshape(k, db)[:, :dl] = m["iner"]

    def finish(self, nb, cen, ine):
        nb, cen, ine = nb.numpy().copy(), cen.numpy().copy(), ine.numpy().copy()
        k, d = self.k, self.d
        ek = 0
        for h in range(k):
            if not (float(nb[h]) > 1e-20):
                ek = h + 1
        self.nbobs_k = nb.astype(np.float32)
        live = [float(nb[h]) > 1e-20 for h in range(k)]
        for h in range(k):
            if live[h]:                                           # (an empty class keeps its centre, nem_mod.c:1405)
                self.center[h] = cen[h]
        if self.disper == "skd":
            for h in range(k):
                if float(nb[h]) > 1e-20:
                    self.disp[h] = (ine[h].astype(np.float32) / np.float32(nb[h])).astype(np.float32)
        elif self.disper == "sk_":
            for h in range(k):
                if nb[h] > 0:
                    sn, si = np.float32(0), np.float32(0)
                    for j in range(d):
                        sn = np.float32(sn + nb[h])
                        si = np.float32(si + ine[h, j])
                    self.disp[h, :] = np.float32(si / sn)
        else:
            raise NotImplementedError(self.disper)
        if self.propor == "pk":
            self.prop = (nb / np.float32(self.n)).astype(np.float32)
        else:
            self.prop = np.full(k, np.float32(1.0 / k), np.float32)
        if ek == 0:
            self.density()
        return ek

    def moved(self, new, old):
        # HasConverged's CVTEST_CLAS, nem_alg.c:2077-2088: max |c - c_old| < threshold, in float
        a, b = new.numpy()[self.lo:self.hi], old.numpy()[self.lo:self.hi]
        return 0 if np.float32(np.max(np.abs(a - b))) < np.float32(self.cvthres) else 1

    def params(self):
        return dict(prop=self.prop, center=self.center, disp=self.disp, nbobs_k=self.nbobs_k)


def _worker(rank, world, initfile, n, d, k, beta, disper, seed, it_max, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle.pyoracle import Oracle
    from pangenomenem_amd import synth
    from pangenomenem_amd.distributed import Comm, ShardedFuzzyNem
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        x, _ = synth.ushaped_pa_matrix(n, d, seed)
        nei = synth.contiguity_graph(n, seed)
        prop, center, disp = synth.default_init(d)
        st = OracleFuzzyStepper(Oracle(), x, nei, k, world, rank, prop, center, disp, disper, "pk", 1e-8)
        job = ShardedFuzzyNem(st, Comm(), n, d, k, beta, cvtest="clas", cvthres=1e-8)
        res = job.run(it_max)
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), c=job.memberships(), iters=res["iters"], converged=res["converged"],
                 status=res["status"], emptyk=res["emptyk"], rounds=res["sweep_rounds"], **st.params())
    finally:
        dist.destroy_process_group()


def _run(world, n, d, k, beta, disper, seed, it_max=12):
    import torch.multiprocessing as mp
    outdir = tempfile.mkdtemp(prefix="nemfz_")
    mp.spawn(_worker, args=(world, os.path.join(outdir, "rdv"), n, d, k, beta, disper, seed, it_max, outdir), nprocs=world, join=True)
    return [np.load(os.path.join(outdir, "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,n,d,beta,disper,it_max", [(2, 700, 15, 0.5, "sk_", 12), (3, 500, 20, 1.0, "skd", 12), (2, 401, 9, 0.0, "sk_", 12),
                                                          (3, 64, 7, 0.5, "skd", 12), (2, 300, 11, 0.5, "sk_", 0)])
def test_sharded_fuzzy_em_equals_the_single_process_oracle(oracle, world, n, d, beta, disper, it_max):
    from pangenomenem_amd import synth
    outs = _run(world, n, d, 3, beta, disper, 4, it_max)
    x, _ = synth.ushaped_pa_matrix(n, d, 4)
    prop, center, disp = synth.default_init(d)
    want = oracle.run(x, synth.contiguity_graph(n, 4), 3, prop, center, disp, algo="nem", beta=beta, disper=disper, it_max=it_max)
    for o in outs:
        assert int(o["status"]) == want["status"] and int(o["iters"]) == want["iters"] and bool(o["converged"]) == want["converged"]
        assert np.array_equal(o["c"].view(np.uint32), want["c"].view(np.uint32))              # memberships: bit for bit
        assert np.array_equal(o["center"], want["center"])
        assert np.array_equal(o["disp"].view(np.uint32), want["disp"].view(np.uint32))
        assert np.array_equal(o["prop"].view(np.uint32), want["prop"].view(np.uint32))
        assert np.array_equal(o["nbobs_k"].view(np.uint32), want["nbobs_k"].view(np.uint32))


def test_organism_bounds_cover_every_organism_once():
    from pangenomenem_amd.distributed import organism_bounds
    for d in (1, 5, 15, 500, 1001):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi, blk = organism_bounds(d, world, r)
                assert 0 <= lo <= hi <= d and hi - lo <= blk
                seen.extend(range(lo, hi))
            assert seen == list(range(d))
