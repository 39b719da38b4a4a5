"""bench.py's first-run safety for the multi-rank modes (CPU): a phase that never ends is ended by the job's watchdog,
rank 0 still prints ONE JSON line -- the record it had, with the reason -- and the process leaves with status 0; and the
brackets of a timed block (barrier + maximum over ranks, sums) over gloo at world 2."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_watchdog_prints_the_record_it_has():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--watchdog-selftest"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 0
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["value"] == 1.0 and rec["fallback_reason"] == "selftest" and "deadline" in rec["error"]


def _ranks_worker(rank, world, initfile, outdir):
    sys.path.insert(0, ROOT)
    import argparse
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        class _Cuda:                                       # (no GPU here: the device synchronisations are no-ops)
            @staticmethod
            def synchronize():
                return None
        r = bench.Ranks(argparse.Namespace(backend="gloo"), rank, world, 0)
        r.torch = type("T", (), {"cuda": _Cuda, "tensor": staticmethod(r.torch.tensor), "float64": r.torch.float64})
        r.sync(None)
        mx = r.sync(1.0 + rank)                            # maximum over ranks of a block's seconds
        sm = r.sum(10.0 * (rank + 1))
        np.save(os.path.join(outdir, "r%d.npy" % rank), np.array([mx, sm]))
    finally:
        dist.destroy_process_group()


def test_block_brackets_over_gloo():
    import torch.multiprocessing as mp
    outdir = tempfile.mkdtemp(prefix="nembg_")
    mp.spawn(_ranks_worker, args=(2, os.path.join(outdir, "rdv"), outdir), nprocs=2, join=True)
    for r in range(2):
        mx, sm = np.load(os.path.join(outdir, "r%d.npy" % r))
        assert mx == 2.0 and sm == 30.0
