"""The chunks of PPanGGOLiN's voting loop, formed on the device (the reference's real use).

``partition()`` solves a pangenome of more than 500 organisms as many NEM problems, each on a random sample of the
organisms (ppanggolin/ppanggolin.py:1045-1086: ``orgs = sample(organisms, chunck_size)``), and writes every sample's
input files from ONE graph (``__write_nem_input_files``, ppanggolin.py:821-930):

  * the matrix columns are the sampled organisms, in sample order (:850);
  * a family with no sampled organism is dropped, the others are numbered in the graph's order (:849-852);
  * an edge's weight is the number of sampled organisms that carry the adjacency (``coverage``, :866-878), an edge none
    of them carries is dropped; a family's neighbours keep the order the master lists them in.

``Master`` puts that one pangenome on the device (``nemgpu_master_create``); ``solve_chunks`` solves any number of
samples in ONE library call (``nemgpu_solve_chunks``): the device forms every sample's problem straight into its engine's
buffers, the lock-step pipeline of ``batch.solve_many`` runs them.  ``form_chunk_host`` is the same formation in numpy --
what the tests hold the device against (through ``batch.solve_many``) and what documents the index maps.
"""
import ctypes as C

import numpy as np

from .engine import ALGO, CVT, DISP, PROP, STATUS_OK, TIE, Config, NemGpuError, Result, load_library


class Chunk(C.Structure):
    """nemgpu_chunk (include/nem_mi355x.h)"""
    _fields_ = [("organisms", C.c_void_p), ("dc", C.c_int), ("n", C.c_int), ("nnz", C.c_int),
                ("keep", C.c_void_p), ("labels", C.c_void_p),
                ("out_prop", C.c_void_p), ("out_center", C.c_void_p), ("out_disp", C.c_void_p), ("out_nbobs_k", C.c_void_p),
                ("result", Result), ("rc", C.c_int)]


def pack_rows(x):
    """uint8 [n][d] of 0/1 -> uint32 bit rows [n][ceil(d/32)] (bit o of a row = column o)"""
    x = np.ascontiguousarray(x, np.uint8)
    n, d = x.shape
    wf = (d + 31) // 32
    rows = np.zeros((n, wf * 4), np.uint8)
    bits = np.packbits(x, axis=1, bitorder="little")
    rows[:, :bits.shape[1]] = bits
    return rows.view(np.uint32)


def form_chunk_host(x, ptr, idx, edge_bits, organisms):
    """One sample's NEM problem the way __write_nem_input_files makes it (ppanggolin.py:821-930), in numpy.
    x: uint8 [n][d]; (ptr, idx): the master graph in CSR; edge_bits: uint32 [nnz][ceil(d/32)], the organisms that carry
    each directed edge; organisms: the sample (column order).  Returns (x_chunk uint8 [n_c][d_c], (ptr_c, idx_c, w_c),
    families int64 [n_c]: the master index of the chunk's family j)."""
    x = np.asarray(x, np.uint8)
    organisms = np.asarray(organisms, np.int64)
    n, d = x.shape
    sub = x[:, organisms]
    keep = sub.any(axis=1)                                    # `not organisms.isdisjoint(node_organisms)`, :849
    families = np.flatnonzero(keep)
    renum = np.full(n, -1, np.int64)
    renum[families] = np.arange(len(families))                # index_fam, :851 (1-based there)
    mask = np.zeros(edge_bits.shape[1] * 32, np.uint8)
    mask[organisms] = 1
    mask_words = np.packbits(mask, bitorder="little").view(np.uint32)
    nnz = len(idx)
    cov = np.zeros(nnz, np.int64)
    if nnz:
        anded = np.bitwise_and(np.asarray(edge_bits, np.uint32), mask_words[None, :])
        cov = np.unpackbits(anded.view(np.uint8), axis=1).sum(axis=1).astype(np.int64)     # coverage, :866-876
    src = np.repeat(np.arange(n), np.diff(ptr))
    ok = (cov > 0) & keep[src] & (renum[np.asarray(idx, np.int64)] >= 0) if nnz else np.zeros(0, bool)   # `if coverage == 0: continue`, :877
    deg = np.bincount(renum[src[ok]], minlength=len(families)) if nnz else np.zeros(len(families), np.int64)
    ptr_c = np.zeros(len(families) + 1, np.int32)
    ptr_c[1:] = np.cumsum(deg)
    idx_c = renum[np.asarray(idx, np.int64)[ok]].astype(np.int32)     # (rows stay in master order: src is sorted)
    w_c = cov[ok].astype(np.float32)
    return np.ascontiguousarray(sub[keep]), (ptr_c, idx_c, w_c), families


class Master:
    """One pangenome on the device: presence/absence matrix, neighbourhood graph, per directed edge its organisms."""

    def __init__(self, x, ptr, idx, edge_bits, device=0):
        self.lib = load_library()
        lib = self.lib
        lib.nemgpu_master_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.nemgpu_master_destroy.argtypes = [C.c_void_p]
        lib.nemgpu_master_destroy.restype = None
        lib.nemgpu_solve_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(Config), C.c_int, C.c_int]
        x = np.asarray(x)
        if x.dtype == np.uint32:
            raise ValueError("Master takes the 0/1 byte matrix (it packs the rows itself)")
        self.n, self.d = x.shape
        self.wf = (self.d + 31) // 32
        rows = pack_rows(x)
        ptr = np.ascontiguousarray(ptr, np.int32)
        idx = np.ascontiguousarray(idx, np.int32)
        edge_bits = np.ascontiguousarray(edge_bits, np.uint32).reshape(len(idx), self.wf) if len(idx) else np.zeros((1, self.wf), np.uint32)
        assert ptr.shape == (self.n + 1,)
        self._h = C.c_void_p()
        rc = lib.nemgpu_master_create(C.byref(self._h), int(device), self.n, self.d, rows.ctypes.data, ptr.ctypes.data,
                                      idx.ctypes.data if len(idx) else None, edge_bits.ctypes.data if len(idx) else None)
        if rc != 0:
            raise NemGpuError("nemgpu_master_create failed (status %d): %s" % (rc, lib.nemgpu_last_error().decode()))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.nemgpu_master_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve_chunks(self, samples, k=3, prop=(0.33333, 0.33333, None), center_k=(1.0, 0.5, 0.0), disp_k=(0.1, 0.5, 0.1),
                     workers=8, group=32, algo="ncem", beta=0.5, disper="sk_", propor="pk", cvtest="clas", cvthres=1e-8,
                     it_max=100, param_fix=False, tie="hash", seed=0, want_params=True):
        """samples: sequences of organism indices (a chunk's columns, in order).  Initial parameters as PPanGGOLiN's
        default .m (ppanggolin.py:893-901): proportions (the last one None = the float remainder ReadParamFile computes,
        nem_exe.c:1022-1034), ONE centre and ONE dispersion per class.  Returns one dict per sample: families (master
        index of the chunk's family j), labels (uint8 per kept family), prop / center / disp / nbobs_k, iters, status, ..."""
        lib = self.lib
        prop = list(prop)
        if prop[-1] is None:
            rem = np.float32(1.0)
            for v in prop[:-1]:
                rem = np.float32(rem - np.float32(v))
            prop[-1] = rem
        prop = np.ascontiguousarray(prop, np.float32)
        center_k = np.ascontiguousarray(center_k, np.float32)
        disp_k = np.ascontiguousarray(disp_k, np.float32)
        assert len(prop) == k and len(center_k) == k and len(disp_k) == k
        cfg = Config(ALGO[algo], beta, DISP[disper], PROP[propor], CVT[cvtest], cvthres, it_max, int(param_fix), TIE[tie], seed)
        arr = (Chunk * len(samples))()
        hold, outs = [], []
        nw64 = (self.n + 63) // 64
        for q, s in zip(arr, samples):
            org = np.ascontiguousarray(s, np.int32)
            dc = len(org)
            o = dict(keep=np.zeros(nw64, np.uint64), labels=np.zeros(self.n, np.uint8))
            q.organisms, q.dc = org.ctypes.data, dc
            q.keep, q.labels = o["keep"].ctypes.data, o["labels"].ctypes.data
            if want_params:
                o.update(prop=np.empty(k, np.float32), center=np.empty((k, dc), np.float32), disp=np.empty((k, dc), np.float32),
                         nbobs_k=np.empty(k, np.float32))
                q.out_prop, q.out_center, q.out_disp, q.out_nbobs_k = (o[f].ctypes.data for f in ("prop", "center", "disp", "nbobs_k"))
            hold.append(org)
            outs.append(o)
        import time
        t0 = time.perf_counter()
        rc = lib.nemgpu_solve_chunks(self._h, arr, len(samples), int(k), prop.ctypes.data, center_k.ctypes.data, disp_k.ctypes.data,
                                     C.byref(cfg), int(workers), int(group))
        self.last_call_seconds = time.perf_counter() - t0     # (the library call alone: what follows is numpy bookkeeping)
        if rc != STATUS_OK:
            raise NemGpuError("nemgpu_solve_chunks failed (status %d): %s" % (rc, lib.nemgpu_last_error().decode()))
        res = []
        for q, o in zip(arr, outs):
            r = q.result
            bits = np.unpackbits(o["keep"].view(np.uint8), bitorder="little")[:self.n]
            fam = np.flatnonzero(bits)
            assert len(fam) == q.n
            meta = dict(status=r.status, iters=r.iters, converged=bool(r.converged), emptyk=r.emptyk, n=q.n, nnz=q.nnz,
                        n_zero_density=r.zero_density_sites, sweep_rounds=r.sweep_rounds, crit=np.array(list(r.crit), np.float32),
                        families=fam, labels=o["labels"][:q.n].copy())
            for f in ("prop", "center", "disp", "nbobs_k"):
                if f in o:
                    meta[f] = o[f]
            res.append(meta)
        return res
