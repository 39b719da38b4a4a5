"""Seeded synthetic NEM problems: presence/absence matrices, contiguity graphs
and initial parameters shaped like what PPanGGOLiN's ``__write_nem_input_files`` emits
(reference: ppanggolin/ppanggolin.py:821-930).  Pure numpy; input generation only.
"""
import numpy as np

# BASELINE.json configs (families N, organisms D, classes K, beta)
CONFIGS = {
    "C1": dict(n=2048, d=15, k=3, beta=0.0, graph=False, seed=1),
    "C2": dict(n=20000, d=500, k=3, beta=0.5, graph=True, seed=2),
    "C3": dict(n=50000, d=1000, k=3, beta=0.5, graph=True, seed=3),
    "C4": dict(n=200000, d=5000, k=3, beta=0.5, graph=True, seed=4),
    "C5": dict(n=20000, d=500, k=None, beta=0.5, graph=True, seed=5, latent=10),
}


def bernoulli_pa_matrix(n, d, seed, mix=(0.30, 0.20, 0.50), p=(0.97, 0.5, 0.03)):
    """3-latent-class (persistent / shell / cloud) 0/1 matrix, uint8 [n, d]; all-zero rows get
    one random 1 (a family absent from every selected organism is never written, ppanggolin.py:849)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    z = rng.choice(len(mix), size=n, p=np.asarray(mix) / np.sum(mix))
    pz = np.asarray(p, np.float32)[z]
    x = np.empty((n, d), np.uint8)
    step = max(1, (1 << 26) // max(d, 1))              # rows per chunk: keeps the float scratch around 256 MB
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        x[r0:r1] = rng.random((r1 - r0, d), dtype=np.float32) < pz[r0:r1, None]
    empty = np.flatnonzero(x.sum(axis=1) == 0)
    x[empty, rng.integers(0, d, size=len(empty))] = 1
    return x, z


def ushaped_pa_matrix(n, d, seed, a=0.3, b=0.3, rows=None):
    """Family-frequency spectrum of a real pangenome: U-shaped (many rare families, many near-universal ones, a
    broad shell in between).  Family i is present in each organism with its own probability p_i ~ Beta(a, b).
    Unlike the three well-separated latent classes of bernoulli_pa_matrix -- on which NEM is at its fixed point
    after ONE iteration from PPanGGOLiN's default .m -- the shell boundaries move for several EM iterations
    (7 at 20 000 x 500 with the reference), so an EM benchmark on it times real pre-convergence iterations.
    rows = (lo, hi): only those rows of the SAME matrix are kept (a rank's shard of a problem too large to hold
    whole on every rank: the generator's stream is consumed in full, the memory is the shard's)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pz = rng.beta(a, b, size=n).astype(np.float32)
    lo, hi = (0, n) if rows is None else (max(0, int(rows[0])), min(n, int(rows[1])))
    x = np.empty((max(hi - lo, 0), d), np.uint8)
    step = max(1, (1 << 26) // max(d, 1))
    empties = []
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        chunk = rng.random((r1 - r0, d), dtype=np.float32) < pz[r0:r1, None]
        empties.append(np.flatnonzero(~chunk.any(axis=1)) + r0)
        a0, a1 = max(r0, lo), min(r1, hi)
        if a0 < a1:
            x[a0 - lo:a1 - lo] = chunk[a0 - r0:a1 - r0]
    empty = np.concatenate(empties) if empties else np.zeros(0, np.int64)
    cols = rng.integers(0, d, size=len(empty))
    mine = (empty >= lo) & (empty < hi)
    x[empty[mine] - lo, cols[mine]] = 1
    return x, pz


def grouped_pa_matrix(n, d, seed, groups=10, p_in=0.95, p_out=0.05):
    """`groups`-latent-class matrix for the K-sweep (C5): class c present with p_in in organism
    group c and p_out elsewhere, so no class empties for K <= groups."""
    rng = np.random.Generator(np.random.PCG64(seed))
    z = rng.integers(0, groups, size=n)
    org_group = (np.arange(d) * groups) // d
    prob = np.where(org_group[None, :] == z[:, None], p_in, p_out)
    x = (rng.random((n, d)) < prob).astype(np.uint8)
    empty = np.flatnonzero(x.sum(axis=1) == 0)
    x[empty, rng.integers(0, d, size=len(empty))] = 1
    return x, z


def contiguity_graph(n, seed, chord_frac=0.05, wmax=8, weights="small", d=None, counts=None):
    """Path i<->i+1 in index order plus n*chord_frac random chords, undirected, every edge listed
    from both endpoints, integer weights uniform in [1, wmax] (keeps beta*sum(w) far below the
    reference's exp overflow at 709).  Returns CSR (ptr int32[n+1], idx int32, w float32);
    neighbour order inside a row = insertion order (the order a .nei line would list them).
    weights="coverage": the weights PPanGGOLiN itself writes (ppanggolin.py:866-878: the number of selected
    organisms that carry the adjacency, the division by len(organisms) is commented out) -- integers uniform in
    [1, d]; with them beta*sum(w) passes 88 (the float exp of criterion Z overflows: M = -inf) and, on sites whose
    neighbours agree, 709 (the double exp of the site's own factor: NaN rows), as SURVEY.md Appendix B records.
    weights="adjacency": coverage weights with the structure a pangenome graph gives them -- an adjacency is carried by
    at most the organisms that carry BOTH families, and an occurrence of a family has one neighbour on either side, so a
    family's weights add up to at most twice its number of organisms: a path edge gets 60-100 % of min(count_i, count_j),
    a chord 1-15 % of it (counts = the families' organism counts, the matrix's row sums).  beta * sum(w) then stays
    below 709 for up to ~600 organisms at beta = 0.5 (finite rows, the run converges) and far above 88 (M = -inf)."""
    if weights == "coverage":
        if d is None:
            raise ValueError("weights='coverage' needs the number of organisms d")
        wmax = int(d)
    elif weights == "adjacency":
        if counts is None:
            raise ValueError("weights='adjacency' needs the families' organism counts")
    elif weights != "small":
        raise ValueError("weights must be 'small', 'coverage' or 'adjacency'")
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    src = [np.arange(n - 1, dtype=np.int64)]
    dst = [np.arange(1, n, dtype=np.int64)]
    nch = int(n * chord_frac)
    if nch > 0 and n > 3:
        a = rng.integers(0, n, size=nch)
        b = rng.integers(0, n, size=nch)
        keep = np.abs(a - b) > 1
        src.append(a[keep]); dst.append(b[keep])
    s = np.concatenate(src); t = np.concatenate(dst)
    # drop duplicate undirected edges, keep first occurrence
    key = np.minimum(s, t) * n + np.maximum(s, t)
    _, first = np.unique(key, return_index=True)
    first.sort()
    s, t = s[first], t[first]
    w = rng.integers(1, wmax + 1, size=len(s)).astype(np.float32)
    if weights == "adjacency":
        cmin = np.minimum(np.asarray(counts)[s], np.asarray(counts)[t]).astype(np.float64)
        u = rng.random(len(s))
        share = np.where(np.abs(s - t) == 1, 0.6 + 0.4 * u, 0.01 + 0.14 * u)
        w = np.maximum(1.0, np.floor(share * cmin)).astype(np.float32)
    # both directions, stable by insertion order
    us = np.concatenate([s, t]); ut = np.concatenate([t, s]); uw = np.concatenate([w, w])
    order = np.concatenate([np.arange(len(s)) * 2, np.arange(len(s)) * 2 + 1])
    perm = np.lexsort((order, us))
    us, ut, uw = us[perm], ut[perm], uw[perm]
    ptr = np.zeros(n + 1, np.int32)
    np.add.at(ptr, us + 1, 1)
    ptr = np.cumsum(ptr).astype(np.int32)
    return ptr, ut.astype(np.int32), uw.astype(np.float32)


def ring_graph(n, seed, deg, wlo, whi):
    """Every site linked to its deg/2 nearest sites on either side of a ring, symmetric integer weights uniform
    in [wlo, whi]: every site's weight sum is about deg*(wlo+whi)/2, so beta*sum(w) can be placed on either side of
    88 (float exp) and 709 (double exp) for ALL sites -- the heavy-weight cases of the golden fixtures and the fuzz
    suite.  Neighbour order inside a row: offsets -deg/2 .. -1, +1 .. +deg/2."""
    h = max(1, deg // 2)
    offs = np.concatenate([np.arange(-h, 0), np.arange(1, h + 1)])
    offs = offs[np.abs(offs) < n] if n > 1 else offs[:0]
    idx = ((np.arange(n)[:, None] + offs[None, :]) % n).astype(np.int64)
    keep = idx != np.arange(n)[:, None]
    ptr = np.zeros(n + 1, np.int32)
    ptr[1:] = np.cumsum(keep.sum(axis=1))
    src = np.repeat(np.arange(n), keep.sum(axis=1))
    dst = idx[keep]
    key = np.minimum(src, dst) * n + np.maximum(src, dst)
    uniq, inv = np.unique(key, return_inverse=True)
    rng = np.random.Generator(np.random.PCG64(seed + 104729))
    wu = rng.integers(wlo, whi + 1, size=len(uniq)).astype(np.float32)
    return ptr, dst.astype(np.int32), wu[inv].astype(np.float32)


def master_pangenome(n, d, seed, chord_frac=0.05, carry_path=0.8, carry_chord=0.08):
    """A whole pangenome for the chunk voting loop (ppanggolin.py:995-1097): the U-shaped presence/absence matrix over ALL
    d organisms, the contiguity graph's structure, and per directed edge the organisms that carry the adjacency -- a
    subset of those where both families are present (a path edge in most of them, a chord in few), the same set in both
    directions.  Returns x uint8 [n][d], (ptr, idx), edge_bits uint32 [nnz][ceil(d/32)]."""
    x, _ = ushaped_pa_matrix(n, d, seed)
    ptr, idx, _ = contiguity_graph(n, seed, chord_frac=chord_frac)
    rng = np.random.Generator(np.random.PCG64(seed + 15485863))
    src = np.repeat(np.arange(n), np.diff(ptr)).astype(np.int64)
    dst = idx.astype(np.int64)
    lo, hi = np.minimum(src, dst), np.maximum(src, dst)
    uniq, inv = np.unique(lo * n + hi, return_inverse=True)
    ulo, uhi = uniq // n, uniq % n
    wf = (d + 31) // 32
    ebits_u = np.zeros((len(uniq), wf * 4), np.uint8)
    step = max(1, (1 << 24) // max(d, 1))
    for e0 in range(0, len(uniq), step):
        e1 = min(len(uniq), e0 + step)
        both = x[ulo[e0:e1]] & x[uhi[e0:e1]]
        p = np.where(np.abs(uhi[e0:e1] - ulo[e0:e1]) == 1, carry_path, carry_chord)[:, None]
        carry = (both.astype(bool) & (rng.random((e1 - e0, d), dtype=np.float32) < p)).astype(np.uint8)
        packed = np.packbits(carry, axis=1, bitorder="little")
        ebits_u[e0:e1, :packed.shape[1]] = packed
    edge_bits = ebits_u.view(np.uint32)[inv]
    return x, (ptr, idx), np.ascontiguousarray(edge_bits)


def default_init(d, low_disp=0.1):
    """PPanGGOLiN's default .m (ppanggolin.py:893-901): pi 0.33333/0.33333/rest, mu 1/0.5/0,
    eps low/0.5/low.  pi_K is computed as ReadParamFile does (float 1 - p0 - p1, nem_exe.c:1022-1034)."""
    p0 = np.float32(0.33333)
    prop = np.array([p0, p0, np.float32(np.float32(np.float32(1.0) - p0) - p0)], np.float32)
    center = np.stack([np.ones(d), np.full(d, 0.5), np.zeros(d)]).astype(np.float32)
    disp = np.stack([np.full(d, low_disp), np.full(d, 0.5), np.full(d, low_disp)]).astype(np.float32)
    return prop, center, disp


def kclass_init(x, k, eps=0.1):
    """Deterministic K-class init for the K-sweep: equal pi written with 4 decimals
    (round(1/K, 4) like ppanggolin.py:907), mu_k = data row floor((k+1/2)N/K) after sorting rows by
    (popcount, index), eps = 0.1 everywhere."""
    n, d = x.shape
    pc = x.sum(axis=1)
    order = np.lexsort((np.arange(n), pc))
    rows = [order[int((c + 0.5) * n / k)] for c in range(k)]
    center = x[rows].astype(np.float32)
    disp = np.full((k, d), eps, np.float32)
    pw = np.float32(round(1.0 / k, 4))
    prop = np.full(k, pw, np.float32)
    rem = np.float32(1.0)
    for c in range(k - 1):
        rem = np.float32(rem - prop[c])
    prop[k - 1] = rem
    return prop, center, disp


def make_config(name, n=None, d=None, k=None):
    """Build one BASELINE config (optionally down-scaled): dict(x, nei, k, beta, prop, center, disp)."""
    cfg = dict(CONFIGS[name])
    n = n or cfg["n"]; d = d or cfg["d"]
    if cfg.get("latent"):
        x, _ = grouped_pa_matrix(n, d, cfg["seed"], groups=cfg["latent"])
        k = k or 5
        prop, center, disp = kclass_init(x, k)
    else:
        x, _ = bernoulli_pa_matrix(n, d, cfg["seed"])
        k = 3
        prop, center, disp = default_init(d)
    nei = contiguity_graph(n, cfg["seed"]) if cfg["graph"] else None
    return dict(name=name, x=x, nei=nei, k=k, beta=cfg["beta"], prop=prop, center=center, disp=disp)
