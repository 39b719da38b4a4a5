"""The formation of a sample's NEM problem from a master pangenome (pangenomenem_amd.chunks.form_chunk_host -- what the
device kernels of csrc/nem_chunks.hip are held against), checked on the CPU against the reference's own recipe
(__write_nem_input_files, ppanggolin.py:821-930) restated family by family with Python sets and dicts."""
import numpy as np
import pytest

from pangenomenem_amd import synth
from pangenomenem_amd.chunks import form_chunk_host, pack_rows


def by_the_book(x, ptr, idx, edge_bits, organisms):
    """ppanggolin.py:843-883 with `organisms` the ordered sample: index_fam in node order, the .dat rows, per family the
    neighbours with their coverage (0: skipped), in the order the master lists them"""
    n, d = x.shape
    orgs = [int(o) for o in organisms]
    index_fam, rows = {}, []
    for i in range(n):
        node_organisms = set(np.flatnonzero(x[i]).tolist())
        if not set(orgs).isdisjoint(node_organisms):
            rows.append([1 if o in node_organisms else 0 for o in orgs])
            index_fam[i] = len(index_fam)
    nei = []
    for i in index_fam:
        row = []
        for e in range(ptr[i], ptr[i + 1]):
            carried = set(np.flatnonzero(np.unpackbits(edge_bits[e].view(np.uint8), bitorder="little")[:d]).tolist())
            coverage = sum(1 for o in orgs if o in carried)
            if coverage == 0 or int(idx[e]) not in index_fam:
                continue
            row.append((index_fam[int(idx[e])], float(coverage)))
        nei.append(row)
    return np.array(rows, np.uint8), nei, list(index_fam)


@pytest.mark.parametrize("n,d,dc,seed", [(300, 70, 20, 1), (500, 33, 33, 2), (257, 100, 7, 3), (64, 40, 1, 4)])
def test_host_formation_follows_the_reference_recipe(n, d, dc, seed):
    x, (ptr, idx), eb = synth.master_pangenome(n, d, seed, chord_frac=0.3)
    rng = np.random.default_rng(seed)
    org = rng.permutation(d)[:dc]
    xc, (pc, ic, wc), fam = form_chunk_host(x, ptr, idx, eb, org)
    rows, nei, index_fam = by_the_book(x, ptr, idx, eb, org)
    assert fam.tolist() == index_fam
    assert np.array_equal(xc, rows)
    assert len(fam) < n or dc == d                           # (the sample does drop families)
    for j, row in enumerate(nei):
        got = list(zip(ic[pc[j]:pc[j + 1]].tolist(), wc[pc[j]:pc[j + 1]].tolist()))
        assert got == row, j
    assert pc[-1] == sum(len(r) for r in nei)


def test_whole_sample_is_the_master_itself():
    """every organism in master order: nothing is dropped but families nobody has and edges nobody carries"""
    n, d = 200, 37
    x, (ptr, idx), eb = synth.master_pangenome(n, d, 9)
    xc, (pc, ic, wc), fam = form_chunk_host(x, ptr, idx, eb, np.arange(d))
    assert np.array_equal(fam, np.flatnonzero(x.any(axis=1))) and np.array_equal(xc, x[fam])
    cov = np.unpackbits(eb.view(np.uint8), axis=1).sum(axis=1)
    assert pc[-1] == int((cov > 0).sum())


def test_pack_rows_layout():
    x = (np.arange(3 * 70).reshape(3, 70) % 3 == 0).astype(np.uint8)
    rows = pack_rows(x)
    assert rows.shape == (3, 3) and rows.dtype == np.uint32
    for i in range(3):
        for o in range(70):
            assert ((int(rows[i, o >> 5]) >> (o & 31)) & 1) == int(x[i, o])
