import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def case_names(files_only=False):
    return [m["name"] for m in manifest() if (m["files"] or not files_only)]


def load_case(name):
    d = os.path.join(GOLDEN, name)
    with open(os.path.join(d, "meta.json")) as f:
        meta = json.load(f)
    inp = np.load(os.path.join(d, "inputs.npz"))
    exp = np.load(os.path.join(d, "expected.npz"))
    n, dd = int(inp["n"]), int(inp["d"])
    x = np.unpackbits(inp["xbits"], axis=1, bitorder="little")[:, :dd].astype(np.uint8)
    nei = (inp["nei_ptr"], inp["nei_idx"], inp["nei_w"]) if bool(inp["has_graph"]) else None
    case = dict(meta=meta, x=x, nei=nei, k=int(inp["k"]), prop=inp["prop"], center=inp["center"], disp=inp["disp"],
                cfg=meta["cfg"], expected={k: exp[k] for k in exp.files})
    for ext in ("uf", "mf"):
        p = os.path.join(d, "ref_%s.txt.gz" % ext)
        if os.path.isfile(p):
            case["ref_" + ext] = gzip.open(p, "rb").read()
    return case
