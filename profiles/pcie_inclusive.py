#!/usr/bin/env python3
"""Host-buffer-inclusive timing of one whole solve through the in-memory C ABI:
upload + device layouts, graph upload, the EM run itself, results back.  Never part of bench.py's `value`."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenomenem_amd import synth  # noqa: E402
from pangenomenem_amd.engine import NemEngine  # noqa: E402


def main():
    out = []
    for name in ("C2", "C3"):
        cfg = synth.make_config(name)
        x = cfg["x"]
        n, d = x.shape
        bits = np.packbits(x, axis=1, bitorder="little")
        pad = (-bits.shape[1]) % 4
        bits = np.ascontiguousarray(np.pad(bits, ((0, 0), (0, pad)))).view(np.uint32)
        eng = NemEngine(n, d, 3)
        eng.set_matrix_bits(bits); eng.set_graph(cfg["nei"]); eng.set_params(cfg["prop"], cfg["center"], cfg["disp"])
        eng.configure(algo="ncem", beta=0.5, disper="sk_", propor="pk")
        eng.run(); eng.run()                        # warm-up: buffers; a batch shape gets its graph the second time it is enqueued
        rec = dict(config=name, families=n, organisms=d)
        t0 = time.perf_counter(); eng.set_matrix_bits(bits); t1 = time.perf_counter()
        eng.set_graph(cfg["nei"]); t2 = time.perf_counter()
        eng.set_params(cfg["prop"], cfg["center"], cfg["disp"]); t3 = time.perf_counter()
        r = eng.run(); t4 = time.perf_counter()
        lab = eng.labels(); t5 = time.perf_counter()
        rec.update(upload_matrix_bits_ms=(t1 - t0) * 1e3, upload_graph_ms=(t2 - t1) * 1e3, upload_params_ms=(t3 - t2) * 1e3,
                   run_ms=(t4 - t3) * 1e3, iterations=int(r["iters"]), labels_back_ms=(t5 - t4) * 1e3,
                   whole_solve_ms=(t5 - t0) * 1e3, matrix_bytes=int(bits.nbytes))
        t0 = time.perf_counter(); eng.set_matrix(x); t1 = time.perf_counter()
        rec.update(upload_matrix_bytes_ms=(t1 - t0) * 1e3)
        out.append(rec)
        eng.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
