"""configs[1] (the headline's matrix and graph) under the two tie rules, 100 restart cycles each, for a kernel trace:
what the reference's own rule (TIE_LIBC, the drop-in's default) costs per launch beside the stateless hash rule."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
x, nei, prop, center, disp, disper, _ = bench.make_workload(20000, 500, 3, "ushape", 2)
for tie in ("hash", "libc"):
    run = bench.EngineRun(x, nei, 3, prop, center, disp, "ncem", 0.5, "sk_", tie=tie)
    run.prime(run.cycle * 4, run.cycle)
    t0 = time.perf_counter()
    run.run_steps(run.cycle * 100)
    dt = time.perf_counter() - t0
    print("%s: %.4f ms per step (%d iterations per cycle)" % (tie, dt * 1e3 / (run.cycle * 100), run.cycle), file=sys.stderr)
    run.eng.close()
