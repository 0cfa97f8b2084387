import sys; sys.path.insert(0, ".")
import bench, json
for evo in (0, 500, 1000, 2000):
    r = bench.run_single("dam1m", 50, 5, 0, evolve=evo, prime=150 if evo == 0 else 0)
    print(evo, "%.4f ms" % r["ms_per_step"], {k: round(v, 4) for k, v in r["phases_ms_per_step"].items()}, r["tile_rounds"], "nb mean %.1f max %d" % (r["mean_neighbors"], r["max_neighbors"]), flush=True)
