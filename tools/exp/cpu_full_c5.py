#!/usr/bin/env python3
"""The WHOLE C5 volume (2 channels x 512 x 2048 x 2048, denoise_size 25, co-localisation) through the CPU oracle, on
whatever cores this machine has (hours on 8 cores: not part of any default run).  Writes the digests of the oracle's
final table and flags the way bench.py writes the GPU's (``table_sha1`` / ``colocs_sha1``): equal digests = the GPU's
full-size C5 table is the oracle's, row for row.  The volume comes from the generator bench.py uses (same on CPU and GPU).

    python tools/exp/cpu_full_c5.py [cores] > profiles/r06_cpu_full_c5.json
"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench

if __name__ == "__main__":
    cores = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
    cfg = bench.CONFIGS["c5"]
    profile = dict(bench._BASE_PROFILE, **cfg["profile"])
    t0 = time.time()
    sample = bench.make_host_sample(cfg["shape"], cfg["seed"], cfg["channels"])
    t_gen = time.time() - t0
    (final, colocs), t_det, t_tot, n_jobs = bench.cpu_baseline(sample, cores, profile, list(range(cfg["channels"])),
                                                               True, with_colocs=True)
    sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "cpu_full_c5_table.npz"), final=final, colocs=colocs)
    print(json.dumps({"workload": "c5, the whole volume " + "x".join(str(v) for v in cfg["shape"]) + " (z,y,x) x 2 channels",
                      "blocks": n_jobs, "cores": cores, "volume_gen_s": round(t_gen, 1), "volume_sha1": bench.volume_sha1(sample),
                      "detection_s": round(t_det, 1), "total_s": round(t_tot, 1),
                      "Mvoxels_per_s": round(float(np.prod(cfg["shape"])) / t_tot / 1e6, 3),
                      "blobs": int(len(final)), "table_sha1": sha(final), "colocs_sha1": sha(colocs),
                      "colocs_dtype": str(colocs.dtype), "colocs_shape": list(colocs.shape)}))
