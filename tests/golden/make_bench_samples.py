#!/usr/bin/env python3
"""Oracle tables of fixed samples of bench.py's workloads (``bench.py --parity-sample``).

    python tests/golden/make_bench_samples.py            # writes tests/golden/bench_sample_{c3,c5}.npz

``bench.py`` checks the HIP path against the oracle on a sample of its workload at every run, which costs about a
minute of CPU work on all cores of the box.  The ``-m gpu`` tests that run the full-size workloads only want the
check, not the CPU timing, so the oracle's answer for a FIXED sample is committed here: the sample's shape and seed,
the SHA-1 of the generated voxels (``bench.make_host_sample``: bench.py regenerates the sample and refuses the table if
the voxels differ), the profile, and the oracle's final table (+ co-localisation flags).  Data only; the oracle is
``oracle/magmap_oracle.py`` through ``bench.cpu_baseline``, i.e. exactly what a bench run without the option computes.
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

#: (workload, sample shape z y x): c3 -- 2 x 2 x 2 blocks of the benchmark geometry (seams along all three axes);
#: c5 -- one layer of 2 x 2 two-channel blocks with preprocessing and co-localisation
SAMPLES = {"c3": (320, 512, 512), "c5": (96, 512, 512)}


def main():
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=sorted(SAMPLES), default=None)
    ap.add_argument("--cores", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    args = argparse.Namespace(config=None, denoise=0, segment_size=0, shape=None)
    for name, shape in SAMPLES.items():
        if a.only and name != a.only:
            continue
        args.config = name
        cfg, profile, _, n_chl, coloc = bench.config_setup(name, args, None)
        sample = bench.make_host_sample(shape, cfg["seed"], n_chl)
        final, t_det, t_tot, n_jobs = bench.cpu_baseline(sample, a.cores, profile, list(range(n_chl)), coloc,
                                                         with_colocs=True)
        final, colocs = final
        out = dict(config=name, shape=np.asarray(shape), seed=cfg["seed"], n_channels=n_chl,
                   profile=json.dumps(profile), volume_sha1=bench.volume_sha1(sample),
                   final=np.zeros((0, 8)) if final is None else final, oracle_seconds=t_tot, blocks=n_jobs)
        if coloc:
            out["colocs"] = np.zeros((0, n_chl), dtype=np.uint8) if colocs is None else colocs
        path = os.path.join(HERE, f"bench_sample_{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{path}: {0 if final is None else len(final)} blobs, {n_jobs} blocks, oracle {t_tot:.1f} s")


if __name__ == "__main__":
    main()
