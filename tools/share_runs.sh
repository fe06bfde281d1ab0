#!/bin/bash
# One rank's step of an N-GPU strong-scaling run of C3, modelled on ONE GPU (bench.py --share k/N: dist.Loopback, a
# recording in place of the wire): N = 1, 2, 4, 8, edge and middle ranks -> gpurun_out/share/*.json and one summary.
#   tools/share_runs.sh [steps]
set -u
steps=${1:-10}
out=gpurun_out/share
mkdir -p $out
python bench.py --no-cpu-baseline --no-sub-records --steps $steps --warmup 3 > $out/n1.json 2> $out/n1.err
full=$(python -c "import json;print(json.load(open('$out/n1.json'))['ms_per_step'])")
for spec in 0/2 1/2 0/4 1/4 3/4 0/8 3/8 4/8 7/8; do
  name=$(echo $spec | tr / _)
  python bench.py --no-cpu-baseline --no-sub-records --steps $steps --warmup 3 --share $spec --full-step-ms $full \
      > $out/share_$name.json 2> $out/share_$name.err || tail -3 $out/share_$name.err
done
python - <<PY
import json, glob, os
rows = []
full = json.load(open("$out/n1.json"))
rows.append({"rank": 0, "of": 1, "step_ms": full["ms_per_step"], "kernels_ms": full["pipeline_roofline"]["gpu_kernel_ms_per_step_rank0"],
             "main_stream_kernels_ms": full["pipeline_roofline"]["main_stream_kernel_ms_per_step_rank0"],
             "tail_after_last_kernel_ms": full["pipeline_roofline"]["tail_after_last_kernel_ms"], "table_sha1": full["table_sha1"]})
for f in sorted(glob.glob("$out/share_*.json")):
    try:
        rec = json.load(open(f))
    except ValueError:
        continue
    sh = dict(rec["share"]); sh.pop("note", None)
    sh["table_sha1"] = rec["table_sha1"]; sh["blobs"] = rec["blobs"]
    sh["fraction_of_linear"] = None if not sh.get("linear_ms") else round(sh["linear_ms"] / sh["step_ms"], 3)
    rows.append(sh)
rows.sort(key=lambda r: (r["of"], r["rank"]))
json.dump({"what": "MODEL: one rank's step of an N-GPU strong-scaling run of C3 (2048x2048x1024, 256 blocks), measured on ONE MI355X "
                   "with a recording in place of the wire (bench.py --share k/N); no RCCL transfer time in it",
           "one_gpu_step_ms": full["ms_per_step"], "one_gpu_table_sha1": full["table_sha1"], "runs": rows},
          open("$out/summary.json", "w"), indent=1)
for r in rows:
    print(r)
PY
