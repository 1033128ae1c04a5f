#!/bin/bash
# One bench.py run per line of a plan file (run under gpurun from the repository root); per cell: throughput,
# ms per step, mean launch time of the dominant kernel, strip rows / views per launch the library chose.
#     tools/cells.sh tools/plans/shapes.txt [variant ...]
# Plan line:  tag | bench.py arguments          ('#' starts a comment; every cell gets --no-cpu-baseline, and
#             --no-planesweep unless it is a "--workload planesweep" cell)
# Variants:   names of build/variants/libamvs_<name>.so (tools/build_variant.sh); "base" = the in-tree library
#             (default).  Every cell runs once per variant, the variants alternating cell by cell: an A/B inside
#             ONE GPU run (box-to-box spread is +-2 %).
# This one script replaces the rounds' one-off tables (ab / th_* / vpl_* / paired_* / shapes / ps_* .sh); their
# cells live on as tools/plans/*.txt.
plan=$1; shift
variants=("$@"); [ ${#variants[@]} -eq 0 ] && variants=(base)
mkdir -p gpurun_out
grep -v '^\s*#' "$plan" | grep '|' | while IFS='|' read -r tag args; do
  tag=$(echo $tag); extra="--no-planesweep"; case "$args" in *planesweep*) extra="";; esac
  for v in "${variants[@]}"; do
    if [ "$v" = base ]; then unset AMVS_LIB; else export AMVS_LIB=$PWD/build/variants/libamvs_$v.so; fi
    timeout -k 10 500 python bench.py --no-cpu-baseline $extra $args > gpurun_out/cell.json 2> gpurun_out/cell.err || { echo "$tag [$v] FAILED"; tail -3 gpurun_out/cell.err; continue; }
    python - "$tag" "$v" <<'PY'
import json, sys
r = json.loads(open("gpurun_out/cell.json").read().strip().splitlines()[-1])
c = r["config"]
print(sys.argv[1], f"[{sys.argv[2]}]", r["value"], "Mpx-hyp/s", r["ms_per_step"], "ms/step", "launch", r["roofline"]["avg_launch_ms"], "ms",
      "frac", r["roofline"]["frac"], "rows", c.get("tile_rows"), "vpl", c.get("views_per_launch"), flush=True)
PY
  done
done
