#!/bin/bash
# Probe builds of libcrychic_hip.so with extra -D flags, built HERE (hipcc cross-compiles) into tools/_probe/ (git-ignored, ships to
# the GPU box with the snapshot), and an A/B bench loop to run ON the box.
#   tools/probes/variants.sh build  NAME "-DCRY_PROBE_X ..."     (repeat per variant)
#   tools/probes/variants.sh bench  TAG "bench flags" NAME...      (on the box: one bench line + kernel stats per variant)
cd "$(dirname "$0")/../.."
R=$(pwd)
mode=$1; shift
if [ "$mode" = build ]; then
  name=$1; defs=$2
  mkdir -p tools/_probe
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function $defs \
    -I include -I crychic_renderer_amd/csrc $(for s in kernels.hip raster.hip api.cpp comm.cpp host_constants.cpp host_geometry.cpp host_textures.cpp; do echo -x hip crychic_renderer_amd/csrc/$s; done) \
    -ldl -o tools/_probe/lib_$name.so && echo "built tools/_probe/lib_$name.so"
  exit $?
fi
export TMPDIR=/tmp
tag=$1; flags=$2; shift; shift
mkdir -p gpurun_out
for name in "$@"; do
  lib=$R/tools/_probe/lib_$name.so
  [ "$name" = base ] && lib=$R/crychic_renderer_amd/libcrychic_hip.so
  L="--steps 100 --warmup 10 --no-cpu-baseline --no-producers --no-legs $flags"
  (cd /tmp && CRYCHIC_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_${name}_prof -- python $R/bench.py $L > $R/gpurun_out/${tag}_${name}.json 2> $R/gpurun_out/${tag}_${name}.err) || { tail -5 gpurun_out/${tag}_${name}.err; exit 1; }
  python - "$tag" "$name" <<'PY'
import csv, glob, json, sys
tag, name = sys.argv[1:3]
f = glob.glob("gpurun_out/%s_%s_prof/**/*kernel_stats.csv" % (tag, name), recursive=True)[0]
d = json.load(open("gpurun_out/%s_%s.json" % (tag, name)))
ks = {r["Name"].split("(")[0].replace("void cry::", "").replace("cry::", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f)) if "cry::" in r["Name"]}
print("%-14s frame %.4f ms  " % (name, d["ms_per_step"]) + "  ".join("%s %.1f" % (k[:22], v) for k, v in sorted(ks.items())))
PY
  rm -rf gpurun_out/${tag}_${name}_prof
done
