#!/bin/bash
# Runs on the GPU box: tools/step_prof.py on the working tree, then on tools/abl/pkg_old (see ab_prepare.sh), then
# on the working tree again; arguments are passed to step_prof.py.
mkdir -p gpurun_out
timeout -k 10 200 python tools/step_prof.py "$@" > gpurun_out/prof_new.json || exit 1
mv gtg-vae_amd gtg-vae_amd.new && cp -r tools/abl/pkg_old gtg-vae_amd
timeout -k 10 200 python tools/step_prof.py "$@" > gpurun_out/prof_old.json; rc=$?
rm -rf gtg-vae_amd && mv gtg-vae_amd.new gtg-vae_amd
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/step_prof.py "$@" > gpurun_out/prof_new2.json || exit 1
python - <<'PY'
import json
a, b, c = (json.load(open("gpurun_out/prof_%s.json" % t)) for t in ("new", "old", "new2"))
print("gradients bit-identical new vs old:", a["grad_sha1"] == b["grad_sha1"], " new vs new:", a["grad_sha1"] == c["grad_sha1"])
for k in b["grad_sha1"]:
    if a["grad_sha1"].get(k) != b["grad_sha1"][k]:
        print("  differs:", k)
for k, v in b["us_per_launch"].items():
    print("%-48s old %8.2f  new %8.2f %8.2f" % (k, v, a["us_per_launch"].get(k, float("nan")), c["us_per_launch"].get(k, float("nan"))))
PY
