#!/bin/bash
# Runs on the GPU box: the joint step (tools/joint_bench.py, default-mode trainer only) of the working tree, of
# tools/abl/pkg_old (see ab_prepare.sh), and of the working tree again.
run() { timeout -k 10 300 python tools/joint_bench.py 20 trainer 2>/dev/null | python -c "import json,sys;print(json.loads(sys.stdin.read().strip().splitlines()[-1])['trainer_f16x2']['ms_per_step'])" || exit 1; }
echo new; run; run
mv gtg-vae_amd gtg-vae_amd.new && cp -r tools/abl/pkg_old gtg-vae_amd && echo old && run && run
rm -rf gtg-vae_amd && mv gtg-vae_amd.new gtg-vae_amd && echo new && run && run
