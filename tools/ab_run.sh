#!/bin/bash
# Runs on the GPU box: three benches of the working tree, three of tools/abl/pkg_old, three of the working tree again.
run() { for i in 1 2 3; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-mst --no-alt > gpurun_out/b.json || exit 1
        python -c "import json;print(json.load(open('gpurun_out/b.json'))['ms_per_step'])"; done; }
mkdir -p gpurun_out
echo new; run
mv gtg-vae_amd gtg-vae_amd.new && cp -r tools/abl/pkg_old gtg-vae_amd && echo old && run
rm -rf gtg-vae_amd && mv gtg-vae_amd.new gtg-vae_amd && echo new && run
