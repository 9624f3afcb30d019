#!/bin/bash
# Build the last commit's package next to the working tree's, for an A/B bench on ONE GPU box
# (boxes differ by ~2 %):  tools/ab_prepare.sh && gpurun -- 'bash tools/ab_run.sh'
set -e
cd "$(dirname "$0")/.."
rm -rf tools/abl && mkdir -p tools/abl
git stash -q
make -C gtg-vae_amd/csrc >/dev/null
cp -r gtg-vae_amd tools/abl/pkg_old
git stash pop -q
make -C gtg-vae_amd/csrc >/dev/null
echo prepared
