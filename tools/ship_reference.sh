#!/bin/bash
# A git-ignored copy of what the reference-side tests import (coot/, nntrainer/, the ActivityNet YAML) inside the repo, so that ONE gpurun
# call can run tests/test_gpu_reference_on_device.py on the GPU box (which has no /root/reference):
#   bash tools/ship_reference.sh && gpurun -- 'COOT_REFERENCE_ROOT=$PWD/_refship python -m pytest tests/test_gpu_reference_on_device.py -s'
# Delete _refship/ afterwards; it is never committed (.gitignore).
set -e
cd "$(dirname "$0")/.."
rm -rf _refship && mkdir -p _refship/config/retrieval/paper2020
cp -r /root/reference/coot /root/reference/nntrainer _refship/
cp /root/reference/config/retrieval/paper2020/anet_coot.yaml _refship/config/retrieval/paper2020/
find _refship -name "__pycache__" -prune -exec rm -rf {} \;
du -sh _refship
