#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/_tmp_bias_probe.py 2>&1 | grep -v amdgpu.ids
