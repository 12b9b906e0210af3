#!/bin/bash
cd $GRAFT_REPO_ROOT; timeout 600 ./tools/microbench > gpurun_out/microbench.jsonl 2>&1; grep -E "transpose|device" gpurun_out/microbench.jsonl
