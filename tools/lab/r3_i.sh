#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
for v in 1 0; do echo "== R29_REDUCE_G2=$v"; ZKMI_R29_REDUCE_G2=$v timeout 300 python tools/lab/dbg_g2rc.py bn128 2>&1 | tail -14; done
echo "== bls default"; timeout 300 python tools/lab/dbg_g2rc.py bls12381 2>&1 | tail -14
