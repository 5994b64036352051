#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4k; mkdir -p $O
echo "== plain C"; tools/bin/fieldbench29 2>&1 | grep -A4 "bn254_fq\|bls12381_fq" | grep -v "^--" | tee $O/plain.txt
echo "== column asm statements"; tools/bin/fieldbench29_cols 2>&1 | grep -A4 "bn254_fq\|bls12381_fq" | grep -v "^--" | tee $O/cols.txt
echo "== per-instruction asm (r03)"; tools/bin/fieldbench29_asm 2>&1 | grep -A4 "bn254_fq\|bls12381_fq" | grep -v "^--" | tee $O/asm.txt
