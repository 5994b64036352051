"""Synthetic workload generators (SURVEY.md 8d recipes) shared by bench.py and tests/: input streams, structurally valid Groth16 zkey / wtns
containers, valid PLONK / FFLONK keys. Nothing here touches the CPU oracle."""
