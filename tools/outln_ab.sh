#!/bin/bash
# A/B of the bf16 mode's out_proj + LayerNorm kernels: round 5's in-register LayerNorm (IEFVAD_OUTLN=r, opt-in) against round 4's
# park-through-LDS persistent kernel (the default).  Prints snippets/s, ms per step and the out_proj stage time of `bench.py --compute bf16`.
QUIET="--no-extra-modes --no-ucf-eval --no-cpu-baseline"
for v in r p r p; do
  IEFVAD_OUTLN=$v python3 bench.py --compute bf16 --steps 3 --warmup 1 $QUIET 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('IEFVAD_OUTLN=$v', round(d['value']), 'snippets/s', round(d['ms_per_step'],2), 'ms/step; out_proj + LN', round(d['stage_ms_per_step']['out_gemm_ms'],2), 'ms/step')"
done
