#!/bin/bash
# A/B (round 5): the persistent out_proj + LayerNorm kernel with every other workgroup started n x 8128 cycles late (IEFVAD_OL_STAGGER = n)
QUIET="--no-extra-modes --no-ucf-eval --no-cpu-baseline"
for v in 0 2 3 4 6 0 2 3 4 6; do
  IEFVAD_OL_STAGGER=$v python3 bench.py --compute bf16 --steps 3 --warmup 1 $QUIET 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('IEFVAD_OL_STAGGER=$v', round(d['value']), 'snippets/s', round(d['ms_per_step'],2), 'ms/step; out_proj + LN', round(d['stage_ms_per_step']['out_gemm_ms'],2), 'ms/step')"
done
