#!/bin/bash
# Timing probe (round 5): how much of the persistent out_proj + LayerNorm kernel's epilogue is workgroup-barrier wait?  Libraries built
# with -DOP_PROBE_NOBAR=1 (the four "quarter consumed" barriers dropped) and =2 (the four "quarter parked" barriers too) give WRONG
# results and valid timings: the upper bound of any re-arrangement of the epilogue's synchronisation.
#   make -C ief-vad_amd/csrc -B EXTRA=-DOP_PROBE_NOBAR=1 OUT=../../build/libiefvad_nobar1.so   (and =2 -> nobar2)
# Other probe builds of the same kernel, passed as arguments: -DOP_PROBE_NOSTORE (the normalised rows are not stored), -DOP_PROBE_NORES (the
# residual rows of quarters 1 - 3 are not requested), both; with -DOC_DIAG added, tools/outproj_pdiag.py prints their phase stamps.
# bench.py accepts non-finite scores only under IEFVAD_TIMING_PROBE=1, which this script sets for the probe libraries alone.
QUIET="--no-extra-modes --no-ucf-eval --no-cpu-baseline"
LIBS=${@:-build/libiefvad_nobar1.so build/libiefvad_nobar2.so}
for v in "" $LIBS "" $LIBS; do
  if [ -n "$v" ] && [ ! -f "$v" ]; then continue; fi
  IEFVAD_TIMING_PROBE=${v:+1} IEFVAD_LIB=${v:+$PWD/$v} python3 bench.py --compute bf16 --steps 3 --warmup 1 $QUIET 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lib=${v:-production}', round(d['value']), 'snippets/s', round(d['ms_per_step'],2), 'ms/step; out_proj + LN', round(d['stage_ms_per_step']['out_gemm_ms'],2), 'ms/step')"
done
