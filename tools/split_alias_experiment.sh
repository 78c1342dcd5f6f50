#!/bin/bash
# Round-3 verdict item 8: the split GEMM with its A operand L2-resident (every row panel reads one of 8 panels, 3 MB:
# -DGS_EXPERIMENT_A_ALIAS=8, tools only) beside the production kernel: time / TF-equivalent, in-kernel clock, FETCH_SIZE.
#   hipcc --offload-arch=gfx950 -O3 [-DGS_EXPERIMENT_A_ALIAS=8] [-DGB2_CLOCK_DIAG] -o tools/gemm_tune_split[_clk][_alias] tools/gemm_tune_split.hip
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/${1:-r04}/split_alias"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for v in "" _alias; do
  "$R/tools/gemm_tune_split$v" 65536 50 > "$O/tune$v.log" 2>&1
  "$R/tools/gemm_tune_split_clk$v" 65536 50 > "$O/clk$v.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch$v" -o p -- "$R/tools/gemm_tune_split$v" 65536 2 > /dev/null 2> "$O/pmc_fetch$v.log"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write$v" -o p -- "$R/tools/gemm_tune_split$v" 65536 2 > /dev/null 2> "$O/pmc_write$v.log"
done
python3 - "$O" <<'PY' > "$O/summary.txt"
import csv, glob, sys
O = sys.argv[1]
print("FETCH_SIZE / WRITE_SIZE per launch of iefvad_gemm_split_n128_kernel (M = 65536, K = 768), KB as the counter reports (gfx950: FETCH_SIZE x 2 = bytes / 1024)")
for v in ("", "_alias"):
    for c in ("fetch", "write"):
        f = glob.glob(f"{O}/pmc_{c}{v}/**/*counter_collection.csv", recursive=True)
        if not f:
            print(v, c, "no csv"); continue
        acc = {}
        for r in csv.DictReader(open(f[0])):
            if "gemm_split_n128" in r["Kernel_Name"] and "f16" not in r["Kernel_Name"]:
                acc.setdefault(r["Grid_Size"], []).append(float(r["Counter_Value"]))
        for g, vals in sorted(acc.items()):
            print(f"{'alias' if v else 'production':10s} {c.upper()}_SIZE grid {g:>8s}: mean {sum(vals)/len(vals):12.0f} KB over {len(vals)} launches")
PY
for v in "" _alias; do echo "== tune$v"; grep "n128" "$O/tune$v.log"; echo "== clk$v"; grep -A3 "128 x 128, two" "$O/clk$v.log" | head -5; done >> "$O/summary.txt"
cat "$O/summary.txt"
