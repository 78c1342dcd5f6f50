#!/bin/bash
# Rebuild libiefvad.so and print the GEMM main-loop instruction mix (MFMA runs collapsed).
set -e
cd "$(dirname "$0")/../ief-vad_amd/csrc"
make 2>&1 | grep -E "error" -A3 || true
make resource-usage 2>&1 | grep -A8 "gemm_f32_kernel" | grep -E "VGPRs:|Spill" || true
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -S --cuda-device-only -o /tmp/iefvad.s iefvad.hip 2>/dev/null
awk '/^_Z22iefvad_gemm_f32_kernel8GemmArgs:/,/s_endpgm/' /tmp/iefvad.s > /tmp/gemm.s
awk '/Inner Loop Header/,/s_cbranch_scc/' /tmp/gemm.s | awk '/v_mfma/{n++; next} {if(n>0){print "   ... " n " mfma"; n=0} print}' | grep -v "s_add\|s_mov\|s_nop\|buffer_load\|s_lshl\|s_xor\|s_cmp"
