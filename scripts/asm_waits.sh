#!/bin/bash
# The wait / barrier / memory-instruction skeleton of a kernel's device code, straight from hipcc's assembly: what round 3 used to
# find the blanket `s_waitcnt lgkmcnt(0)` in gemm2p_kernel's K loop (DESIGN.md tuning log).  No GPU needed.
#   scripts/asm_waits.sh reverb_amd/csrc/gemm2.hip _ZN3rvb13gemm2p_kernelIttLb0ELb0EEEvNS_8GemmArgsE [first_line [last_line]]
# Without a symbol: lists the kernels of the file with their VGPR / scratch use.
set -eu
SRC=${1:?source file}; SYM=${2:-}
S=/tmp/asm_waits_$$.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip --cuda-device-only -S "$SRC" -o $S 2>/dev/null
if [ -z "$SYM" ]; then
  grep -E "^\s+\.name:|\.vgpr_count:|\.private_segment_fixed_size:|\.group_segment_fixed_size:" $S | paste - - - - | sed 's/\s\+/ /g' | head -80
  rm -f $S; exit 0
fi
K=/tmp/asm_waits_$$.k
awk -v s="^$SYM:" '$0 ~ s {f=1} f {print} /^\.Lfunc_end/ {if (f) exit}' $S > $K
A=${3:-1}; B=${4:-$(wc -l < $K)}
echo "$(wc -l < $K) lines; loop headers:"; grep -n "Loop Header" $K | head -20
sed -n "${A},${B}p" $K | grep -v "^\s*;" | grep -E "v_mfma|ds_read|ds_write|s_waitcnt|s_barrier|global_load|global_store|buffer_|scratch_|v_exp|v_rcp|s_sleep|s_setprio" \
  | awk '{print $1, ($1=="s_waitcnt" ? $2" "$3 : "")}' | uniq -c
rm -f $S $K
