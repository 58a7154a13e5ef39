# Instruction mix of one kernel, without a GPU:  bash tools/isa_audit.sh <header-or-.hip> '<explicit instantiation or empty>' <mangled-name-substring> [extra hipcc flags]
#   bash tools/isa_audit.sh conv_direct.h 'template __global__ void conv_direct_f32<DirectCfg<7, 1>>(const DirectKParams);' conv_direct_f32
#   bash tools/isa_audit.sh warp.hip '' warp_concat_kernelILi1 -ffp-contract=off
# Prints registers / spills / LDS of the kernel and its instruction histogram (how the per-element branch patterns of
# HISTORY.md section 3.9b were found: count v_cndmask / s_cbranch / s_waitcnt against the arithmetic the kernel exists for).
SRC=$1; INST=$2; PAT=$3; shift 3
CS=$(dirname $0)/../animateportrait_amd/csrc
OUT=${TMPDIR:-/tmp}/isa_audit; mkdir -p $OUT
if [ "${SRC##*.}" = "hip" ]; then
  cp $CS/$SRC $CS/_isa_audit.hip
else
  printf '#include "%s"\nnamespace apamd { %s }\n' "$SRC" "$INST" > $CS/_isa_audit.hip
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$CS "$@" -S --cuda-device-only -o $OUT/k.s $CS/_isa_audit.hip 2>&1 | grep -E "error" | head
rm -f $CS/_isa_audit.hip
SYM=$(grep -oE "^_Z[A-Za-z0-9_]*${PAT}[A-Za-z0-9_]*:" $OUT/k.s | head -1 | tr -d ':')
[ -z "$SYM" ] && { echo "no kernel matching $PAT"; exit 1; }
echo "kernel $SYM"
awk -v s="$SYM:" 'index($0, s)==1{f=1} f{print} /s_endpgm/{if(f)exit}' $OUT/k.s > $OUT/body.s
grep -A40 "^\s*\.name:\s*$SYM" $OUT/k.s | grep -E "vgpr_count|vgpr_spill|sgpr_count|group_segment_fixed|private_segment_fixed" | head -5
echo "lines $(wc -l < $OUT/body.s)  (full listing: $OUT/body.s)"
grep -oE "^\s*[vs]_[a-z0-9_]+|^\s*ds_[a-z0-9_]+|^\s*global_[a-z_0-9]+|^\s*buffer_[a-z_0-9]+" $OUT/body.s | sort | uniq -c | sort -rn | head -${ISA_TOP:-25}
