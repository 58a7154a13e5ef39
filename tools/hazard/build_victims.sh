# Victim-side variants of the forward warp kernel for the co-residency lab: the product objects with warp.o
# rebuilt four ways -> animateportrait_amd/libapamd_v_{wpe2,wpe8,nop,fz}.so (select with APAMD_LIB).  Needs `make` done.
ROOT=$(cd $(dirname $0)/../.. && pwd)
cd $ROOT/animateportrait_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -Wno-unused-result -ffp-contract=off"
mkdir -p build_hz
build() {  # name, extra flags...
  local name=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c warp.hip -o build_hz/warp_$name.o 2>/dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libapamd_v_$name.so $(ls build/*.o | grep -v "/warp.o") build_hz/warp_$name.o
}
build wpe2 -DAPAMD_WARP_WPE=2 &
build wpe8 -DAPAMD_WARP_WPE=8 &
build nop -DAPAMD_HZ_NOP &
build fz -mllvm -amdgpu-waitcnt-forcezero &
wait
ls -la ../libapamd_v_*.so
