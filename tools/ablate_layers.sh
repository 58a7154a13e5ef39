cd $GRAFT_REPO_ROOT
export APAMD_LIB=$GRAFT_REPO_ROOT/animateportrait_amd/libapamd_ablate.so
for L in "down 64->128 k3s2" "down 128->256 k3s2" "up 256->128" "stem 3->64"; do
  for a in 0 1 2 8 9 11; do echo -n "ABLATE=$a  "; APAMD_ABLATE=$a python tools/conv_bench.py 20 "$L" 2>&1 | tail -1; done
done
