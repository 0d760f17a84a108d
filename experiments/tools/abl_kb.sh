for v in NOA NOOPD NOMFMA; do
  echo "== $v"
  DIFFAB_HIP_LIB=$GRAFT_REPO_ROOT/experiments/build_abl/kb_$v/libdiffab_hip.so TOP=40 bash tools/kstats_top.sh --train --steps 6 --warmup 2 | grep keys_nn
done
