cd $GRAFT_REPO_ROOT
tools/hbm_streams 2>/dev/null | grep '"1R:1W"' | sort -t: -k6 | head -3
for lib in "" gcn10_amd/libgcn10_gpu_diag3.so; do
  echo "== lib=$lib"
  GCN10_GPU_LIB=$lib python tools/tune_strip.py --quick 2>/dev/null | grep -E '"workload": "config2"' | cut -c1-150
done
