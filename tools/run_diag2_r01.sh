cd $GRAFT_REPO_ROOT
for lib in "" gcn10_amd/libgcn10_gpu_diag1.so gcn10_amd/libgcn10_gpu_diag2.so; do
  echo "== lib=$lib"
  GCN10_GPU_LIB=$lib python tools/tune_strip.py --quick 2>/dev/null | grep -E '"workload": "config2"' | cut -c1-140
done
