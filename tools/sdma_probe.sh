cd $GRAFT_REPO_ROOT
env | grep -E "^HSA|^GPU_|^ROC|^HIP|^AMD" 
for e in "X=1" "GPU_FORCE_BLIT_COPY_SIZE=0" "HSA_ENABLE_SDMA=1" "HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0" "HSA_ENABLE_SDMA=0"; do
  echo "## $e"
  env $e timeout 120 python tools/sps10_skips.py 2>&1 | grep -E "product|no tail"
  env $e URH_SPS=100 timeout 120 python tools/sps10_skips.py 2>&1 | grep -E "product" | head -1
done
