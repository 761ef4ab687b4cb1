cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in liburhgpu.so liburhgpu_noee.so; do
  echo "## $lib sps10"; URHGPU_LIB=$GRAFT_REPO_ROOT/urh_amd/$lib timeout 200 python tools/sps10_skips.py 2>&1 | grep -E "product|expand"
  echo "## $lib sps100"; URH_SPS=100 URHGPU_LIB=$GRAFT_REPO_ROOT/urh_amd/$lib timeout 200 python tools/sps10_skips.py 2>&1 | grep -E "product"
done; done
for lib in liburhgpu.so liburhgpu_noee.so liburhgpu.so liburhgpu_noee.so; do echo "## $lib variants"; URHGPU_LIB=$GRAFT_REPO_ROOT/urh_amd/$lib timeout 200 python tools/dtype_mask_sweep.py 4 2>&1 | grep -v "amdgpu.ids\|^#"; done
