cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/multimodal-registration_amd/csrc
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_net.py -x -q 2>&1 | tail -2
for v in _prev "" _prev "" _prev ""; do
  echo "== lib$v"
  MMR_LIB=$C/libmmr_hip$v.so timeout -k 10 300 python tools/time_upfold.py 2>&1 | grep "forward bf16" | head -3
done
