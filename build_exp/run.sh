echo == PF8; python tools/pv_modes.py 2>/dev/null | grep bf16x3 | cut -c1-40
echo == PF4; RANGE_LIB_PATH=$GRAFT_REPO_ROOT/build_exp/librange_PF4.so python tools/pv_modes.py 2>/dev/null | grep bf16x3 | cut -c1-200
