cd $GRAFT_REPO_ROOT
export E3DGE_LIB_PATH=$PWD/cvpr23-e3dge_amd/lib/variants/lib_blurnew.so
echo "== new"; timeout 300 python tools/bench_ops.py 2>&1 | grep -i "blur" | cut -c1-220
