export TMPDIR=/tmp
for lib in libndp_hip.so libndp_pf10.so libndp_pf14.so libndp_hip.so libndp_pf10.so; do
echo $lib
NDP_LIB_PATH=$GRAFT_REPO_ROOT/ndivplanning_amd/lib/$lib NDP_FM_SIDE_STREAM=0 N=8 python scripts/probe/fm_time.py 2>&1 | grep "ms/step, \|wgrad\[deconv3\]\|wgrad\[deconv6\]"
NDP_LIB_PATH=$GRAFT_REPO_ROOT/ndivplanning_amd/lib/$lib NDP_FM_SIDE_STREAM=0 N=32 STEPS=10 python scripts/probe/fm_time.py 2>&1 | grep "ms/step, \|wgrad\[deconv3\]\|wgrad\[deconv6\]"
done
