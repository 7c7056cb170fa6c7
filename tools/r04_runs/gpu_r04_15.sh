cd $GRAFT_REPO_ROOT
bash tools/ab.sh variants/lib_a.so pixelspointspolygons_amd/libp3hip.so --lean
