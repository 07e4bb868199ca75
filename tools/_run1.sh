cd $GRAFT_REPO_ROOT
for v in ngspeciesid_amd/libngsid_hip.so build_alt/libngsid_hip_st8.so build_alt/libngsid_hip_notb.so build_alt/libngsid_hip_st8notb.so; do echo "== $v"; timeout 300 python tools/micro/time_ed2.py $v 400000 2>&1 | grep -E "k_ed_align|pairs" | tail -4; done
