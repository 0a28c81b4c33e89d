# on-box A/B of two builds of libpam_hip.so on ONE device: A = the shipped build, B = the same sources with -D$1 (built into /tmp).
# usage (through gpurun): bash tools/ab_build.sh PAM_SOME_MACRO 'python tools/bench_block.py --n 20' [rounds]
SRC=$GRAFT_REPO_ROOT/part-aware_measurement_for_3d_pose_estimation_and_tracking_amd/csrc
rm -rf /tmp/csrcB; mkdir -p /tmp/csrcB/x/y; cp $SRC/*.hip $SRC/*.hpp $SRC/Makefile /tmp/csrcB/x/y/; mkdir -p /tmp/csrcB/include; cp $GRAFT_REPO_ROOT/include/pam.h /tmp/csrcB/include/
(cd /tmp/csrcB/x/y && make -j8 EXTRA=-D$1 > /tmp/csrcB/build.log 2>&1) || { tail -5 /tmp/csrcB/build.log; exit 1; }
for r in $(seq 1 ${3:-2}); do
  echo "== A (shipped build)"; eval "$2" 2>&1 | grep -v amdgpu.ids
  echo "== B (-D$1)"; PAM_LIB=/tmp/csrcB/x/y/libpam_hip.so eval "$2" 2>&1 | grep -v amdgpu.ids
done
