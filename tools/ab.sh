# on-box A/B of two builds of pam_conv.hip: A = default, B = with $1 defined.  usage: bash tools/ab.sh PAM_SOMETHING [crops]
cd $GRAFT_REPO_ROOT/part-aware_measurement_for_3d_pose_estimation_and_tracking_amd/csrc
for v in A B A B; do
  rm -f pam_conv.o libpam_hip.so
  if [ $v = A ]; then make -j8 > /dev/null 2>&1; else make -j8 EXTRA=-D$1 > /dev/null 2>&1; fi
  echo "variant $v (B = -D$1)"
  for i in 1 2 3; do python $GRAFT_REPO_ROOT/tools/bench_hrnet.py --backends hip --iters 30 --n ${2:-20} | tail -1; done
done
