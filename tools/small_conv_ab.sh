# deep layers of a SMALL batch: two-sweep (5) against six-sweep (9) kernel, sweeps in one workgroup or split over workgroups
cd $GRAFT_REPO_ROOT
for streams in 64 128 256; do
  for v in "0 0" "1 1"; do
    set -- $v
    echo "== $streams streams, 257 frames, STITO_W43S2_SWSPLIT=$1 STITO_W43S3_SWSPLIT=$2 (modes 5 = two sweeps, 9 = six sweeps)"
    STITO_W43S2_SWSPLIT=$1 STITO_W43S3_SWSPLIT=$2 python tools/conv_bench.py --streams $streams --frames 257 --modes 5,9 --reps 5 2>/dev/null | grep -E "^(16x8|8x4|32x16) (512|1024|2048)|->(1024|2048)" 
  done
done
