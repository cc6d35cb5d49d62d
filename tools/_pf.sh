cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" pf1 pf4 pf16; do
  lib=""; [ -n "$v" ] && lib=$GRAFT_REPO_ROOT/build_variants/liblocgpu_$v.so
  echo "== ${v:-default}"
  LOCGPU_LIB=$lib LAT_ONLY=p2plane_eager timeout 200 python tools/latency_microbench.py 2>/dev/null | grep '^{' | tail -1
done; done
for v in "" pf1 pf4; do
  lib=""; [ -n "$v" ] && lib=$GRAFT_REPO_ROOT/build_variants/liblocgpu_$v.so
  echo "== detail ${v:-default}"
  LOCGPU_LIB=$lib timeout 200 python tools/latency_detail.py 2>/dev/null | tail -12
done
