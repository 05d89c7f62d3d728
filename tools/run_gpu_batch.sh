set -u
timeout 900 python -m pytest tests/test_pv.py -m gpu -q -x 2>&1 | tail -3
for lib in shipped melonix_amd/lib/variants/pv_prev.so; do
  if [ "$lib" = shipped ]; then unset MX_AB_LIB; else export MX_AB_LIB=$lib; fi
  echo "== $lib"
  python tools/pv_ab.py 60 3 sweep 2>&1 | grep "^pv"
  python tools/pv_ab.py 60 3 rich 2>&1 | grep "^pv"
  STATS_ONLY=1 bash tools/profile_pv.sh ab sweep | grep -E "^pv_(analysis|synthesis)"
done
