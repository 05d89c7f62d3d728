set -u
timeout 900 python -m pytest tests/test_pv.py tests/test_gpu_fullsize.py tests/test_gpu_facade.py -m gpu -q -x -k "pv or facade" 2>&1 | tail -2
python tools/pv_ab.py 60 3 sweep 2>&1 | grep "^pv"
python tools/pv_ab.py 60 3 rich 2>&1 | grep "^pv"
STATS_ONLY=1 bash tools/profile_pv.sh ab sweep | grep -E "^pv_(analysis|synthesis)"
