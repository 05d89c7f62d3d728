cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np, sys
sys.path.insert(0,'tests')
from conftest import accum_sweep
accum_sweep(10*48000).tofile('/tmp/audio.f32')
PY
make -s -C melonix_amd/cpp NO_GL=1
g++ -std=c++20 -O1 -DMELONIX_AMD_NO_GL -I melonix_amd/cpp -I include tests/cpp/facade_driver.cpp -o /tmp/facade_driver -L melonix_amd/lib -lmelonix_facade -lmelonix_amd -Wl,-rpath,$PWD/melonix_amd/lib -lpthread
mkdir -p /tmp/fo
MELONIX_TIMING=1 /tmp/facade_driver /tmp/audio.f32 /tmp/fo 32768 2>&1 | grep -E "Spec worker|cold_screen|stage" | tail -20
