python - <<'PY'
import torch, os, sys
sys.path.insert(0, '.')
torch.cuda.init()
from frb_baseband_amd import _lib
_lib.load()
libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'libhsa' in l})
print("\n".join(libs))
PY
