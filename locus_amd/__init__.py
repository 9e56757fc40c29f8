"""locus_amd: the MI355X-native GICP hot path of LOCUS behind include/locus_hip.h (ctypes bindings in capi.py)."""
import os

# The scheduler overlaps up to sixteen groups of pairs, one HIP stream each; the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware
# queues (four by default) and reads the variable once, when it initialises -- which `import torch` may already do.  Importing this package
# first therefore sets the default the library was measured with (locus_amd/csrc/lh_api.hip lh_runtime_defaults: 9.7 k pairs/s at 4 queues,
# 12.2 k at 24); a value the deployment set is kept.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
