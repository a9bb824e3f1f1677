"""scsfm_hip: MI355X (gfx950) kernels for the SC-SfMLearner warp + loss hot path.

Layout
  csrc/*.hip         hand-written HIP kernels + the C ABI (include/scsfm_hip.h)
  scsfm_hip/_lib.py  ctypes loader of libscsfm_hip.so (fails loudly, no fallback)
  scsfm_hip/capi.py  tensor-level wrappers of the C ABI
  scsfm_hip/ops.py   torch.autograd.Function wrappers used by ../loss_functions.py, ../inverse_warp.py
  scsfm_hip/synth.py seeded synthetic batches (no dataset is reachable)
"""
__all__ = ["_lib", "capi", "synth"]
