"""rmcl_amd: MI355X-native hot path of the RMCL training step behind the reference's
``vilt.modules.ViLTransformerSS`` / ``training_step`` API.  All compute runs in
``lib/librmcl_hip.so`` (hand-written HIP for gfx950, C ABI in ``include/rmcl.h``); PyTorch is
used for device memory, streams and ``torch.distributed`` only.  There is no CPU fallback:
importing :mod:`rmcl_amd.runtime` fails loudly when the HIP library is missing."""
__version__ = "0.1.0"
