"""tools/ only: honour LQ_HIP_LIB (another build of the same C ABI: `make -C learned_quantization_amd/csrc dev`, experiment
builds).  The product loader reads no environment variable; the tools that want a development build say so in code, here:

    import _devlib  # noqa: F401   (after learned_quantization_amd is importable, before its first op)
"""
import os

from learned_quantization_amd import _hip

_p = os.environ.get("LQ_HIP_LIB")
if _p:
    _hip.use_library(_p)
    print(f"[tools/_devlib] using {_p}", flush=True)
