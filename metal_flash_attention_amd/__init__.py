"""metal_flash_attention_amd -- MI355X (gfx950) FlashAttention kernel suite behind the host API
of philipturner/metal-flash-attention.  Hand-written HIP kernels + a C-ABI library
(libmfa_hip.so, include/mfa.h); this package is the thin Python mirror of the Swift types.
No CPU / PyTorch compute fallback exists: importing works without the library, using it raises.
"""
from ._abi import LIB_PATH, MFAError  # noqa: F401
from .attention import (  # noqa: F401
    AttentionDescriptor,
    AttentionKernel,
    AttentionKernelDescriptor,
    AttentionKernelType,
    AttentionOperand,
    GEMMOperandPrecision,
    deviceCount,
    deviceName,
    parameterFile,
    resetParameterFiles,
    selectParameterRow,
    setParameterFile,
)
from .gemm import GEMMDescriptor, GEMMKernel, GEMMKernelDescriptor  # noqa: F401,E402
