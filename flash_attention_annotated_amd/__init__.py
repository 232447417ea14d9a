"""MI355X-native (gfx950) FlashAttention (forward hot path, backward, KV-cache decode) behind the reference's operator boundary.

Public surface (same names as the reference's `flash_attn` package, flash_attn/__init__.py:1-11):
flash_attn_func, flash_attn_varlen_func, the packed variants and flash_attn_with_kvcache.  The compute path is the
hand-written HIP kernels in csrc/, reached through the C-ABI of include/fa_fwd.h and include/fa_bwd.h.
"""
__version__ = "0.1.0"

from .flash_attn_interface import (  # noqa: F401
    flash_attn_func,
    flash_attn_kvpacked_func,
    flash_attn_qkvpacked_func,
    flash_attn_varlen_func,
    flash_attn_varlen_kvpacked_func,
    flash_attn_varlen_qkvpacked_func,
    flash_attn_with_kvcache,
)
