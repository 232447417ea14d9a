"""Top-level alias so that `import flash_attn_2_cuda` (flash_attn/flash_attn_interface.py:15 of the
reference) resolves to the gfx950 back-end when this repository root is on sys.path."""
from flash_attention_annotated_amd.flash_attn_2_cuda import bwd, fwd, fwd_kvcache, varlen_bwd, varlen_fwd  # noqa: F401
