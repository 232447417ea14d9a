"""ctypes binding of the C-ABI in include/fa_fwd.h and include/fa_bwd.h (libfa_fwd_gfx950.so).

The shared library is the product: there is no Python/CPU fallback.  If it is
missing or the GPU is absent, every compute entry point raises.
"""
import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libfa_fwd_gfx950.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

FA_ABI_VERSION = 12
FA_FLAG_FA3_WINDOW = 1
FA_FLAG_SDMASK_SIGNED = 2
FA_DTYPE_FP16, FA_DTYPE_BF16, FA_DTYPE_FP8_E4M3 = 0, 1, 2

# every symbol include/fa_fwd.h declares (tests check the .so exports all of them)
EXPORTED_SYMBOLS = (
    "fa_fwd",
    "fa_fwd_validate",
    "fa_fwd_workspace_size",
    "fa_strerror",
    "fa_fwd_params_size",
    "fa_abi_version",
    "fa_fwd_tile_shape",
    "fa_set_default_variant",
    "fa_set_persist_mode",
    "fa_kvcache_append",
    "fa_kvcache_append_params_size",
    "fa_rotary_apply",
    "fa_rotary_params_size",
    "fa_fwd_combine",
    "fa_combine_params_size",
    # include/fa_bwd.h
    "fa_bwd",
    "fa_bwd_validate",
    "fa_bwd_params_size",
)


class FaFwdParams(ctypes.Structure):
    """Field-for-field mirror of `struct fa_fwd_params` (include/fa_fwd.h)."""

    _fields_ = [
        ("abi_version", ctypes.c_uint32),
        ("struct_size", ctypes.c_uint32),
        ("q", ctypes.c_void_p),
        ("k", ctypes.c_void_p),
        ("v", ctypes.c_void_p),
        ("o", ctypes.c_void_p),
        ("softmax_lse", ctypes.c_void_p),
        ("q_batch_stride", ctypes.c_int64),
        ("q_row_stride", ctypes.c_int64),
        ("q_head_stride", ctypes.c_int64),
        ("k_batch_stride", ctypes.c_int64),
        ("k_row_stride", ctypes.c_int64),
        ("k_head_stride", ctypes.c_int64),
        ("v_batch_stride", ctypes.c_int64),
        ("v_row_stride", ctypes.c_int64),
        ("v_head_stride", ctypes.c_int64),
        ("o_batch_stride", ctypes.c_int64),
        ("o_row_stride", ctypes.c_int64),
        ("o_head_stride", ctypes.c_int64),
        ("b", ctypes.c_int32),
        ("seqlen_q", ctypes.c_int32),
        ("seqlen_k", ctypes.c_int32),
        ("h", ctypes.c_int32),
        ("h_k", ctypes.c_int32),
        ("d", ctypes.c_int32),
        ("total_q", ctypes.c_int32),
        ("dtype", ctypes.c_int32),
        ("cu_seqlens_q", ctypes.c_void_p),
        ("cu_seqlens_k", ctypes.c_void_p),
        ("seqused_q", ctypes.c_void_p),
        ("seqused_k", ctypes.c_void_p),
        ("softmax_scale", ctypes.c_float),
        ("softcap", ctypes.c_float),
        ("is_causal", ctypes.c_int32),
        ("window_size_left", ctypes.c_int32),
        ("window_size_right", ctypes.c_int32),
        ("q_descale", ctypes.c_void_p),
        ("k_descale", ctypes.c_void_p),
        ("v_descale", ctypes.c_void_p),
        ("q_descale_batch_stride", ctypes.c_int64),
        ("q_descale_head_stride", ctypes.c_int64),
        ("k_descale_batch_stride", ctypes.c_int64),
        ("k_descale_head_stride", ctypes.c_int64),
        ("v_descale_batch_stride", ctypes.c_int64),
        ("v_descale_head_stride", ctypes.c_int64),
        ("kernel_variant", ctypes.c_int32),
        ("total_k", ctypes.c_int32),
        ("workspace", ctypes.c_void_p),
        ("workspace_bytes", ctypes.c_uint64),
        ("alibi_slopes", ctypes.c_void_p),
        ("alibi_slopes_batch_stride", ctypes.c_int64),
        ("kv_batch_idx", ctypes.c_void_p),
        ("block_table", ctypes.c_void_p),
        ("block_table_batch_stride", ctypes.c_int64),
        ("page_block_size", ctypes.c_int32),
        ("num_splits", ctypes.c_int32),
        ("leftpad_k", ctypes.c_void_p),
        ("p_dropout", ctypes.c_float),
        ("flags", ctypes.c_int32),
        ("rng_state", ctypes.c_void_p),
        ("s_dmask", ctypes.c_void_p),
        ("s_dmask_rows", ctypes.c_int32),
        ("s_dmask_cols", ctypes.c_int32),
        ("s_dmask_block_n", ctypes.c_int32),
        ("attention_chunk", ctypes.c_int32),
        ("d_v", ctypes.c_int32),
        ("reserved_v12", ctypes.c_int32),
    ]


class FaCombineParams(ctypes.Structure):
    """Field-for-field mirror of `struct fa_combine_params` (include/fa_fwd.h)."""

    _fields_ = (
        [("abi_version", ctypes.c_uint32), ("struct_size", ctypes.c_uint32)]
        + [(n, ctypes.c_void_p) for n in ("out_partial", "lse_partial", "out", "softmax_lse")]
        + [(f"op_{s}_stride", ctypes.c_int64) for s in ("split", "batch", "row", "head")]
        + [(f"lp_{s}_stride", ctypes.c_int64) for s in ("split", "batch", "row", "head")]
        + [(f"o_{s}_stride", ctypes.c_int64) for s in ("batch", "row", "head")]
        + [(f"lse_{s}_stride", ctypes.c_int64) for s in ("batch", "row", "head")]
        + [(n, ctypes.c_int32) for n in ("num_splits", "b", "seqlen", "h", "d", "out_dtype")]
    )


class FaKvcacheAppendParams(ctypes.Structure):
    """Field-for-field mirror of `struct fa_kvcache_append_params` (include/fa_fwd.h)."""

    _fields_ = (
        [("abi_version", ctypes.c_uint32), ("struct_size", ctypes.c_uint32)]
        + [(n, ctypes.c_void_p) for n in ("k_new", "v_new", "k_cache", "v_cache")]
        + [(f"{t}_{s}_stride", ctypes.c_int64) for t in ("knew", "vnew", "kcache", "vcache")
           for s in ("batch", "row", "head")]
        + [(n, ctypes.c_int32) for n in ("b", "seqlen_new", "seqlen_cache", "h_k", "d", "reserved")]
        + [("cache_seqlens", ctypes.c_void_p), ("cache_batch_idx", ctypes.c_void_p)]
        + [("block_table", ctypes.c_void_p), ("block_table_batch_stride", ctypes.c_int64),
           ("page_block_size", ctypes.c_int32), ("dtype", ctypes.c_int32)]
        + [("rotary_cos", ctypes.c_void_p), ("rotary_sin", ctypes.c_void_p),
           ("rotary_dim", ctypes.c_int32), ("rotary_interleaved", ctypes.c_int32), ("rotary_seqlens", ctypes.c_void_p)]
    )


class FaRotaryParams(ctypes.Structure):
    """Field-for-field mirror of `struct fa_rotary_params` (include/fa_fwd.h)."""

    _fields_ = (
        [("abi_version", ctypes.c_uint32), ("struct_size", ctypes.c_uint32), ("src", ctypes.c_void_p), ("dst", ctypes.c_void_p)]
        + [(f"{t}_{s}_stride", ctypes.c_int64) for t in ("src", "dst") for s in ("batch", "row", "head")]
        + [(n, ctypes.c_int32) for n in ("b", "s", "h", "d", "dtype", "rotary_dim", "rotary_interleaved", "per_row_positions")]
        + [("rotary_cos", ctypes.c_void_p), ("rotary_sin", ctypes.c_void_p), ("seqlen_offsets", ctypes.c_void_p)]
    )


class FaBwdParams(ctypes.Structure):
    """Field-for-field mirror of `struct fa_bwd_params` (include/fa_bwd.h)."""

    _fields_ = (
        [("abi_version", ctypes.c_uint32), ("struct_size", ctypes.c_uint32)]
        + [(n, ctypes.c_void_p) for n in ("q", "k", "v", "o", "dout", "softmax_lse", "dq", "dk", "dv", "softmax_d")]
        + [(f"{t}_{s}_stride", ctypes.c_int64) for t in ("q", "k", "v", "o", "do", "dq", "dk", "dv")
           for s in ("batch", "row", "head")]
        + [("softmax_d_row_len", ctypes.c_int64)]
        + [(n, ctypes.c_int32) for n in ("b", "seqlen_q", "seqlen_k", "h", "h_k", "d", "total_q", "total_k", "dtype")]
        + [("cu_seqlens_q", ctypes.c_void_p), ("cu_seqlens_k", ctypes.c_void_p)]
        + [("softmax_scale", ctypes.c_float), ("softcap", ctypes.c_float)]
        + [("is_causal", ctypes.c_int32), ("window_size_left", ctypes.c_int32), ("window_size_right", ctypes.c_int32)]
        + [("alibi_slopes", ctypes.c_void_p), ("alibi_slopes_batch_stride", ctypes.c_int64)]
        + [("deterministic", ctypes.c_int32), ("p_dropout", ctypes.c_float), ("rng_state", ctypes.c_void_p)]
        + [("flags", ctypes.c_int32), ("d_v", ctypes.c_int32)]
    )


HASH_PATH = os.path.join(_HERE, "libfa_fwd_gfx950.srchash")


def _deps():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp"))) + [
        os.path.join(INCLUDE, "fa_fwd.h"), os.path.join(INCLUDE, "fa_bwd.h")]


def source_hash():
    """sha256 over the sources the library is built from (what build() records next to the library)."""
    import hashlib
    h = hashlib.sha256()
    for path in _deps():
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()


def is_stale():
    """True when the built library is missing or was not built from the sources in the tree (content hash, not mtimes:
    the tree is copied to the GPU box)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH) or not os.path.exists(binding_path()):
        return True
    return open(HASH_PATH).read().strip() != source_hash()


def build(force=False, verbose=False):
    """Compile csrc/ for gfx950 into the in-tree shared library (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, "fa_fwd_api.hip"), os.path.join(CSRC, "fa_bwd_api.hip")]
    if not force and not is_stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", INCLUDE, "-I", CSRC, *srcs, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    build_binding(verbose)
    with open(HASH_PATH, "w") as f:
        f.write(source_hash() + "\n")
    return LIB_PATH


def binding_path():
    import sysconfig
    return os.path.join(_HERE, "flash_attn_2_cuda_C" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_binding(verbose=False):
    """The compiled `flash_attn_2_cuda` surface (csrc/torch_binding.cpp): host-only C++ over the C-ABI, plain g++ against
    the torch headers (no hipify: there is no device code in it), linked to the library next to it."""
    import sysconfig
    import torch
    tdir = os.path.dirname(torch.__file__)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(CSRC, "torch_binding.cpp"),
           "-DTORCH_EXTENSION_NAME=flash_attn_2_cuda_C", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-I", INCLUDE, "-I", os.path.join(tdir, "include"), "-I", os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
           "-I", "/opt/rocm/include", "-I", sysconfig.get_paths()["include"],
           "-L", os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-ltorch_python",
           "-L", _HERE, f"-l:{LIB_NAME}", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{os.path.join(tdir, 'lib')}",
           "-o", binding_path()]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return binding_path()


_lib = None


def load():
    """dlopen the library and declare prototypes.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_NAME} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(expected at {LIB_PATH}); there is no CPU fallback")
    lib = ctypes.CDLL(os.environ.get("FA_FWD_LIB", LIB_PATH))  # FA_FWD_LIB: developer override (ablation builds)
    lib.fa_fwd.argtypes = [ctypes.POINTER(FaFwdParams), ctypes.c_void_p]
    lib.fa_fwd.restype = ctypes.c_int
    lib.fa_fwd_validate.argtypes = [ctypes.POINTER(FaFwdParams)]
    lib.fa_fwd_validate.restype = ctypes.c_int
    lib.fa_fwd_workspace_size.argtypes = [ctypes.POINTER(FaFwdParams)]
    lib.fa_fwd_workspace_size.restype = ctypes.c_int64
    lib.fa_strerror.argtypes = [ctypes.c_int]
    lib.fa_strerror.restype = ctypes.c_char_p
    lib.fa_fwd_params_size.argtypes = []
    lib.fa_fwd_params_size.restype = ctypes.c_uint32
    lib.fa_abi_version.argtypes = []
    lib.fa_abi_version.restype = ctypes.c_uint32
    lib.fa_fwd_tile_shape.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                      ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    lib.fa_fwd_tile_shape.restype = ctypes.c_int
    lib.fa_set_default_variant.argtypes = [ctypes.c_int32]
    lib.fa_set_default_variant.restype = None
    lib.fa_set_persist_mode.argtypes = [ctypes.c_int32]
    lib.fa_set_persist_mode.restype = None
    lib.fa_kvcache_append.argtypes = [ctypes.POINTER(FaKvcacheAppendParams), ctypes.c_void_p]
    lib.fa_kvcache_append.restype = ctypes.c_int
    lib.fa_kvcache_append_params_size.argtypes = []
    lib.fa_kvcache_append_params_size.restype = ctypes.c_uint32
    if lib.fa_kvcache_append_params_size() != ctypes.sizeof(FaKvcacheAppendParams):
        raise RuntimeError("fa_kvcache_append_params layout mismatch between include/fa_fwd.h and _lib")
    lib.fa_rotary_apply.argtypes = [ctypes.POINTER(FaRotaryParams), ctypes.c_void_p]
    lib.fa_rotary_apply.restype = ctypes.c_int
    lib.fa_rotary_params_size.argtypes = []
    lib.fa_rotary_params_size.restype = ctypes.c_uint32
    if lib.fa_rotary_params_size() != ctypes.sizeof(FaRotaryParams):
        raise RuntimeError("fa_rotary_params layout mismatch between include/fa_fwd.h and _lib")
    lib.fa_fwd_combine.argtypes = [ctypes.POINTER(FaCombineParams), ctypes.c_void_p]
    lib.fa_fwd_combine.restype = ctypes.c_int
    lib.fa_combine_params_size.argtypes = []
    lib.fa_combine_params_size.restype = ctypes.c_uint32
    if lib.fa_combine_params_size() != ctypes.sizeof(FaCombineParams):
        raise RuntimeError("fa_combine_params layout mismatch between include/fa_fwd.h and _lib")
    lib.fa_bwd.argtypes = [ctypes.POINTER(FaBwdParams), ctypes.c_void_p]
    lib.fa_bwd.restype = ctypes.c_int
    lib.fa_bwd_validate.argtypes = [ctypes.POINTER(FaBwdParams)]
    lib.fa_bwd_validate.restype = ctypes.c_int
    lib.fa_bwd_params_size.argtypes = []
    lib.fa_bwd_params_size.restype = ctypes.c_uint32
    if lib.fa_bwd_params_size() != ctypes.sizeof(FaBwdParams):
        raise RuntimeError("fa_bwd_params layout mismatch between include/fa_bwd.h and _lib.FaBwdParams")
    if lib.fa_fwd_params_size() != ctypes.sizeof(FaFwdParams):
        raise RuntimeError("fa_fwd_params layout mismatch between include/fa_fwd.h and _lib.FaFwdParams")
    if lib.fa_abi_version() != FA_ABI_VERSION:
        raise RuntimeError("fa_fwd ABI version mismatch")
    if os.environ.get("FA_FWD_VARIANT"):  # developer override of the kernel shape (never changes results)
        lib.fa_set_default_variant(int(os.environ["FA_FWD_VARIANT"]))
    if os.environ.get("FA_FWD_PERSIST"):  # developer override: -1 never / 1 always the persistent 256-row kernel (never changes results)
        lib.fa_set_persist_mode(int(os.environ["FA_FWD_PERSIST"]))
    _lib = lib
    return lib


def strerror(status):
    return load().fa_strerror(status).decode()


def new_bwd_params():
    p = FaBwdParams()
    p.abi_version = FA_ABI_VERSION
    p.struct_size = ctypes.sizeof(FaBwdParams)
    return p


def new_params():
    p = FaFwdParams()
    p.abi_version = FA_ABI_VERSION
    p.struct_size = ctypes.sizeof(FaFwdParams)
    return p
