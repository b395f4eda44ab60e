"""ctypes binding of libzstd_amd.so (include/zstd_amd.h).

There is deliberately no fallback: if the HIP library is missing, importing the engine raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZSTD_AMD_LIB") or os.path.join(_HERE, "libzstd_amd.so")   # override: A/B builds of the same ABI

c_size_p = ctypes.POINTER(ctypes.c_size_t)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "zstd_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU implementation to fall back to." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    S, V, I = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int
    sig = {
        "zsa_compressBound": (S, [S]),
        "zsa_compress": (S, [V, S, V, S, I]),
        "zsa_decompress": (S, [V, S, V, S]),
        "zsa_getFrameContentSize": (ctypes.c_ulonglong, [V, S]),
        "zsa_isError": (ctypes.c_uint, [S]),
        "zsa_getErrorName": (ctypes.c_char_p, [S]),
        "zsa_createCDict": (V, [V, S, I]),
        "zsa_freeCDict": (S, [V]),
        "zsa_createDDict": (V, [V, S]),
        "zsa_freeDDict": (S, [V]),
        "zsa_compress_usingCDict": (S, [V, S, V, S, V]),
        "zsa_decompress_usingDDict": (S, [V, S, V, S, V]),
        "zsa_compress_batch": (S, [c_void_pp, c_size_p, c_void_pp, c_size_p, c_size_p, S, I, V]),
        "zsa_decompress_batch": (S, [c_void_pp, c_size_p, c_void_pp, c_size_p, c_size_p, S, V]),
        "zsa_batch_create": (V, [S]),
        "zsa_batch_free": (None, [V]),
        "zsa_batch_set_frame": (S, [V, S, V, S, V, S]),
        "zsa_batch_set_frames_strided": (S, [V, S, V, S, c_size_p, V, S, S]),
        "zsa_batch_link_src_sizes": (S, [V, V, V]),
        "zsa_batch_upload": (S, [V, S, V]),
        "zsa_compress_enqueue": (S, [V, I, V, V]),
        "zsa_decompress_enqueue": (S, [V, V, V]),
        "zsa_batch_results": (S, [V, c_size_p, V]),
        "zsa_batch_device_results": (V, [V]),
        "zsa_batch_export_sizes_i32": (S, [V, V, V]),
        "zsa_decompress_frames_noalloc": (S, [V, V, S, V, V]),
        "zsa_thread_release": (None, []),
        "zsa_noalloc_prepare": (S, []),
        "zsa_engine_failure": (I, []),
        "zsa_host_alloc": (V, [S]),
        "zsa_host_free": (None, [V]),
        "zsa_batch_last_launch": (None, [V, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]),
        "zsa_batch_profile": (S, [V, I, ctypes.POINTER(ctypes.c_uint64), V]),
        "zsa_getCParams": (I, [I, ctypes.c_ulonglong, S, I, ctypes.POINTER(ctypes.c_uint)]),
        "zsa_device_count": (I, []),
        "zsa_version": (ctypes.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError here == the ABI in include/zstd_amd.h is not fully exported
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


lib, EXPORTS = _load()
