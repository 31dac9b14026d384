"""ctypes binding of libjova_hip.so (the C ABI declared in include/jova_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is missing or an entry point fails,
the call raises.  PyTorch only supplies device memory (tensors) and the HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JOVA_LIB") or os.path.join(_HERE, "libjova_hip.so")    # JOVA_LIB: instrumented dev builds

# signature codes: p = pointer, i = int, q = int64, f = float, Q = uint64
_SIGS = {
    "jova_graphconv_fwd": "piqqiipppp" + "ippppp",
    "jova_atoms_to_f32": "pqiipp",
    "jova_graphconv_bwd": "piqqiippp" + "pppp" + "ppppppp",
    "jova_adj_symmetry": "qpppp",
    "jova_collate_gconv": "ippppp" + "i" + "ppppppp" + "pip",
    "jova_graphpool_fwd": "pqipp" + "ppppp",
    "jova_row_table": "qpppppp",
    "jova_block_order": "pqp",
    "jova_graphpool_bwd": "ppqipp" + "ppipppp",
    "jova_adj_reverse_pos": "qpppp",
    "jova_gather2d_fwd": "ppiiipp",
    "jova_gather2d_bwd": "ppiiipp",
    "jova_nonsat_fwd": "pqfipppp" + "p",
    "jova_nonsat_fwd_noalloc": "pqfippp" + "p",
    "jova_mul": "ppqpp",
    "jova_gemm_f32": "pqqpqqpq" + "iiif" + "ppqii" + "ii" + "qqqqqq" + "p",
    "jova_gemm_f32_img": "pqqpqqpq" + "iiif" + "ppqii" + "ii" + "qqqqqq" + "pp",
    "jova_gemm_b_image": "pqqiipp",
    "jova_colstats": "pqipp",
    "jova_batchnorm_fwd": "pqippp" + "ff" + "ppp" + "ippp",
    "jova_batchnorm_bwd": "ppppp" + "qii" + "ppp" + "pp",
    "jova_colsum_add": "pqiqpp",
    "jova_layernorm_fwd": "pppp" + "qif" + "ppp",
    "jova_layernorm_bwd": "ppppp" + "qi" + "pppp",
    "jova_softmax_fwd": "pqiqpp",
    "jova_softmax_bwd": "pppfqiqpp",
    "jova_dropout": "pqiqfQQpp",
    "jova_prot2vec_fwd": "pppp" + "iiiiii" + "ppp",
    "jova_prot2vec_bwd": "ppppp" + "iiiiii" + "pp",
    "jova_viewpool_fwd": "ppiiipp",
    "jova_viewpool_bwd": "ppiiiipp",
    "jova_mse_loss": "pppippp",
    "jova_sgd_step": "pppq" + "ffffi" + "pip",
    "jova_adamax_step": "ppppq" + "ffffffi" + "pip",
    "jova_optim_step": "ipppp" + "qfffi" + "fffff" + "pip",
    "jova_nonfinite_flag": "pqpp",
    "jova_dti_metrics": "pppi" + "dd" + "ppp",
    "jova_dti_metrics_tasks": "pppii" + "dd" + "pppp",
    "jova_copy2d": "pqiqpqiqip", "jova_linear_bwd_weight": "pqpqqiippip",
    "jova_gather_rows": "ppqiqpp", "jova_scatter_add_rows": "ppqiqpp", "jova_gauss_hist_fwd": "pqfpp",
    "jova_gauss_hist_bwd": "ppqfpp",
    "jova_lstm_fwd": "ppiiipppp", "jova_lstm_bwd": "ppppiiipp",
    "jova_embedding_fwd": "ppqiqppp", "jova_embedding_bwd": "ppqiqpp", "jova_rowscale_add": "pppqipp",
    "jova_spmm_csr": "ppppqiqpp",
    "jova_relu_fwd": "pqpp",
    "jova_relu_bwd": "ppqpp",
    "jova_axpy": "pfqpp",
    "jova_zero": "pqp",
    "jova_scale_dev": "ppqpp",
}
_CT = {"p": ctypes.c_void_p, "i": ctypes.c_int, "q": ctypes.c_int64, "f": ctypes.c_float, "Q": ctypes.c_uint64,
       "d": ctypes.c_double}

EXPORTS = sorted(list(_SIGS) + ["jova_last_error", "jova_abi_version", "jova_nonsat_workspace_bytes",
                                "jova_graphconv_workspace_floats", "jova_graphconv_bwd_scratch_floats",
                                "jova_nonsat_set_spin_limit", "jova_nonsat_noalloc_max_n", "jova_gemm_set_bf16x3", "jova_gemm_get_bf16x3",
                                "jova_gemm_b_image_bytes"])

_lib = None


def lib():
    """Loads the shared library once; raises if it has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("jova_amd: %s is missing -- build it with `make -C jova_amd/csrc` "
                               "(there is no CPU fallback for the HIP path)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, sig in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = [_CT[c] for c in sig]
            fn.restype = ctypes.c_int
        L.jova_last_error.restype = ctypes.c_char_p
        L.jova_last_error.argtypes = []
        L.jova_abi_version.restype = ctypes.c_int
        L.jova_nonsat_workspace_bytes.restype = ctypes.c_size_t
        L.jova_nonsat_workspace_bytes.argtypes = [ctypes.c_int64]
        L.jova_graphconv_workspace_floats.restype = ctypes.c_size_t
        L.jova_graphconv_workspace_floats.argtypes = [ctypes.c_int, ctypes.c_int]
        L.jova_graphconv_bwd_scratch_floats.restype = ctypes.c_size_t
        L.jova_graphconv_bwd_scratch_floats.argtypes = []
        L.jova_nonsat_noalloc_max_n.restype = ctypes.c_int64
        L.jova_nonsat_noalloc_max_n.argtypes = []
        L.jova_nonsat_set_spin_limit.restype = None
        L.jova_nonsat_set_spin_limit.argtypes = [ctypes.c_uint64]
        L.jova_gemm_set_bf16x3.restype = None
        L.jova_gemm_set_bf16x3.argtypes = [ctypes.c_int]
        L.jova_gemm_get_bf16x3.restype = ctypes.c_int
        L.jova_gemm_get_bf16x3.argtypes = []
        L.jova_gemm_b_image_bytes.restype = ctypes.c_size_t
        L.jova_gemm_b_image_bytes.argtypes = [_CT[c] for c in "pqqpqqiiiiii"]
        _lib = L
    return _lib


def call(name, *args):
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, L.jova_last_error().decode()))


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous and live on the GPU."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "jova_amd kernels take contiguous GPU tensors"
    return ctypes.c_void_p(t.data_ptr())


def ptr_array(tensors, n=None):
    """Host array of device pointers (the `const T* const*` arguments of the ABI)."""
    n = len(tensors) if n is None else n
    arr = (ctypes.c_void_p * n)()
    for k, t in enumerate(tensors):
        arr[k] = None if t is None else t.data_ptr()
    return arr


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and torch.is_tensor(t) and not t.is_cuda:
            raise RuntimeError("jova_amd: tensors must live on the GPU (the HIP path has no CPU fallback)")
