"""ctypes binding of libmi355_isdf.so (include/mi355_isdf.h).

Mirrors the reference's loader idiom (pyscf/lib/misc.py:91-104 ``load_library`` + raw
``ctypes.c_void_p`` pointers, e.g. pyscf/pbc/gto/eval_gto.py:140-151).  There is no CPU fallback:
if the shared library is missing or no GPU is visible, ``load()`` / ``Handle()`` raise.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libmi355_isdf.so')
ABI_VERSION = 19

_lib = None

c_int, c_i64, c_dbl, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p

# name -> (restype, argtypes); every symbol declared in include/mi355_isdf.h
SIGNATURES = {
    'isdf_abi_version': (c_int, []),
    'isdf_create': (c_int, [c_int, ctypes.POINTER(c_vp)]),
    'isdf_destroy': (c_int, [c_vp]),
    'isdf_set_stream': (c_int, [c_vp, c_vp]),
    'isdf_last_error': (ctypes.c_char_p, [c_vp]),
    'isdf_workspace_bytes': (c_i64, [c_vp]),
    'isdf_set_coulomb_omega': (c_int, [c_vp, c_dbl]),
    'isdf_set_coulomb_cutoff': (c_int, [c_vp, c_dbl]),
    'isdf_set_coulomb_ws': (c_int, [c_vp, c_dbl, c_vp, c_vp, c_vp, c_vp]),
    'isdf_set_option': (c_int, [c_vp, ctypes.c_char_p, c_int]),
    'isdf_release_workspace': (c_int, [c_vp]),
    'isdf_prof_enable': (c_int, [c_vp, c_int]),
    'isdf_prof_reset': (c_int, [c_vp]),
    'isdf_prof_count': (c_int, [c_vp]),
    'isdf_prof_get': (c_int, [c_vp, c_int, ctypes.c_char_p, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl)]),
    'isdf_eval_ao': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_i64]),
    'isdf_block_row_absmax': (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_vp, c_vp]),
    'isdf_gather_cols': (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_i64, c_vp, c_i64]),
    'isdf_partition_by_atom': (c_int, [c_vp, c_vp, c_i64, c_vp, c_int, c_vp, c_dbl, c_vp]),
    'isdf_select_ip': (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_dbl, c_dbl, c_vp, c_i64, c_vp, c_vp]),
    'isdf_select_ip_gram': (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_dbl, c_dbl, c_int, c_vp, ctypes.POINTER(ctypes.c_int32)]),
    'isdf_fit_from_chol': (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_vp]),
    'isdf_fit_prepare': (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_int, c_dbl, c_vp, c_vp, ctypes.POINTER(c_dbl)]),
    'isdf_fit_apply': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_i64, c_i64, c_int, c_vp, c_i64]),
    'isdf_gather_aoP': (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_int, c_vp]),
    'isdf_gram_sq': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    'isdf_pair_gram_rows': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_i64, c_vp, c_i64]),
    'isdf_gram_prod': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    'isdf_pair_prod_rows': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64]),
    'isdf_factor_solve_half': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_i64, c_i64]),
    'isdf_block_chol': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_dbl, c_vp, ctypes.POINTER(c_dbl)]),
    'isdf_block_solve': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_int, c_int, c_vp, c_i64, c_i64]),
    'isdf_block_invert': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    'isdf_block_apply': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_i64]),
    'isdf_pair_rows_block_apply': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_i64, c_i64, c_vp, c_i64, c_int, c_vp, c_vp, c_i64]),
    'isdf_shift_diag': (c_int, [c_vp, c_vp, c_int, c_dbl]),
    'isdf_chol_inplace': (c_int, [c_vp, c_vp, c_int, c_dbl, c_vp, ctypes.POINTER(c_dbl)]),
    'isdf_factor_solve': (c_int, [c_vp, c_vp, c_int, c_vp, c_i64, c_i64]),
    'isdf_bj_probe_vectors': (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_vp]),
    'isdf_rows_combine': (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_vp, c_i64, c_i64, c_vp, c_i64, c_int]),
    'isdf_bj_probe_rows': (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64]),
    'isdf_gather_T': (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_vp]),
    'isdf_W_from_factor': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_i64]),
    'isdf_fit_global': (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_vp, c_int, c_dbl, c_vp, c_i64, c_vp, ctypes.POINTER(c_dbl)]),
    'isdf_coulomb_W': (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_i64]),
    'isdf_coulomb_rows': (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_vp, c_int, c_vp, c_i64]),
    'isdf_symmetrize_upper': (c_int, [c_vp, c_vp, c_int, c_i64]),
    'isdf_symmetrize_mean': (c_int, [c_vp, c_vp, c_int, c_i64, c_int]),
    'isdf_get_j': (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_vp, c_vp, c_vp, c_int, c_vp]),
    'isdf_rho': (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_vp, c_int, c_vp, c_i64]),
    'isdf_coulomb_potential': (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_vp]),
    'isdf_vj_from_vR': (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_vp, c_int, c_i64, c_vp]),
    'isdf_eval_ao_k': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_i64]),
    'isdf_select_ip_cplx': (c_int, [c_vp, c_vp, c_int, c_int, c_i64, c_int, c_vp, c_vp, c_dbl, c_dbl, c_vp, c_i64, c_vp, c_vp]),
    'isdf_fit_prepare_cplx': (c_int, [c_vp, c_vp, c_int, c_int, c_i64, c_vp, c_int, c_dbl, c_vp, c_vp, ctypes.POINTER(c_dbl)]),
    'isdf_fit_apply_cplx': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_i64, c_int, c_vp, c_i64]),
    'isdf_coulG_q': (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_dbl, c_vp]),
    'isdf_coulomb_Wq': (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_vp, c_dbl, c_int, c_int, c_int, c_int, c_vp, c_vp, c_i64]),
    'isdf_symmetrize_hermitian': (c_int, [c_vp, c_vp, c_vp, c_int, c_i64]),
    'isdf_finish_Wq': (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_vp, c_vp]),
    'isdf_get_k_pair': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_dbl, c_vp]),
    'isdf_get_k_exact_kpt': (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_vp, c_vp, c_int, c_i64, c_vp, c_vp, c_dbl, c_int, c_int, c_int, c_vp, c_vp]),
    'isdf_coulG_half': (c_int, [c_vp, c_vp, c_vp, c_vp]),
    'isdf_spectral_supported': (c_int, [c_vp, c_vp, c_int, ctypes.POINTER(c_int)]),
    'isdf_spectral_rows': (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_i64]),
    'isdf_coulomb_rows_q': (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_vp]),
    'isdf_nyquist_spectra': (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_int, c_vp, c_vp]),
    'isdf_zhadamard_planes': (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_i64]),
    'isdf_rho_k': (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_vp, c_vp, c_dbl, c_vp]),
    'isdf_vj_k': (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_vp, c_vp, c_vp]),
    'isdf_pp_local_potential': (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'isdf_pp_projector_overlaps': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    'isdf_gemm_nn': (c_int, [c_vp, c_int, c_i64, c_int, c_dbl, c_vp, c_i64, c_vp, c_i64, c_dbl, c_vp, c_i64]),
    'isdf_hadamard_rows': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_i64]),
    'isdf_eval_ao_deriv1': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64]),
    'isdf_eval_ao_k_deriv1': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64]),
    'isdf_uniform_grid': (c_int, [c_vp, c_vp, c_vp, c_vp]),
    'isdf_rho_pair': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_i64, c_i64, c_vp, c_int, c_vp, c_i64]),
    'isdf_mg_embed_density': (c_int, [c_vp, c_vp, c_int, c_vp, c_dbl, c_vp, c_vp, c_int]),
    'isdf_mg_restrict_potential': (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_dbl, c_vp]),
    'isdf_mg_coulomb_kernel': (c_int, [c_vp, c_vp, c_int, c_vp, c_vp]),
    'isdf_lda_exchange': (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    'isdf_lda_vwn_add': (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    'isdf_gga_b88': (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64]),
    'isdf_lda_exchange_fxc': (c_int, [c_vp, c_vp, c_i64, c_vp]),
    'isdf_dot': (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    'isdf_gemm_nt': (c_int, [c_vp, c_int, c_int, c_i64, c_dbl, c_vp, c_i64, c_vp, c_i64, c_vp, c_dbl, c_vp, c_i64]),
    'isdf_get_k_exact': (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    'isdf_get_k': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_i64, c_int, c_int, c_vp, c_int, c_vp]),
}


def load():
    """Load the shared library (once).  torch is imported first so that the HIP runtime, rocBLAS,
    rocSOLVER and hipFFT resolve to the single copy the process already holds."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'libmi355_isdf.so not found at %s — build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` or `make -C pyscf_isdf_amd/csrc`.  There is no CPU fallback.' % LIB_PATH)
    try:
        import torch  # noqa: F401  (side effect: loads libamdhip64 & co.)
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    v = lib.isdf_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError('libmi355_isdf.so ABI version %d != binding version %d' % (v, ABI_VERSION))
    _lib = lib
    return lib


class IsdfError(RuntimeError):
    pass


class Handle:
    """Owns one isdf_handle (one GPU)."""

    def __init__(self, device=0):
        self.lib = load()
        h = c_vp()
        rc = self.lib.isdf_create(int(device), ctypes.byref(h))
        if rc != 0 or not h.value:
            raise IsdfError('isdf_create(device=%d) failed with status %d (no MI355X visible?)' % (device, rc))
        self.h = h
        self.device = device

    def call(self, name, *args):
        rc = getattr(self.lib, name)(self.h, *args)
        if rc != 0:
            msg = self.lib.isdf_last_error(self.h)
            raise IsdfError('%s failed (status %d): %s' % (name, rc, msg.decode() if msg else ''))

    def close(self):
        if getattr(self, 'h', None) is not None and self.h.value:
            self.lib.isdf_destroy(self.h)
            self.h = c_vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
