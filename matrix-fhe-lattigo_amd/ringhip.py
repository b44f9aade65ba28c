"""ctypes binding of libringhip.so + thin mirror of the reference's ring.Ring API.

Names follow the reference (ring/ring.go, ring/ntt.go, ring/operations.go, ring/basis_extension.go):
Ring.NTT / NTTLazy / INTT / INTTLazy, Ring.Add / Sub / Neg / Reduce / MulCoeffsMontgomery / ..., SubRing.NTT etc.
Polynomials live on the device as (npoly, level+1, N) uint64 blocks (DevicePoly)."""
import ctypes as C
import os
import re
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
U64P = C.POINTER(C.c_uint64)
# rh_allgather_fn (ringhip.h): (ctx, send_dev, recv_dev, send_words, hip_stream) -> 0 on success
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

Standard = 0   # ring.Standard (ring/ring.go Type)
ConjugateInvariant = 1   # ring.ConjugateInvariant, Z[X+X^-1]/(X^2N+1)
Matrix3N = 2   # 3N-cyclotomic ring (ring.Matrix, ring/ring.go:299-304)


class RingHipError(RuntimeError):
    """Raised for any non-zero rh_status; the Go wrapper panics / returns error at the same places."""


def library_path():
    """RINGHIP_LIB overrides the in-tree build (A/B runs of kernel variants); there is no other search path."""
    return os.environ.get("RINGHIP_LIB") or os.path.join(_HERE, "lib", "libringhip.so")


_lib = None


def _opcodes():
    txt = open(os.path.join(_ROOT, "include", "ringhip_ops.h")).read()
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"RH_OP_([A-Z0-9_]+)\s*=\s*(\d+)", txt)}


OPS = _opcodes()
# opcodes that read their output operand as well (z = f(x, y, z)): its layout tag takes part like an input's
_OPS_READ_Z = {v for k, v in OPS.items() if ("THEN_ADD" in k or "THEN_SUB" in k) and k != "MUL_SCALAR_MONT_THEN_ADD_SCALAR"}


def _share_hip_runtime_with_torch():
    """One HIP/HSA runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7).
    When torch is imported first, this library's NEEDED libamdhip64.so.7 binds to torch's copy by SONAME and all is
    well; when this library is loaded first it pulls in /opt/rocm's copy, torch later loads its own next to it, and
    the second runtime finds no GPU.  So: if torch is installed but not yet imported, load torch's libamdhip64.so
    first -- the same binding as in the other order.  Nothing is imported from torch; no torch -> nothing to do."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand)
        except OSError:
            pass


def lib():
    """Loads the HIP library.  No fallback: a missing library is a hard error."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RingHipError("libringhip.so not built (%s): run `make` or __graft_entry__.build()" % path)
    _share_hip_runtime_with_torch()
    L = C.CDLL(path)
    i, vp, sz = C.c_int, C.c_void_p, C.c_size_t
    sig = {
        "rh_last_error": (C.c_char_p, []),
        "rh_device_count": (i, []),
        "rh_ring_create": (i, [C.POINTER(vp), i, i, i, i, U64P, U64P, U64P, U64P, U64P, U64P, U64P]),
        "rh_ring_create_auto": (i, [C.POINTER(vp), i, i, i, i, U64P, U64P]),
        "rh_ring_destroy": (None, [vp]),
        "rh_ring_n": (i, [vp]), "rh_ring_limbs": (i, [vp]),
        "rh_ring_get_constants": (i, [vp, U64P, U64P, U64P, U64P, U64P, U64P, U64P]),
        "rh_ring_set_stream": (i, [vp, vp]), "rh_ring_sync": (i, [vp]), "rh_ring_reserve": (i, [vp, i]),
        "rh_dev_alloc": (i, [vp, sz, C.POINTER(vp)]), "rh_dev_free": (i, [vp, vp]),
        "rh_dev_upload": (i, [vp, vp, U64P, sz]), "rh_dev_download": (i, [vp, U64P, vp, sz]),
        "rh_ring_copy_rows": (i, [vp, vp, i, vp, i, i, i]),
        "rh_ring_vec_op_bcast": (i, [vp, i, vp, i, vp, vp, i, i, i]), "rh_ring_vec_op_halves": (i, [vp, i, vp, vp, i, i, U64P, U64P]),
        "rh_ring_shift": (i, [vp, i, vp, vp, i, i]), "rh_ring_automorphism_ntt_index": (i, [vp, i, vp, vp, vp, i, i]), "rh_ring_mult_by_monomial": (i, [vp, i, vp, vp, i, i]),
        "rh_ring_unfold_ci_to_standard": (i, [vp, i, vp, vp, i]), "rh_ring_fold_standard_to_ci": (i, [vp, i, vp, vp, vp, i]),
        "rh_ring_pad_default_to_ci": (i, [vp, i, vp, i, vp, i]),
        "rh_ntt_forward": (i, [vp, i, U64P, U64P]), "rh_ntt_forward_lazy": (i, [vp, i, U64P, U64P]),
        "rh_ntt_backward": (i, [vp, i, U64P, U64P]), "rh_ntt_backward_lazy": (i, [vp, i, U64P, U64P]),
        "rh_ntt_poly_forward": (i, [vp, i, C.POINTER(vp), C.POINTER(vp), i]), "rh_ntt_poly_backward": (i, [vp, i, C.POINTER(vp), C.POINTER(vp), i]),
        "rh_host_alloc": (i, [sz, C.POINTER(vp)]), "rh_host_free": (i, [vp]), "rh_host_register": (i, [vp, sz]), "rh_host_unregister": (i, [vp]),
        "rh_ring_ntt": (i, [vp, vp, vp, i, i, i]), "rh_ring_intt": (i, [vp, vp, vp, i, i, i]),
        "rh_ring_ntt_phase": (i, [vp, vp, vp, i, i, i, i]),
        "rh_ring_ntt3n_block_order_supported": (i, [vp]), "rh_ring_ntt_layout": (i, [vp, vp, i, vp, i, i, i, i, i]),
        "rh_ring_div_by_last_modulus_many_ntt_layout": (i, [vp, i, i, i, vp, vp, i, i, i]),
        "rh_ring_stats": (i, [vp, C.c_char_p, C.POINTER(C.c_long)]),
        "rh_ring_ntt_rows": (i, [vp, vp, i, vp, i, i, i, i]), "rh_ring_intt_rows": (i, [vp, vp, i, vp, i, i, i, i]),
        "rh_ring_vec_op_rows": (i, [vp, i, vp, i, vp, i, vp, i, i, i, U64P, U64P]),
        "rh_ring_intt_mul": (i, [vp, vp, vp, vp, i, i]),
        "rh_ring_polymul": (i, [vp, vp, vp, vp, i, i]),
        "rh_ring_ntt_many": (i, [vp, vp, vp, vp, i, i]),
        "rh_ring_ntt3n_reorder": (i, [vp, vp, vp, i, i, i]),
        "rh_ring_set_tuning": (i, [vp, C.c_char_p, C.c_long]),
        "rh_ring_vec_op": (i, [vp, i, vp, vp, vp, i, i, U64P, U64P]),
        "rh_ring_div_by_last_modulus_many": (i, [vp, i, i, i, vp, vp, i, i]),
        "rh_ring_div_by_last_modulus_many_ntt": (i, [vp, i, i, i, vp, vp, i, i]),
        "rh_ring_automorphism_ntt": (i, [vp, i, vp, C.c_uint64, vp, i, i]),
        "rh_ring_automorphism": (i, [vp, i, vp, C.c_uint64, vp, i]),
        "rh_ring_tensor_degree1": (i, [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i]),
        "rh_bext_create": (i, [C.POINTER(vp), vp, vp]), "rh_bext_destroy": (None, [vp]), "rh_bext_reserve": (i, [vp, i]),
        "rh_bext_modup_q_to_p": (i, [vp, i, i, vp, vp, i]), "rh_bext_modup_p_to_q": (i, [vp, i, i, vp, vp, i]),
        "rh_bext_moddown_qp_to_q": (i, [vp, i, i, vp, vp, vp, i]),
        "rh_bext_moddown_qp_to_q_ntt": (i, [vp, i, i, vp, vp, vp, i]),
        "rh_bext_moddown_qp_to_p": (i, [vp, i, i, vp, vp, vp, i]),
        "rh_bext_decompose_and_split": (i, [vp, i, i, i, i, vp, vp, vp, i]),
        "rh_bext_gadget_product": (i, [vp, i, i, vp, vp, vp, i, vp, vp, i]),
        "rh_bext_gadget_product_then_add": (i, [vp, i, i, vp, vp, vp, i, vp, vp, vp, vp, i]),
        "rh_bext_gadget_product_coeff": (i, [vp, i, i, vp, vp, vp, i, vp, vp, i]),
        "rh_bext_gadget_product_single_p": (i, [vp, i, i, vp, i, i, C.POINTER(i), vp, vp, i, vp, vp, i]),
        "rh_bext_gadget_product_single_p_lazy": (i, [vp, i, i, vp, i, i, C.POINTER(i), vp, vp, i, i, vp, vp, vp, vp, i]),
        "rh_bext_decompose_ntt": (i, [vp, i, i, vp, i, vp, vp, i]),
        "rh_bext_gadget_product_hoisted": (i, [vp, i, i, vp, vp, vp, vp, i, vp, vp, i]),
        "rh_bext_gadget_product_hoisted_lazy": (i, [vp, i, i, vp, vp, vp, vp, i, vp, vp, vp, vp, i]),
        "rh_bext_moddown_qp_to_q_ntt_pair": (i, [vp, i, i, vp, vp, vp, vp, vp, vp, i]),
        "rh_bext_gadget_product_hoisted_then_add": (i, [vp, i, i, vp, vp, vp, vp, i, vp, vp, vp, vp, i]),
        "rh_kshard_create": (i, [C.POINTER(vp), vp, vp, U64P, i, U64P, i, C.POINTER(i), i, C.POINTER(i), i]),
        "rh_kshard_destroy": (None, [vp]), "rh_kshard_num_digits": (i, [vp]),
        "rh_kshard_digit_range": (i, [vp, i, C.POINTER(i), C.POINTER(i)]),
        "rh_kshard_digit": (i, [vp, i, vp, vp, vp, vp, vp, vp, vp, vp, i]),
        "rh_kshard_product": (i, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i]),
        "rh_kshard_moddown": (i, [vp, vp, vp, vp, i]),
        "rh_kshard_set_world": (i, [vp, i, i, C.POINTER(i)]), "rh_kshard_exchange_words": (i, [vp, i, i, C.POINTER(sz)]),
        "rh_kshard_set_exchange": (i, [vp, vp, sz]),
        "rh_kshard_gadget_product": (i, [vp, vp, vp, vp, vp, vp, i, ALLGATHER_FN, vp, i]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise RingHipError("ringhip status %d: %s" % (rc, lib().rh_last_error().decode()))


def _u64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def _p(a):
    return a.ctypes.data_as(U64P) if a is not None else None


class DevicePoly:
    """Device-resident block of `npoly` polynomials with `limbs` limbs of N words: replaces Poly.Coeffs [][]uint64
    (ring/poly.go:13-24).  Either owns hipMalloc'd memory or wraps an external device pointer (e.g. a torch tensor)."""

    layout = None      # 3N rings: "block" when the block holds NTT-domain data in block order (Ring.ntt3n_layout); None: the reference's order
                       # (or coefficient-domain data).  Set by the Ring methods that produce the block; the host only ever sees reference order.

    def __init__(self, ring, npoly, limbs, ptr=None, owner=None):
        self.ring, self.npoly, self.limbs = ring, int(npoly), int(limbs)
        self.words = self.npoly * self.limbs * ring.N
        self._own = ptr is None
        self._owner = owner
        if ptr is None:
            p = C.c_void_p()
            _check(lib().rh_dev_alloc(ring._h, max(self.words, 1), C.byref(p)))
            ptr = p.value
        self.ptr = int(ptr)

    @classmethod
    def from_numpy(cls, ring, arr):
        arr = _u64(arr)
        if arr.ndim == 2:
            arr = arr[None]
        assert arr.ndim == 3 and arr.shape[2] == ring.N, arr.shape
        p = cls(ring, arr.shape[0], arr.shape[1])
        _check(lib().rh_dev_upload(ring._h, p.ptr, _p(arr), arr.size))
        return p

    @classmethod
    def from_torch(cls, ring, t):
        """wraps an int64/uint64 CUDA tensor of shape (npoly, limbs, N) without copying"""
        assert t.is_cuda and t.is_contiguous() and t.element_size() == 8 and t.dim() == 3 and t.shape[2] == ring.N
        return cls(ring, t.shape[0], t.shape[1], ptr=t.data_ptr(), owner=t)

    def numpy(self):
        """the block on the host -- always in the reference's order: a block-order NTT-domain block (3N rings) is converted on the way out"""
        out = np.empty((self.npoly, self.limbs, self.ring.N), dtype=np.uint64)
        src = self
        if self.layout == "block":
            src = DevicePoly(self.ring, self.npoly, self.limbs)
            _check(lib().rh_ring_ntt3n_reorder(self.ring._h, self.ptr, src.ptr, self.npoly, self.limbs - 1, 1))
        _check(lib().rh_dev_download(self.ring._h, _p(out), src.ptr, out.size))
        return out

    def free(self):
        if self._own and self.ptr:
            lib().rh_dev_free(None, self.ptr)      # the ring handle may already be closed; the free does not need it
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedBuffer:
    """page-locked host words (rh_host_alloc) as a numpy array: backing store for Poly.Coeffs that the engine DMAs without staging"""

    def __init__(self, shape):
        n = int(np.prod(shape))
        p = C.c_void_p()
        _check(lib().rh_host_alloc(n, C.byref(p)))
        self.ptr = p.value
        self.array = np.ctypeslib.as_array((C.c_uint64 * n).from_address(self.ptr)).reshape(shape)

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            lib().rh_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class SubRing:
    """One modulus of a Ring: the NumberTheoreticTransformer seam (ring/ntt.go:17-22, ring/subring_ops.go:235-252).
    Host slices in, host slices out, one limb per call -- the call shape the Go interface has."""

    def __init__(self, ring, idx):
        self.ring, self.idx = ring, idx
        self.N = ring.N
        self.Modulus = int(ring.moduli[idx])

    def _call(self, fn, p1):
        p1 = _u64(p1)
        if p1.size < self.N:
            raise RingHipError("cannot NTT: ensure that len(p1)=%d >= N=%d" % (p1.size, self.N))
        p2 = np.empty(self.N, dtype=np.uint64)
        _check(fn(self.ring._h, self.idx, _p(p1), _p(p2)))
        return p2

    def NTT(self, p1):
        return self._call(lib().rh_ntt_forward, p1)

    def NTTLazy(self, p1):
        return self._call(lib().rh_ntt_forward_lazy, p1)

    def INTT(self, p1):
        return self._call(lib().rh_ntt_backward, p1)

    def INTTLazy(self, p1):
        return self._call(lib().rh_ntt_backward_lazy, p1)


class Ring:
    """ring.Ring (ring/ring.go:76-89) on one MI355X.  NewRing(N, moduli) == Ring(N, moduli);
    NewRingFromType(N, moduli, ring.Matrix) == Ring(N, moduli, kind=Matrix3N).
    `constants` (dict with mred, bred, ninv, roots_fwd, roots_bwd / omega3n) plays the role of the SubRing fields the
    Go shim hands over through rh_ring_create; without it the engine generates them (rh_ring_create_auto)."""

    def __init__(self, N, moduli, kind=Standard, device=0, constants=None, omega3n=None):
        self.N, self.kind, self.device = int(N), kind, device
        self.moduli = _u64(moduli)
        self.L = len(self.moduli)
        h = C.c_void_p()
        if constants is None:
            om = _u64(omega3n) if omega3n is not None else None
            _check(lib().rh_ring_create_auto(C.byref(h), device, kind, self.N, self.L, _p(self.moduli), _p(om)))
        else:
            g = lambda k: _u64(constants[k]) if constants.get(k) is not None else None
            keep = [g(k) for k in ("mred", "bred", "ninv", "roots_fwd", "roots_bwd", "omega3n")]
            _check(lib().rh_ring_create(C.byref(h), device, kind, self.N, self.L, _p(self.moduli), *[_p(a) for a in keep]))
        self._h = h
        self.level = self.L - 1
        self._shared = {"ntt3n_layout": None}     # one dict for the handle and all its AtLevel views
        self.SubRings = [SubRing(self, i) for i in range(self.L)]

    # ---- construction helpers -------------------------------------------------------------------------------
    def constants(self):
        mred = np.zeros(self.L, dtype=np.uint64); bred = np.zeros(2 * self.L, dtype=np.uint64)
        ninv = np.zeros(self.L, dtype=np.uint64)
        out = {"mred": mred, "bred": bred.reshape(self.L, 2), "ninv": ninv}
        if self.kind in (Standard, ConjugateInvariant):
            TN = self.N if self.kind == Standard else 2 * self.N
            rf = np.zeros((self.L, TN), dtype=np.uint64); rb = np.zeros((self.L, TN), dtype=np.uint64)
            _check(lib().rh_ring_get_constants(self._h, None, _p(mred), _p(bred), _p(ninv), _p(rf), _p(rb), None))
            out.update(roots_fwd=rf, roots_bwd=rb)
        else:
            om = np.zeros(self.L, dtype=np.uint64)
            _check(lib().rh_ring_get_constants(self._h, None, _p(mred), _p(bred), None, None, None, _p(om)))
            out.update(omega3n=om)
        return out

    def AtLevel(self, level):
        """view restricted to limbs 0..level (ring/ring.go:194-213): the same engine handle, smaller loops"""
        if level < 0 or level >= self.L:
            raise RingHipError("level %d out of range" % level)
        v = object.__new__(Ring)
        v.__dict__.update(self.__dict__)
        v.level = level
        v._view = True
        return v

    def NewPoly(self, npoly=1):
        return DevicePoly(self, npoly, self.level + 1)

    def set_stream(self, stream_ptr):
        _check(lib().rh_ring_set_stream(self._h, stream_ptr))

    def set_tuning(self, key, value):
        _check(lib().rh_ring_set_tuning(self._h, key.encode(), int(value)))

    def ntt_phase(self, p1, p2, inverse=False, phase=0):
        _check(lib().rh_ring_ntt_phase(self._h, p1.ptr, p2.ptr, p1.npoly, self.level, int(inverse), int(phase)))

    def stats(self, key):
        """rh_ring_stats: how the rows-per-poly (AtLevel) transforms of this handle were served: rows_direct / rows_compacted / rows_poly_by_poly"""
        v = C.c_long()
        _check(lib().rh_ring_stats(self._h, key.encode(), C.byref(v)))
        return v.value

    def sync(self):
        _check(lib().rh_ring_sync(self._h))

    def reserve(self, npoly):
        """pre-size the ring's lazily grown scratch (rescale, 3N workspace) for batches of npoly polys: no allocation afterwards"""
        _check(lib().rh_ring_reserve(self._h, int(npoly)))

    def close(self):
        if getattr(self, "_h", None) and not getattr(self, "_view", False):
            lib().rh_ring_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- NTT (ring/ntt.go:127-152) -----------------------------------------------------------------------------
    def _chk(self, *polys, rows_ok=False):
        """The batched C entry points stride every block by level+1 rows per poly (ringhip.h "Data model").  The reference's
        ring.AtLevel(l) on polys with MORE limbs (ring/ring.go:192-213) goes through the *_rows entry points (NTT, INTT, the
        element-wise family: rows_ok); the other batched calls accept it for a single poly only (leading limbs contiguous) and
        refuse a batch instead of striding wrongly.  Returns True when a poly has more limbs than the view's level."""
        more = False
        for p in polys:
            if p is None:
                continue
            if p.limbs < self.level + 1:
                raise RingHipError("poly has %d limbs, ring level needs %d" % (p.limbs, self.level + 1))
            if p.limbs != self.level + 1:
                more = True
                if p.npoly > 1 and not rows_ok:
                    raise RingHipError("batch of %d polys has %d limbs per poly but the ring view is at level %d: this call strides "
                                       "blocks by level+1 rows, allocate the batch at that level" % (p.npoly, p.limbs, self.level))
        return more

    # ---- 3N rings: the device NTT domain's layout, tagged per block ---------------------------------------------------------
    @property
    def ntt3n_layout(self):
        """None (default): Ring.NTT writes the reference's ascending-totative order (ring/ntt_3n.go:82-109).  "block": Ring.NTT writes BLOCK
        order (no permutation pass: 2 HBM passes per transform instead of 3) and tags its output (DevicePoly.layout); every later Ring call
        reads the tags -- INTT takes either layout, the coefficient-wise calls keep it, a block-order operand that meets a reference-order
        one is converted first, and DevicePoly.numpy() hands the host the reference's order.  Set on a ring or any of its views."""
        return self._shared["ntt3n_layout"]

    @ntt3n_layout.setter
    def ntt3n_layout(self, value):
        if value not in (None, "block"):
            raise RingHipError("ntt3n_layout must be None or 'block'")
        if value == "block" and not (self.kind == Matrix3N and lib().rh_ring_ntt3n_block_order_supported(self._h)):
            raise RingHipError("block order needs a 3N ring with N = 3 * 2^k, k >= 13")
        self._shared["ntt3n_layout"] = value

    def ToReferenceOrder(self, p):
        """a block-order NTT-domain block back into the reference's order, in place (through a temporary); no-op for any other block"""
        if getattr(p, "layout", None) != "block":
            return p
        v = self.AtLevel(p.limbs - 1)
        tmp = DevicePoly(v, p.npoly, p.limbs)
        _check(lib().rh_ring_ntt3n_reorder(self._h, p.ptr, tmp.ptr, p.npoly, p.limbs - 1, 1))
        _check(lib().rh_ring_copy_rows(self._h, p.ptr, p.limbs, tmp.ptr, p.limbs, p.npoly, p.limbs - 1))
        p.layout = None
        return p

    def _lay(self, out, *ins):
        """layout of a coefficient-wise call's output: its operands' common layout; mixed layouts are brought to the reference's first.
        An output block that is also an operand keeps participating through its tag."""
        if self.kind != Matrix3N:
            return
        tags = {getattr(p, "layout", None) for p in ins if p is not None}
        if len(tags) > 1:
            for p in ins:
                if p is not None:
                    self.ToReferenceOrder(p)
            tags = {None}
        lay = tags.pop() if tags else None
        for o in (out if isinstance(out, (list, tuple)) else [out]):
            o.layout = lay

    def _ntt(self, p1, p2, inverse, lazy):
        L = lib()
        more = self._chk(p1, p2, rows_ok=True)
        # 3N rings, tagged path: a forward transform writes the RING's layout (its input is coefficient-domain data, whatever stale tag the
        # block carries), an inverse transform reads the OPERAND's tag
        if self.kind == Matrix3N and ((getattr(p1, "layout", None) == "block") if inverse else (self.ntt3n_layout == "block")):
            _check(L.rh_ring_ntt_layout(self._h, p1.ptr, p1.limbs, p2.ptr, p2.limbs, p1.npoly, self.level, 1 if inverse else 0, 1))
            p2.layout = None if inverse else "block"
            return
        if more:
            f = L.rh_ring_intt_rows if inverse else L.rh_ring_ntt_rows
            _check(f(self._h, p1.ptr, p1.limbs, p2.ptr, p2.limbs, p1.npoly, self.level, lazy))
        else:
            f = L.rh_ring_intt if inverse else L.rh_ring_ntt
            _check(f(self._h, p1.ptr, p2.ptr, p1.npoly, self.level, lazy))
        if self.kind == Matrix3N:
            p2.layout = None

    def NTT(self, p1, p2): self._ntt(p1, p2, False, 0)
    def NTTLazy(self, p1, p2): self._ntt(p1, p2, False, 1)
    def INTT(self, p1, p2): self._ntt(p1, p2, True, 0)
    def INTTLazy(self, p1, p2): self._ntt(p1, p2, True, 1)

    # ---- whole host polys: Ring.NTT(p1, p2 Poly) as the reference's callers issue it (ring/ntt.go:127-152) --------------------
    def _host_poly(self, fn, p1, p2, lazy):
        """p1 / p2: sequences of level+1 host limbs (numpy uint64 arrays of >= N words: Poly.Coeffs); p2[i] may be p1[i]"""
        n = self.level + 1
        if len(p1) < n or len(p2) < n:
            raise RingHipError("cannot NTT: poly has %d / %d limbs, ring level needs %d" % (len(p1), len(p2), n))
        for a in list(p1[:n]) + list(p2[:n]):
            if a.dtype != np.uint64 or not a.flags["C_CONTIGUOUS"] or a.size < self.N:
                raise RingHipError("cannot NTT: ensure that every limb is a contiguous uint64 slice of len >= N=%d" % self.N)
        ins = (C.c_void_p * n)(*[a.ctypes.data for a in p1[:n]])
        outs = (C.c_void_p * n)(*[a.ctypes.data for a in p2[:n]])
        _check(fn(self._h, self.level, ins, outs, lazy))

    def NTTHost(self, p1, p2): self._host_poly(lib().rh_ntt_poly_forward, p1, p2, 0)
    def NTTLazyHost(self, p1, p2): self._host_poly(lib().rh_ntt_poly_forward, p1, p2, 1)
    def INTTHost(self, p1, p2): self._host_poly(lib().rh_ntt_poly_backward, p1, p2, 0)
    def INTTLazyHost(self, p1, p2): self._host_poly(lib().rh_ntt_poly_backward, p1, p2, 1)

    def NTT3NReorder(self, p1, p2, to_reference=True):
        """3N rings: NTT-domain block between block order (tuning ntt3n_block_order) and the Go transformer's order; out of place"""
        self._chk(p1, p2); _check(lib().rh_ring_ntt3n_reorder(self._h, p1.ptr, p2.ptr, p1.npoly, self.level, 1 if to_reference else 0))
        if self.ntt3n_layout == "block":            # tagged use: the output carries its layout (the handle-wide tuning key leaves tags alone)
            p2.layout = None if to_reference else "block"

    def CopyLvl(self, p1, p2):
        """p2 <- limbs 0..level of p1 (Poly.CopyLvl, ring/poly.go), blocks may carry more limbs per poly"""
        if p1.npoly != p2.npoly:
            raise RingHipError("CopyLvl: blocks differ in poly count")
        _check(lib().rh_ring_copy_rows(self._h, p2.ptr, p2.limbs, p1.ptr, p1.limbs, p1.npoly, self.level))
        if self.kind == Matrix3N:
            p2.layout = getattr(p1, "layout", None)

    def NTTMany(self, pairs):
        """Ring.NTT(p1, p2) for every (p1, p2) of `pairs` in one call (rh_ring_ntt_many): one software pipeline through all blocks"""
        n = len(pairs)
        for p1, p2 in pairs:
            self._chk(p1, p2)
        ins = (C.c_void_p * n)(*[p1.ptr for p1, _ in pairs]); outs = (C.c_void_p * n)(*[p2.ptr for _, p2 in pairs])
        cnt = (C.c_int * n)(*[p1.npoly for p1, _ in pairs])
        _check(lib().rh_ring_ntt_many(self._h, ins, outs, cnt, n, self.level))

    def INTTMul(self, p1, p2, p3):
        """p3 = INTT(p1 . p2) for NTT-domain p1, p2: the values of MForm(p1, t); MulCoeffsMontgomery(t, p2, p3); INTT(p3, p3)
        (schemes/ckks/evaluator.go:821-834 + INTT), with the product formed on load inside the inverse transform"""
        self._chk(p1, p2, p3)
        if self.kind == Matrix3N and (getattr(p1, "layout", None) or getattr(p2, "layout", None)):      # tagged operands: the three calls as they are
            self.MForm(p1, p3); self.MulCoeffsMontgomery(p3, p2, p3); self.INTT(p3, p3)
            return
        _check(lib().rh_ring_intt_mul(self._h, p1.ptr, p2.ptr, p3.ptr, p1.npoly, self.level))

    def PolyMul(self, p1, p2, p3):
        """p3 = INTT(NTT(p1) . NTT(p2)) for COEFFICIENT-domain p1, p2 (BASELINE config 3): the canonical values of NTT; NTT; MForm;
        MulCoeffsMontgomery; INTT with the tile-stage middle of all three transforms as one kernel (rh_ring_polymul).  p1 and p2 are consumed
        (they hold column-stage intermediates afterwards).  Raises RingHipError for shapes the fused kernel does not cover."""
        self._chk(p1, p2, p3); _check(lib().rh_ring_polymul(self._h, p1.ptr, p2.ptr, p3.ptr, p1.npoly, self.level))

    # ---- automorphisms (ring/automorphism.go) ---------------------------------------------------------------
    def TensorDegree1(self, a0, a1, b0, b1, c0, c1, c2, mform_first=True):
        """degree-1 x degree-1 tensoring as one kernel: ckks mulRelin (schemes/ckks/evaluator.go:821-834) with mform_first,
        matrix_ckks.Evaluator.Mul (schemes/matrix_ckks/evaluator.go:166-173) without"""
        self._chk(a0, a1, b0, b1, c0, c1, c2)
        self._lay([c0, c1, c2], a0, a1, b0, b1)
        _check(lib().rh_ring_tensor_degree1(self._h, a0.ptr, a1.ptr, b0.ptr, b1.ptr, c0.ptr, c1.ptr, c2.ptr, a0.npoly, self.level,
                                            1 if mform_first else 0))

    def AutomorphismNTT(self, polIn, gen, polOut):
        self._chk(polIn, polOut)
        _check(lib().rh_ring_automorphism_ntt(self._h, self.level, polIn.ptr, int(gen), polOut.ptr, polIn.npoly, 0))

    def AutomorphismNTTThenAddLazy(self, polIn, gen, polOut):
        self._chk(polIn, polOut)
        _check(lib().rh_ring_automorphism_ntt(self._h, self.level, polIn.ptr, int(gen), polOut.ptr, polIn.npoly, 1))

    def _index_table(self, index, bound=1):
        """the lookup table as a device block: a DevicePoly of N words, or a host array uploaded for the call"""
        if isinstance(index, DevicePoly):
            return index                   # a table already on the device is the caller's responsibility, like a Go slice index
        arr = _u64(index).reshape(-1)
        if arr.size != self.N or int(arr.max()) >= bound * self.N:
            # the reference panics (index out of range) on such a table; the kernel would read out of bounds instead
            raise RingHipError("index table needs N = %d entries below %d" % (self.N, bound * self.N))
        return DevicePoly.from_numpy(self.AtLevel(0), arr.reshape(1, 1, self.N))

    def AutomorphismNTTWithIndex(self, polIn, index, polOut):
        """ring/automorphism.go:50-78: polOut[j] = polIn[index[j]] on every limb (index: AutomorphismNTTIndex or any permutation)"""
        self._chk(polIn, polOut)
        t = self._index_table(index)
        _check(lib().rh_ring_automorphism_ntt_index(self._h, self.level, polIn.ptr, t.ptr, polOut.ptr, polIn.npoly, 0))

    def AutomorphismNTTWithIndexThenAddLazy(self, polIn, index, polOut):
        """:82-117: polOut[j] += polIn[index[j]] (no reduction)"""
        self._chk(polIn, polOut)
        t = self._index_table(index)
        _check(lib().rh_ring_automorphism_ntt_index(self._h, self.level, polIn.ptr, t.ptr, polOut.ptr, polIn.npoly, 1))

    def Automorphism(self, polIn, gen, polOut):
        self._chk(polIn, polOut)
        _check(lib().rh_ring_automorphism(self._h, self.level, polIn.ptr, int(gen), polOut.ptr, polIn.npoly))

    # ---- rescale (ring/scaling.go).  Output poly may have level+1 or fewer limbs, like the reference -----------
    def DivFloorByLastModulusMany(self, nbRescales, p0, p1):
        _check(lib().rh_ring_div_by_last_modulus_many(self._h, 0, self.level, nbRescales, p0.ptr, p1.ptr, p1.limbs, p0.npoly))

    def DivRoundByLastModulusMany(self, nbRescales, p0, p1):
        _check(lib().rh_ring_div_by_last_modulus_many(self._h, 1, self.level, nbRescales, p0.ptr, p1.ptr, p1.limbs, p0.npoly))

    def DivFloorByLastModulus(self, p0, p1): self.DivFloorByLastModulusMany(1, p0, p1)
    def DivRoundByLastModulus(self, p0, p1): self.DivRoundByLastModulusMany(1, p0, p1)

    def _div_ntt(self, rnd, nbRescales, p0, p1):
        if self.kind == Matrix3N and getattr(p0, "layout", None) == "block":       # the NTT-domain form on tagged data: the layout goes with the call
            _check(lib().rh_ring_div_by_last_modulus_many_ntt_layout(self._h, rnd, self.level, nbRescales, p0.ptr, p1.ptr, p1.limbs, p0.npoly, 1))
            p1.layout = "block"
            return
        _check(lib().rh_ring_div_by_last_modulus_many_ntt(self._h, rnd, self.level, nbRescales, p0.ptr, p1.ptr, p1.limbs, p0.npoly))
        if self.kind == Matrix3N:
            p1.layout = None

    def DivFloorByLastModulusManyNTT(self, nbRescales, p0, p1): self._div_ntt(0, nbRescales, p0, p1)
    def DivRoundByLastModulusManyNTT(self, nbRescales, p0, p1): self._div_ntt(1, nbRescales, p0, p1)

    def DivFloorByLastModulusNTT(self, p0, p1): self.DivFloorByLastModulusManyNTT(1, p0, p1)
    def DivRoundByLastModulusNTT(self, p0, p1): self.DivRoundByLastModulusManyNTT(1, p0, p1)

    # ---- element-wise (ring/operations.go -> ring/vec_ops.go) --------------------------------------------------
    def vec_op(self, op, p1, p2, p3, s0=None, s1=None):
        code = OPS[op] if isinstance(op, str) else int(op)
        more = self._chk(p1, p2, p3, rows_ok=True)
        if self.kind == Matrix3N:
            self._lay(p3, p1, p2, p3 if code in _OPS_READ_Z else None)
        a = _u64(s0) if s0 is not None else None
        b = _u64(s1) if s1 is not None else None
        if more:
            _check(lib().rh_ring_vec_op_rows(self._h, code, p1.ptr if p1 is not None else None, p1.limbs if p1 is not None else 0,
                                             p2.ptr if p2 is not None else None, p2.limbs if p2 is not None else 0,
                                             p3.ptr, p3.limbs, p3.npoly, self.level, _p(a), _p(b)))
            return
        _check(lib().rh_ring_vec_op(self._h, code, p1.ptr if p1 is not None else None, p2.ptr if p2 is not None else None,
                                    p3.ptr, p3.npoly, self.level, _p(a), _p(b)))

    def Add(self, p1, p2, p3): self.vec_op("ADD", p1, p2, p3)
    def AddLazy(self, p1, p2, p3): self.vec_op("ADD_LAZY", p1, p2, p3)
    def Sub(self, p1, p2, p3): self.vec_op("SUB", p1, p2, p3)
    def SubLazy(self, p1, p2, p3): self.vec_op("SUB_LAZY", p1, p2, p3)
    def Neg(self, p1, p2): self.vec_op("NEG", p1, None, p2)
    def Reduce(self, p1, p2): self.vec_op("REDUCE", p1, None, p2)
    def ReduceLazy(self, p1, p2): self.vec_op("REDUCE_LAZY", p1, None, p2)
    def MulCoeffsBarrett(self, p1, p2, p3): self.vec_op("MUL_BARRETT", p1, p2, p3)
    def MulCoeffsBarrettLazy(self, p1, p2, p3): self.vec_op("MUL_BARRETT_LAZY", p1, p2, p3)
    def MulCoeffsBarrettThenAdd(self, p1, p2, p3): self.vec_op("MUL_BARRETT_THEN_ADD", p1, p2, p3)
    def MulCoeffsBarrettThenAddLazy(self, p1, p2, p3): self.vec_op("MUL_BARRETT_THEN_ADD_LAZY", p1, p2, p3)
    def MulCoeffsMontgomery(self, p1, p2, p3): self.vec_op("MUL_MONT", p1, p2, p3)
    def MulCoeffsMontgomeryLazy(self, p1, p2, p3): self.vec_op("MUL_MONT_LAZY", p1, p2, p3)
    def MulCoeffsMontgomeryThenAdd(self, p1, p2, p3): self.vec_op("MUL_MONT_THEN_ADD", p1, p2, p3)
    def MulCoeffsMontgomeryThenAddLazy(self, p1, p2, p3): self.vec_op("MUL_MONT_THEN_ADD_LAZY", p1, p2, p3)
    def MulCoeffsMontgomeryLazyThenAddLazy(self, p1, p2, p3): self.vec_op("MUL_MONT_LAZY_THEN_ADD_LAZY", p1, p2, p3)
    def MulCoeffsMontgomeryThenSub(self, p1, p2, p3): self.vec_op("MUL_MONT_THEN_SUB", p1, p2, p3)
    def MulCoeffsMontgomeryThenSubLazy(self, p1, p2, p3): self.vec_op("MUL_MONT_THEN_SUB_LAZY", p1, p2, p3)
    def MulCoeffsMontgomeryLazyThenSubLazy(self, p1, p2, p3): self.vec_op("MUL_MONT_LAZY_THEN_SUB_LAZY", p1, p2, p3)
    def MulCoeffsMontgomeryLazyThenNeg(self, p1, p2, p3): self.vec_op("MUL_MONT_LAZY_THEN_NEG", p1, p2, p3)
    def MForm(self, p1, p2): self.vec_op("MFORM", p1, None, p2)
    def MFormLazy(self, p1, p2): self.vec_op("MFORM_LAZY", p1, None, p2)
    def IMForm(self, p1, p2): self.vec_op("IMFORM", p1, None, p2)
    def MulRNSScalarMontgomery(self, p1, scalar, p2): self.vec_op("MUL_SCALAR_MONT", p1, None, p2, s0=scalar)
    def AddScalar(self, p1, scalar, p2): self.vec_op("ADD_SCALAR", p1, None, p2, s0=self._per_limb(scalar))
    def SubScalar(self, p1, scalar, p2): self.vec_op("SUB_SCALAR", p1, None, p2, s0=self._per_limb(scalar))

    # ---- scalar forms of ring/operations.go: the per-limb constant is formed on the host exactly as the reference forms it ----
    def _per_limb(self, scalar):
        """a uint64 scalar handed to every limb as it is (AddScalar / SubScalar :151-155, :186-190), or an RNSScalar (one word per limb)"""
        if np.ndim(scalar) == 0:
            return [int(scalar)] * (self.level + 1)
        return scalar

    def _qs(self):
        return [int(q) for q in self.moduli[:self.level + 1]]

    def _mform_scalars(self, values):
        """MForm(v_i, q_i) = v_i * 2^64 mod q_i for the limbs of this view (ring/modular_reduction.go:11-24)"""
        return [(int(v) << 64) % q for v, q in zip(values, self._qs())]

    def AddScalarBigint(self, p1, scalar, p2):
        """:158-163"""
        self.vec_op("ADD_SCALAR", p1, None, p2, s0=[int(scalar) % q for q in self._qs()])

    def SubScalarBigint(self, p1, scalar, p2):
        """:193-198"""
        self.vec_op("SUB_SCALAR", p1, None, p2, s0=[int(scalar) % q for q in self._qs()])

    def MulScalar(self, p1, scalar, p2):
        """:201-205: MulScalarMontgomery by MForm(scalar)"""
        self.vec_op("MUL_SCALAR_MONT", p1, None, p2, s0=self._mform_scalars([int(scalar)] * (self.level + 1)))

    def MulScalarThenAdd(self, p1, scalar, p2):
        """:208-212: p2 += p1 * scalar"""
        self.vec_op("MUL_SCALAR_MONT_THEN_ADD", p1, None, p2, s0=self._mform_scalars([int(scalar)] * (self.level + 1)))

    def MulScalarThenSub(self, p1, scalar, p2):
        """:223-228: p2 -= p1 * scalar, as p2 += p1 * MForm(q - BRedAdd(scalar))"""
        qs = self._qs()
        self.vec_op("MUL_SCALAR_MONT_THEN_ADD", p1, None, p2, s0=self._mform_scalars([q - int(scalar) % q for q in qs]))

    def MulScalarBigint(self, p1, scalar, p2):
        """:231-237"""
        self.vec_op("MUL_SCALAR_MONT", p1, None, p2, s0=self._mform_scalars([int(scalar) % q for q in self._qs()]))

    def MulScalarBigintThenAdd(self, p1, scalar, p2):
        """:240-247"""
        self.vec_op("MUL_SCALAR_MONT_THEN_ADD", p1, None, p2, s0=self._mform_scalars([int(scalar) % q for q in self._qs()]))

    def _halves(self, op, p1, s_lo, s_hi, p2):
        self._chk(p1, p2)
        if getattr(p1, "layout", None) == "block":      # "the first / second N/2 coefficients" means positions of the reference's order
            self.ToReferenceOrder(p1)
        self._lay(p2, p1)
        a, b = _u64(s_lo), _u64(s_hi)
        _check(lib().rh_ring_vec_op_halves(self._h, OPS[op], p1.ptr, p2.ptr, p1.npoly, self.level, _p(a), _p(b)))

    def AddDoubleRNSScalar(self, p1, scalar0, scalar1, p2):
        """:167-173: scalar0 on coefficients [0, N/2), scalar1 on [N/2, N)"""
        self._halves("ADD_SCALAR", p1, scalar0, scalar1, p2)

    def SubDoubleRNSScalar(self, p1, scalar0, scalar1, p2):
        """:177-183"""
        self._halves("SUB_SCALAR", p1, scalar0, scalar1, p2)

    def MulDoubleRNSScalar(self, p1, scalar0, scalar1, p2):
        """:250-256: MulScalarMontgomery by MForm(scalar_k[i])"""
        self._halves("MUL_SCALAR_MONT", p1, self._mform_scalars(scalar0), self._mform_scalars(scalar1), p2)

    def MulDoubleRNSScalarThenAdd(self, p1, scalar0, scalar1, p2):
        """:260-266"""
        self._halves("MUL_SCALAR_MONT_THEN_ADD", p1, self._mform_scalars(scalar0), self._mform_scalars(scalar1), p2)

    def MulByVectorMontgomery(self, p1, vector, p2):
        """:366-370: every limb of every poly times the same N-word vector (a 1-poly, 1-limb DevicePoly)"""
        self._chk(p1, p2, rows_ok=True)
        self._lay(p2, p1, vector)
        _check(lib().rh_ring_vec_op_bcast(self._h, OPS["MUL_MONT"], p1.ptr, p1.limbs, vector.ptr, p2.ptr, p2.limbs, p1.npoly, self.level))

    def MulByVectorMontgomeryThenAddLazy(self, p1, vector, p2):
        """:373-377"""
        self._chk(p1, p2, rows_ok=True)
        self._lay(p2, p1, vector, p2)
        _check(lib().rh_ring_vec_op_bcast(self._h, OPS["MUL_MONT_THEN_ADD_LAZY"], p1.ptr, p1.limbs, vector.ptr, p2.ptr, p2.limbs, p1.npoly, self.level))

    def Shift(self, p1, k, p2):
        """:278-282: p2[j] = p1[(j + k) mod N] on every limb; in place through a temporary"""
        self._chk(p1, p2)
        src = p1
        if p1.ptr == p2.ptr:
            src = self.NewPoly(p1.npoly); self.CopyLvl(p1, src)
        _check(lib().rh_ring_shift(self._h, self.level, src.ptr, p2.ptr, int(k), p1.npoly))

    def MultByMonomial(self, p1, k, p2):
        """:306-363: p2 = p1 * X^k (X^N = -1), the reference's representatives (q - 0 is written as q); in place through a temporary"""
        self._chk(p1, p2)
        src = p1
        if p1.ptr == p2.ptr:
            src = self.NewPoly(p1.npoly); self.CopyLvl(p1, src)
        _check(lib().rh_ring_mult_by_monomial(self._h, self.level, src.ptr, p2.ptr, int(k), p1.npoly))

    # ---- big-integer converters and Equal (ring/ring.go:433-558): host-side in the reference too (math/big) -----------
    def SetCoefficientsBigint(self, coeffs, p1, poly=0):
        """:433-447: limb i of poly `poly` of the block gets coeff mod q_i (Go's big.Int.Mod: the non-negative residue) for the
        first len(coeffs) coefficients; the others and the other polys of the block keep what they hold"""
        coeffs = [int(c) for c in coeffs]
        if len(coeffs) > self.N:
            raise RingHipError("SetCoefficientsBigint: %d coefficients for a ring of degree %d" % (len(coeffs), self.N))
        host = p1.numpy()
        for i in range(self.level + 1):
            q = int(self.moduli[i])
            host[poly, i, :len(coeffs)] = np.array([c % q for c in coeffs], dtype=np.uint64)
        _check(lib().rh_dev_upload(self._h, p1.ptr, _p(host), host.size))
        p1.layout = None                               # the block was read back (reference order) and rewritten whole

    def _crt(self):
        mods = [int(q) for q in self.moduli[:self.level + 1]]
        Q = 1
        for q in mods:
            Q *= q
        return mods, Q, [(Q // q) * pow(Q // q, -1, q) for q in mods]

    def PolyToBigint(self, p1, gap=1, poly=0):
        """:467-496: CRT reconstruction of coefficients 0, gap, 2 gap, ... of poly `poly` modulo Q_level, as Python integers"""
        mods, Q, crt = self._crt()
        host = p1.numpy()[poly]
        return [sum(int(host[k, j]) * crt[k] for k in range(len(mods))) % Q for j in range(0, self.N, gap)]

    def PolyToBigintCentered(self, p1, gap=1, poly=0):
        """:503-543: the same, centred: values >= floor(Q/2) have Q subtracted"""
        mods, Q, crt = self._crt()
        half = Q >> 1
        return [v - Q if v >= half else v for v in self.PolyToBigint(p1, gap, poly)]

    def Equal(self, p1, p2):
        """:546-558: Reduce both operands IN PLACE (as the reference does), then compare limbs 0..level"""
        self._chk(p1, p2, rows_ok=True)
        if p1.npoly != p2.npoly:
            return False
        self.Reduce(p1, p1)
        self.Reduce(p2, p2)
        a, b = p1.numpy(), p2.numpy()
        return bool(np.array_equal(a[:, :self.level + 1], b[:, :self.level + 1]))

    # ---- standard <-> conjugate-invariant bridges (ring/conjugate_invariant.go) ------------------------------------
    def _bridge_chk(self, small, big, who):
        if big.ring.N != 2 * small.ring.N:
            raise RingHipError("cannot %s: the ring degree of one operand must be twice the other's (%d, %d)" % (who, small.ring.N, big.ring.N))
        for p in (small, big):
            if p.limbs != self.level + 1 or p.npoly != small.npoly:
                raise RingHipError("%s: dense blocks of level+1 = %d limbs and the same number of polys" % (who, self.level + 1))

    def UnfoldConjugateInvariantToStandard(self, polyConjugateInvariant, polyStandard):
        """:8-26, receiver = the standard ring of degree 2n: std[j] = ci[j], std[n + k] = ci[n - 1 - k] on every limb"""
        self._bridge_chk(polyConjugateInvariant, polyStandard, "UnfoldConjugateInvariantToStandard")
        if polyStandard.ring.N != self.N:
            raise RingHipError("UnfoldConjugateInvariantToStandard: the receiver is the standard ring of polyStandard")
        _check(lib().rh_ring_unfold_ci_to_standard(self._h, self.level, polyConjugateInvariant.ptr, polyStandard.ptr, polyStandard.npoly))

    def FoldStandardToConjugateInvariant(self, polyStandard, permuteNTTIndexInv, polyConjugateInvariant):
        """:31-49, receiver = the conjugate-invariant ring of degree n: ci[j] = CRed(std[index[j]] + std[j]); index: n entries < 2n
        (the standard ring's AutomorphismNTTIndex of the inverse Galois element, schemes/ckks/bridge.go:48-50)"""
        self._bridge_chk(polyConjugateInvariant, polyStandard, "FoldStandardToConjugateInvariant")
        if polyConjugateInvariant.ring.N != self.N:
            raise RingHipError("FoldStandardToConjugateInvariant: the receiver is the ring of polyConjugateInvariant")
        if isinstance(permuteNTTIndexInv, DevicePoly):
            t = permuteNTTIndexInv
        else:
            t = self._index_table(_u64(permuteNTTIndexInv).reshape(-1)[:self.N], bound=2)
        _check(lib().rh_ring_fold_standard_to_ci(self._h, self.level, polyStandard.ptr, t.ptr, polyConjugateInvariant.ptr, polyStandard.npoly))

    def PadDefaultRingToConjugateInvariant(self, polyStandard, IsNTT, polyConjugateInvariant):
        """:52-80, receiver = a ring of polyStandard's degree n: the first n words of every limb of polyConjugateInvariant (rows of 2n
        words) as the reference's in-place loop leaves them; words n..2n-1 are not written"""
        self._bridge_chk(polyStandard, polyConjugateInvariant, "PadDefaultRingToConjugateInvariant")
        if polyStandard.ring.N != self.N:
            raise RingHipError("PadDefaultRingToConjugateInvariant: the receiver has polyStandard's degree")
        _check(lib().rh_ring_pad_default_to_ci(self._h, self.level, polyStandard.ptr, 1 if IsNTT else 0, polyConjugateInvariant.ptr, polyStandard.npoly))

    def EvalPolyScalar(self, p1, scalar, p2):
        """:269-275: p2 = p1[0] + p1[1] * scalar + ... by Horner (p1: list of blocks)"""
        self.CopyLvl(p1[-1], p2)
        for i in range(len(p1) - 1, 0, -1):
            self.MulScalar(p2, scalar, p2)
            self.Add(p2, p1[i - 1], p2)


def AutomorphismNTTIndex(N, NthRoot, GalEl):
    """ring/automorphism.go:12-34: the lookup table of the NTT-domain automorphism X -> X^GalEl"""
    if N & (N - 1) or NthRoot & (NthRoot - 1):
        raise RingHipError("N and NthRoot must be powers of two")
    lg = (int(NthRoot) - 1).bit_length() - 1
    mask = int(NthRoot) - 1
    rev = lambda x: int(format(x, "0%db" % lg)[::-1], 2) if lg else 0
    out = np.empty(N, dtype=np.uint64)
    for i in range(N):
        t1 = 2 * rev(i) + 1
        t2 = (((int(GalEl) * t1) & mask) - 1) >> 1
        out[i] = rev(t2)
    return out


class BasisExtender:
    """ring.BasisExtender (ring/basis_extension.go:13-79) over a (ringQ, ringP) pair on the same device."""

    def __init__(self, ringQ, ringP=None):
        self.ringQ, self.ringP = ringQ, ringP
        h = C.c_void_p()
        _check(lib().rh_bext_create(C.byref(h), ringQ._h, ringP._h if ringP is not None else None))
        self._h = h

    def reserve(self, npoly):
        _check(lib().rh_bext_reserve(self._h, int(npoly)))

    def ModUpQtoP(self, levelQ, levelP, polQ, polP):
        _check(lib().rh_bext_modup_q_to_p(self._h, levelQ, levelP, polQ.ptr, polP.ptr, polQ.npoly))

    def ModUpPtoQ(self, levelP, levelQ, polP, polQ):
        _check(lib().rh_bext_modup_p_to_q(self._h, levelP, levelQ, polP.ptr, polQ.ptr, polP.npoly))

    def ModDownQPtoQ(self, levelQ, levelP, p1Q, p1P, p2Q):
        _check(lib().rh_bext_moddown_qp_to_q(self._h, levelQ, levelP, p1Q.ptr, p1P.ptr, p2Q.ptr, p1Q.npoly))

    def ModDownQPtoQNTT(self, levelQ, levelP, p1Q, p1P, p2Q):
        _check(lib().rh_bext_moddown_qp_to_q_ntt(self._h, levelQ, levelP, p1Q.ptr, p1P.ptr, p2Q.ptr, p1Q.npoly))

    def ModDownQPtoP(self, levelQ, levelP, p1Q, p1P, p2P):
        _check(lib().rh_bext_moddown_qp_to_p(self._h, levelQ, levelP, p1Q.ptr, p1P.ptr, p2P.ptr, p1Q.npoly))

    def DecomposeAndSplit(self, levelQ, levelP, nbPi, digit, p0Q, p1Q, p1P):
        _check(lib().rh_bext_decompose_and_split(self._h, levelQ, levelP, nbPi, digit, p0Q.ptr, p1Q.ptr, p1P.ptr, p0Q.npoly))

    def GadgetProduct(self, levelQ, levelP, cx, evkQ_ptr, evkP_ptr, beta_key, ct0, ct1):
        """rlwe.Evaluator.GadgetProduct for NTT-domain cx; evk*_ptr: device pointers of the key blocks (see ringhip.h)"""
        _check(lib().rh_bext_gadget_product(self._h, levelQ, levelP, cx.ptr, evkQ_ptr, evkP_ptr, beta_key, ct0.ptr, ct1.ptr, cx.npoly))

    def ShallowCopy(self):
        """basis_extension.go:166-183: an extender over the same rings with its own scratch (the handle itself is safe to share)"""
        return BasisExtender(self.ringQ, self.ringP)

    def GadgetProductThenAdd(self, levelQ, levelP, cx, evkQ_ptr, evkP_ptr, beta_key, add0, add1, ct0, ct1):
        """ct_c = add_c + GadgetProduct(cx)_c (rh_bext_gadget_product_then_add); add_c may be None and may alias ct_c"""
        _check(lib().rh_bext_gadget_product_then_add(self._h, levelQ, levelP, cx.ptr, evkQ_ptr, evkP_ptr, beta_key,
                                                     add0.ptr if add0 is not None else None, add1.ptr if add1 is not None else None,
                                                     ct0.ptr, ct1.ptr, cx.npoly))

    def close(self):
        if getattr(self, "_h", None):
            lib().rh_bext_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
