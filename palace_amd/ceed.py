"""Host-side mirror of `palace::ceed::Operator` / `BilinearForm::PartialAssemble` over the C ABI.

Reference: palace/fem/libceed/operator.hpp:32-65 (Mult / AddMult / AssembleDiagonal on L-vectors),
palace/fem/bilinearform.cpp:27-107 (loop over geometries x integrators adding sub-operators),
palace/fem/integ/{curlcurl,vecfemass,curlcurlmass}.cpp (QFunction + eval modes + context per
integrator), palace/fem/libceed/coefficient.cpp:51-131 (context packing).

Vectors are float64 torch tensors on the GPU; torch is only the owner of device memory and the
stream — every FLOP happens in libpalace_amd.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .fem.basis1d import Tables1D
from .fem.fespace import H1HexSpace, NDHexSpace
from .fem.mesh import HexMesh, _q2_1d

QF_HDIV_33, QF_HCURL_33, QF_HDIVMASS_33, QF_HCURLMASS_33, QF_H1_1, QF_HCURL_22, QF_L2_1, QF_HDIVMASS_22, QF_HCURL_32 = range(9)
QF_HCURLHDIV_ERROR_33, QF_HDIVHCURL_ERROR_33 = 11, 12
QF_HDIVMASS_32, QF_HCURLMASS_22, QF_HCURLMASS_32 = 13, 14, 15  # the remaining 2-D / boundary-element pair forms
QF_HCURL_21, QF_HCURL_31, QF_HCURLMASS_21, QF_HCURLMASS_31 = 16, 17, 18, 19  # line elements in the plane / in space
QF_HCURLHDIV_33, QF_HDIVHCURL_33 = 9, 10  # weak curl (trial Interp, test Curl) / mixed curl (trial Curl, test Interp)
QF_HCURLHDIV_22, QF_HDIVHCURL_22, QF_HCURLHDIV_ERROR_22, QF_HDIVHCURL_ERROR_22 = 20, 21, 22, 23  # two spaces, plane elements
QF_HDIV_22 = 24  # mass of a plane H(div) space (FE_HDIV block, Interp)
QF_L2H1_ERROR = 25  # element error between two scalar fields (ElementErrorIntegrator with two scalar blocks)
# the contravariant members on boundary / line elements, div-div + mass, the gradient form (include/palace_amd.h)
QF_HDIV_32, QF_HDIV_21, QF_HDIV_31 = 26, 27, 28
QF_L2MASS_22, QF_L2MASS_33, QF_L2MASS_32, QF_L2MASS_21, QF_L2MASS_31 = 29, 30, 31, 32, 33
QF_HCURLHDIV_32, QF_HDIVHCURL_32, QF_HCURLHDIV_21, QF_HDIVHCURL_21, QF_HCURLHDIV_31, QF_HDIVHCURL_31 = 34, 35, 36, 37, 38, 39
QF_HCURLH1D_22, QF_HCURLH1D_33, QF_HCURLH1D_32, QF_HCURLH1D_21, QF_HCURLH1D_31 = 40, 41, 42, 43, 44
EVAL_WEIGHT, EVAL_NONE, EVAL_INTERP, EVAL_GRAD, EVAL_DIV, EVAL_CURL = (1 << i for i in range(6))
FE_H1, FE_HCURL, FE_HDIV = 0, 1, 2


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _stream():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def coefficient_context(dim, attr_mat=None, mat_coeff=None, a=1.0):
    """PopulateCoefficientContext (coefficient.cpp:51-118) -> 8-byte-slot blob (float64 view).

    attr_mat[i] = material index of (1-based) attribute i+1, negative = unassigned (zero);
    mat_coeff[k] = scalar or dim x dim matrix.  attr_mat=None: identity scaled by a."""
    d2 = dim * dim
    if attr_mat is None:
        raw = np.zeros(2 + d2)
        iv = raw.view(np.int32).reshape(-1, 2)
        iv[0, 0], iv[1, 0] = 0, 1
        raw[2:] = (a * np.eye(dim)).reshape(-1)
        return raw
    attr_mat = np.asarray(attr_mat, dtype=np.int32)
    nmat = len(mat_coeff)
    raw = np.zeros(2 + attr_mat.size + d2 * (nmat + 1))
    iv = raw.view(np.int32).reshape(-1, 2)
    iv[0, 0] = attr_mat.size
    iv[1 : 1 + attr_mat.size, 0] = np.where(attr_mat < 0, nmat, attr_mat)
    iv[1 + attr_mat.size, 0] = nmat + 1
    base = 2 + attr_mat.size
    for k, m in enumerate(mat_coeff):
        m = np.asarray(m, dtype=np.float64)
        full = a * float(m.reshape(-1)[0]) * np.eye(dim) if m.size == 1 else a * m.reshape(dim, dim)
        raw[base + d2 * k : base + d2 * (k + 1)] = full.T.reshape(-1)  # column-major
    return raw


class GeomFactorData:
    """ceed::CeedGeomFactorData (fem/mesh.hpp:27-69) for one hex block, built on the device."""

    def __init__(self, mesh: HexMesh, q1d: int):
        from .fem.basis1d import gauss_legendre

        self.mesh, self.q1d = mesh, q1d
        qx, qw = gauss_legendre(q1d)
        B, G = _q2_1d(qx)
        self._keep = dict(
            off=np.ascontiguousarray(mesh.elem_nodes, dtype=np.int32),
            nodes=np.ascontiguousarray(mesh.x, dtype=np.float64),
            attr=np.ascontiguousarray(mesh.attr, dtype=np.int32),
            B=np.ascontiguousarray(B), G=np.ascontiguousarray(G), w=np.ascontiguousarray(qw))
        k = self._keep
        desc = _lib.MeshDesc(mesh.ne, 2, q1d, mesh.x.shape[0], _ptr(k["off"]), _ptr(k["nodes"]),
                             _ptr(k["attr"]), _ptr(k["B"]), _ptr(k["G"]), _ptr(k["w"]))
        self.handle = C.c_void_p()
        L = _lib.load()
        _lib.check(L.pa_geom_create(C.byref(desc), _stream(), C.byref(self.handle)))
        self.Q = q1d**3

    def to_numpy(self):
        """Copy the [ne][11][Q] geometry data back (tests / diagnostics)."""
        import torch

        L = _lib.load()
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(L.pa_geom_data(self.handle, C.byref(p), C.byref(n)))

        class _View:  # device memory owned by the library, exposed through the array interface
            __cuda_array_interface__ = dict(shape=(n.value,), typestr="<f8", data=(p.value, False), version=2)

        torch.cuda.synchronize()
        data = torch.as_tensor(_View(), device="cuda").cpu().numpy().reshape(self.mesh.ne, 11, self.Q)
        order = np.zeros(self.mesh.ne, dtype=np.int32)  # the library keeps the elements in its own order
        _lib.check(L.pa_geom_element_order(self.handle, order.ctypes.data_as(C.c_void_p)))
        out = np.empty_like(data)
        out[order] = data
        return out

    def __del__(self):
        try:
            if self.handle:
                _lib.load().pa_geom_destroy(self.handle)
        except Exception:
            pass


class DenseGeomFactorData:
    """ceed::CeedGeomFactorData for one NON-tensor element block (tetrahedra, ...): the mesh nodal basis
    enters through its dense gradient table at the quadrature points (fem/mesh.cpp:146-209 with a
    non-tensor mesh basis).  elem_nodes [ne, npe], nodes [nn, 3], attr [ne] (1-based),
    mesh_grad [3, Q, npe], qweight [Q]."""

    def __init__(self, elem_nodes, nodes, attr, mesh_grad, qweight):
        self.ne, self.npe = elem_nodes.shape
        self.Q = len(qweight)
        self.dim = int(np.asarray(mesh_grad).shape[0])  # element dimension: 3, or 2 (2-D cases, boundary elements)
        self.space_dim = int(np.asarray(nodes).shape[1])  # 3 with dim 2: boundary elements (8 geometry rows)
        self._keep = dict(off=np.ascontiguousarray(elem_nodes, dtype=np.int32),
                          nodes=np.ascontiguousarray(nodes, dtype=np.float64),
                          attr=np.ascontiguousarray(attr, dtype=np.int32),
                          grad=np.ascontiguousarray(mesh_grad, dtype=np.float64),
                          w=np.ascontiguousarray(qweight, dtype=np.float64))
        k = self._keep
        assert k["grad"].shape == (self.dim, self.Q, self.npe)
        desc = _lib.MeshDenseDesc(self.ne, self.npe, self.Q, k["nodes"].shape[0], _ptr(k["off"]), _ptr(k["nodes"]),
                                  _ptr(k["attr"]), _ptr(k["grad"]), _ptr(k["w"]), self.dim, self.space_dim)
        self.handle = C.c_void_p()
        _lib.check(_lib.load().pa_geom_create_dense(C.byref(desc), _stream(), C.byref(self.handle)))

    def to_numpy(self):
        """The geometry data in the reference layout [ne][11][Q] (undoing the 16-element blocking)."""
        import torch

        L = _lib.load()
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(L.pa_geom_data(self.handle, C.byref(p), C.byref(n)))
        lay = (C.c_int32 * 4)()
        _lib.check(L.pa_geom_layout(self.handle, lay))
        ne, Q, Qpad, eb = list(lay)
        rows = L.pa_geom_num_rows(self.handle)

        class _View:
            __cuda_array_interface__ = dict(shape=(n.value,), typestr="<f8", data=(p.value, False), version=2)

        torch.cuda.synchronize()
        raw = torch.as_tensor(_View(), device="cuda").cpu().numpy().reshape(-1, rows, Qpad, eb)
        return np.ascontiguousarray(raw.transpose(0, 3, 1, 2).reshape(-1, rows, Qpad)[:ne, :, :Q])

    def __del__(self):
        try:
            if self.handle:
                _lib.load().pa_geom_destroy(self.handle)
        except Exception:
            pass


class ElementErrorIntegrator:
    """AssembleCeedElementErrorIntegrator (fem/libceed/integrator.cpp:550-626) on the device (pa_error_op_*): per-element
    int |C_2 u_2 - C_1 u_1|^2 for L-vectors of two spaces; qf = QF_HCURLHDIV_ERROR_33 (first input in H(curl)) or
    QF_HDIVHCURL_ERROR_33; ctx_pair = the two coefficient contexts, first input's first."""

    def __init__(self, geom, first: "DenseBlock", second: "DenseBlock", qf, ctx_pair):
        r1, b1 = first.descs()
        r2, b2 = second.descs()
        ctx = np.ascontiguousarray(ctx_pair)
        self.handle = C.c_void_p()
        self._keep = (geom, first, second)
        _lib.check(_lib.load().pa_error_op_create(geom.handle, C.byref(r1), C.byref(b1), C.byref(r2), C.byref(b2),
                                                  C.c_int32(qf), _ptr(ctx), C.c_size_t(ctx.nbytes), C.byref(self.handle)))
        self.ne = int(_lib.load().pa_error_op_num_elem(self.handle))

    def apply_add(self, u1, u2, estimates):
        _lib.check(_lib.load().pa_error_op_apply_add(self.handle, C.c_void_p(u1.data_ptr()), C.c_void_p(u2.data_ptr()),
                                                     C.c_void_p(estimates.data_ptr()), _stream()))
        return estimates

    def __del__(self):
        try:  # (module globals may already be gone at interpreter shutdown)
            if getattr(self, "handle", None):
                _lib.load().pa_error_op_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class DenseBlock:
    """What Palace hands to libCEED for one (geometry, space) pair on the non-tensor path: the native
    restriction (fem/libceed/restriction.cpp:207-385: offsets + bool orients, or + int8 tridiagonal
    curl_orients [ne, P, 3]) and the dense basis tables (fem/libceed/basis.cpp:40-85:
    interp [qcomp, Q, P], deriv [3, Q, P])."""

    def __init__(self, fe_type, lsize, offsets, interp, deriv, orients=None, curl_orients=None):
        self.fe_type, self.lsize = fe_type, int(lsize)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        self.ne, self.P = self.offsets.shape
        self.orients = None if orients is None else np.ascontiguousarray(orients, dtype=np.uint8)
        self.curl_orients = None if curl_orients is None else np.ascontiguousarray(curl_orients, dtype=np.int8)
        self.interp = None if interp is None else np.ascontiguousarray(interp, dtype=np.float64)
        self.deriv = None if deriv is None else np.ascontiguousarray(deriv, dtype=np.float64)
        self.Q = (self.deriv if self.deriv is not None else self.interp).shape[-2]

    def descs(self):
        r = _lib.RestrictionDesc(self.ne, self.P, self.lsize, _ptr(self.offsets), _ptr(self.orients),
                                 _ptr(self.curl_orients))
        b = _lib.DenseBasisDesc(self.fe_type, self.P, self.Q, _ptr(self.interp), _ptr(self.deriv))
        return r, b


def _basis_desc(space, q1d, dense=None):
    p = space.p
    t = Tables1D(p, q1d)
    keep = dict(Bc=np.ascontiguousarray(t.Bc), Gc=np.ascontiguousarray(t.Gc),
                Bo=np.ascontiguousarray(t.Bo))
    if isinstance(space, NDHexSpace):
        keep["dof_map"] = np.ascontiguousarray(space.dof_map_native(), dtype=np.int32)
        fe = FE_HCURL
    else:
        keep["dof_map"] = None
        fe = FE_H1
    interp = deriv = None
    if dense is not None:
        keep["interp"] = interp = np.ascontiguousarray(dense[0])
        keep["deriv"] = deriv = np.ascontiguousarray(dense[1])
    desc = _lib.BasisDesc(fe, p, q1d, _ptr(keep["Bc"]), _ptr(keep["Gc"]), _ptr(keep["Bo"]),
                          _ptr(keep["dof_map"]), _ptr(interp), _ptr(deriv))
    return desc, keep


def _restriction_desc(space):
    if isinstance(space, NDHexSpace):
        off, ori = space.native_restriction()
        keep = dict(off=np.ascontiguousarray(off, dtype=np.int32),
                    ori=np.ascontiguousarray(ori, dtype=np.uint8))
    else:
        keep = dict(off=np.ascontiguousarray(space.elem_dof_lex, dtype=np.int32), ori=None)
    desc = _lib.RestrictionDesc(space.mesh.ne, space.P, space.ndofs, _ptr(keep["off"]), _ptr(keep["ori"]))
    return desc, keep


class DeviceCsr:
    """Owner of a pa_csr (device arrays of a fully assembled local operator)."""

    def __init__(self, handle):
        self.handle = handle
        L = _lib.load()
        n, nnz = C.c_int32(), C.c_int64()
        _lib.check(L.pa_csr_get(self.handle, C.byref(n), C.byref(nnz), None, None, None))
        self.nrows, self.nnz = n.value, nnz.value
        L.pa_csr_num_cols.restype = C.c_int
        L.pa_csr_num_cols.argtypes = [C.c_void_p]
        self.ncols = int(L.pa_csr_num_cols(self.handle))

    def __del__(self):
        try:
            L = _lib.load()
            L.pa_csr_destroy.restype = None
            L.pa_csr_destroy.argtypes = [C.c_void_p]
            L.pa_csr_destroy(self.handle)
        except Exception:
            pass


class Operator:
    """palace::ceed::Operator: a sum of partially assembled sub-operators on L-vectors."""

    def __init__(self, height, width, handle=None):
        self.height, self.width = height, width
        self.handle = C.c_void_p() if handle is None else handle
        if handle is None:
            _lib.check(_lib.load().pa_op_create(height, width, C.byref(self.handle)))

    def add_integrator(self, geom: GeomFactorData, space, qf, ctx_blob, ops, dense=None, test_ops=None):
        r, k1 = _restriction_desc(space)
        b, k2 = _basis_desc(space, geom.q1d, dense)
        ctx = np.ascontiguousarray(ctx_blob)
        _lib.check(_lib.load().pa_op_add_sub(self.handle, geom.handle, C.byref(r), C.byref(b),
                                             C.c_int32(qf), _ptr(ctx), C.c_size_t(ctx.nbytes),
                                             C.c_uint32(ops), C.c_uint32(ops if test_ops is None else test_ops)))
        return self

    def add_dense_integrator(self, geom: DenseGeomFactorData, block: DenseBlock, qf, ctx_blob, ops):
        """AddSubOperator for a non-tensor element block (dense tables on the FP64 matrix cores)."""
        r, b = block.descs()
        ctx = np.ascontiguousarray(ctx_blob)
        _lib.check(_lib.load().pa_op_add_sub_dense(self.handle, geom.handle, C.byref(r), C.byref(b),
                                                   C.c_int32(qf), _ptr(ctx), C.c_size_t(ctx.nbytes),
                                                   C.c_uint32(ops), C.c_uint32(ops)))
        return self

    def add_dense_mixed_integrator(self, geom: DenseGeomFactorData, trial: DenseBlock, test: DenseBlock, qf, ctx_blob):
        """BilinearForm(trial_fespace, test_fespace) + VectorFEMassIntegrator between an H(curl) and an H(div) space
        (pa_op_add_sub_dense_mixed): qf = QF_HCURLHDIV_33 (H(curl) trial) or QF_HDIVHCURL_33 (H(div) trial).  An H1 block on
        an H(curl) side enters with its gradient table: qf = QF_HCURL_33 with an H1 trial and an H(curl) test block is
        MixedVectorGradientIntegrator (C grad phi, v) (fem/integ/mixedvecgrad.cpp:43-76).  Plane elements: the _22 QFunctions."""
        r1, b1 = trial.descs()
        r2, b2 = test.descs()
        ctx = np.ascontiguousarray(ctx_blob)
        _lib.check(_lib.load().pa_op_add_sub_dense_mixed(self.handle, geom.handle, C.byref(r1), C.byref(b1), C.byref(r2),
                                                         C.byref(b2), C.c_int32(qf), _ptr(ctx), C.c_size_t(ctx.nbytes)))
        return self

    def add_dense_gradient_integrator(self, geom: DenseGeomFactorData, trial: DenseBlock, test: DenseBlock, comp_stride, qf, ctx_blob):
        """GradientIntegrator (fem/integ/grad.cpp:16-72, pa_op_add_sub_dense_gradient): `trial` a scalar H1 block (gradient
        table), `test` ONE component of the vector H1 test space (value table; its lsize = the size of the whole vector
        L-vector), the other components `comp_stride` entries further; qf = QF_HCURLH1D_* of the geometry data."""
        r1, b1 = trial.descs()
        r2, b2 = test.descs()
        ctx = np.ascontiguousarray(ctx_blob)
        L = _lib.load()
        _lib.check(L.pa_op_add_sub_dense_gradient(self.handle, geom.handle, C.byref(r1), C.byref(b1), C.byref(r2), C.byref(b2),
                                                  C.c_int32(int(comp_stride)), C.c_int32(qf), _ptr(ctx), C.c_size_t(ctx.nbytes)))
        return self

    def add_dense_vector_mass_integrator(self, geom: DenseGeomFactorData, comp: DenseBlock, num_comp, comp_stride, ctx_blob):
        """MassIntegrator on a vector H1 space with 2 or 3 components (f_apply_h1_2 | _3, pa_op_add_sub_dense_vector_mass): `comp`
        one component of the space (its lsize = the size of the whole vector L-vector)."""
        r, b = comp.descs()
        ctx = np.ascontiguousarray(ctx_blob)
        _lib.check(_lib.load().pa_op_add_sub_dense_vector_mass(self.handle, geom.handle, C.byref(r), C.byref(b), C.c_int32(int(num_comp)),
                                                               C.c_int32(int(comp_stride)), _ptr(ctx), C.c_size_t(ctx.nbytes)))
        return self

    @staticmethod
    def _sum_args(terms):
        qfs = (C.c_int32 * len(terms))(*[int(t[1]) for t in terms])
        blobs = [np.ascontiguousarray(t[2]) for t in terms]
        ptrs = (C.c_void_p * len(terms))(*[b.ctypes.data for b in blobs])
        sizes = (C.c_size_t * len(terms))(*[b.nbytes for b in blobs])
        coeffs = (C.c_double * len(terms))(*[float(t[0]) for t in terms])
        return qfs, ptrs, sizes, coeffs, blobs

    def add_integrator_sum(self, geom: GeomFactorData, space, terms):
        """sum_k a_k * integrator_k as ONE sub-operator (pa_op_add_sub_sum): terms = [(a, qf, ctx_blob), ...] over
        the H(curl) integrators K (QF_HDIV_33), M / C (QF_HCURL_33), K + M (QF_HDIVMASS_33) of one space."""
        r, k1 = _restriction_desc(space)
        b, k2 = _basis_desc(space, geom.q1d)
        qfs, ptrs, sizes, coeffs, keep = self._sum_args(terms)
        _lib.check(_lib.load().pa_op_add_sub_sum(self.handle, geom.handle, C.byref(r), C.byref(b), C.c_int32(len(terms)),
                                                 qfs, ptrs, sizes, coeffs))
        return self

    def add_dense_integrator_sum(self, geom: DenseGeomFactorData, block: DenseBlock, terms):
        """The same for a non-tensor element block (pa_op_add_sub_dense_sum)."""
        r, b = block.descs()
        qfs, ptrs, sizes, coeffs, keep = self._sum_args(terms)
        _lib.check(_lib.load().pa_op_add_sub_dense_sum(self.handle, geom.handle, C.byref(r), C.byref(b),
                                                       C.c_int32(len(terms)), qfs, ptrs, sizes, coeffs))
        return self

    def finalize(self):
        _lib.check(_lib.load().pa_op_finalize(self.handle))
        return self

    def coarsen(self, geom: GeomFactorData, space_coarse):
        """CeedOperatorCoarsen (operator.cpp:525-585)."""
        r, k1 = _restriction_desc(space_coarse)
        b, k2 = _basis_desc(space_coarse, geom.q1d)
        h = C.c_void_p()
        _lib.check(_lib.load().pa_op_coarsen(self.handle, C.byref(r), C.byref(b), C.byref(h)))
        return Operator(space_coarse.ndofs, space_coarse.ndofs, handle=h)

    def coarsen_dense(self, block: "DenseBlock"):
        """CeedOperatorCoarsen for dense sub-operators: coarse restriction + tables at the fine rule."""
        r, b = block.descs()
        h = C.c_void_p()
        _lib.check(_lib.load().pa_op_coarsen_dense(self.handle, C.byref(r), C.byref(b), C.byref(h)))
        return Operator(block.lsize, block.lsize, handle=h)

    def mult(self, x, y):
        _lib.check(_lib.load().pa_op_mult(self.handle, C.c_void_p(x.data_ptr()),
                                          C.c_void_p(y.data_ptr()), _stream()))
        return y

    def set_essential(self, ess):
        """pa_op_set_essential: fuse a list of essential dofs into the operator's index tables (masked applies)."""
        e = np.ascontiguousarray(ess, dtype=np.int32)
        _lib.check(_lib.load().pa_op_set_essential(self.handle, e.ctypes.data_as(C.c_void_p), int(e.size)))
        self._ess_keep = e

    def supports_split(self):
        return bool(_lib.load().pa_op_supports_split(self.handle))

    def mult_split(self, x, xg0, y, yg, ess_policy=-1, xg1=None, sel=None):
        """pa_op_mult_split: true dofs in x / y (len n_true), ghosts read from xg0 (or xg1 when the uint64 device scalar `sel`
        is odd) and written to yg."""
        L = _lib.load()
        L.pa_op_mult_split.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p]
        _lib.check(L.pa_op_mult_split(self.handle, C.c_void_p(x.data_ptr()), C.c_void_p(xg0.data_ptr()),
                                      C.c_void_p(xg1.data_ptr()) if xg1 is not None else None,
                                      C.c_void_p(sel.data_ptr()) if sel is not None else None, C.c_void_p(y.data_ptr()),
                                      C.c_void_p(yg.data_ptr()), int(x.numel()), int(ess_policy), _stream()))
        return y, yg

    def mult2(self, x0, x1, y0, y1):
        """y0 = A x0, y1 = A x1 in one pass over the element data (pa_op_mult2)."""
        _lib.check(_lib.load().pa_op_mult2(self.handle, C.c_void_p(x0.data_ptr()), C.c_void_p(x1.data_ptr()),
                                           C.c_void_p(y0.data_ptr()), C.c_void_p(y1.data_ptr()), _stream()))
        return y0, y1

    def mult_transpose(self, x, y):
        """y = A^T x (Operator::MultTranspose, fem/libceed/operator.cpp:214-224)."""
        _lib.check(_lib.load().pa_op_mult_transpose(self.handle, C.c_void_p(x.data_ptr()),
                                                    C.c_void_p(y.data_ptr()), _stream()))
        return y

    def is_symmetric(self):
        return bool(_lib.load().pa_op_is_symmetric(self.handle))

    def dense_affine(self):
        """Number of dense sub-operators running in the affine (constant Jacobian) form (pa_op_dense_affine)."""
        return int(_lib.load().pa_op_dense_affine(self.handle))

    def streams(self):
        """True when y = A x runs on the streaming kernels (pa_op_streams)."""
        return bool(_lib.load().pa_op_streams(self.handle))

    def stream_affine(self):
        """(elements, affine elements, of them compressed) of the streaming H(curl) hex kernel (pa_op_stream_affine)."""
        out = (C.c_int32 * 3)()
        _lib.check(_lib.load().pa_op_stream_affine(self.handle, out))
        return int(out[0]), int(out[1]), int(out[2])

    def dense_gather_form(self):
        """(E-vector rows by element?, lanes per dof) of the first dense-table block's E^T gather (pa_op_dense_gather_form)."""
        out = (C.c_int32 * 2)()
        _lib.check(_lib.load().pa_op_dense_gather_form(self.handle, out))
        return bool(out[0]), int(out[1])

    def add_mult(self, x, y, a=1.0):
        if a != 1.0:  # operator.cpp:194
            raise _lib.PalaceAmdError("ceed::Operator::AddMult only supports coefficient = 1.0!")
        _lib.check(_lib.load().pa_op_apply_add(self.handle, C.c_void_p(x.data_ptr()),
                                               C.c_void_p(y.data_ptr()), _stream()))
        return y

    def assemble_diagonal(self, diag):
        _lib.check(_lib.load().pa_op_assemble_diagonal(self.handle, C.c_void_p(diag.data_ptr()), _stream()))
        return diag

    def full_assemble_device(self, skip_zeros=False):
        """CeedOperatorFullAssemble, kept in HBM: a DeviceCsr for linalg.AssembledParOperator."""
        h = C.c_void_p()
        _lib.check(_lib.load().pa_op_full_assemble(self.handle, int(skip_zeros), _stream(), C.byref(h)))
        return DeviceCsr(h)

    def full_assemble(self, skip_zeros=False):
        """CeedOperatorFullAssemble -> scipy.sparse.csr_matrix (values copied back from the device)."""
        import scipy.sparse as sp
        import torch

        L = _lib.load()
        h = C.c_void_p()
        _lib.check(L.pa_op_full_assemble(self.handle, int(skip_zeros), _stream(), C.byref(h)))
        n, nnz = C.c_int32(), C.c_int64()
        rp, ci, va = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(L.pa_csr_get(h, C.byref(n), C.byref(nnz), C.byref(rp), C.byref(ci), C.byref(va)))

        def view(ptr, count, typestr, dtype):
            class _V:
                __cuda_array_interface__ = dict(shape=(count,), typestr=typestr, data=(ptr.value, False), version=2)
            return torch.as_tensor(_V(), device="cuda").cpu().numpy().astype(dtype)

        torch.cuda.synchronize()
        rowptr = view(rp, n.value + 1, "<i4", np.int32)
        col = view(ci, max(1, nnz.value), "<i4", np.int32)[: nnz.value]
        val = view(va, max(1, nnz.value), "<f8", np.float64)[: nnz.value]
        L.pa_csr_destroy.restype = None
        L.pa_csr_destroy.argtypes = [C.c_void_p]
        L.pa_csr_num_cols.restype = C.c_int
        L.pa_csr_num_cols.argtypes = [C.c_void_p]
        ncols = int(L.pa_csr_num_cols(h))  # = rows unless the operator has two spaces
        L.pa_csr_destroy(h)
        return sp.csr_matrix((val, col, rowptr), shape=(n.value, ncols))

    def algorithmic_bytes(self):
        return _lib.load().pa_op_algorithmic_bytes(self.handle)

    def __del__(self):
        try:
            if self.handle:
                _lib.load().pa_op_destroy(self.handle)
        except Exception:
            pass


# ---- the integrators of the hot path (fem/integ/*.cpp) ----------------------------------------

def curlcurl_operator(geom, nd: NDHexSpace, ctx_curl, dense=None):
    """CurlCurlIntegrator (fem/integ/curlcurl.cpp:23-75): f_apply_hdiv_33, Curl/Curl."""
    return Operator(nd.ndofs, nd.ndofs).add_integrator(geom, nd, QF_HDIV_33, ctx_curl, EVAL_CURL, dense).finalize()


def ndmass_operator(geom, nd: NDHexSpace, ctx_mass, dense=None):
    """VectorFEMassIntegrator (fem/integ/vecfemass.cpp): f_apply_hcurl_33, Interp/Interp."""
    return Operator(nd.ndofs, nd.ndofs).add_integrator(geom, nd, QF_HCURL_33, ctx_mass, EVAL_INTERP, dense).finalize()


def curlcurlmass_operator(geom, nd: NDHexSpace, ctx_mass, ctx_curl, dense=None):
    """CurlCurlMassIntegrator (fem/integ/curlcurlmass.cpp:16-68): f_apply_hdivmass_33; the paired
    context is mass first, then curl-curl (coefficient.cpp:120-131)."""
    ctx = np.concatenate([ctx_mass, ctx_curl])
    return Operator(nd.ndofs, nd.ndofs).add_integrator(
        geom, nd, QF_HDIVMASS_33, ctx, EVAL_CURL | EVAL_INTERP, dense).finalize()


def weakcurl_operator(geom, nd: NDHexSpace, ctx):
    """MixedVectorWeakCurlIntegrator on one H(curl) space (fem/integ/mixedveccurl.cpp:75-120): (C u, curl v),
    f_apply_hcurlhdiv_33, trial Interp / test Curl."""
    return Operator(nd.ndofs, nd.ndofs).add_integrator(geom, nd, QF_HCURLHDIV_33, ctx, EVAL_INTERP, test_ops=EVAL_CURL).finalize()


def mixedcurl_operator(geom, nd: NDHexSpace, ctx):
    """MixedVectorCurlIntegrator on one H(curl) space (fem/integ/mixedveccurl.cpp:21-73): (C curl u, v),
    f_apply_hdivhcurl_33, trial Curl / test Interp."""
    return Operator(nd.ndofs, nd.ndofs).add_integrator(geom, nd, QF_HDIVHCURL_33, ctx, EVAL_CURL, test_ops=EVAL_INTERP).finalize()


def diffusion_operator(geom, h1: H1HexSpace, ctx_diff, dense=None):
    """DiffusionIntegrator (fem/integ/diffusion.cpp): f_apply_hcurl_33 on grad u, Grad/Grad."""
    return Operator(h1.ndofs, h1.ndofs).add_integrator(geom, h1, QF_HCURL_33, ctx_diff, EVAL_GRAD, dense).finalize()


def h1mass_operator(geom, h1: H1HexSpace, ctx_mass1, dense=None):
    """MassIntegrator (fem/integ/mass.cpp): f_apply_h1_1, Interp/Interp, scalar (dim-1) context."""
    return Operator(h1.ndofs, h1.ndofs).add_integrator(geom, h1, QF_H1_1, ctx_mass1, EVAL_INTERP, dense).finalize()


def diffusionmass_operator(geom, h1: H1HexSpace, ctx_mass1, ctx_diff, dense=None):
    """DiffusionMassIntegrator (fem/integ/diffusionmass.cpp): f_apply_hcurlmass_33, mass context (dim 1)
    first, then the diffusion one (dim 3)."""
    ctx = np.concatenate([ctx_mass1, ctx_diff])
    return Operator(h1.ndofs, h1.ndofs).add_integrator(
        geom, h1, QF_HCURLMASS_33, ctx, EVAL_GRAD | EVAL_INTERP, dense).finalize()
