"""ctypes face of the C++ host layer that mirrors palace::linalg (include/palace_amd_linalg.h,
palace_amd/csrc/linalg.hpp): ParOperator, Chebyshev / Jacobi smoothers, CG / GMRES, geometric
multigrid and the p-prolongation, all on float64 CUDA tensors owned by torch."""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from . import lib as _lib
from .ceed import Operator, _basis_desc, _ptr, _restriction_desc
from .fem.basis1d import gauss_legendre, gauss_lobatto, lagrange_eval

DIAG_ZERO, DIAG_ONE = 0, 1


def _L():
    L = _lib.load()
    if not getattr(L, "_linalg_typed", False):
        for name in ("pa_context_destroy", "pa_halo_destroy", "pa_par_op_destroy", "pa_solver_destroy",
                     "pa_interp_destroy"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_void_p]
        L.pa_vec_axpby.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_void_p, C.c_int]
        L.pa_bench_mfma_f64.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        L.pa_chebyshev_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.pa_cg_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int,
                                   C.c_void_p]
        L.pa_gmres_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p]
        L.pa_gmg_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.pa_vec_set_random.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64]
        L._linalg_typed = True
    return L


class Context:
    """Stream + (optional) RCCL communicator."""

    def __init__(self, stream=None):
        import torch

        self.handle = C.c_void_p()
        if stream is None and torch.cuda.current_stream().cuda_stream == 0:
            # PyTorch is on the legacy null stream: let the context create its own blocking stream (same ordering against
            # the null stream, and recordable into HIP graphs)
            _lib.check(_L().pa_context_create(None, C.byref(self.handle)))
            raw = C.c_void_p()
            _lib.check(_L().pa_context_stream(self.handle, C.byref(raw)))
            self.torch_stream = torch.cuda.ExternalStream(raw.value)
        else:
            self.torch_stream = torch.cuda.current_stream() if stream is None else stream
            _lib.check(_L().pa_context_create(C.c_void_p(self.torch_stream.cuda_stream), C.byref(self.handle)))
        self.rank, self.size = 0, 1

    def init_comm_from_torch_distributed(self):
        """Create the RCCL communicator; the 128-byte unique id travels over torch.distributed.  Halo exchanges and global sums
        then move to the peer transport if -- and only if -- EVERY rank could map every arena and the transport passed its
        self-test on this machine; the decision is collective (a rank never ends up on another transport than its peers)."""
        import torch
        import torch.distributed as dist

        L = _L()
        rank, size = dist.get_rank(), dist.get_world_size()
        buf = C.create_string_buffer(128)
        if rank == 0:
            _lib.check(L.pa_comm_unique_id(buf))
        t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).cuda()
        dist.broadcast(t, src=0)
        raw = bytes(t.cpu().numpy().tobytes())
        _lib.check(L.pa_context_init_comm(self.handle, rank, size, raw))
        self.rank, self.size = rank, size
        self.transport_report = {"transport": "rccl"}
        if size > 1 and os.environ.get("PALACE_AMD_HALO", "peer") != "rccl":
            self._peer_bring_up(fallback=True)

    # ---- peer transport: collective bring-up ------------------------------------------------------------------------------
    def _all_ok(self, ok):
        """Logical AND of `ok` over the ranks (one small all-reduce on torch.distributed's own backend)."""
        import torch
        import torch.distributed as dist

        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def _peer_bring_up(self, fallback):
        """Map the arenas of all ranks and prove the transport on this machine, every step decided by all ranks together:
        (1) every rank exports its arena handle, (2) the table is gathered, (3) every rank maps every arena, (4) self-test with
        the relaxed ordering protocol, (5) if -- and only if -- that saw wrong VALUES (no lost message), the same test with
        system-scope fences; otherwise (6) every rank disconnects and RCCL carries the exchanges (`fallback`; a communicator
        without RCCL raises instead).  Local failures never skip a collective call: they only turn this rank's vote to 'no'."""
        import sys

        import torch
        import torch.distributed as dist

        L = _L()
        L.pa_comm_peer_set_timeout.argtypes = [C.c_double]
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        rep = {"transport": "peer", "ordering": "relaxed"}
        why = ""
        buf = C.create_string_buffer(64)
        ok = L.pa_comm_peer_handle(self.handle, buf) == 0
        if not ok:
            why = _lib.last_error()
        mine = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).to(dev)
        parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine)  # (also by a rank whose export failed: its vote below says so)
        ok = self._all_ok(ok)
        if ok:
            table = b"".join(bytes(t.cpu().numpy().tobytes()) for t in parts)
            ok = L.pa_comm_peer_connect(self.handle, table) == 0
            if not ok:
                why = _lib.last_error()
            ok = self._all_ok(ok)
            if not ok and L.pa_comm_peer_ready(self.handle):
                L.pa_comm_peer_disconnect(self.handle)  # (some other rank could not map: nobody uses the transport)
        if ok:
            # a lost message in the self-test must not stall the start-up for a minute per wait
            user_timeout = float(os.environ.get("PALACE_AMD_PEER_TIMEOUT_S", "60"))
            _lib.check(L.pa_comm_peer_set_timeout(min(user_timeout, 10.0)))
            ok, values_only, rep["self_test"] = self._peer_self_test()
            if not ok and values_only and not L.pa_comm_peer_fenced():
                _lib.check(L.pa_comm_peer_set_fenced(1))
                rep["ordering"] = "system-scope fences (the relaxed protocol failed its self-test here)"
                ok, values_only, rep["self_test_fenced"] = self._peer_self_test()
            _lib.check(L.pa_comm_peer_set_timeout(user_timeout))
            if not ok:
                why = "self-test failed: " + json.dumps(rep)
        if not ok:
            if L.pa_comm_peer_ready(self.handle) and fallback:
                _lib.check(L.pa_comm_peer_disconnect(self.handle))
            if not fallback:
                raise RuntimeError(f"peer transport not available ({why or 'another rank failed'})")
            rep = {"transport": "rccl", "peer_failed": why or "on another rank"}
            if self.rank == 0:
                print(f"palace_amd: peer transport not available ({why or 'another rank failed'}); using RCCL send / receive",
                      file=sys.stderr)
        self.transport_report = rep
        return ok

    def _ring_plan(self, n):
        """Ring plan over a vector of 2 n entries: owned [0, n) go to rank + 1, ghosts [n, 2 n) are owned by rank - 1."""
        rank, size = self.rank, self.size
        left, right = (rank - 1) % size, (rank + 1) % size
        own, gh = np.arange(n, dtype=np.int32), np.arange(n, 2 * n, dtype=np.int32)
        if size == 2:
            return Halo(self, [left], [own], [gh])
        return Halo(self, [left, right], [np.zeros(0, np.int32), own], [gh, np.zeros(0, np.int32)])

    def peer_stress(self, rounds, n=4096, direct=False, graph=False, ring=None):
        """pa_comm_peer_stress on a ring plan: the number of wrong values this rank saw (raises on a timed-out wait)."""
        L = _L()
        L.pa_comm_peer_stress.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
        h = ring if ring is not None else self._ring_plan(n)
        bad = C.c_longlong(0)
        _lib.check(L.pa_comm_peer_stress(self.handle, h.handle, int(n), int(rounds), int(bool(direct)), int(bool(graph)),
                                         C.byref(bad)))
        return int(bad.value)

    def _peer_self_test(self, rounds=None):
        """The transport has to prove itself on the machine it runs on before the solvers rely on it: one global sum and three
        ring exchanges with known answers, then `rounds` (default 1 000 each, PALACE_AMD_PEER_SELFTEST_ROUNDS) back-to-back
        rounds of P / P^T / all-reduce with payloads that change every round and are verified on the device -- through the local
        vector, in the direct form of ParOperator::Mult (ghosts read from the mailbox in place), and as a recorded round replayed
        as a HIP graph.  Returns (ok on every rank, failures were wrong values only -- no timed-out wait --, report)."""
        import torch

        rank, size = self.rank, self.size
        rounds = int(os.environ.get("PALACE_AMD_PEER_SELFTEST_ROUNDS", "1000")) if rounds is None else int(rounds)
        ok, lost, rep = True, False, {"rounds": rounds}
        try:
            v = torch.tensor([float(rank + 1), 0.5 * rank], dtype=torch.float64, device="cuda")
            _lib.check(_L().pa_allreduce_sum(self.handle, C.c_void_p(v.data_ptr()), 2))
            ok = bool(torch.allclose(v.cpu(), torch.tensor([size * (size + 1) / 2.0, 0.25 * size * (size - 1)], dtype=torch.float64)))
            n = 8
            left = (rank - 1) % size
            h = self._ring_plan(n)
            for rep_i in range(3):  # (more than two: both mailbox buffers and the acknowledgements)
                lx = torch.zeros(2 * n, dtype=torch.float64, device="cuda")
                lx[:n] = torch.arange(n, dtype=torch.float64, device="cuda") + 100.0 * rank + rep_i
                _lib.check(_L().pa_halo_prolongate(self.handle, h.handle, C.c_void_p(lx.data_ptr())))
                want = torch.arange(n, dtype=torch.float64) + 100.0 * left + rep_i
                ok = ok and bool(torch.equal(lx[n:].cpu(), want))
                _lib.check(_L().pa_halo_restrict_add(self.handle, h.handle, C.c_void_p(lx.data_ptr())))
                want = (torch.arange(n, dtype=torch.float64) + 100.0 * rank + rep_i) * 2.0
                ok = ok and bool(torch.equal(lx[:n].cpu(), want))
            self.peer_check()
            del h
            rep["known_answers"] = ok
            ring = self._ring_plan(4096)
            for name, direct, graph in (("lvector", False, False), ("direct", True, False), ("direct_graph", True, True)):
                bad = self.peer_stress(rounds, 4096, direct=direct, graph=graph, ring=ring)
                rep[name + "_wrong_values"] = bad
                ok = ok and bad == 0
            del ring
        except Exception as exc:  # noqa: BLE001 -- a timed-out wait or a set-up failure: this rank votes 'no'
            ok, lost = False, True
            rep["error"] = str(exc)[:200]
        all_ok = self._all_ok(ok)
        values_only = self._all_ok(not lost)
        return all_ok, values_only, rep

    def init_comm_local(self, group, rank):
        """Rank `rank` of an in-process group of rank THREADS on one GPU (pa_local_group_*: test harness of the multi-rank paths)."""
        _lib.check(_L().pa_context_init_comm_local(self.handle, int(rank), group.handle))
        self.rank, self.size, self._group = int(rank), group.size, group

    def init_comm_peer_from_torch_distributed(self):
        """Communicator on the peer transport alone (no RCCL): halo exchanges and global sums as direct stores between the
        ranks' device arenas.  torch.distributed (any backend, e.g. gloo) only carries the 64-byte IPC handles once.  Works
        for several processes on ONE GPU as well, which RCCL refuses.  Raises (on every rank) if the transport cannot be
        brought up: there is nothing to fall back to."""
        import torch.distributed as dist

        rank, size = dist.get_rank(), dist.get_world_size()
        _lib.check(_L().pa_context_init_comm_peer(self.handle, rank, size))
        self.rank, self.size = rank, size
        self.transport_report = {"transport": "peer"}
        if size > 1:
            self._peer_bring_up(fallback=False)

    def init_comm_peer_single(self):
        """One-rank communicator on the peer transport (a plan naming rank 0 as its own neighbour exercises every kernel of
        an exchange without a second rank: scripts/time_halo_mult.py)."""
        _lib.check(_L().pa_context_init_comm_peer(self.handle, 0, 1))
        self.rank, self.size = 0, 1

    def peer_ready(self):
        """True when halo exchanges and global sums run over the peer transport (arenas connected)."""
        return bool(_L().pa_comm_peer_ready(self.handle))

    def peer_check(self):
        """Raises if a wait of the peer transport has timed out."""
        _lib.check(_L().pa_comm_peer_check(self.handle))

    def init_comm_single(self):
        """One-rank communicator (exercises RCCL init / allreduce without a second GPU)."""
        L = _L()
        buf = C.create_string_buffer(128)
        _lib.check(L.pa_comm_unique_id(buf))
        _lib.check(L.pa_context_init_comm(self.handle, 0, 1, buf.raw))

    def synchronize(self):
        _lib.check(_L().pa_context_synchronize(self.handle))

    def dot(self, x, y):
        out = C.c_double()
        _lib.check(_L().pa_vec_dot(self.handle, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                   C.c_int(x.numel()), C.byref(out)))
        return out.value

    def sum(self, x):
        """linalg::Sum (vector.cpp:687-699 + global sum)."""
        out = C.c_double()
        _lib.check(_L().pa_vec_sum(self.handle, C.c_void_p(x.data_ptr()), C.c_int(x.numel()), C.byref(out)))
        return out.value

    def sqrt(self, x, s=1.0):
        """linalg::Sqrt: x = sqrt(s x) (vector.cpp:774-781)."""
        _lib.check(_L().pa_vec_sqrt(self.handle, C.c_void_p(x.data_ptr()), C.c_int(x.numel()), C.c_double(s)))
        return x

    def axpby(self, a, x, b, y):
        """y = a x + b y (linalg::AXPBY, vector.cpp:530-557)."""
        _lib.check(_L().pa_vec_axpby(self.handle, a, C.c_void_p(x.data_ptr()), b, C.c_void_p(y.data_ptr()),
                                     C.c_int(x.numel())))
        return y

    def bench_mfma_f64(self, iters, n_blocks, scratch):
        """Launch the FP64 matrix-core peak micro-kernel once; returns the flops of the launch."""
        fl = C.c_double(0.0)
        _lib.check(_L().pa_bench_mfma_f64(self.handle, C.c_int(iters), C.c_int(n_blocks), C.c_void_p(scratch.data_ptr()),
                                          C.byref(fl)))
        return fl.value

    def orthogonalize_column(self, kind, V, w, weight=None):
        """OrthogonalizeColumnMGS / CGS / CGS2 (orthog.hpp:41-89): returns H, updates w in place.  kind in
        {"MGS", "CGS", "CGS2"}; V = list of device vectors; weight = ParOperator or None."""
        m = len(V)
        k = {"MGS": 0, "CGS": 1, "CGS2": 2}[kind]
        ptrs = (C.c_void_p * max(m, 1))(*[v.data_ptr() for v in V])
        H = np.zeros(max(m, 1), dtype=np.float64)
        _lib.check(_L().pa_orthogonalize_column(self.handle, C.c_int(k), C.c_int(m), ptrs, C.c_void_p(w.data_ptr()),
                                                C.c_int(w.numel()), _ptr(H), weight.handle if weight else None))
        return H[:m]

    def orthogonalize_column_complex(self, kind, Vr, Vi, wr, wi, weight=None):
        """The complex form (ComplexVector = separate real / imaginary vectors): returns complex H."""
        m = len(Vr)
        k = {"MGS": 0, "CGS": 1, "CGS2": 2}[kind]
        pr = (C.c_void_p * max(m, 1))(*[v.data_ptr() for v in Vr])
        pi = (C.c_void_p * max(m, 1))(*[v.data_ptr() for v in Vi])
        H = np.zeros(2 * max(m, 1), dtype=np.float64)
        _lib.check(_L().pa_orthogonalize_column_complex(self.handle, C.c_int(k), C.c_int(m), pr, pi,
                                                        C.c_void_p(wr.data_ptr()), C.c_void_p(wi.data_ptr()),
                                                        C.c_int(wr.numel()), _ptr(H), weight.handle if weight else None))
        return H[0:2 * m:2] + 1j * H[1:2 * m:2]

    @staticmethod
    def set_device_orthogonalization(on):
        _lib.check(_L().pa_set_device_orthogonalization(C.c_int(1 if on else 0)))

    @staticmethod
    def resident_columns():
        """Columns orthogonalised so far with w resident in the register file (orthog.hip: k_mgs_resident)."""
        f = _L().pa_orthog_resident_columns
        f.restype = C.c_longlong
        return int(f())

    def orthonormalize_column(self, kind, V, w):
        """One Arnoldi column (iterative.cpp:629-633): orthogonalise, norm, normalise; returns (H, hn)."""
        m = len(V)
        k = {"MGS": 0, "CGS": 1, "CGS2": 2}[kind]
        ptrs = (C.c_void_p * max(m, 1))(*[v.data_ptr() for v in V])
        H = np.zeros(max(m, 1), dtype=np.float64)
        hn = C.c_double(0.0)
        _lib.check(_L().pa_orthonormalize_column(self.handle, C.c_int(k), C.c_int(m), ptrs, C.c_void_p(w.data_ptr()),
                                                 C.c_int(w.numel()), _ptr(H), C.byref(hn)))
        return H[:m], hn.value

    def orthonormalize_column_complex(self, kind, Vr, Vi, wr, wi):
        m = len(Vr)
        k = {"MGS": 0, "CGS": 1, "CGS2": 2}[kind]
        pr = (C.c_void_p * max(m, 1))(*[v.data_ptr() for v in Vr])
        pi = (C.c_void_p * max(m, 1))(*[v.data_ptr() for v in Vi])
        H = np.zeros(2 * max(m, 1), dtype=np.float64)
        hn = C.c_double(0.0)
        _lib.check(_L().pa_orthonormalize_column_complex(self.handle, C.c_int(k), C.c_int(m), pr, pi,
                                                         C.c_void_p(wr.data_ptr()), C.c_void_p(wi.data_ptr()),
                                                         C.c_int(wr.numel()), _ptr(H), C.byref(hn)))
        return H[0:2 * m:2] + 1j * H[1:2 * m:2], hn.value

    def set_random(self, x, seed):
        _lib.check(_L().pa_vec_set_random(self.handle, C.c_void_p(x.data_ptr()), x.numel(), seed))
        return x

    def __del__(self):
        try:
            _L().pa_context_destroy(self.handle)
        except Exception:
            pass


class phase_range:
    """`with phase_range("Time Stepping"): ...` -- a named roctx range around a phase (pa_range_push / pa_range_pop)."""

    def __init__(self, name):
        self.name, self.h = name, None

    def __enter__(self):
        L = _L()
        L.pa_range_push.restype = C.c_void_p
        L.pa_range_push.argtypes = [C.c_char_p]
        self.h = L.pa_range_push(self.name.encode())
        return self

    def __exit__(self, *exc):
        L = _L()
        L.pa_range_pop.restype = None
        L.pa_range_pop.argtypes = [C.c_void_p]
        L.pa_range_pop(self.h)
        return False


class LocalGroup:
    """pa_local_group: the rendezvous object shared by the rank threads of an in-process group."""

    def __init__(self, size):
        L = _L()
        L.pa_local_group_destroy.restype = None
        L.pa_local_group_destroy.argtypes = [C.c_void_p]
        self.size = int(size)
        self.handle = C.c_void_p()
        _lib.check(L.pa_local_group_create(self.size, C.byref(self.handle)))

    def abort(self):
        """Called by a failing rank thread: the others leave their barriers with an error."""
        L = _L()
        L.pa_local_group_abort.restype = None
        L.pa_local_group_abort.argtypes = [C.c_void_p]
        L.pa_local_group_abort(self.handle)

    def __del__(self):
        try:
            _L().pa_local_group_destroy(self.handle)
        except Exception:
            pass


class Halo:
    def __init__(self, ctx: Context, nbr, send_lists, recv_lists):
        """send_lists[k] / recv_lists[k]: local dof indices exchanged with neighbour nbr[k]."""
        self.ctx = ctx
        nbr = np.ascontiguousarray(nbr, dtype=np.int32)
        so = np.zeros(len(nbr) + 1, dtype=np.int32)
        ro = np.zeros(len(nbr) + 1, dtype=np.int32)
        so[1:] = np.cumsum([len(s) for s in send_lists])
        ro[1:] = np.cumsum([len(r) for r in recv_lists])
        si = np.ascontiguousarray(np.concatenate(send_lists) if len(nbr) else np.zeros(0), dtype=np.int32)
        ri = np.ascontiguousarray(np.concatenate(recv_lists) if len(nbr) else np.zeros(0), dtype=np.int32)
        self.handle = C.c_void_p()
        _lib.check(_L().pa_halo_create(ctx.handle, len(nbr), _ptr(nbr), _ptr(so), _ptr(si), _ptr(ro), _ptr(ri),
                                       C.byref(self.handle)))

    def __del__(self):
        try:
            _L().pa_halo_destroy(self.handle)
        except Exception:
            pass


class ParOperator:
    """palace::ParOperator (rap.cpp:154-234) on T-vectors."""

    def direct_form(self):
        """1: the multi-rank Mult runs without L-vector copies (peer transport + split-vector apply), 0: available, off, -1: n/a."""
        return int(_L().pa_par_op_direct_form(self.handle))

    def set_direct(self, on=True):
        _lib.check(_L().pa_par_op_set_direct(self.handle, int(bool(on))))

    def __init__(self, ctx: Context, local: Operator, ess_tdofs, diag_policy=DIAG_ONE, n_true=None, halo=None):
        self.ctx, self.local, self.halo = ctx, local, halo
        self.n = local.height if n_true is None else n_true
        ess = np.ascontiguousarray(ess_tdofs, dtype=np.int32)
        self.ess = ess
        self.handle = C.c_void_p()
        _lib.check(_L().pa_par_op_create(ctx.handle, local.handle, self.n, _ptr(ess), ess.size, diag_policy,
                                         halo.handle if halo else None, C.byref(self.handle)))

    def mult(self, x, y):
        _lib.check(_L().pa_par_op_mult(self.handle, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())))
        return y

    def add_mult(self, x, y, a=1.0):
        _lib.check(_L().pa_par_op_add_mult(self.handle, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                           C.c_double(a)))
        return y

    def eliminate_rhs(self, x, b):
        _lib.check(_L().pa_par_op_eliminate_rhs(self.handle, C.c_void_p(x.data_ptr()), C.c_void_p(b.data_ptr())))
        return b

    def assemble_diagonal(self, d):
        _lib.check(_L().pa_par_op_assemble_diagonal(self.handle, C.c_void_p(d.data_ptr())))
        return d

    def mult_transpose(self, x, y):
        """ParOperator::MultTranspose (rap.cpp:236-275)."""
        _lib.check(_L().pa_par_op_mult_transpose(self.handle, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())))
        return y

    def __del__(self):
        try:
            _L().pa_par_op_destroy(self.handle)
        except Exception:
            pass


class AssembledParOperator(ParOperator):
    """ParOperator around an assembled local operator (ParOperator::ParallelAssemble, rap.cpp:84-152): the coarsest
    level as a device CSR matrix; `csr` is a ceed.DeviceCsr from Operator.full_assemble_device()."""

    def __init__(self, ctx: Context, csr, ess_tdofs, diag_policy=DIAG_ONE, n_true=None, halo=None):
        self.ctx, self.local, self.halo = ctx, csr, halo
        self.n = csr.nrows if n_true is None else n_true
        ess = np.ascontiguousarray(ess_tdofs, dtype=np.int32)
        self.ess = ess
        self.handle = C.c_void_p()
        _lib.check(_L().pa_par_op_create_assembled(ctx.handle, csr.handle, self.n, _ptr(ess), ess.size, diag_policy,
                                                   halo.handle if halo else None, C.byref(self.handle)))


class ParSumOperator(ParOperator):
    """BuildParSumOperator (rap.cpp:843-919): ParOperator around sum_k coeffs[k] * locals[k]."""

    def __init__(self, ctx: Context, locals_, coeffs, ess_tdofs, diag_policy=DIAG_ONE, n_true=None, halo=None):
        self.ctx, self.local, self.halo = ctx, list(locals_), halo
        self.n = self.local[0].height if n_true is None else n_true
        ess = np.ascontiguousarray(ess_tdofs, dtype=np.int32)
        self.ess = ess
        hs = (C.c_void_p * len(self.local))(*[o.handle for o in self.local])
        cs = np.ascontiguousarray(coeffs, dtype=np.float64)
        self.handle = C.c_void_p()
        _lib.check(_L().pa_par_sum_op_create(ctx.handle, len(self.local), hs, _ptr(cs), self.n, _ptr(ess), ess.size,
                                             diag_policy, halo.handle if halo else None, C.byref(self.handle)))


class Solver:
    def __init__(self, ctx, handle, keep=()):
        self.ctx, self.handle, self._keep = ctx, handle, keep
        self._owned_by_parent = False

    def mult(self, b, x, initial_guess=False):
        _lib.check(_L().pa_solver_mult(self.handle, C.c_void_p(b.data_ptr()), C.c_void_p(x.data_ptr()),
                                       int(initial_guess)))
        return x

    def stats(self):
        its, conv = C.c_int(), C.c_int()
        r0, r1 = C.c_double(), C.c_double()
        _lib.check(_L().pa_solver_stats(self.handle, C.byref(its), C.byref(r0), C.byref(r1), C.byref(conv)))
        return dict(iterations=its.value, initial_res=r0.value, final_res=r1.value, converged=bool(conv.value))

    def check_status(self):
        """Raise what a nested solver noted on the device while it ran unattended (Solver::CheckStatus)."""
        _lib.check(_L().pa_solver_check_status(self.handle))

    def mult2(self, x, y, transpose=False, initial_guess=False):
        """Solver::Mult2 / MultTranspose2: y <- y + B (x - A y)."""
        _lib.check(_L().pa_solver_mult2(self.handle, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), int(transpose),
                                        int(initial_guess)))
        return y

    def dist_relaxation_lambda_max(self):
        a, b = C.c_double(), C.c_double()
        _lib.check(_L().pa_dist_relaxation_lambda_max(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_lookahead(self, lookahead=1, host_scalars=False):
        _lib.check(_L().pa_cg_set_lookahead(self.handle, int(lookahead), int(bool(host_scalars))))
        return self

    def gmg_lambda_max(self, level):
        v = C.c_double()
        _lib.check(_L().pa_gmg_smoother_lambda_max(self.handle, int(level), C.byref(v)))
        return v.value

    def lambda_max(self):
        v = C.c_double()
        _lib.check(_L().pa_chebyshev_lambda_max(self.handle, C.byref(v)))
        return v.value

    def fused_step(self, level=-1):
        """True when the Chebyshev smoother (this solver, or level `level` of a multigrid solver) runs its steps inside the
        operator's E^T (pa_chebyshev_fused_step)."""
        v = C.c_int()
        _lib.check(_L().pa_chebyshev_fused_step(self.handle, level, C.byref(v)))
        return bool(v.value)

    def __del__(self):
        try:
            if not self._owned_by_parent:
                _L().pa_solver_destroy(self.handle)
        except Exception:
            pass


def chebyshev(ctx, A: ParOperator, order, smooth_it=1, sf_max=1.0, fourth_kind=True, sf_min=0.0):
    """ChebyshevSmoother (4th kind, chebyshev.cpp:160-220) or ChebyshevSmoother1stKind (:222-293)."""
    h = C.c_void_p()
    if fourth_kind:
        _lib.check(_L().pa_chebyshev_create(ctx.handle, A.handle, smooth_it, order, sf_max, 1, C.byref(h)))
    else:
        L = _L()
        L.pa_chebyshev_create_1st_kind.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double,
                                                   C.c_void_p]
        _lib.check(L.pa_chebyshev_create_1st_kind(ctx.handle, A.handle, smooth_it, order, sf_max, sf_min, C.byref(h)))
    return Solver(ctx, h, (A,))


def dist_relaxation(ctx, A: ParOperator, A_aux: ParOperator, G, smooth_it=1, cheby_smooth_it=1, cheby_order=4, sf_max=1.0,
                    sf_min=0.0, fourth_kind=True):
    """DistRelaxationSmoother (linalg/distrelaxation.cpp:14-151) on its own."""
    L = _L()
    L.pa_dist_relaxation_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.c_double, C.c_double, C.c_int, C.c_void_p]
    h = C.c_void_p()
    _lib.check(L.pa_dist_relaxation_create(ctx.handle, A.handle, A_aux.handle, G.handle, smooth_it, cheby_smooth_it,
                                           cheby_order, sf_max, sf_min, int(fourth_kind), C.byref(h)))
    return Solver(ctx, h, (A, A_aux, G))


def jacobi(ctx, A: ParOperator):
    h = C.c_void_p()
    _lib.check(_L().pa_jacobi_create(ctx.handle, A.handle, C.byref(h)))
    return Solver(ctx, h, (A,))


class _AmgOptions(C.Structure):
    _fields_ = [("max_levels", C.c_int), ("coarse_size", C.c_int), ("smooth_order", C.c_int), ("theta", C.c_double)]


class _AmsOptions(C.Structure):
    _fields_ = [("cycle_it", C.c_int), ("smooth_order", C.c_int), ("singular", C.c_int), ("amg", _AmgOptions)]


def amg(ctx, csr, ess_tdofs=(), max_levels=0, coarse_size=0, smooth_order=0, theta=0.0):
    """Native algebraic multigrid V-cycle on an assembled matrix (`csr`: ceed.DeviceCsr), where the reference calls
    BoomerAMG (linalg/amg.cpp:12-49).  Zero options: the library's defaults."""
    ess = np.ascontiguousarray(ess_tdofs, dtype=np.int32)
    opt = _AmgOptions(max_levels, coarse_size, smooth_order, theta)
    h = C.c_void_p()
    L = _L()
    L.pa_amg_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _lib.check(L.pa_amg_create(ctx.handle, csr.handle, _ptr(ess), ess.size, C.byref(opt), C.byref(h)))
    return Solver(ctx, h, (csr,))


def replicated(ctx, gather_halo, inner, mine_global, n_global, sign=None):
    """ReplicatedSolver: `inner` (a solver of the global problem, the same on every rank) applied to the right-hand side gathered
    from all ranks through `gather_halo` (a Halo on the global-numbered vector); mine_global[i] = global number of true dof i,
    sign[i] = +-1 its orientation relative to the global dof (None: all +1)."""
    mine = np.ascontiguousarray(mine_global, dtype=np.int32)
    sg = None if sign is None else np.ascontiguousarray(sign, dtype=np.float64)
    h = C.c_void_p()
    L = _L()
    L.pa_replicated_solver_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    _lib.check(L.pa_replicated_solver_create(ctx.handle, gather_halo.handle, inner.handle, _ptr(mine),
                                             _ptr(sg) if sg is not None else None, mine.size, int(n_global), C.byref(h)))
    return Solver(ctx, h, (gather_halo, inner, mine, sg))


def replicated_coarse(ctx, level0: "ParOperator", G=None, nv_true=0, xyz_true=None, cycle_it=1, singular=False):
    """ReplicatedCoarseSolver (ksp.hpp): the coarsest level of a multi-rank hierarchy solved redundantly by every rank with the
    native AMS (G: the level's discrete gradient, xyz_true [nv_true, dim]: this rank's true vertices) or AMG (G = None) cycle,
    assembled by the C++ layer from the ranks' own pieces -- what KspSolver's LinearSolver::AMS / BOOMER_AMG do on a space with a
    halo.  Collective."""
    L = _L()
    L.pa_replicated_coarse_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p]
    xyz = None if xyz_true is None else np.ascontiguousarray(xyz_true, dtype=np.float64)
    dim = 3 if xyz is None else int(xyz.shape[1])
    h = C.c_void_p()
    _lib.check(L.pa_replicated_coarse_create(ctx.handle, level0.handle, G.handle if G is not None else None, int(nv_true),
                                             _ptr(xyz) if xyz is not None else None, dim, int(cycle_it), int(bool(singular)), C.byref(h)))
    s = Solver(ctx, h, (level0, G))
    dist, lev = C.c_int(0), C.c_int(0)
    L.pa_replicated_coarse_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    _lib.check(L.pa_replicated_coarse_info(h, C.byref(dist), C.byref(lev)))
    s.distributed, s.algebraic_levels = bool(dist.value), int(lev.value)  # (amg_dist.hpp; PALACE_AMD_COARSE_SOLVE=replicated: False)
    return s


def ams(ctx, csr, ess_tdofs, G, coords, cycle_it=0, smooth_order=0, singular=False, amg_coarse_size=0, amg_smooth_order=0,
        amg_theta=0.0):
    """Native auxiliary-space (Hiptmair-Xu) preconditioner for an assembled lowest-order H(curl) matrix, where the reference
    calls HYPRE's AMS (linalg/ams.cpp:18-224).  G: scipy CSR discrete gradient [edges x vertices], coords: [vertices, dim]."""
    ess = np.ascontiguousarray(ess_tdofs, dtype=np.int32)
    G = G.tocsr()
    G.sort_indices()
    rp = np.ascontiguousarray(G.indptr, dtype=np.int32)
    ci = np.ascontiguousarray(G.indices, dtype=np.int32)
    va = np.ascontiguousarray(G.data, dtype=np.float64)
    xy = np.ascontiguousarray(coords, dtype=np.float64)
    assert xy.shape[0] == G.shape[1]
    opt = _AmsOptions(cycle_it, smooth_order, int(singular), _AmgOptions(0, amg_coarse_size, amg_smooth_order, amg_theta))
    h = C.c_void_p()
    L = _L()
    L.pa_ams_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _lib.check(L.pa_ams_create(ctx.handle, csr.handle, _ptr(ess), ess.size, G.shape[1], _ptr(rp), _ptr(ci), _ptr(va),
                               _ptr(xy), xy.shape[1], C.byref(opt), C.byref(h)))
    return Solver(ctx, h, (csr,))


def amg_hierarchy(solver, which=0):
    """Host copies of an AMG hierarchy (which: 0 the solver itself, 1 / 2 the gradient- / nodal-space solver of an AMS solver):
    (A_levels, P_levels, dense inverse of the last level or None) as scipy / numpy objects."""
    import scipy.sparse as sp

    L = _L()
    L.pa_amg_get_matrix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]
    nl = C.c_int()
    _lib.check(L.pa_amg_num_levels(solver.handle, which, C.byref(nl)))

    def get(level, kind):
        nr, nc, nnz = C.c_int32(), C.c_int32(), C.c_int64()
        _lib.check(L.pa_amg_get_matrix(solver.handle, which, level, kind, C.byref(nr), C.byref(nc), C.byref(nnz), None, None,
                                       None))
        if kind == 2:
            val = np.empty(nnz.value)
            _lib.check(L.pa_amg_get_matrix(solver.handle, which, level, kind, None, None, None, None, None, _ptr(val)))
            return val.reshape(nr.value, nr.value)
        rp, ci, va = np.empty(nr.value + 1, np.int32), np.empty(nnz.value, np.int32), np.empty(nnz.value)
        _lib.check(L.pa_amg_get_matrix(solver.handle, which, level, kind, None, None, None, _ptr(rp), _ptr(ci), _ptr(va)))
        return sp.csr_matrix((va, ci, rp), shape=(nr.value, nc.value))

    A = [get(l, 0) for l in range(nl.value)]
    P = [get(l, 1) for l in range(nl.value - 1)]
    try:
        cinv = get(0, 2)
    except RuntimeError:
        cinv = None
    return A, P, cinv


def cg(ctx, A: ParOperator, precond=None, rel_tol=0.0, abs_tol=0.0, max_it=100, print_level=0):
    h = C.c_void_p()
    _lib.check(_L().pa_cg_create(ctx.handle, A.handle, precond.handle if precond else None, rel_tol, abs_tol,
                                 max_it, print_level, C.byref(h)))
    return Solver(ctx, h, (A, precond))


def gmres(ctx, A: ParOperator, precond=None, rel_tol=0.0, abs_tol=0.0, max_it=100, restart=-1, flexible=False,
          print_level=0, orthogonalization="MGS", pc_side="left"):
    """GmresSolver (iterative.cpp:543-705; pc_side 'left' | 'right') / FgmresSolver (flexible=True, :733-871)."""
    h = C.c_void_p()
    _lib.check(_L().pa_gmres_create(ctx.handle, A.handle, precond.handle if precond else None, rel_tol, abs_tol,
                                    max_it, restart, int(flexible), print_level, C.byref(h)))
    _lib.check(_L().pa_gmres_set_orthogonalization(h, {"MGS": 0, "CGS": 1, "CGS2": 2}[orthogonalization]))
    if not flexible:
        _lib.check(_L().pa_gmres_set_pc_side(h, {"left": 0, "right": 1}[pc_side]))
    return Solver(ctx, h, (A, precond))


class Interp:
    """p-prolongation between two spaces on the same mesh (bilinearform.cpp:203-282)."""

    def __init__(self, ctx, space_c, space_f, coarse_halo=None, n_true_c=None, n_true_f=None):
        self.ctx = ctx
        pc, pf = space_c.p, space_f.p
        Ic, _ = lagrange_eval(gauss_lobatto(pc + 1), gauss_lobatto(pf + 1))
        Io, _ = lagrange_eval(gauss_legendre(pc)[0], gauss_legendre(pf)[0])
        Ic, Io = np.ascontiguousarray(Ic), np.ascontiguousarray(Io)
        rc, k1 = _restriction_desc(space_c)
        rf, k2 = _restriction_desc(space_f)
        bc, k3 = _basis_desc(space_c, pf + 1)
        bf, k4 = _basis_desc(space_f, pf + 1)
        self.handle = C.c_void_p()
        _lib.check(_L().pa_interp_create(ctx.handle, C.byref(rc), C.byref(bc), C.byref(rf), C.byref(bf), _ptr(Ic),
                                         _ptr(Io), coarse_halo.handle if coarse_halo else None,
                                         space_c.ndofs if n_true_c is None else n_true_c,
                                         space_f.ndofs if n_true_f is None else n_true_f, C.byref(self.handle)))

    def mult(self, x, y):
        _lib.check(_L().pa_interp_mult(self.handle, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())))
        return y

    def mult_transpose(self, x, y):
        _lib.check(_L().pa_interp_mult_transpose(self.handle, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())))
        return y

    def __del__(self):
        try:
            _L().pa_interp_destroy(self.handle)
        except Exception:
            pass


class DenseInterp(Interp):
    """Element-matrix interpolator between two spaces on the same non-tensor mesh (p-prolongation or
    discrete gradient on tetrahedra): pa_interp_create_dense.  `dom` / `rng` are dicts with `offsets`
    [ne, P], `lsize`, and optionally `orients` (bool) or `curl_orients` (int8 [ne, P, 3]; for the range the
    dual-inverse form, restriction.cpp:318-336); M is the [P_range, P_domain] element matrix."""

    def __init__(self, ctx, dom, rng, M, dom_halo=None, n_true_dom=None, n_true_rng=None):
        self.ctx = ctx
        keep = []

        def desc(r):
            off = np.ascontiguousarray(r["offsets"], dtype=np.int32)
            ori = None if r.get("orients") is None else np.ascontiguousarray(r["orients"], dtype=np.uint8)
            cor = None if r.get("curl_orients") is None else np.ascontiguousarray(r["curl_orients"], dtype=np.int8)
            keep.extend([off, ori, cor])
            return _lib.RestrictionDesc(off.shape[0], off.shape[1], int(r["lsize"]), _ptr(off), _ptr(ori), _ptr(cor))

        rd, rr = desc(dom), desc(rng)
        M = np.ascontiguousarray(M, dtype=np.float64)
        assert M.shape == (rng["offsets"].shape[1], dom["offsets"].shape[1])
        self.handle = C.c_void_p()
        _lib.check(_L().pa_interp_create_dense(ctx.handle, C.byref(rd), C.byref(rr), _ptr(M),
                                               dom_halo.handle if dom_halo else None,
                                               int(dom["lsize"]) if n_true_dom is None else n_true_dom,
                                               int(rng["lsize"]) if n_true_rng is None else n_true_rng,
                                               C.byref(self.handle)))


class RefinementTransfer(Interp):
    """Prolongation between the spaces of one collection on a mesh and on its uniform refinement (mfem::TransferOperator for
    two levels on different meshes, fem/fespace.cpp:246-251): pa_interp_create_refinement.  `dom`: dict(offsets [ne_fine, P]
    = the PARENT's dofs per fine element, lsize[, orients]), `rng`: the fine space's restriction, M [nmat, P, P] the local
    interpolation matrices, mat_id [ne_fine] (fem/htransfer.py builds all four)."""

    def __init__(self, ctx, dom, rng, M, mat_id, dom_halo=None, n_true_dom=None, n_true_rng=None):
        self.ctx = ctx
        keep = []

        def desc(r):
            off = np.ascontiguousarray(r["offsets"], dtype=np.int32)
            ori = None if r.get("orients") is None else np.ascontiguousarray(r["orients"], dtype=np.uint8)
            keep.extend([off, ori])
            return _lib.RestrictionDesc(off.shape[0], off.shape[1], int(r["lsize"]), _ptr(off), _ptr(ori), None)

        rd, rr = desc(dom), desc(rng)
        M = np.ascontiguousarray(M, dtype=np.float64)
        mid = np.ascontiguousarray(mat_id, dtype=np.uint8)
        assert M.ndim == 3 and M.shape[1:] == (rng["offsets"].shape[1], dom["offsets"].shape[1]) and mid.size == rd.num_elem
        self.handle = C.c_void_p()
        L = _L()
        L.pa_interp_create_refinement.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_int, C.c_int, C.c_void_p]
        _lib.check(L.pa_interp_create_refinement(ctx.handle, C.byref(rd), C.byref(rr), M.shape[0], _ptr(M), _ptr(mid),
                                                 dom_halo.handle if dom_halo else None,
                                                 int(dom["lsize"]) if n_true_dom is None else n_true_dom,
                                                 int(rng["lsize"]) if n_true_rng is None else n_true_rng, C.byref(self.handle)))


class Gradient(Interp):
    """Discrete gradient G : H1(p) -> ND(p) (the auxiliary-space transfer of the Hiptmair smoother)."""

    def __init__(self, ctx, h1_space, nd_space, h1_halo=None, n_true_h1=None, n_true_nd=None):
        self.ctx = ctx
        p = nd_space.p
        assert h1_space.p == p
        _, Dg = lagrange_eval(gauss_lobatto(p + 1), gauss_legendre(p)[0])  # [p][p+1]
        Dg = np.ascontiguousarray(Dg)
        rh, k1 = _restriction_desc(h1_space)
        rn, k2 = _restriction_desc(nd_space)
        bh, k3 = _basis_desc(h1_space, p + 1)
        bn, k4 = _basis_desc(nd_space, p + 1)
        self.handle = C.c_void_p()
        _lib.check(_L().pa_gradient_create(ctx.handle, C.byref(rh), C.byref(bh), C.byref(rn), C.byref(bn), _ptr(Dg),
                                           h1_halo.handle if h1_halo else None,
                                           h1_space.ndofs if n_true_h1 is None else n_true_h1,
                                           nd_space.ndofs if n_true_nd is None else n_true_nd, C.byref(self.handle)))


def gmg(ctx, A_levels, P_levels, coarse: Solver, cycle_it=1, smooth_it=1, cheby_order=4, sf_max=1.0, sf_min=0.0,
        fourth_kind=True, A_aux=None, G=None):
    """GeometricMultigridSolver (gmg.cpp); takes ownership of `coarse`.  With A_aux (H1 ParOperators)
    and G (discrete gradients) per level the smoother is the Hiptmair DistRelaxationSmoother."""
    n = len(A_levels)
    Ah = (C.c_void_p * n)(*[a.handle for a in A_levels])
    Ph = (C.c_void_p * max(1, n - 1))(*[p.handle for p in P_levels])
    h = C.c_void_p()
    if G is None:
        _lib.check(_L().pa_gmg_create(ctx.handle, n, Ah, Ph, coarse.handle, cycle_it, smooth_it, cheby_order,
                                      sf_max, sf_min, int(fourth_kind), C.byref(h)))
    else:
        Xh = (C.c_void_p * n)(*[(a.handle if a is not None else None) for a in A_aux])
        Gh = (C.c_void_p * n)(*[(g.handle if g is not None else None) for g in G])
        L = _L()
        L.pa_gmg_create_aux.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                        C.c_void_p]
        _lib.check(L.pa_gmg_create_aux(ctx.handle, n, Ah, Ph, Xh, Gh, coarse.handle, cycle_it, smooth_it,
                                       cheby_order, sf_max, sf_min, int(fourth_kind), C.byref(h)))
    coarse._owned_by_parent = True
    return Solver(ctx, h, (A_levels, P_levels, coarse, A_aux, G))


# ---- complex layer (ComplexVector = a pair of real tensors) ------------------------------------

class ComplexOperator:
    """Persistent ComplexWrapperOperator (linalg/operator.cpp:58-134): y = (Ar + i Ai) x."""

    def __init__(self, ctx, Ar, Ai):
        self.ctx, self._keep = ctx, (Ar, Ai)
        self.handle = C.c_void_p()
        _lib.check(_L().pa_complex_op_create(ctx.handle, Ar.handle if Ar else None, Ai.handle if Ai else None,
                                             C.byref(self.handle)))

    def mult(self, xr, xi, yr, yi):
        _lib.check(_L().pa_complex_op_apply(self.handle, C.c_void_p(xr.data_ptr()), C.c_void_p(xi.data_ptr()),
                                            C.c_void_p(yr.data_ptr()), C.c_void_p(yi.data_ptr())))
        return yr, yi

    def __del__(self):
        try:
            L = _L()
            L.pa_complex_op_destroy.restype = None
            L.pa_complex_op_destroy.argtypes = [C.c_void_p]
            L.pa_complex_op_destroy(self.handle)
        except Exception:
            pass


def complex_mult(ctx, Ar, Ai, xr, xi, yr, yi):
    """ComplexWrapperOperator::Mult (linalg/operator.cpp:98-134): y = (Ar + i Ai) x."""
    _lib.check(_L().pa_complex_op_mult(ctx.handle, Ar.handle if Ar else None, Ai.handle if Ai else None,
                                       C.c_void_p(xr.data_ptr()), C.c_void_p(xi.data_ptr()),
                                       C.c_void_p(yr.data_ptr()), C.c_void_p(yi.data_ptr())))
    return yr, yi


class ComplexGmres:
    """GmresSolver<ComplexOperator> (linalg/iterative.cpp:543-705), real preconditioner on both parts."""

    def __init__(self, ctx, Ar, Ai, precond=None, rel_tol=1e-8, abs_tol=0.0, max_it=200, restart=-1, print_level=0):
        L = _L()
        L.pa_complex_gmres_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                              C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pa_csolver_destroy.restype = None
        L.pa_csolver_destroy.argtypes = [C.c_void_p]
        self.ctx, self._keep = ctx, (Ar, Ai, precond)
        self.handle = C.c_void_p()
        _lib.check(L.pa_complex_gmres_create(ctx.handle, Ar.handle if Ar else None, Ai.handle if Ai else None,
                                             precond.handle if precond else None, rel_tol, abs_tol, max_it, restart,
                                             print_level, C.byref(self.handle)))

    def mult(self, br, bi, xr, xi, initial_guess=False):
        _lib.check(_L().pa_csolver_mult(self.handle, C.c_void_p(br.data_ptr()), C.c_void_p(bi.data_ptr()),
                                        C.c_void_p(xr.data_ptr()), C.c_void_p(xi.data_ptr()), int(initial_guess)))
        return xr, xi

    def stats(self):
        its, conv = C.c_int(), C.c_int()
        r0, r1 = C.c_double(), C.c_double()
        _lib.check(_L().pa_csolver_stats(self.handle, C.byref(its), C.byref(r0), C.byref(r1), C.byref(conv)))
        return dict(iterations=its.value, initial_res=r0.value, final_res=r1.value, converged=bool(conv.value))

    def __del__(self):
        try:
            _L().pa_csolver_destroy(self.handle)
        except Exception:
            pass


class ComplexParOperator:
    """palace::ComplexParOperator (linalg/rap.cpp:393-749) over two local ceed operators (either may be None)."""

    def __init__(self, ctx, local_r, local_i, ess_tdofs=(), diag_policy=DIAG_ONE, n_true=None, halo=None):
        L = _L()
        L.pa_complex_par_op_destroy.restype = None
        L.pa_complex_par_op_destroy.argtypes = [C.c_void_p]
        L.pa_complex_par_op_mult.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double] + [C.c_void_p] * 4
        L.pa_complex_par_op_local_mult.argtypes = L.pa_complex_par_op_mult.argtypes
        self.ctx, self._keep = ctx, (local_r, local_i, halo)
        any_op = local_r if local_r is not None else local_i
        self.n = any_op.height if n_true is None else n_true
        self.handle = C.c_void_p()
        _lib.check(L.pa_complex_par_op_create(ctx.handle, local_r.handle if local_r else None,
                                              local_i.handle if local_i else None, self.n,
                                              halo.handle if halo else None, C.byref(self.handle)))
        ess = np.ascontiguousarray(ess_tdofs, dtype=np.int32)
        if ess.size:
            _lib.check(L.pa_complex_par_op_set_essential(self.handle, _ptr(ess), ess.size, diag_policy))

    _MODES = {"N": 0, "T": 1, "H": 2}

    def mult(self, xr, xi, yr, yi, mode="N", a=None, local=False):
        """y = op(A) x (a is None) or y += a op(A) x; mode 'N' | 'T' | 'H'; local=True applies the L-vector
        ComplexWrapperOperator instead."""
        fn = _L().pa_complex_par_op_local_mult if local else _L().pa_complex_par_op_mult
        av = complex(a) if a is not None else 0j
        _lib.check(fn(self.handle, self._MODES[mode], int(a is not None), av.real, av.imag, C.c_void_p(xr.data_ptr()),
                      C.c_void_p(xi.data_ptr()), C.c_void_p(yr.data_ptr()), C.c_void_p(yi.data_ptr())))
        return yr, yi

    def assemble_diagonal(self, dr, di):
        _lib.check(_L().pa_complex_par_op_assemble_diagonal(self.handle, C.c_void_p(dr.data_ptr()),
                                                            C.c_void_p(di.data_ptr())))
        return dr, di

    def __del__(self):
        try:
            _L().pa_complex_par_op_destroy(self.handle)
        except Exception:
            pass


class ComplexParGmres(ComplexGmres):
    """GmresSolver / FgmresSolver <ComplexOperator> (linalg/iterative.cpp:543-871) on a ComplexParOperator.  `precond`: a real
    Solver (applied to both parts) or None; set_complex_preconditioner installs a ComplexSmoother instead."""

    def set_complex_preconditioner(self, smoother):
        _lib.check(_L().pa_csolver_set_complex_preconditioner(self.handle, smoother.handle))
        self._keep = self._keep + (smoother,)
        return self

    def __init__(self, ctx, A: ComplexParOperator, precond=None, rel_tol=1e-8, abs_tol=0.0, max_it=200, restart=-1,
                 flexible=False, pc_side="left", orthogonalization="MGS", print_level=0):
        L = _L()
        L.pa_complex_gmres_create_par.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pa_csolver_destroy.restype = None
        L.pa_csolver_destroy.argtypes = [C.c_void_p]
        self.ctx, self._keep = ctx, (A, precond)
        self.handle = C.c_void_p()
        _lib.check(L.pa_complex_gmres_create_par(ctx.handle, A.handle, precond.handle if precond else None, rel_tol,
                                                 abs_tol, max_it, restart, int(flexible),
                                                 {"left": 0, "right": 1}[pc_side],
                                                 {"MGS": 0, "CGS": 1, "CGS2": 2}[orthogonalization], print_level,
                                                 C.byref(self.handle)))


class ComplexParCg(ComplexGmres):
    """CgSolver<ComplexOperator> (linalg/iterative.cpp:360-486) on a ComplexParOperator (Hermitian positive definite systems)."""

    def __init__(self, ctx, A: ComplexParOperator, precond=None, rel_tol=1e-8, abs_tol=0.0, max_it=200, print_level=0):
        L = _L()
        L.pa_complex_cg_create_par.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int,
                                               C.c_void_p]
        L.pa_csolver_destroy.restype = None
        L.pa_csolver_destroy.argtypes = [C.c_void_p]
        self.ctx, self._keep = ctx, (A, precond)
        self.handle = C.c_void_p()
        _lib.check(L.pa_complex_cg_create_par(ctx.handle, A.handle, precond.handle if precond else None, rel_tol, abs_tol,
                                              max_it, print_level, C.byref(self.handle)))


class ComplexSmoother:
    """Solver<ComplexOperator> smoothers on a ComplexParOperator: 'jacobi', 'chebyshev' (4th kind), 'chebyshev1'
    (linalg/jacobi.cpp, chebyshev.cpp:160-293 with the complex inverse diagonal)."""

    def __init__(self, ctx, A: ComplexParOperator, kind="chebyshev", order=4, smooth_it=1, sf_max=1.0, sf_min=0.0):
        L = _L()
        L.pa_complex_smoother_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                                 C.c_void_p]
        L.pa_complex_smoother_destroy.restype = None
        L.pa_complex_smoother_destroy.argtypes = [C.c_void_p]
        self.ctx, self._keep = ctx, A
        self.handle = C.c_void_p()
        _lib.check(L.pa_complex_smoother_create(ctx.handle, A.handle, {"jacobi": 0, "chebyshev": 1, "chebyshev1": 2}[kind],
                                                smooth_it, order, sf_max, sf_min, C.byref(self.handle)))

    def lambda_max(self):
        v = C.c_double()
        _lib.check(_L().pa_complex_smoother_lambda_max(self.handle, C.byref(v)))
        return v.value

    def mult(self, xr, xi, yr, yi, initial_guess=False):
        _lib.check(_L().pa_complex_smoother_mult(self.handle, C.c_void_p(xr.data_ptr()), C.c_void_p(xi.data_ptr()),
                                                 C.c_void_p(yr.data_ptr()), C.c_void_p(yi.data_ptr()), int(initial_guess)))
        return yr, yi

    def __del__(self):
        try:
            _L().pa_complex_smoother_destroy(self.handle)
        except Exception:
            pass


class ComplexGmg(ComplexSmoother):
    """GeometricMultigridSolver<ComplexOperator> (gmg.cpp:16-205): ComplexParOperators per level (coarsest first), real
    prolongations, complex Chebyshev smoothers, a real coarse solver on both parts; takes ownership of `coarse`."""

    def __init__(self, ctx, A_levels, P_levels, coarse, cycle_it=1, smooth_it=1, cheby_order=4, sf_max=1.0, sf_min=0.0,
                 fourth_kind=True, A_aux=None, G=None):
        L = _L()
        L.pa_complex_gmg_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.pa_complex_smoother_destroy.restype = None
        L.pa_complex_smoother_destroy.argtypes = [C.c_void_p]
        n = len(A_levels)
        Ah = (C.c_void_p * n)(*[a.handle for a in A_levels])
        Ph = (C.c_void_p * max(1, n - 1))(*[q.handle for q in P_levels])
        self.ctx, self._keep = ctx, (A_levels, P_levels, coarse, A_aux, G)
        self.handle = C.c_void_p()
        Xh = Gh = None
        if G is not None:
            Xh = (C.c_void_p * n)(*[a.handle if a is not None else None for a in A_aux])
            Gh = (C.c_void_p * n)(*[g.handle if g is not None else None for g in G])
        _lib.check(L.pa_complex_gmg_create(ctx.handle, n, Ah, Ph, Xh, Gh, coarse.handle, cycle_it, smooth_it, cheby_order, sf_max,
                                           sf_min, int(fourth_kind), C.byref(self.handle)))
        coarse._owned_by_parent = True

    def level_lambda_max(self, level):
        """(primary, auxiliary) eigenvalue estimates of the level's smoother (auxiliary = 0 for plain Chebyshev)."""
        v = (C.c_double * 2)()
        _lib.check(_L().pa_complex_gmg_smoother_lambda_max(self.handle, int(level), v))
        return v[0], v[1]
