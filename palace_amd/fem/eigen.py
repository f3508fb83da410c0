"""BASELINE config 2's shape on the device: the eigenmode driver's inner loop around the device solver layer.

What the reference does per outer (Krylov-Schur / ARPACK) step (drivers/eigensolver.cpp:331-345, :469-477; linalg/arpack.cpp:661-670,
linalg/slepc.cpp shift-and-invert): y = (K - sigma^2 M)^-1 M x through its KspSolver, with M-weighted inner products, after a
divergence-free projection of the start vector (linalg/divfree.cpp: x -= G (G^T M G)^-1 G^T M x).  The OUTER eigensolver
(ARPACK / SLEPc) is out of scope; its stand-in here is a plain shift-and-invert Lanczos iteration with full
re-orthogonalisation, ~40 lines of host code -- every vector operation, operator apply, inner solve and inner product of the loop
is a call into the C ABI (device), only the tridiagonal eigenproblem of a few dozen rows is solved with numpy.

Host-side test / bench harness (Python, like the rest of palace_amd/fem): it stands where Palace's EigenSolver driver would call
the C++ layer (KspSolver::Mult, ParOperator::Mult, linalg::OrthogonalizeColumn with the weight operator)."""
import time

import numpy as np

C0 = 299792458.0  # m / s (utils/constants.hpp)


class HexEigenSystem:
    """K u = lambda eps M u on an order-p Nedelec space of a hex27 mesh, PEC on the whole boundary (essential rows: DIAG_ONE in K,
    DIAG_ZERO in M, drivers/eigensolver.cpp:41-43), lambda = (omega L0 / c0)^2 in mesh units:
      A = K - sigma^2 eps M          ParOperator over ONE fused curl-curl + mass ceed::Operator (negative mass coefficient)
      B ~ (K + sigma^2 eps M)^-1     p-multigrid (p = 1..order), Hiptmair or plain Chebyshev smoothers, native AMS on the assembled
                                     order-1 level -- the positive-shift preconditioner (config "PCMatShifted")
      inner solve                    FGMRES(A, B) to `tol`
      M                              ParOperator(mass), DIAG_ZERO
      div-free projection            H1 order-p diffusion (= G^T M G), PCG + p-multigrid + native AMG, discrete gradient G."""

    def __init__(self, ctx, mesh, order, target_ghz, eps_r=2.08, L0=1.0e-2, tol=1.0e-8, max_it=200, hiptmair=True,
                 divfree_tol=1.0e-10):
        from .. import ceed, linalg
        from .fespace import H1HexSpace, NDHexSpace, lowest_order_gradient, vertex_coordinates

        self.ctx, self.mesh, self.order, self.L0, self.eps_r = ctx, mesh, order, L0, eps_r
        self.sigma2 = (2.0 * np.pi * target_ghz * 1.0e9 * L0 / C0) ** 2
        q1d = order + 1
        orders = list(range(1, order + 1))
        nds = [NDHexSpace(mesh, q) for q in orders]
        h1s = [H1HexSpace(mesh, q) for q in orders]
        self.nd, self.n = nds[-1], nds[-1].ndofs
        ess = [s.ess_dofs() for s in nds]
        self.ess = ess[-1]
        geom = ceed.GeomFactorData(mesh, q1d)
        ident = ceed.coefficient_context(3)
        shift = self.sigma2 * eps_r
        cpos = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([shift])])
        cneg = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([-shift])])
        ceps = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([eps_r])])
        # operator of the inner solve and the mass operator of the inner products
        self.A_local = ceed.curlcurlmass_operator(geom, nds[-1], cneg, ident)
        self.A = linalg.ParOperator(ctx, self.A_local, ess[-1], linalg.DIAG_ONE)
        self.M_local = ceed.ndmass_operator(geom, nds[-1], ceps)
        self.M = linalg.ParOperator(ctx, self.M_local, ess[-1], linalg.DIAG_ZERO)
        self.K_local = ceed.curlcurl_operator(geom, nds[-1], ident)
        self.K = linalg.ParOperator(ctx, self.K_local, ess[-1], linalg.DIAG_ONE)
        # preconditioner: p-multigrid of K + sigma^2 eps M
        fine = ceed.curlcurlmass_operator(geom, nds[-1], cpos, ident)
        local = [fine.coarsen(geom, s) for s in nds[:-1]] + [fine]
        Al = [linalg.ParOperator(ctx, op, e, linalg.DIAG_ONE) for op, e in zip(local, ess)]
        keep = [geom, local, nds, h1s]
        if len(Al) > 1:
            csr0 = local[0].full_assemble_device()
            Al[0] = linalg.AssembledParOperator(ctx, csr0, ess[0], linalg.DIAG_ONE)
            P = [linalg.Interp(ctx, nds[l], nds[l + 1]) for l in range(len(Al) - 1)]
            aux = {}
            if hiptmair:
                fine_h1 = ceed.diffusion_operator(geom, h1s[-1], cpos)
                loc_h1 = [fine_h1.coarsen(geom, s) for s in h1s[:-1]] + [fine_h1]
                A_h1 = [linalg.ParOperator(ctx, op, s.ess_dofs(), linalg.DIAG_ONE) for op, s in zip(loc_h1, h1s)]
                G = [linalg.Gradient(ctx, h, n) for h, n in zip(h1s, nds)]
                aux = dict(A_aux=A_h1, G=G)
                keep += [loc_h1, A_h1, G]
            coarse = linalg.ams(ctx, csr0, ess[0], lowest_order_gradient(h1s[0], nds[0]), vertex_coordinates(h1s[0]))
            self.B = linalg.gmg(ctx, Al, P, coarse, cheby_order=max(2 * order, 4), **aux)
            keep += [csr0, P]
        else:
            self.B = linalg.jacobi(ctx, Al[0])
        keep += [Al]
        self.ksp = linalg.gmres(ctx, self.A, self.B, rel_tol=tol, max_it=max_it, restart=max_it, flexible=True)
        # divergence-free projection (linalg/divfree.cpp): G^T M G = the H1 diffusion operator with the mass coefficient
        d_fine = ceed.diffusion_operator(geom, h1s[-1], ceps)
        d_loc = [d_fine.coarsen(geom, s) for s in h1s[:-1]] + [d_fine]
        h1_ess = [s.ess_dofs() for s in h1s]
        D = [linalg.ParOperator(ctx, op, e, linalg.DIAG_ONE) for op, e in zip(d_loc, h1_ess)]
        if len(D) > 1:
            dcsr0 = d_loc[0].full_assemble_device()
            D[0] = linalg.AssembledParOperator(ctx, dcsr0, h1_ess[0], linalg.DIAG_ONE)
            DP = [linalg.Interp(ctx, h1s[l], h1s[l + 1]) for l in range(len(D) - 1)]
            DB = linalg.gmg(ctx, D, DP, linalg.amg(ctx, dcsr0, h1_ess[0]), cheby_order=max(2 * order, 4))
            keep += [dcsr0, DP]
        else:
            DB = linalg.jacobi(ctx, D[0])
        self.div_solver = linalg.cg(ctx, D[-1], DB, rel_tol=divfree_tol, max_it=200)
        self.Gf = linalg.Gradient(ctx, h1s[-1], nds[-1])
        self.h1_ess = h1_ess[-1]
        self.n_h1 = h1s[-1].ndofs
        keep += [d_loc, D, DB]
        self._keep = keep
        self.inner_its, self.inner_solves, self.inner_seconds = 0, 0, 0.0

    # ---- pieces of the outer iteration, all on the device ------------------------------------------------------------------
    def _vec(self, n=None):
        import torch

        return torch.zeros(self.n if n is None else n, dtype=torch.float64, device="cuda")

    def project_divfree(self, x):
        """x -= G (G^T M G)^-1 G^T M x (divfree.cpp:95-118); returns the PCG iteration count."""
        import torch

        mx, rhs, phi, gphi = self._vec(), self._vec(self.n_h1), self._vec(self.n_h1), self._vec()
        self.M.mult(x, mx)
        self.Gf.mult_transpose(mx, rhs)
        rhs[torch.from_numpy(self.h1_ess.astype(np.int64)).cuda()] = 0.0
        self.div_solver.mult(rhs, phi)
        self.Gf.mult(phi, gphi)
        x -= gphi
        x[torch.from_numpy(self.ess.astype(np.int64)).cuda()] = 0.0
        return self.div_solver.stats()["iterations"]

    def apply_shift_invert(self, x, y, mx):
        """y = (K - sigma^2 M)^-1 M x (the operator ARPACK / SLEPc iterate with, arpack.cpp:661-670)."""
        import torch

        self.M.mult(x, mx)
        y.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.ksp.mult(mx, y)
        torch.cuda.synchronize()
        self.inner_seconds += time.perf_counter() - t0
        st = self.ksp.stats()
        assert st["converged"], f"inner solve did not converge: {st}"
        self.inner_its += st["iterations"]
        self.inner_solves += 1

    def frequency_ghz(self, lam):
        return np.sqrt(np.maximum(lam, 0.0)) * C0 / self.L0 / (2.0 * np.pi) / 1.0e9

    def lanczos(self, steps, nev=3, seed=1, res_tol=1.0e-6, orthog="CGS2"):
        """Shift-and-invert Lanczos in the M inner product, full re-orthogonalisation by linalg::OrthogonalizeColumn with the weight
        operator (what the reference's eigensolver wrappers call for B-orthogonalisation); stops when the `nev` Ritz values nearest
        the target above it have residual estimates below res_tol (relative to theta) or after `steps` steps.  Returns a dict."""
        import torch

        ctx = self.ctx
        v = self._vec()
        ctx.set_random(v, seed)
        v[torch.from_numpy(self.ess.astype(np.int64)).cuda()] = 0.0
        div_its = self.project_divfree(v)
        mx, w = self._vec(), self._vec()
        self.M.mult(v, mx)
        v /= float(torch.sqrt(v @ mx))
        V, T = [v], np.zeros((steps + 1, steps))
        out, t0 = {}, time.perf_counter()
        k_done = 0
        for k in range(steps):
            self.apply_shift_invert(V[k], w, mx)
            h = ctx.orthogonalize_column(orthog, V, w, weight=self.M)
            T[: k + 1, k] = h
            self.M.mult(w, mx)
            beta = float(torch.sqrt(w @ mx))
            T[k + 1, k] = beta
            V.append(w / beta)
            w = self._vec()
            k_done = k + 1
            Tk = 0.5 * (T[:k_done, :k_done] + T[:k_done, :k_done].T)
            theta, S = np.linalg.eigh(Tk)
            lam = self.sigma2 + 1.0 / theta
            res = np.abs(beta * S[-1, :]) / np.abs(theta)
            phys = np.nonzero(theta > 0)[0]  # above the target (gradient fields sit at theta = -1 / sigma^2 < 0)
            order = phys[np.argsort(lam[phys])]
            if order.size >= nev and res[order[:nev]].max() < res_tol:
                break
        out["steps"] = k_done
        out["seconds"] = time.perf_counter() - t0
        out["lambda"] = lam[order]
        out["residual_estimates"] = res[order]
        out["frequencies_ghz"] = self.frequency_ghz(lam[order])
        out["divfree_pcg_iterations"] = div_its
        out["inner_iterations"] = self.inner_its
        out["inner_solves"] = self.inner_solves
        out["inner_seconds"] = self.inner_seconds
        # Ritz vector of the lowest mode and its Rayleigh quotient with the DEVICE operators (an independent check of lambda_0)
        if order.size:
            s0 = S[:, order[0]]
            x = self._vec()
            for j in range(k_done):
                x += float(s0[j]) * V[j]
            kx = self._vec()
            self.K.mult(x, kx)
            self.M.mult(x, mx)
            ess = torch.from_numpy(self.ess.astype(np.int64)).cuda()
            kx[ess] = 0.0
            out["rayleigh_quotient_0"] = float(x @ kx) / float(x @ mx)
        return out
