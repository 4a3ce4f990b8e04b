#!/usr/bin/env python3
"""Lid-driven cavity by Chorin projection on a MAC grid -- the CALLER of the hot path.

What the reference's FVM example does per time step (ldc_solver_common.py:137-224): wall ghost values, explicit
convection + diffusion predictor for the staggered velocities, pressure Poisson solve A p = div(u*)/dt, projection.
This is an independent implementation of that scheme whose pressure solve goes through THIS package's sparse path:
the pressure matrix is built once as CSR (`create_ldc_pressure_csr`, the matrix of ldc_solver_common.py:90-135), so
every step reuses the same device handle (coded SpMV form, built once), optionally warm-started from the previous
pressure -- where the reference hands a dense n x n matrix to the solver each step (ldc_solver_module_a.py:19-21).

    python examples/ldc_projection.py --nx 100 --steps 200 --method gmres [--warm-start] [--device cuda]
"""
import argparse
import os
import sys
import time
from dataclasses import dataclass, field

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_sparse_solver.module_a import bicgstab, get_last_stats, gmres  # noqa: E402
from pytorch_sparse_solver.utils.matrix_utils import create_ldc_pressure_csr  # noqa: E402


@dataclass
class StepReport:
    mass_residual: float
    info: int
    matvecs: int
    solve_seconds: float


@dataclass
class LidDrivenCavity:
    """Unit square, nx x nx cells, lid speed 1 at the top.  Arrays carry one ghost layer: shape (nx + 2, nx + 2),
    index [j, i] = (row of cells, column of cells); u lives on vertical faces, v on horizontal faces."""
    nx: int = 100
    Re: float = 400.0
    method: str = "gmres"
    device: str = "cpu"
    warm_start: bool = False
    solver_kwargs: dict = field(default_factory=lambda: dict(tol=1e-10, maxiter=1000))
    restart: int = 30

    def __post_init__(self):
        n1 = self.nx + 2
        self.h = 1.0 / self.nx
        self.nu = 1.0 / self.Re                                   # lid speed and cavity size are 1
        self.dt = min(0.25 * self.h ** 2 / self.nu, 4.0 * self.nu)   # diffusive / convective limits (ref :59-61)
        mk = lambda: torch.zeros((n1, n1), dtype=torch.float64, device=self.device)
        self.u, self.v, self.us, self.vs, self.p = mk(), mk(), mk(), mk(), mk()
        self.A = create_ldc_pressure_csr(self.nx, device=self.device)   # built once: one solver handle for all steps
        self.rhs_log = []
        self.solve_seconds = 0.0

    # -- boundary values through the ghost layer (no slip; moving lid at the top)
    def set_wall_values(self):
        u, v = self.u, self.v
        u[1:-1, 1] = 0.0
        u[1:-1, -1] = 0.0
        u[-1, 1:] = 2.0 - u[-2, 1:]
        u[0, 1:] = -u[1, 1:]
        v[1:, 0] = -v[1:, 1]
        v[1:, -1] = -v[1:, -2]
        v[1, 1:-1] = 0.0
        v[-1, 1:-1] = 0.0

    # -- explicit predictor: u* = u + dt (-div(u u) + nu lap u)
    def predictor(self):
        h, dt, nu, u, v = self.h, self.dt, self.nu, self.u, self.v
        avg = lambda a, b: 0.5 * (a + b)
        c = u[1:-1, 2:-1]                                          # interior u faces
        u_e, u_w = avg(u[1:-1, 3:], c), avg(c, u[1:-1, 1:-2])
        u_n, u_s = avg(u[2:, 2:-1], c), avg(c, u[:-2, 2:-1])
        v_n, v_s = avg(v[2:, 2:-1], v[2:, 1:-2]), avg(v[1:-1, 2:-1], v[1:-1, 1:-2])
        conv = -(u_e * u_e - u_w * u_w) / h - (u_n * v_n - u_s * v_s) / h
        lap = ((u[1:-1, 3:] - 2 * c + u[1:-1, 1:-2]) / h ** 2 + (u[2:, 2:-1] - 2 * c + u[:-2, 2:-1]) / h ** 2)
        self.us[1:-1, 2:-1] = c + dt * (conv + nu * lap)

        c = v[2:-1, 1:-1]                                          # interior v faces
        v_e, v_w = avg(v[2:-1, 2:], c), avg(c, v[2:-1, :-2])
        u_e, u_w = avg(u[2:-1, 2:], u[1:-2, 2:]), avg(u[2:-1, 1:-1], u[1:-2, 1:-1])
        v_n, v_s = avg(v[3:, 1:-1], c), avg(c, v[1:-2, 1:-1])
        conv = -(u_e * v_e - u_w * v_w) / h - (v_n * v_n - v_s * v_s) / h
        lap = ((v[2:-1, 2:] - 2 * c + v[2:-1, :-2]) / h ** 2 + (v[3:, 1:-1] - 2 * c + v[1:-2, 1:-1]) / h ** 2)
        self.vs[2:-1, 1:-1] = c + dt * (conv + nu * lap)

    def divergence(self, u, v):
        return (u[1:-1, 2:] - u[1:-1, 1:-1]) / self.h + (v[2:, 1:-1] - v[1:-1, 1:-1]) / self.h

    # -- the hot path: A p = div(u*) / dt on the CSR matrix
    def solve_pressure(self) -> StepReport:
        rhs = ((1.0 / self.dt) * self.divergence(self.us, self.vs)).reshape(-1).contiguous()
        self.rhs_log.append(rhs.clone())
        x0 = self.p[1:-1, 1:-1].reshape(-1).contiguous() if self.warm_start else None
        if rhs.is_cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        if self.method == "gmres":
            p, info = gmres(self.A, rhs, x0=x0, restart=self.restart, **self.solver_kwargs)
        else:
            p, info = bicgstab(self.A, rhs, x0=x0, **self.solver_kwargs)
        if rhs.is_cuda:
            torch.cuda.synchronize()
        dt_solve = time.perf_counter() - t0
        self.solve_seconds += dt_solve
        self.p.zero_()
        self.p[1:-1, 1:-1] = p.reshape(self.nx, self.nx)
        st = get_last_stats()
        return StepReport(0.0, int(info), int(getattr(st, "matvecs", 0)), dt_solve)

    def project(self):
        h, dt, p = self.h, self.dt, self.p
        self.u[1:-1, 2:-1] = self.us[1:-1, 2:-1] - dt * (p[1:-1, 2:-1] - p[1:-1, 1:-2]) / h
        self.v[2:-1, 1:-1] = self.vs[2:-1, 1:-1] - dt * (p[2:-1, 1:-1] - p[1:-2, 1:-1]) / h

    def step(self) -> StepReport:
        self.set_wall_values()
        self.predictor()
        rep = self.solve_pressure()
        self.project()
        rep.mass_residual = float(torch.linalg.norm(self.divergence(self.u, self.v)))
        return rep


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--nx", type=int, default=100)
    ap.add_argument("--Re", type=float, default=400.0)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--method", choices=("gmres", "bicgstab"), default="gmres")
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--warm-start", action="store_true")
    a = ap.parse_args()
    sim = LidDrivenCavity(nx=a.nx, Re=a.Re, method=a.method, device=a.device, warm_start=a.warm_start)
    t0 = time.perf_counter()
    mv = 0
    for k in range(a.steps):
        rep = sim.step()
        mv += rep.matvecs
        if k % max(1, a.steps // 10) == 0:
            print(f"step {k:5d}  mass residual {rep.mass_residual:.3e}  info {rep.info}  operator applications {rep.matvecs}")
    wall = time.perf_counter() - t0
    print(f"{a.steps} steps, nx={a.nx}, {a.method}{' (warm start)' if a.warm_start else ''} on {a.device}: "
          f"{wall:.2f} s total, {sim.solve_seconds:.2f} s in pressure solves, {mv} operator applications, "
          f"{sim.solve_seconds / a.steps * 1e3:.2f} ms per solve")


if __name__ == "__main__":
    main()
