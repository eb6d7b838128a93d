// Coarse-mesh incompressible solver (host, float64): the producer of the SR call's input.
//
// Restates the numba kernels and the SIMPLE-type time loop of the reference's lid-driven-cavity solver
//   PyCFD_ML_accelerated.py:110-328 (copy_new_to_old, apply_bc_configured, linear_interpolation, quick_scheme /
//   simple_upwind, diffusive_flux, update_flux, solve_momentum_*, solve_pressure, correct_velocity) and
//   :377-395, 433-505 (CFDSolver._initialize_fields, _implicit_solve, _convergence_check, solve)
// so that BASELINE config 1's coarse 10x10 input (Re = 400, which the reference checkout does not contain) can be produced
// here; and, with case_type = SRCFD_CASE_BFS, the backward-facing-step variant of the same loop that produces BASELINE
// config 3's input (bfs_ml_accelerated.py:233-468 kernels, :523-562 mixed wall / parabolic inlet on the left boundary,
// :626-673 _implicit_solve with under-relaxation of u, v and p, :675-707 _convergence_check).  A 10x10 problem is a few kB and a few thousand sweeps: this is host code on purpose; the hot path starts after it.
//
// Differences from the reference, on purpose: its point sweeps run inside numba `prange` loops that update Var in place
// (a benign data race: rows are relaxed in whatever order the threads run, the reference's own runs differ by ~2e-7);
// here the sweep is the serial one (i outer, j inner, in place).  Both converge to the same fixed point to the
// solver's own tolerance; tests compare with the reference's five stored coarse fields.  Index arithmetic that leaves the
// array in the reference (QUICK's i+2 / j-2 stencil at a boundary cell whose face flux is negative) wraps as numba's does;
// with wall boundaries those face fluxes are exactly zero and the branch is never taken.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/srcfd.h"
#include "abi_guard.h"

namespace srcfd {
void set_error(const std::string& m);
}

namespace {

struct Grid {
  int nx, ny, sx, sy;  // Var[k][i][j] at k*sx + i*sy + j, (nx+2) x (ny+2) per variable
  double dx, dy, volp;
  std::vector<double> Var, Old, Ff;
  double& v(int k, int i, int j) { return Var[(size_t)k * sx + (size_t)i * sy + j]; }
  double& f(int k, int i, int j) { return Ff[(size_t)k * sx + (size_t)i * sy + j]; }
  // numba semantics for the stencil reads that can leave the row / plane: negative indices wrap per axis, indices past the
  // end run on in flat memory (PyCFD_ML_accelerated.py:189-226 at i = 1, nx and j = 1, ny)
  double vw(int k, int i, int j) const {
    if (i < 0) i += nx + 2;
    if (j < 0) j += ny + 2;
    size_t idx = (size_t)k * sx + (size_t)i * sy + j;
    if (idx >= Var.size()) idx = Var.size() - 1;
    return Var[idx];
  }
};

void apply_bc(Grid& g, int k, const int* t, const double* val) {  // PyCFD_ML_accelerated.py:118-146
  for (int j = 1; j <= g.ny; ++j) {
    g.v(k, 0, j) = t[0] == 0 ? 2 * val[0] - g.v(k, 1, j) : g.v(k, 1, j);
    g.v(k, g.nx + 1, j) = t[1] == 0 ? 2 * val[1] - g.v(k, g.nx, j) : g.v(k, g.nx, j);
  }
  for (int i = 1; i <= g.nx; ++i) {
    g.v(k, i, g.ny + 1) = t[2] == 0 ? 2 * val[2] - g.v(k, i, g.ny) : g.v(k, i, g.ny);
    g.v(k, i, 0) = t[3] == 0 ? 2 * val[3] - g.v(k, i, 1) : g.v(k, i, 1);
  }
}

void linear_interpolation(Grid& g) {  // :148-155
  for (int i = 1; i <= g.nx; ++i)
    for (int j = 1; j <= g.ny; ++j) {
      g.f(0, i, j) = (g.v(0, i, j) + g.v(0, i + 1, j)) * g.dy * 0.5;
      g.f(1, i, j) = (g.v(1, i, j) + g.v(1, i, j + 1)) * g.dx * 0.5;
      g.f(2, i, j) = -(g.v(0, i, j) + g.v(0, i - 1, j)) * g.dy * 0.5;
      g.f(3, i, j) = -(g.v(1, i, j) + g.v(1, i, j - 1)) * g.dx * 0.5;
    }
}

void convective(Grid& g, bool quick, int k, int i, int j, double& Fc, double& ap_c) {  // :157-233
  const double fe = g.f(0, i, j), fn = g.f(1, i, j), fw = g.f(2, i, j), fs = g.f(3, i, j);
  const double c = g.v(k, i, j);
  double ue, uw, un, us, sum = 0.0;
  if (!quick) {
    if (fe >= 0) { ue = c; sum += fe; } else ue = g.v(k, i + 1, j);
    if (fw >= 0) { uw = c; sum += fw; } else uw = g.v(k, i - 1, j);
    if (fn >= 0) { un = c; sum += fn; } else un = g.v(k, i, j + 1);
    if (fs >= 0) { us = c; sum += fs; } else us = g.v(k, i, j - 1);
  } else {
    if (fe >= 0) { ue = 0.75 * c + 0.375 * g.v(k, i + 1, j) - 0.125 * g.v(k, i - 1, j); sum += 0.75 * fe; }
    else { ue = 0.75 * g.v(k, i + 1, j) + 0.375 * c - 0.125 * g.vw(k, i + 2, j); sum += 0.375 * fe; }
    if (fw >= 0) { uw = 0.75 * c + 0.375 * g.v(k, i - 1, j) - 0.125 * g.v(k, i + 1, j); sum += 0.75 * fw; }
    else { uw = 0.75 * g.v(k, i - 1, j) + 0.375 * c - 0.125 * g.vw(k, i - 2, j); sum += 0.375 * fw; }
    if (fn >= 0) { un = 0.75 * c + 0.375 * g.v(k, i, j + 1) - 0.125 * g.v(k, i, j - 1); sum += 0.75 * fn; }
    else { un = 0.75 * g.v(k, i, j + 1) + 0.375 * c - 0.125 * g.vw(k, i, j + 2); sum += 0.375 * fn; }
    if (fs >= 0) { us = 0.75 * c + 0.375 * g.v(k, i, j - 1) - 0.125 * g.v(k, i, j + 1); sum += 0.75 * fs; }
    else { us = 0.75 * g.v(k, i, j - 1) + 0.375 * c - 0.125 * g.vw(k, i, j - 2); sum += 0.375 * fs; }
  }
  Fc = ue * fe + uw * fw + un * fn + us * fs;
  ap_c = sum * g.volp;
}

inline void diffusive(Grid& g, int k, int i, int j, double& Fd, double& ap_d) {  // :235-240
  Fd = g.volp * ((g.v(k, i + 1, j) - 2.0 * g.v(k, i, j) + g.v(k, i - 1, j)) / (g.dx * g.dx) +
                 (g.v(k, i, j + 1) - 2.0 * g.v(k, i, j) + g.v(k, i, j - 1)) / (g.dy * g.dy));
  ap_d = -g.volp * (2.0 / (g.dx * g.dx) + 2.0 / (g.dy * g.dy));
}

void solve_momentum(Grid& g, bool quick, int k, double dt, double nu) {  // :251-297
  for (int it = 0; it < 1000; ++it) {
    double rms = 0.0;
    for (int i = 1; i <= g.nx; ++i)
      for (int j = 1; j <= g.ny; ++j) {
        double Fc, ap_c, Fd, ap_d;
        convective(g, quick, k, i, j, Fc, ap_c);
        diffusive(g, k, i, j, Fd, ap_d);
        const double R = -(g.volp / dt * (g.v(k, i, j) - g.Old[(size_t)k * g.sx + (size_t)i * g.sy + j]) + Fc + (-nu) * Fd);
        const double ap = g.volp / dt + ap_c + (-nu) * ap_d;
        g.v(k, i, j) = g.v(k, i, j) + R / ap;
        rms += R * R;
      }
    if (std::sqrt(rms / (g.nx * g.ny)) < 1e-6) break;
  }
}

void solve_pressure(Grid& g, double dt, double rho) {  // :299-321
  for (int it = 0; it < 1000; ++it) {
    double rms = 0.0;
    for (int i = 1; i <= g.nx; ++i)
      for (int j = 1; j <= g.ny; ++j) {
        double Fd, ap_d;
        diffusive(g, 2, i, j, Fd, ap_d);
        const double RHS = rho / dt * (g.f(0, i, j) + g.f(1, i, j) + g.f(2, i, j) + g.f(3, i, j));
        const double R = RHS - Fd;
        g.v(2, i, j) = g.v(2, i, j) + R / ap_d;
        rms += R * R;
      }
    if (std::sqrt(rms / (g.nx * g.ny)) < 1e-6) break;
  }
}

// CFDSolver._apply_bfs_inlet, bfs_ml_accelerated.py:523-562: below the step the left boundary is a no-slip wall, above it a
// parabolic u profile with v = 0.  Applying it for k = 0 also rewrites v's ghost cells of the open part, as the reference does.
void apply_bfs_inlet(Grid& g, int k, double step_h, double h, double Ub) {
  if (k > 1) return;
  for (int j = 1; j <= g.ny; ++j) {
    const double y = (j - 0.5) * g.dy;
    if (y < step_h) { g.v(k, 0, j) = -g.v(k, 1, j); continue; }
    if (k == 1) { g.v(1, 0, j) = -g.v(1, 1, j); continue; }
    double yp = y - step_h;
    if (yp < 0.0) yp = 0.0;
    if (yp > h) yp = h;
    const double u_in = 6.0 * Ub * (yp / h) * (1.0 - (yp / h));
    g.v(0, 0, j) = 2.0 * u_in - g.v(0, 1, j);
    g.v(1, 0, j) = -g.v(1, 1, j);
  }
}

void under_relax(Grid& g, int k, double alpha) {  // bfs_ml_accelerated.py:371-375
  for (int i = 1; i <= g.nx; ++i)
    for (int j = 1; j <= g.ny; ++j) {
      const double o = g.Old[(size_t)k * g.sx + (size_t)i * g.sy + j];
      g.v(k, i, j) = o + alpha * (g.v(k, i, j) - o);
    }
}

}  // namespace

extern "C" int srcfd_coarse_solve(const srcfd_coarse_problem* pb, double* var_out, int* iterations, double rms_out[3]) {
  return srcfd::abi_guard("srcfd_coarse_solve", [&]() -> int {
    using srcfd::set_error;
    if (!pb || !var_out) { set_error("srcfd_coarse_solve: bad arguments"); return SRCFD_EINVAL; }
    if (pb->nx < 3 || pb->ny < 3 || pb->nx > 4096 || pb->ny > 4096 || !(pb->lx > 0) || !(pb->ly > 0) || !(pb->reynolds > 0) || !(pb->rho > 0) ||
        !(pb->dt > 0) || pb->max_iterations < 0 || (pb->scheme != SRCFD_SCHEME_QUICK && pb->scheme != SRCFD_SCHEME_UPWIND) ||
        (pb->case_type != SRCFD_CASE_LDC && pb->case_type != SRCFD_CASE_BFS) || (pb->case_type == SRCFD_CASE_BFS && !(pb->channel_height > 0))) {
      set_error("srcfd_coarse_solve: bad problem description");
      return SRCFD_EINVAL;
    }
    Grid g;
    g.nx = pb->nx; g.ny = pb->ny; g.sy = g.ny + 2; g.sx = (g.nx + 2) * (g.ny + 2);
    g.dx = pb->lx / g.nx; g.dy = pb->ly / g.ny; g.volp = g.dx * g.dy;   // MeshParameters, :69-77
    g.Var.assign((size_t)3 * g.sx, 0.0); g.Old.assign((size_t)3 * g.sx, 0.0); g.Ff.assign((size_t)4 * g.sx, 0.0);
    const double nu = 1.0 / pb->reynolds;  // FluidProperties, :79-85
    const bool quick = pb->scheme == SRCFD_SCHEME_QUICK;
    const bool bfs = pb->case_type == SRCFD_CASE_BFS;
    auto bc = [&](int k) {   // _apply_bc_wrapper (bfs_ml_accelerated.py:564-569); the cavity solver has no inlet override
      apply_bc(g, k, pb->bc_type[k], pb->bc_value[k]);
      if (bfs) apply_bfs_inlet(g, k, pb->step_height, pb->channel_height, pb->bulk_velocity);
    };
    // _initialize_fields, :377-390
    for (int k = 0; k < 3; ++k) bc(k);
    g.Old = g.Var;
    linear_interpolation(g);
    int count = 0;
    bool converged = false;
    double rms[3] = {0, 0, 0};
    while (!converged && count < pb->max_iterations) {   // solve, :411-424
      ++count;
      // _implicit_solve, :433-470
      // (BFS: each solve is followed by under-relaxation against the previous iterate, bfs_ml_accelerated.py:642-659)
      for (int k = 0; k < 2; ++k) {
        solve_momentum(g, quick, k, pb->dt, nu);
        if (bfs) under_relax(g, k, pb->relax[k]);
        bc(k);
      }
      linear_interpolation(g);
      solve_pressure(g, pb->dt, pb->rho);
      if (bfs) under_relax(g, 2, pb->relax[2]);
      bc(2);
      double res[3] = {0, 0, 0};
      for (int i = 1; i <= g.nx; ++i)       // correct_velocity, :323-335
        for (int j = 1; j <= g.ny; ++j) {
          g.v(0, i, j) = g.v(0, i, j) - pb->dt / pb->rho * (g.v(2, i + 1, j) - g.v(2, i - 1, j)) / (2 * g.dx);
          g.v(1, i, j) = g.v(1, i, j) - pb->dt / pb->rho * (g.v(2, i, j + 1) - g.v(2, i, j - 1)) / (2 * g.dy);
          for (int k = 0; k < 3; ++k) {
            const double d = g.v(k, i, j) - g.Old[(size_t)k * g.sx + (size_t)i * g.sy + j];
            res[k] += d * d;
          }
        }
      bc(0);
      bc(1);
      for (int i = 1; i <= g.nx; ++i)       // update_flux, :242-249
        for (int j = 1; j <= g.ny; ++j) {
          g.f(0, i, j) += -pb->dt / pb->rho * (g.v(2, i + 1, j) - g.v(2, i, j)) * g.dy / g.dx;
          g.f(1, i, j) += -pb->dt / pb->rho * (g.v(2, i, j + 1) - g.v(2, i, j)) * g.dx / g.dy;
          g.f(2, i, j) += -pb->dt / pb->rho * (g.v(2, i - 1, j) - g.v(2, i, j)) * g.dy / g.dx;
          g.f(3, i, j) += -pb->dt / pb->rho * (g.v(2, i, j - 1) - g.v(2, i, j)) * g.dx / g.dy;
        }
      // _convergence_check, :472-505
      converged = true;
      for (int k = 0; k < 3; ++k) {
        rms[k] = std::sqrt(res[k] / (g.nx * g.ny)) / pb->dt;
        if (!std::isfinite(rms[k])) {
          set_error("srcfd_coarse_solve: NaN or Inf in the residuals (solver instability)");   // the reference raises ValueError here
          return SRCFD_EINVAL;
        }
        if (rms[k] > pb->tolerance[k]) converged = false;
      }
      if (!converged) g.Old = g.Var;
    }
    std::memcpy(var_out, g.Var.data(), g.Var.size() * sizeof(double));
    if (iterations) *iterations = count;
    if (rms_out) for (int k = 0; k < 3; ++k) rms_out[k] = rms[k];
    return SRCFD_OK;
  });
}
