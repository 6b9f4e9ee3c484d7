"""Independent numpy restatements used to pin the oracle (no shared code with it)."""
import numpy as np


def rot_axis(a, th):
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]], dtype=float)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def fk_homogeneous(chain, q):
    """4x4 homogeneous chain product; returns T_ee, joint origins, joint axes (world)."""
    T = np.eye(4)
    origins, axes = [], []
    for i in range(6):
        M = np.eye(4)
        M[:3, :3] = chain.place[i, :9].reshape(3, 3)
        M[:3, 3] = chain.place[i, 9:]
        T = T @ M
        origins.append(T[:3, 3].copy())
        axes.append(T[:3, :3] @ chain.axis[i])
        Rj = np.eye(4)
        Rj[:3, :3] = rot_axis(chain.axis[i], q[i])
        T = T @ Rj
    M = np.eye(4)
    M[:3, :3] = chain.place[6, :9].reshape(3, 3)
    M[:3, 3] = chain.place[6, 9:]
    return T @ M, origins, axes


def spatial_jacobian_fd(chain, q, eps=1e-6):
    """World-frame spatial Jacobian from d/dq (T) T^-1 (definition of
    pin.ReferenceFrame.WORLD, prediction_model.py:163-164), by central differences."""
    T, _, _ = fk_homogeneous(chain, q)
    Ti = np.linalg.inv(T)
    J = np.zeros((6, 6))
    for i in range(6):
        d = np.zeros(6)
        d[i] = eps
        Tp, _, _ = fk_homogeneous(chain, q + d)
        Tm, _, _ = fk_homogeneous(chain, q - d)
        V = ((Tp - Tm) / (2 * eps)) @ Ti
        J[:3, i] = V[:3, 3]
        J[3:, i] = [V[2, 1], V[0, 2], V[1, 0]]
    return J


def lti_matrices(wcv, Ts):
    """prediction_model.py:87-115 restated with numpy."""
    wcv = np.asarray(wcv, dtype=float)
    a22 = np.exp(-wcv * Ts)
    a12 = (1 - a22) / wcv
    A = np.eye(12)
    A[:6, 6:] = np.diag(a12)
    A[6:, 6:] = np.diag(a22)
    B = np.zeros((12, 6))
    B[:6] = np.diag(Ts - a12)
    B[6:] = np.diag(1 - a22)
    return A, B


def task_g_numpy(chain, coeffs, q, qd, t_ee=(0, 0, 0.1)):
    """g1..g5 of trajectory_optimizer.py:120-124 from the 4x4 FK and the FD Jacobian-free
    analytic spatial Jacobian columns [o x z; z]."""
    T, origins, axes = fk_homogeneous(chain, q)
    R, p = T[:3, :3], T[:3, 3]
    tw = R @ np.asarray(t_ee, dtype=float)
    pt = p + tw
    a, b, c, d, e, f = coeffs
    X, Y = pt[0], pt[1]
    S = a * X * X + b * Y * Y + c * X * Y + d * X + e * Y + f
    m = np.array([2 * a * X + c * Y + d, 2 * b * Y + c * X + e, -1.0])
    n = m / np.linalg.norm(m)
    vl = sum(np.cross(o, z) * v for o, z, v in zip(origins, axes, qd))
    om = sum(z * v for z, v in zip(axes, qd))
    vt = R.T @ (vl + np.cross(om, tw))
    return np.array([S - pt[2], n @ R[:, 2], R[0, 1], pt[0], vt[1]])


def qp_kkt_residuals(H, g, b, A, B, lb, ub, dx0, w, pi, lam, t):
    """KKT residuals of the OCP-QP at (w, pi, lam): an optimality certificate that does not
    depend on how the point was computed.  Returns dict of inf-norms."""
    N = H.shape[0] - 1
    stat = prim = comp = dual = feas = 0.0
    for k in range(N + 1):
        r = H[k] @ w[k] + g[k]
        if k < N:
            r[:6] += B.T @ pi[k]
            r[6:] += A.T @ pi[k]
        if k >= 1:
            r[6:] -= pi[k - 1]
        for j in range(12):
            has = (k < N) if j < 6 else True
            if has and lb[k, j] > -1e29:
                r[j] -= lam[k, j]
                feas = max(feas, lb[k, j] - w[k, j])
                comp = max(comp, abs(lam[k, j] * (w[k, j] - lb[k, j])))
                dual = max(dual, -lam[k, j])
            if has and ub[k, j] < 1e29:
                r[j] += lam[k, 12 + j]
                feas = max(feas, w[k, j] - ub[k, j])
                comp = max(comp, abs(lam[k, 12 + j] * (ub[k, j] - w[k, j])))
                dual = max(dual, -lam[k, 12 + j])
        if k == 0:
            r[6:] = 0
        if k == N:
            r[:6] = 0
        stat = max(stat, np.abs(r).max())
        if k < N:
            prim = max(prim, np.abs(A @ w[k, 6:] + B @ w[k, :6] + b[k] - w[k + 1, 6:]).max())
    prim = max(prim, np.abs(w[0, 6:] - dx0).max())
    return dict(stat=stat, prim=prim, comp=comp, dual=dual, feas=feas)


def random_ocp_qp(rng, N, wcv=200.0, Ts=0.01, scale_g=1.0, tight=False):
    """Random strictly convex OCP-QP with this problem's block structure."""
    A, B = lti_matrices([wcv] * 6, Ts)
    H = np.zeros((N + 1, 18, 18))
    g = rng.normal(size=(N + 1, 18)) * scale_g
    b = rng.normal(size=(N, 12)) * 1e-3
    lb = np.full((N + 1, 12), -1e30)
    ub = np.full((N + 1, 12), 1e30)
    for k in range(N):
        J = rng.normal(size=(5, 18))
        J[:, :6] = 0
        H[k] = Ts * 50 * J.T @ J
        H[k][:6, :6] += np.diag(rng.uniform(0.5, 2.0, 6))
        H[k][12:, 12:] += np.diag(rng.uniform(0.5, 2.0, 6))
        lim = 0.05 if tight else 2.0
        lb[k, :6] = -lim * rng.uniform(0.5, 1.5, 6)
        ub[k, :6] = lim * rng.uniform(0.5, 1.5, 6)
        if k >= 1:
            lq = 0.01 if tight else 6.0
            lb[k, 6:] = -lq
            ub[k, 6:] = lq
    g[N] = 0
    dx0 = rng.normal(size=12) * (1e-3 if tight else 1e-2)
    return H, g, b, A, B, lb, ub, dx0


def oracle_runner(cfgs, chain):
    """A SimulationManager runner backed by the CPU oracle (tests only): lets the host layer
    (queueing, bucketing, sharding, gather, analysis) be exercised without a GPU."""
    from oracle import orc

    rb = orc.make_robot(chain, cfgs[0]["t_ee"]) if cfgs else None
    from robotic_mpc_amd import analysis

    recs = [orc.run(rb, orc.make_params(c)) for c in cfgs]
    for r, c in zip(recs, cfgs):   # the engine logs the task errors with every column; here they come from the numpy analysis
        e = analysis.compute_errors(r["ee_pose"], r["ee_vel"], c["coeffs"], c["t_ee"], c["px_ref"], c["vy_ref"])
        r["errors"] = np.stack([e[k] for k in ("e1", "e2", "e3", "e4", "e5", "p_task_z", "p_ee_y")])
        r["plant_time"] = np.zeros_like(r["solver_time"])
    keys = ("z", "u", "ee_pose", "ee_rpy", "ee_vel", "status", "sqp_iter", "qp_iter", "residuals", "cost", "solver_time",
            "errors", "plant_time")
    return {k: np.stack([r[k] for r in recs]) for k in keys}


def reference_errors_loop(ee_pose, ee_vel, coeffs, t_ee, px_ref, vy_ref):
    """simulator.py:280-344 restated literally (per-step loop, numpy ops as written there)."""
    p_ee_all = ee_pose[:3, :]
    R_flat_all = ee_pose[3:12, :]
    n_steps = p_ee_all.shape[1]
    e = {k: np.zeros(n_steps) for k in ("e1", "e2", "e3", "e4", "e5", "p_task_z")}
    a, b, c, d, ee, f = coeffs
    R_ee_t = np.eye(3)
    t_ee = np.asarray(t_ee).reshape(3,)
    for i in range(n_steps):
        p_ee = p_ee_all[:, i]
        R_w_ee = R_flat_all[:, i].reshape(3, 3)
        vee = ee_vel[:, i]
        v_ee_w, w_ee_w = vee[:3], vee[3:6]
        translation_w = R_w_ee @ t_ee
        R_w_t = R_w_ee @ R_ee_t
        p_t = p_ee + R_w_ee @ t_ee
        v_task = R_w_t @ (v_ee_w + w_ee_w.T @ translation_w)
        R_task_y, R_task_z = R_w_t[:, 1], R_w_t[:, 2]
        x, y, z = p_t
        nx, ny, nz = 2 * a * x + c * y + d, 2 * b * y + c * x + ee, -1.0
        nn = np.sqrt(nx ** 2 + ny ** 2 + nz ** 2)
        n = np.array([nx, ny, nz]) / nn
        z_surf = a * x ** 2 + b * y ** 2 + c * x * y + d * x + ee * y + f
        e["e1"][i] = z_surf - z
        e["e2"][i] = 1.0 - float(n @ R_task_z)
        e["e3"][i] = float(R_task_y[0])
        e["e4"][i] = px_ref - x
        e["e5"][i] = vy_ref - v_task[1]
        e["p_task_z"][i] = z
    e["p_ee_y"] = p_ee_all[1, :]
    return e


def random_parameter_cfgs(n, seed, **kw):
    """Seeded random draws over every per-simulation parameter of the parameter record: weights, references, surface, bandwidths,
    start state, integrator, bounds from wide to tight (some active from the first step), and the sampling time dt (slot [0]:
    5e-4 .. 0.02, i.e. wcv * dt from 0.03 to 2.4 -- the reference's own regime is dt = 5e-4, main.py:14).  `simulation_time` in
    `kw` is read as a number of steps at dt = 0.01: every simulation of the returned list runs that many steps (Nsim is a
    property of the launch, dt of the simulation)."""
    from robotic_mpc_amd import config

    rng = np.random.default_rng(seed)
    steps = int(round(float(kw.pop("simulation_time", config.BASE_PARAMS["simulation_time"])) / 0.01))
    out = []
    for i in range(n):
        tight = rng.uniform(0.6, 3.2)
        p = dict(
            q_0=config.BASE_PARAMS["q_0"] + rng.uniform(-0.15, 0.15, 6), qdot_0=rng.uniform(-0.5, 0.5, 6),
            wcv=rng.uniform(60.0, 250.0, 6), w_u=float(10 ** rng.uniform(-3, -1.5)), w_qddot=float(10 ** rng.uniform(-2.3, -1)),
            px_ref=float(rng.uniform(0.3, 0.55)), vy_ref=float(rng.uniform(-0.05, 0.08)),
            surface_coeffs={k: float(v) for k, v in zip("abcdef", rng.normal([-0.1, 0.1, -0.01, 0.01, 0.01, 0.0], 0.03))},
            qdot_min=np.full(6, -tight), qdot_max=np.full(6, tight),
            integration_method=["RK4", "RK4", "Euler", "RK2", "RK3"][int(rng.integers(5))])
        dt = [5e-4, 2e-3, 0.01, 0.01, 0.02][int(rng.integers(5))]
        # the plant integrators are stable for wcv * dt < 2 (Euler) .. 2.785 (RK4): keep wcv * dt <= 1.9 at the coarse steps
        p["wcv"] = np.minimum(p["wcv"], (1.9 if p["integration_method"] == "Euler" else 2.4) / dt)
        out.append(config.resolve_config(config.base_params(dt=dt, simulation_time=(steps + 0.5) * dt, **p, **kw)))
        assert out[-1]["Nsim"] == steps
    return out
