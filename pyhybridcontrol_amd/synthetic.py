"""Seeded synthetic MLD-MPC workloads (BASELINE.json configs 1-5).

An *agent* is a cluster of ``n_h`` domestic electric water heaters (DEWH) plus one grid tie, composed
into a single MLD system.  Per-tank rows follow the reference's DEWH model
(examples/residential_mg_with_pv_and_dewhs/modelling/micro_grid_models.py:27-100, const_heat=True,
evaluated with parameters.py:10-27 and per-tank jitter); the tie rows follow its grid model
(micro_grid_models.py:137-172) with ``y = sum_i P_h_Nom_i u_i + omega_load``.

Per step: nx=n_h, nu=n_h (all binary), ndelta=1, nz=1, nmu=2 n_h, nomega=n_h+1, ny=1, nc=2 n_h+6.
Only numpy is used here: this module only *creates inputs*; it computes nothing on the hot path.
"""
import datetime as _dt

import numpy as np

# name -> (n_h, N_p, batch) ; N_tilde = N_p + 1   
CONFIGS = {
    "cfg1": dict(n_h=1, N_p=4, batch=1, tie=False, seed=1),
    "cfg2": dict(n_h=3, N_p=24, batch=256, tie=True, seed=2),
    "cfg3": dict(n_h=7, N_p=24, batch=1024, tie=True, seed=3),
    "cfg4": dict(n_h=7, N_p=24, batch=4096, tie=True, seed=4, n_agents=64),
    "cfg5": dict(n_h=15, N_p=48, batch=8192, tie=True, seed=5),
}

TS = 900.0  # control period, s  (parameters.py:8)
C_W = 4.1816e3  # J/kg/K
T_W = 15.0
T_INF = 25.0
T_H_NOM = 45.0
A_H = 2.35
EPS = float(np.finfo(float).eps)  # parameters.py:51

# time-of-use import prices, c/kWh (micro_grid_control_simulation.py:86-87)
TARIFF = dict(low_off_peak=48.40, low_stnd=76.28, low_peak=110.84,
              high_off_peak=55.90, high_stnd=102.95, high_peak=339.77)


def tou_price(t):
    """Import price (c/kWh) at datetime ``t`` -- the time-of-day rules of tariff_generator.py:62-111."""
    high = (6, 1) <= (t.month, t.day) <= (8, 31)
    pre = "high" if high else "low"
    wd, hour = t.weekday(), t.hour
    if wd == 6:
        kind = "off_peak"
    elif wd == 5:
        if hour >= 20 or hour < 7 or 12 <= hour < 18:
            kind = "off_peak"
        else:
            kind = "stnd"
    elif high:
        if hour >= 22 or hour < 6:
            kind = "off_peak"
        elif 9 <= hour < 17 or 19 <= hour < 22:
            kind = "stnd"
        else:
            kind = "peak"
    else:
        if hour >= 22 or hour < 6:
            kind = "off_peak"
        elif 6 <= hour < 7 or 10 <= hour < 18 or 20 <= hour < 22:
            kind = "stnd"
        else:
            kind = "peak"
    return TARIFF[pre + "_" + kind]


def price_vector(N_tilde, t0=None, ts=TS):
    """q_z profile: price_k * ts / 3.6e8  (micro_grid_control_simulation.py:89-90)."""
    t0 = t0 or _dt.datetime(2018, 12, 10, 5, 0)
    p = np.array([tou_price(t0 + _dt.timedelta(seconds=ts * k)) for k in range(N_tilde)])
    return p / 3600.0 / 100.0 / 1000.0 * ts


def dewh_discrete(m_h, P_h_Nom, U_h, ts=TS):
    """Exact zero-order-hold discretisation of the const-heat DEWH (micro_grid_models.py:37-56)."""
    p1 = U_h * A_H
    p2 = m_h * C_W
    A_c = -p1 / p2
    A = np.exp(A_c * ts)
    em = (A - 1.0) / A_c
    B1 = em * P_h_Nom / p2
    B4 = em * C_W * (T_W - T_H_NOM) / p2
    b5 = em * p1 * T_INF / p2
    return A, B1, B4, b5


def make_agent(n_h, rng, tie=True):
    """System matrices (dict name -> 2-D float64 array) + dims of one synthetic agent."""
    m_h = rng.uniform(120.0, 250.0, n_h)
    P_h = rng.choice([2000.0, 3000.0, 4000.0], n_h)
    U_h = rng.uniform(0.7, 1.1, n_h)
    T_max = rng.choice([65.0, 80.0], n_h)
    T_min = np.full(n_h, 50.0)
    A = np.zeros((n_h, n_h))
    B1 = np.zeros((n_h, n_h))
    B4 = np.zeros((n_h, n_h + (1 if tie else 0)))
    b5 = np.zeros((n_h, 1))
    for i in range(n_h):
        a, b1, b4, b_5 = dewh_discrete(m_h[i], P_h[i], U_h[i])
        A[i, i], B1[i, i], B4[i, i], b5[i, 0] = a, b1, b4, b_5
    nmu = 2 * n_h
    nc = 2 * n_h + (6 if tie else 0)
    E = np.zeros((nc, n_h))
    Psi = np.zeros((nc, nmu))
    f5 = np.zeros((nc, 1))
    for i in range(n_h):
        E[2 * i, i], E[2 * i + 1, i] = 1.0, -1.0
        Psi[2 * i, 2 * i], Psi[2 * i + 1, 2 * i + 1] = -1.0, -1.0
        f5[2 * i, 0], f5[2 * i + 1, 0] = T_max[i], -T_min[i]
    mats = dict(A=A, B1=B1, B4=B4, b5=b5, E=E, Psi=Psi, f5=f5, F1=np.zeros((nc, n_h)))
    dims = dict(nx=n_h, nu=n_h, ndelta=0, nz=0, nmu=nmu, nomega=n_h, ny=0, nc=nc, nu_l=n_h, nmu_l=0)
    if tie:
        Pg_max = 1.0e4 * n_h
        Pg_min = -Pg_max
        r0 = 2 * n_h
        F2 = np.zeros((nc, 1))
        F3 = np.zeros((nc, 1))
        G = np.zeros((nc, 1))
        F2[r0:, 0] = [-Pg_min, -(Pg_max + EPS), -Pg_max, Pg_min, -Pg_min, Pg_max]
        F3[r0:, 0] = [0, 0, 1, -1, 1, -1]
        f5[r0:, 0] = [-Pg_min, -EPS, 0, 0, -Pg_min, Pg_max]
        G[r0:, 0] = [-1, 1, 0, 0, -1, 1]
        D1 = P_h.reshape(1, n_h).copy()
        D4 = np.zeros((1, n_h + 1))
        D4[0, n_h] = 1.0
        C = np.zeros((1, n_h))
        mats.update(F2=F2, F3=F3, G=G, D1=D1, D4=D4, C=C, F4=np.zeros((nc, n_h + 1)),
                    d5=np.zeros((1, 1)), D2=np.zeros((1, 1)), D3=np.zeros((1, 1)),
                    B2=np.zeros((n_h, 1)), B3=np.zeros((n_h, 1)))
        dims.update(ndelta=1, nz=1, nomega=n_h + 1, ny=1)
    else:
        mats.update(C=np.eye(n_h), F4=np.zeros((nc, n_h)))
        dims.update(ny=n_h)
    params = dict(m_h=m_h, P_h_Nom=P_h, U_h=U_h, T_h_max=T_max, T_h_min=T_min)
    return mats, dims, params


def make_cost(n_h, N_tilde, params, tie=True, quadratic=False, t0=None, soft_top=10.0, soft_bot=1.0):
    """Objective-atom weights in the reference's string-keyed form (objective_atoms.py:453-496).

    q_z = price_k ts/3.6e8 (time varying, length N_tilde*nz);  q_mu = (10,1) * sum_k price_k * P_h_Nom
    (micro_grid_control_simulation.py:194-198).  Without a tie the energy price is put on u directly.
    ``quadratic`` adds the MIQP variant's Q_x = 1e-3 I.
    """
    price = price_vector(N_tilde, t0=t0)
    atoms = {}
    P_h = params["P_h_Nom"]
    q_mu = np.empty(2 * n_h)
    q_mu[0::2] = soft_top * price.sum() * P_h
    q_mu[1::2] = soft_bot * price.sum() * P_h
    atoms["q_mu"] = q_mu
    if tie:
        atoms["q_z"] = price.reshape(-1, 1).copy()
    else:
        atoms["q_u"] = np.kron(price, P_h).reshape(-1, 1)
    if quadratic:
        atoms["Q_x"] = 1e-3 * np.eye(n_h)
    return atoms


def make_scenarios(n_h, N_tilde, batch, rng, tie=True):
    """x0 (batch, nx) and omega_tilde (batch, N_tilde*nomega), step-major like the reference."""
    x0 = rng.integers(55, 65, size=(batch, n_h)).astype(np.float64)
    # hot water draw: Bernoulli(0.15) x LogNormal, scaled to a 200 L/day mean, in kg/s
    steps_per_day = int(86400 / TS)
    on = rng.random((batch, N_tilde, n_h)) < 0.15
    ln = rng.lognormal(mean=0.0, sigma=0.6, size=(batch, N_tilde, n_h))
    mean_per_step = 200.0 / steps_per_day / TS  # kg/s averaged over a step
    draw = on * ln * (mean_per_step / (0.15 * np.exp(0.18)))
    nomega = n_h + (1 if tie else 0)
    om = np.zeros((batch, N_tilde, nomega))
    om[:, :, :n_h] = draw
    if tie:
        r = np.clip(rng.normal(1.0, 0.38, size=(batch, N_tilde)), 0.2, None)
        phase = rng.uniform(0, 2 * np.pi, size=(batch, 1))
        k = np.arange(N_tilde)[None, :]
        p = np.clip(np.sin(2 * np.pi * k / steps_per_day + phase), 0.0, None)
        om[:, :, n_h] = 1200.0 * n_h * r - 2000.0 * n_h * p
    return x0, om.reshape(batch, N_tilde * nomega)


def make_workload(name, batch=None, n_agents=None, quadratic=False):
    """Full workload for a BASELINE config: list of agents (mats, dims, atoms) + scenarios per agent."""
    cfg = dict(CONFIGS[name])
    n_h, N_p, tie, seed = cfg["n_h"], cfg["N_p"], cfg["tie"], cfg["seed"]
    N_tilde = N_p + 1
    n_agents = n_agents or cfg.get("n_agents", 1)
    batch = batch or cfg["batch"]
    agents = []
    for a in range(n_agents):
        rng = np.random.Generator(np.random.PCG64(seed * 1000 + a))
        mats, dims, params = make_agent(n_h, rng, tie=tie)
        atoms = make_cost(n_h, N_tilde, params, tie=tie, quadratic=quadratic)
        x0, om = make_scenarios(n_h, N_tilde, batch, rng, tie=tie)
        agents.append(dict(mats=mats, dims=dims, params=params, atoms=atoms, x0=x0, omega=om))
    return dict(name=name, n_h=n_h, N_p=N_p, N_tilde=N_tilde, agents=agents, batch=batch)
