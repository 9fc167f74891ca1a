"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer on the steady-state closed-loop fixture (VERDICT r3 item 1 / ADVICE r3 high):
the oracle restates the dual simplex (incl. the long-step ratio test, ORC_BFRT), the perturbation, the cut loop and the search the kernel runs,
so an index bug of the ALGORITHM (negative index, list overrun) shows here on the CPU.  Builds into /tmp (never into the tree).

    python scripts/cpu_sanitize.py [n_instances=256] [bfrt=1] [with_start=1] [n_miqp]

Re-executes itself with libasan preloaded (python itself is not instrumented)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SAN_DIR = "/tmp/mld_san"
SAN_LIB = os.path.join(SAN_DIR, "libmldoracle.so")


def build():
    os.makedirs(SAN_DIR, exist_ok=True)
    src = os.path.join(ROOT, "oracle", "mld_oracle.c")
    if not os.path.exists(SAN_LIB) or os.path.getmtime(SAN_LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                               "-fPIC", "-std=c11", "-fopenmp", "-shared", "-o", SAN_LIB, src, "-lm"])


def main():
    n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    bfrt = sys.argv[2] if len(sys.argv) > 2 else "1"
    with_start = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    if os.environ.get("MLD_SAN_CHILD") != "1":
        build()
        asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
        env = dict(os.environ, MLD_SAN_CHILD="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", ORC_BFRT=bfrt, OMP_NUM_THREADS="1")
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), str(n_inst), bfrt, str(with_start)] + sys.argv[4:5], env=env))
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    from oracle import orc, condense_np as cn, tighten_np
    orc._LIB = SAN_LIB          # the sanitized build instead of oracle/_build
    orc.build = lambda force=False: SAN_LIB
    z = np.load(os.path.join(ROOT, "tests", "golden", "closed_loop_cfg4_inputs.npz"))
    agents, N_p, N_t, _, _, _ = bench.make_shard(64, 1, 0)
    forms = {}
    done = 0
    for i in range(min(n_inst, z["x0"].shape[0])):
        a = int(z["model_idx"][i])
        if a not in forms:
            d = agents[a]["dims"]
            forms[a] = cn.standard_form(tighten_np.tighten(agents[a]["mats"], d, nu_l=d["nu_l"]), agents[a]["atoms"], N_p, N_t, nu_l=d["nu_l"])
        sf = forms[a]
        x0, om = z["x0"][i], z["omega"][i]
        h, q = cn.rhs(sf["evo"], x0, om), cn.lin_cost(sf["cost"], x0, om)
        kw = dict(gap_rel=1e-2, max_nodes=800, presolve=4, max_pivots=40000)      # (presolve bit 2: the per-instance presolve the kernel runs)
        r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], **kw)
        if with_start and r["x"] is not None and np.all(np.isfinite(r["x"])):
            # the MIP-start path (leaf evaluation of a given assignment) with the plan just found, moved on by one step
            nv = sf["G"].shape[1] // N_t
            xs = np.concatenate([r["x"][nv:], r["x"][-nv:]])
            orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], x_start=xs, **kw)
        done += 1
        if done % 16 == 0:
            print("sanitized oracle: %d instances clean" % done, flush=True)
    print("sanitized oracle: %d instances, no report (ORC_BFRT=%s)" % (done, os.environ.get("ORC_BFRT")))
    if len(sys.argv) > 4:
        # the quadratic-cost path on instances of the MIQP bench leg (cfg3 shape, Q_x = 1e-3 I): 2068 is the instance whose dive overflowed the search stack
        # before fixed binaries were excluded from branching, 1817 / 651 the ones whose primal simplex cycled
        from pyhybridcontrol_amd import synthetic as syn
        wl = syn.make_workload("cfg3", batch=4096, quadratic=True)
        ag = wl["agents"][0]; d = ag["dims"]
        sf = cn.standard_form(tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"]), ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
        idx = [2068, 1817, 651, 365, 1486, 2009] + list(range(int(sys.argv[4])))
        for k, i in enumerate(idx):
            q, h = cn.lin_cost(sf["cost"], ag["x0"][i], ag["omega"][i]), cn.rhs(sf["evo"], ag["x0"][i], ag["omega"][i])
            for pre in (4, 0):
                r = orc.solve_miqp(sf["cost"]["P"], q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], max_nodes=800, presolve=pre, gap_rel=1e-2, max_pivots=40000)
                assert r["status"] in ("optimal", "node_limit") and np.isfinite(r["obj"]), (i, pre, r["status"], r["obj"])
            if (k + 1) % 8 == 0:
                print("sanitized oracle (MIQP): %d instances clean" % (k + 1), flush=True)
        print("sanitized oracle (MIQP): %d instances, every one with an incumbent, no report" % len(idx))


if __name__ == "__main__":
    main()
