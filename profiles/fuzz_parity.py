"""Randomised parity hunt (not part of the test suite): random small problems, storages, waves per
replica, replica counts, sweep counts, dense / CSR, forced forms -- GPU engine vs the CPU oracle,
bit for bit.  usage: fuzz_parity.py <seconds> [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
import spin_glass_anneal_rl_amd as sg

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
n_cases = n_fail = 0
t_note = time.time()
while time.time() < t_end:
    if time.time() - t_note > 60.0:  # a long run keeps talking (the GPU pool takes silence for a hang)
        print(f"... {n_cases} cases, {n_fail} failures so far", flush=True)
        t_note = time.time()
    kind = rng.choice(["dense", "dense", "csr"])
    sizes = [1, 2, 3, 5, 8, 17, 64, 65, 200, 255, 256, 257, 700, 1023, 1025, 2049, 3000, 5000]
    if os.environ.get("FUZZ_BIG"):
        sizes = [1500, 2049, 3000, 4097, 5000, 7000, 9000, 12000]
    n = int(rng.choice(sizes))
    if kind == "csr":
        n = max(n, 2)
    R = int(rng.choice([1, 2, 3, 7, 16, 33, 70]))  # (>= 32: energies from the matrix-core pass)
    ns = int(rng.choice([1, 2, 3, 5]))
    integer = rng.rand() < 0.6
    fixed = (not integer) and rng.rand() < 0.5   # real values on a 2^-10 grid: the fp64-exact row-sum forms
    dens = rng.choice([0.05, 0.3, 1.0]) if kind == "dense" else min(1.0, rng.choice([4, 30, 200, 700]) / n)
    vals = rng.randint(-2, 3, (n, n)) if integer else (np.rint(rng.randn(n, n) * 1024.0) / 1024.0 if fixed else rng.randn(n, n))
    J = np.triu(vals * (rng.rand(n, n) < dens), 1).astype(np.float32)
    J = J + J.T
    h = (rng.randint(-2, 3, n) if integer else rng.randn(n)).astype(np.float32)
    if integer and rng.rand() < 0.3:
        h = h + np.float32(0.5)   # half-integer fields: the accept table at twice the resolution (CSR)
    if rng.rand() < 0.2:
        h[:] = 0
    storage = "auto"
    if kind == "dense" and integer:
        storage = str(rng.choice(["auto", "f32", "i8"]))
        if np.abs(J).max() <= 1 and rng.rand() < 0.5:
            storage = "t2"
    waves = int(rng.choice([0, 0, 1, 2, 3, 4, 8, 16]))
    env = {}
    if rng.rand() < 0.3:
        env["SGA_NO_LOOK_AHEAD"] = "1"
    if kind == "csr" and rng.rand() < 0.5:
        env["SGA_FORCE_CSR_BIG"] = "1"
    # cached-local-field sweep (round 3): "auto" takes it wherever the problem allows, any waves per replica
    # (round 4: AUTO starts on the row kernels until it has seen the acceptance, so "on" is in the mix -- a problem that
    #  does not qualify answers "cached local fields ..." and the case is skipped; the several-accepts-per-round form too)
    cache = str(rng.choice(["off", "auto", "on", "on"]))
    if cache != "off" and rng.rand() < 0.6:
        env["SGA_CLF_WAVES"] = str(rng.choice([1, 2, 3, 4, 8]))
    if cache != "off" and rng.rand() < 0.5:
        env["SGA_CLF_BATCHED"] = "1"
    if kind == "csr" and rng.rand() < 0.5:  # (unset: several updates per step wherever the form applies)
        env["SGA_CSR_PAIR_AHEAD"] = str(rng.choice([0, 1, 2, 4, 8]))
    seed = int(rng.randint(1, 1 << 30))
    temps = np.geomspace(3.0 * max(1.0, np.sqrt(n)), 0.2, R) if R > 1 else np.asarray([1.5])
    mode = str(rng.choice(["plain", "plain", "rules", "pt", "batch", "tsp", "wolff"]))
    rule = int(rng.choice([0, 1, 2]))
    site_mode = int(rng.choice([0, 1]))
    arith = int(rng.choice([0, 1])) if rule == 0 else 0
    n_lad = int(rng.choice([1, 2, 3]))
    if mode == "pt":
        R = n_lad * int(rng.choice([2, 3, 4, 7]))
        temps = np.tile(np.geomspace(3.0 * max(1.0, np.sqrt(n)), 0.2, R // n_lad), n_lad)
    slot_temps = temps.copy()
    desc = f"{mode} rule={rule} site={site_mode} arith={arith} lad={n_lad} {kind} n={n} R={R} ns={ns} int={integer} fixed={fixed} dens={dens:.3g} storage={storage} waves={waves} cache={cache} env={env} seed={seed}"
    for k, v in env.items():
        os.environ[k] = v
    if mode == "tsp":   # TSP-structured couplings never stored vs the oracle on the CSR its restatement writes
        try:
            from spin_glass_anneal_rl_amd import encoders as enc
            nc = int(rng.choice([3, 4, 5, 9, 17, 33, 70]))
            xy = rng.rand(nc, 2) * 100.0
            d = np.hypot(xy[:, None, 0] - xy[None, :, 0], xy[:, None, 1] - xy[None, :, 1])
            if rng.rand() < 0.3:
                d = d + rng.rand(nc, nc) * 5.0            # asymmetric
            if integer:
                d = np.rint(d / 4.0) * 4.0
            d32, A, B, hh, _ = enc.tsp_structure(d, float(rng.choice([100.0, 200.0, 52.0])), 120.0,
                                                 auto_scale=not integer)
            Rt = int(rng.choice([1, 2, 5]))
            tt = np.geomspace(150.0, 3.0, Rt) if Rt > 1 else np.asarray([20.0])
            rp, ci, vv = oracle.tsp_to_csr(d32, A, B)
            prob = oracle.Problem(csr=(rp.astype(np.int32), ci, vv), h=hh)
            st = oracle.init_spins(nc * nc, Rt, seed)
            ref = oracle.sweeps(prob, st, tt, ns, seed=seed, trace=True)
            with sg.AnnealEngine(0) as e:
                e.set_tsp(d32, A, B, hh)
                e.init_replicas(Rt, seed=seed)
                e.set_temperatures(tt)
                out = e.sweep(ns, trace=True)
                ok = (np.array_equal(out["accept_trace"], ref["accept_trace"]) and np.array_equal(out["dE_trace"], ref["dE_trace"])
                      and np.array_equal(e.spins(), st))
                # the production sweep (several updates per step, one per wave: sweep_tsp_par_kernel) from the same start
                par = str(rng.choice([0, 2, 4, 8]))
                os.environ["SGA_TSP_PARALLEL"] = par
                e.init_replicas(Rt, seed=seed)
                e.set_temperatures(tt)
                e.sweep(ns)
                os.environ.pop("SGA_TSP_PARALLEL", None)
                ok = ok and np.array_equal(e.spins(), st) and np.array_equal(e.stats()[0], ref["n_accepted"])
                if not ok:
                    n_fail += 1
                    print("MISMATCH tsp", desc, f"cities={nc} par={par}", "|", e.describe(), flush=True)
        except Exception as ex:
            n_fail += 1
            print("ERROR tsp", desc, "|", str(ex)[:200], flush=True)
        finally:
            for k in env:
                os.environ.pop(k, None)
        n_cases += 1
        continue
    if mode == "wolff" and n >= 2:   # cluster moves, Philox uniforms, dense or CSR
        try:
            nw_ = min(n, 400)
            Jw, hw = np.ascontiguousarray(J[:nw_, :nw_]), np.ascontiguousarray(h[:nw_])
            if kind == "csr":
                rowptr = np.concatenate([[0], np.cumsum((Jw != 0).sum(1))]).astype(np.int32)
                col = np.concatenate([np.nonzero(Jw[i])[0] for i in range(nw_)] + [np.zeros(0, int)]).astype(np.int32)
                val = np.concatenate([Jw[i][Jw[i] != 0] for i in range(nw_)] + [np.zeros(0)]).astype(np.float32)
                prob = oracle.Problem(csr=(rowptr, col, val), h=hw)
            else:
                prob = oracle.Problem(J=Jw, h=hw)
            Rw = min(R, 4)
            tw = np.geomspace(6.0, 0.8, Rw) if Rw > 1 else np.asarray([2.0])
            sw = oracle.init_spins(nw_, Rw, seed)
            ref = oracle.sweeps(prob, sw, tw, 1, rule=oracle.RULE_WOLFF, seed=seed, recompute_energy=True)
            with sg.AnnealEngine(0) as e:
                if kind == "csr":
                    e.set_csr(rowptr, col, val, hw)
                else:
                    e.set_dense(Jw, hw, storage="auto" if storage == "t2" else storage)
                e.set_update_rule(3)
                e.init_replicas(Rw, seed=seed)
                e.set_temperatures(tw)
                e.sweep(1)
                ok = np.array_equal(e.spins(), sw) and np.array_equal(e.stats()[0], ref["n_accepted"])
                if not ok:
                    n_fail += 1
                    print("MISMATCH wolff", desc, "|", e.describe(), flush=True)
        except Exception as ex:
            if not any(t in str(ex) for t in ("not integer", "ternary")):
                n_fail += 1
                print("ERROR wolff", desc, "|", str(ex)[:200], flush=True)
        finally:
            for k in env:
                os.environ.pop(k, None)
        n_cases += 1
        continue
    if mode in ("tsp", "wolff"):
        mode = "plain"
    if mode == "batch" and kind == "dense" and n >= 2:
        try:
            M, kk = int(rng.choice([2, 3, 5])), int(rng.choice([1, 2, 4]))
            Js = np.stack([np.triu(rng.randint(-1, 2, (n, n)) if integer else rng.randn(n, n), 1) for _ in range(M)]).astype(np.float32)
            Js = Js + Js.transpose(0, 2, 1)
            hs = (rng.randint(-1, 2, (M, n)) if integer else rng.randn(M, n)).astype(np.float32)
            Rb = M * kk
            tb = np.tile(np.geomspace(4.0, 0.3, kk) if kk > 1 else np.asarray([1.0]), M)
            with sg.AnnealEngine(0) as e:
                if waves:
                    e.set_tuning(waves_per_replica=waves)
                e.set_dense_batch(Js, hs, storage="auto" if storage == "t2" else storage)
                e.init_replicas(Rb, seed=seed)
                e.set_temperatures(tb)
                out = e.sweep(ns, energy_trace=True)
                spins = e.spins()
                ok = True
                for m in range(M):
                    sl = slice(m * kk, (m + 1) * kk)
                    sm = oracle.init_spins(n, kk, seed, replica0=m * kk)
                    ref = oracle.sweeps(oracle.Problem(J=Js[m], h=hs[m]), sm, tb[sl], ns, seed=seed, replica0=m * kk)
                    ok = ok and np.array_equal(out["energy_trace"][:, sl], ref["energy_trace"]) and np.array_equal(spins[sl], sm)
                if not ok:
                    n_fail += 1
                    print("MISMATCH batch", desc, "|", e.describe(), flush=True)
        except Exception as ex:
            if not any(t in str(ex) for t in ("not integer", "ternary", "waves", "tuning", "cached local fields")):
                n_fail += 1
                print("ERROR batch", desc, "|", str(ex)[:200], flush=True)
        finally:
            for k in env:
                os.environ.pop(k, None)
        n_cases += 1
        continue
    if mode == "batch":
        mode = "plain"
    out = ref = None
    try:
        if kind == "csr":
            rowptr = np.concatenate([[0], np.cumsum((J != 0).sum(1))]).astype(np.int32)
            col = np.concatenate([np.nonzero(J[i])[0] for i in range(n)] + [np.zeros(0, int)]).astype(np.int32)
            val = np.concatenate([J[i][J[i] != 0] for i in range(n)] + [np.zeros(0)]).astype(np.float32)
            prob = oracle.Problem(csr=(rowptr, col, val), h=h)
        else:
            prob = oracle.Problem(J=J, h=h)
        s = oracle.init_spins(n, R, seed)
        if mode == "plain":
            ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=4)
        elif mode == "rules":
            u_seq = rng.rand(R, ns * n).astype(np.float32) if site_mode == 1 else None
            ref = oracle.sweeps(prob, s, temps, ns, site_mode=site_mode, arith=arith, rule=rule, seed=seed,
                                replay_u=u_seq, trace=True, n_threads=4)
        with sg.AnnealEngine(0) as e:
            if waves:
                try:
                    e.set_tuning(waves_per_replica=min(waves, 8 if kind == "csr" else 16))
                except Exception:
                    pass
            e.set_field_cache(cache)
            if kind == "csr":
                e.set_csr(rowptr, col, val, h)
            else:
                e.set_dense(J, h, storage=storage)
            e.init_replicas(R, seed=seed)
            e.set_temperatures(temps)
            tuned = bool(rng.rand() < 0.3 and kind == "dense")
            if tuned:
                e.autotune()
            geom = e.describe()  # the geometry the case really ran with (autotune picks by timing)
            if mode == "plain":
                out = e.sweep(ns, energy_trace=True)
                ok = (np.array_equal(out["energy_trace"], ref["energy_trace"]) and np.array_equal(e.spins(), s)
                      and np.array_equal(e.stats()[0], ref["n_accepted"]))
                for r in range(R):
                    be, bs, _ = e.best(r)
                    ok = ok and be == ref["best_energy"][r] and np.array_equal(bs, ref["best_spins"][r])
            elif mode == "rules":  # other rule / sequential sites / fp32 operator arithmetic
                e.set_update_rule(rule)
                out = e.sweep(ns, site_mode=site_mode, arith=arith, replay_u=u_seq, energy_trace=True, trace=True)
                ok = (np.array_equal(out["accept_trace"], ref["accept_trace"])
                      and np.array_equal(out["dE_trace"], ref["dE_trace"]) and np.array_equal(e.spins(), s))
            else:  # tempering: sweeps and exchange rounds on 1-3 ladders
                e.set_ladder(slot_temps, n_lad)
                slot = np.arange(R, dtype=np.int32)
                rep_T = slot_temps.copy()
                s2 = oracle.init_spins(n, R, seed)
                energy = None
                ok, done = True, 0
                for rnd in range(3):
                    res = oracle.sweeps(prob, s2, rep_T, ns, seed=seed, sweep0=done, energy=energy, n_threads=4)
                    energy, done = res["energy"], done + ns
                    e.sweep(ns)
                    L = R // n_lad
                    acc = 0
                    for lad in range(n_lad):
                        sub = slot[lad * L:(lad + 1) * L].copy()
                        acc += oracle.pt_exchange_round(slot_temps[lad * L:(lad + 1) * L], energy, sub,
                                                        seed=seed, round_=rnd, ladder=lad)
                        slot[lad * L:(lad + 1) * L] = sub
                    rep_T[slot] = slot_temps
                    got = e.exchange()
                    ok = ok and got == acc and np.array_equal(e.slot_map(), slot) and np.array_equal(e.energies(), energy)
                ok = ok and np.array_equal(e.spins(), s2) and np.array_equal(e.temperatures(), rep_T)
            if not ok:
                n_fail += 1
                # self-describing record: what ran (geometry after autotune included) and the inputs
                os.makedirs("gpurun_out", exist_ok=True)
                dump = f"gpurun_out/fuzz_fail_{n_fail}.npz"
                np.savez(dump, J=J, h=h, temps=temps, desc=np.asarray(desc), geometry=np.asarray(geom),
                         autotuned=np.asarray(tuned), spins_gpu=e.spins(), spins_ref=s,
                         **{k: v for k, v in (out or {}).items() if v is not None},
                         **{"ref_" + k: v for k, v in (ref or {}).items() if isinstance(v, np.ndarray)})
                print("MISMATCH", desc, f"autotuned={tuned} |", geom, "| inputs and both results in", dump, flush=True)
    except Exception as ex:
        msg = str(ex)
        if "not integer" in msg or "ternary" in msg or "waves" in msg or "tuning" in msg or (cache == "on" and "cached local fields" in msg):
            pass  # an impossible request (e.g. t2 for non-ternary, the field cache for real couplings), not a parity failure
        else:
            n_fail += 1
            print("ERROR", desc, "|", msg[:200], flush=True)
    finally:
        for k in env:
            os.environ.pop(k, None)
    n_cases += 1
print(f"{n_cases} cases, {n_fail} failures")
