#!/usr/bin/env python3
"""bench.py — shuffle proofs/sec (prove + verify) at ell = 252 on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--ell 252]

A "step" = one pass of the hot path over one batch: CurdleproofsProof::new followed by
CurdleproofsProof::verify for B independent ell = 252 instances (BASELINE configs[1] instance shape,
B of them = the per-GPU share of configs[3]) whose public inputs are already resident in HBM.
Weak scaling: every rank / GPU processes its own B instances, no data-path collective.
Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before anything touches HIP: hardware queues for the engine contexts' streams (curdleproofs_amd/__init__.py)
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
README_I7_PROOFS_PER_S = 1.0 / (0.560 + 0.035)   # reference README.md:49, i7-8550U (context only)


def synth_scalars(seed, n):
    """n synthetic scalars in wire form (arkworks' in-memory Fr, 4 x u64 LE), seeded: 32 random bytes with the top two bits cleared,
    i.e. uniform below 2^254 < r.  The wire form is the Montgomery image of the scalar and the map is a bijection, so the scalars themselves are
    spread over all of Fr; numpy draws them 200x faster than `params.random_fr_wire`'s Python integers (12.7 M draws per 16384-proof batch: half
    a minute of every rank's set-up until round 5).  Benchmark inputs only — never verification factors."""
    import numpy as np
    a = np.frombuffer(np.random.default_rng(seed).bytes(32 * n), dtype=np.uint8).reshape(n, 32).copy()
    a[:, 31] &= 0x3F   # little-endian: the top byte keeps 6 bits
    return a.tobytes()


def synth_batch(ctx, ell, B, seed, helpers=()):
    """Synthetic CRS + B shuffle instances built through the product API only (GPU scale / msm).  `helpers`: further contexts whose host threads share
    the per-proof util::msm calls (M of every instance: 0.8 ms per call, one call at a time per context)."""
    import threading
    import numpy as np
    import curdleproofs_amd as cpx
    from curdleproofs_amd import params as pr
    t_start = time.perf_counter()
    n = ell + 4
    gen = pr.g1_generator_wire()
    crs = ctx.scale(gen * (ell + 7), synth_scalars(seed * 16 + 1, ell + 7))
    ctx.set_crs(ell, crs)
    vec_R = ctx.scale(gen * (B * ell), synth_scalars(seed * 16 + 2, B * ell))
    vec_S = ctx.scale(vec_R, synth_scalars(seed * 16 + 3, B * ell))
    k_wire = synth_scalars(seed * 16 + 4, B)
    mbl_all = synth_scalars(seed * 16 + 5, 4 * B)
    AFF, FR = cpx.AFF, cpx.FR
    k_rep = np.repeat(np.frombuffer(k_wire, dtype=np.uint8).reshape(B, FR), ell, axis=0).tobytes()
    kR_all = np.frombuffer(ctx.scale(vec_R, k_rep), dtype=np.uint8).reshape(B, ell, AFF)   # util.rs:94-95, all proofs in one launch
    kS_all = np.frombuffer(ctx.scale(vec_S, k_rep), dtype=np.uint8).reshape(B, ell, AFF)
    del k_rep
    t_points = time.perf_counter()
    rng = np.random.default_rng(seed)
    perm_all = np.stack([rng.permutation(ell) for _ in range(B)]).astype(np.uint32)      # [B][ell]
    rows = np.arange(B)[:, None]
    vec_T = kR_all[rows, perm_all].tobytes()                                               # T_i = k R_sigma(i)
    vec_U = kS_all[rows, perm_all].tobytes()
    del kR_all, kS_all
    idx_wire = np.frombuffer(b"".join(pr.fr_to_wire(i) for i in range(ell)), dtype=np.uint8).reshape(ell, FR)
    sigma_all = idx_wire[perm_all]                                                         # [B][ell][32]: util.rs:99-103
    crs_G = ctx.marshal(crs[:n * AFF])
    Ms = [None] * B
    workers = [ctx] + list(helpers)

    def commit(w):
        c = workers[w]
        for p in range(w, B, len(workers)):
            Ms[p] = c.msm(crs_G, sigma_all[p].tobytes() + mbl_all[p * 4 * FR:(p + 1) * 4 * FR])
    th = [threading.Thread(target=commit, args=(w,)) for w in range(1, len(workers))]
    for t in th:
        t.start()
    commit(0)
    for t in th:
        t.join()
    t_end = time.perf_counter()
    return dict(crs=crs, vec_R=vec_R, vec_S=vec_S, vec_T=vec_T, vec_U=vec_U, M=b"".join(Ms), perms=perm_all.reshape(-1).tolist(),
                k=k_wire, mbl=mbl_all, prand=synth_scalars(seed * 16 + 6, B * (3 * n + 9)), vrand=synth_scalars(seed * 16 + 7, B * 8),
                setup_s=dict(total=time.perf_counter() - t_start, crs_and_scalar_multiplications=t_points - t_start, per_proof_permute_and_commit=t_end - t_points,
                             commit_threads=len(workers)))


def _instance(data, ell, i):
    """instance i of the synthetic batch in the oracle's argument form"""
    import curdleproofs_amd as cpx
    n = ell + 4
    AFF, FR, JAC = cpx.AFF, cpx.FR, cpx.JAC
    cut = lambda key, unit: data[key][i * unit:(i + 1) * unit]
    return dict(ell=ell, vec_R=cut("vec_R", ell * AFF), vec_S=cut("vec_S", ell * AFF), vec_T=cut("vec_T", ell * AFF), vec_U=cut("vec_U", ell * AFF), M=cut("M", JAC),
                permutation=data["perms"][i * ell:(i + 1) * ell], k=cut("k", FR), vec_m_blinders=cut("mbl", 4 * FR), prover_rand=cut("prand", (3 * n + 9) * FR),
                verifier_rand=cut("vrand", 8 * FR))


def cpu_baseline(ell, data, gpu_proofs, budget_s=12.0, sample=8):
    """Times the CPU oracle ('port' of the reference's algorithm) on instance 0, single thread: the build with the Montgomery product of
    ark-ff's `asm` feature (mulx / adcx / adox, oracle/Makefile FASTFLAGS) where the host CPU has BMI2 + ADX — `value` — and the portable
    x86-64-v2 build beside it.  gpu_proofs (the timed region's proofs of context 0, or None): a sample of `sample` of them spread over the
    batch is compared with the oracle's bytes."""
    from tests.oracle_lib import Oracle, fast_build_runs_here, oracle_build_flags
    psz = 48 * (18 + 10 * ((ell + 4).bit_length() - 1)) + 7 * 32
    builds = [("portable", Oracle())]
    if fast_build_runs_here():
        builds.append(("mulx_adx", Oracle(fast=True)))
    inst = _instance(data, ell, 0)
    res = {}
    for name, orc in builds:
        ok, tp, tv = orc.time_prove_verify(inst, data["crs"], 1)
        reps = max(1, min(30, int(budget_s / len(builds) / max(tp + tv, 1e-3))))
        ok, tp, tv = orc.time_prove_verify(inst, data["crs"], reps)
        res[name] = dict(value=reps / (tp + tv), prove_s=tp / reps, verify_s=tv / reps, reps=reps, accepted=bool(ok), build_flags=oracle_build_flags(fast=orc.fast))
    best_name, orc = builds[-1]
    checked, parity = [], True
    if gpu_proofs is not None:
        gpu_proofs = bytes(gpu_proofs)
        nb = len(gpu_proofs) // psz
        checked = sorted(set(int(round(j * (nb - 1) / max(1, sample - 1))) for j in range(min(sample, nb))))
        for i in checked:
            it = _instance(data, ell, i)
            want = orc.prove(ell, data["crs"], it["vec_R"], it["vec_S"], it["vec_T"], it["vec_U"], it["M"], it["permutation"], it["k"], it["vec_m_blinders"], it["prover_rand"])
            parity &= bytes(gpu_proofs[i * psz:(i + 1) * psz]) == want
    b = res[best_name]
    return dict(value=b["value"], unit="proofs/s", cores=1, kind="port", build=best_name, build_flags=b["build_flags"],
                sample="%d x (CurdleproofsProof::new + verify) of instance 0 of the bench batch at ell=%d, oracle/ single thread "
                       "(prove %.3f s, verify %.3f s each)" % (b["reps"], ell, b["prove_s"], b["verify_s"]),
                prove_s=b["prove_s"], verify_s=b["verify_s"], accepted=all(r["accepted"] for r in res.values()), by_build=res,
                gpu_proof_bit_exact=bool(parity), gpu_proofs_compared=checked,
                note="value = the faster build (ark-ff's asm-feature product where the host has BMI2 + ADX); by_build.portable = x86-64-v2 with unsigned __int128 products "
                     "(what rounds 1-4 reported)")


def effective_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container can see 256
    logical CPUs and be throttled to a handful)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2
            q, per = f.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:      # cgroup v1
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def cpu_baseline_all_cores(ell, data, budget_s=10.0):
    """the CPU oracle on every host core at once — one independent CurdleproofsProof::new + verify per thread
    (the embarrassingly parallel form of BASELINE configs[3]; the oracle is a C library behind ctypes, so Python threads run it in
    parallel).  Returns proofs/s over all threads."""
    import threading
    from tests.oracle_lib import Oracle, fast_build_runs_here, oracle_build_flags
    orc = Oracle(fast=fast_build_runs_here())
    inst = _instance(data, ell, 0)
    cores = effective_cores()
    ok, tp, tv = orc.time_prove_verify(inst, data["crs"], 1)
    reps = max(1, int(budget_s / max(tp + tv, 1e-3) / 1.5))
    done = [0] * cores

    def work(t):
        ok_t, _, _ = orc.time_prove_verify(inst, data["crs"], reps)
        done[t] = reps if ok_t else 0
    th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    return dict(value=sum(done) / dt, unit="proofs/s", cores=cores, logical_cpus=os.cpu_count(), kind="port", cpu_model=model, build_flags=oracle_build_flags(fast=orc.fast),
                sample="%d threads x %d x (CurdleproofsProof::new + verify) of instance 0 at ell=%d, oracle/ (one independent proof per thread), %.1f s wall"
                       % (cores, reps, ell, dt))


def run_config5(args, ctxs, shares, data, ell, B, K, world, rank, dist, torch, red_dev="cuda"):
    """BASELINE config 5: batched verifier.  Proofs are produced once (untimed); a step = every context's fused
    verification of its share (one accumulated MSM per context), the per-GPU partial sums added on the GPU, one
    all-gather of 144-byte partials over the ranks and the final identity test."""
    import threading
    import curdleproofs_amd as cpx
    from curdleproofs_amd import params as pr
    from curdleproofs_amd.dist import all_gather_partials
    rng = random.Random(99 + rank)
    proofs, rands = [], []
    for k, c in enumerate(ctxs):
        sh = shares[k]
        proofs.append(c.prove_batch(sh["perms"], sh["k"], sh["mbl"], sh["prand"], raw=True))
        rands.append(c.marshal(pr.random_fr_wire(rng, 12 * (B // K))))

    def step():
        out = [None] * K

        def one(k):
            out[k] = ctxs[k].verify_batch_fused_partial(proofs[k], rands[k])
        th = [threading.Thread(target=one, args=(k,)) for k in range(K)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        part, _ = ctxs[0].sum_jac(b"".join(o[0] for o in out))              # this GPU's partial
        parts, bad = all_gather_partials(part, sum(o[1] for o in out), dist, red_dev)
        return bad == 0 and ctxs[0].sum_jac(parts)[1]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ok = step()
    for c in ctxs:
        c.set_profiling(True)
        c.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ok = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stats = {}
    for c in ctxs:
        for name, st in c.stats().items():
            acc = stats.setdefault(name, dict(launches=0, ms=0.0, alg_bytes=0.0, units=0.0))
            for f in acc:
                acc[f] += st[f]
        c.set_profiling(False)
    if rank == 0:
        n = ell + 4
        value = world * B * args.steps / dt
        st = stats["k_msm_tblw<2, true>"]   # groups of proofs as tasks of the endomorphism bucket-list kernel
        avg_ms = st["ms"] / max(st["launches"], 1)
        bpl = st["alg_bytes"] / max(st["launches"], 1)
        ach = bpl / (avg_ms * 1e-3) / 1e9 if avg_ms else 0.0
        out = dict(metric="shuffle proofs/sec checked by the batched verifier (BASELINE config 5) at ell=%d, BLS12-381" % ell, value=value, unit="proofs/s",
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="int64", data="synthetic",
                   config=dict(workload="ell=%d: %d proofs per GPU per step in ONE accumulated check (%d points per context), "
                                        "all-gather of one 144-byte partial sum per rank" % (ell, B, (B // K) * (4 * ell + 103 + 10 * (n.bit_length() - 1))),
                               ell=ell, batch_per_gpu=B, global_batch=B * world, engine_contexts_per_gpu=K,
                               parallelism="proofs sharded over %d GPU(s); one all-gather (RCCL) of %d x 148 bytes" % (world, world)),
                   all_verified=bool(ok), kernel_ms_per_step={k: v["ms"] / args.steps for k, v in stats.items() if v["launches"]},
                   roofline=dict(bound="hbm", kernel="k_msm_tblw<2, true> over groups of proofs",
                                 achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=None, avg_launch_ms=avg_ms,
                                 launches=st["launches"], alg_bytes_per_launch=bpl, units_per_launch=st["units"] / max(st["launches"], 1)),
                   cpu_baseline=None)
        if not args.no_cpu_baseline:
            base = cpu_baseline(ell, data, None)
            out["cpu_baseline"] = dict(value=1.0 / base["verify_s"], unit="proofs/s", cores=1, kind="port",
                                       sample="CurdleproofsProof::verify of instance 0 (oracle/, single thread, %.3f s each); the oracle has no batched verifier" % base["verify_s"])
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


def load_first(c0, data, ell, nb):
    """context c0 <- the first nb instances of the synthetic batch"""
    import curdleproofs_amd as cpx
    AFF, JAC = cpx.AFF, cpx.JAC
    c0.load_batch(data["vec_R"][:nb * ell * AFF], data["vec_S"][:nb * ell * AFF], data["vec_T"][:nb * ell * AFF], data["vec_U"][:nb * ell * AFF], data["M"][:nb * JAC])


def shape_passes(c0, data, ell, nb, min_s=1.0):
    """steady-state prove + verify passes over the first nb instances on one engine context"""
    import curdleproofs_amd as cpx
    FR, n_ = cpx.FR, ell + 4
    load_first(c0, data, ell, nb)
    a_ = [c0.marshal(x) for x in (data["perms"][:nb * ell], data["k"][:nb * FR], data["mbl"][:nb * 4 * FR], data["prand"][:nb * (3 * n_ + 9) * FR],
                                  data["vrand"][:nb * 8 * FR])]
    reps, okall = 0, True
    t0 = time.perf_counter()
    while reps < 3 or (time.perf_counter() - t0 < min_s and reps < 50):
        pr_ = c0.prove_batch(a_[0], a_[1], a_[2], a_[3], raw=True)
        okall &= all(v == cpx.CPX_OK for v in c0.verify_batch(pr_, a_[4]))
        reps += 1
    dt_ = time.perf_counter() - t0
    return dict(value=nb * reps / dt_, unit="proofs/s", ms_per_pass=1e3 * dt_ / reps, passes=reps, all_verified=bool(okall))


def shape_passes_two(c0, c1, data, ell, nb, min_s=1.0):
    """the same, driven the way the headline is: two engine contexts with nb proofs each (instances 0..nb-1 and nb..2nb-1 of the bench batch), one host
    thread each, passes back to back"""
    import threading
    import curdleproofs_amd as cpx
    AFF, FR, JAC, n_ = cpx.AFF, cpx.FR, cpx.JAC, ell + 4
    arg = []
    for j, c in enumerate((c0, c1)):
        cut = lambda key, unit: data[key][j * nb * unit:(j + 1) * nb * unit]
        c.load_batch(cut("vec_R", ell * AFF), cut("vec_S", ell * AFF), cut("vec_T", ell * AFF), cut("vec_U", ell * AFF), cut("M", JAC))
        arg.append([c.marshal(x) for x in (data["perms"][j * nb * ell:(j + 1) * nb * ell], cut("k", FR), cut("mbl", 4 * FR), cut("prand", (3 * n_ + 9) * FR), cut("vrand", 8 * FR))])
    ok = [True, True]

    def passes(j, n):
        c, a = (c0, c1)[j], arg[j]
        try:
            for _ in range(n):
                pr_ = c.prove_batch(a[0], a[1], a[2], a[3], raw=True)
                ok[j] &= all(v == cpx.CPX_OK for v in c.verify_batch(pr_, a[4]))
        except Exception:
            ok[j] = False
            raise

    def both(n):
        th = threading.Thread(target=passes, args=(1, n))
        th.start()
        passes(0, n)
        th.join()
    both(2)
    t0 = time.perf_counter()
    both(1)
    n2 = max(3, min(50, int(round(min_s / max(time.perf_counter() - t0, 1e-3)))))
    t0 = time.perf_counter()
    both(n2)
    dt = time.perf_counter() - t0
    return dict(value=2 * nb * n2 / dt, unit="proofs/s", batch_per_context=nb, ms_per_pass_of_both=1e3 * dt / n2, passes=n2, all_verified=bool(all(ok)))


def fused_check(c0, proofs_raw, nb, ell, peak_mm, prod_per_add, brief=False):
    """BASELINE configs[4] on ONE engine context: the nb proofs loaded into c0 (bytes in proofs_raw) go through ONE accumulated check
    (cpx_batch_verify_fused + cpx_g1_sum_jac), verdict included; a batch with one wrong proof must be refused."""
    from curdleproofs_amd import params as pr_
    L = (ell + 4).bit_length() - 1
    frand = c0.marshal(pr_.random_fr_wire(random.Random(55), 12 * nb))

    def fused_once(blob):
        part, bad = c0.verify_batch_fused_partial(blob, frand)
        return bad == 0 and c0.sum_jac(part)[1]
    ok5 = fused_once(proofs_raw)   # warm-up (plans, buffers)
    c0.set_profiling(True)
    c0.reset_stats()
    reps5 = 0
    t0 = time.perf_counter()
    while reps5 < 3 or (time.perf_counter() - t0 < 1.0 and reps5 < 20):
        ok5 = fused_once(proofs_raw) and ok5
        reps5 += 1
    dt5 = time.perf_counter() - t0
    st5 = c0.stat("k_msm_tblw<2, true>")
    c0.set_profiling(False)
    bad_proofs = bytearray(bytes(proofs_raw))
    bad_proofs[48:96], bad_proofs[0:48] = bad_proofs[0:48], bad_proofs[48:96]      # proof 0 with two points swapped: the batch must be refused
    rejects = not fused_once(bytes(bad_proofs))
    npts5 = nb * (4 * ell + 6 + 18 + 10 * L)
    mm5 = st5["units"] * 32 * 255.0 / 256 * prod_per_add / (st5["ms"] * 1e-3) if st5["ms"] else 0.0
    d = dict(value=nb * reps5 / dt5, unit="proofs/s", proofs_per_check=nb, ms_per_check=1e3 * dt5 / reps5, checks=reps5, all_verified=bool(ok5),
             rejects_a_batch_with_one_wrong_proof=bool(rejects), points_in_the_accumulated_msm=npts5 + ell + 4,
             algorithmic_bytes_per_check=128.0 * (npts5 + ell + 4), kernel_ms_per_check=st5["ms"] / max(reps5, 1),
             valu=dict(achieved=mm5, peak=peak_mm, frac=mm5 / peak_mm if peak_mm else None, unit="381-bit modular products/s"))
    if not brief:
        d = dict(dict(metric="shuffle proofs/sec checked by the batched verifier (BASELINE configs[4]) at ell=%d" % ell, kernel="k_msm_tblw<2, true>"), **d)
        d["hbm"] = dict(achieved=128.0 * st5["units"] / (st5["ms"] * 1e-3) / 1e9 if st5["ms"] else None, peak=HBM_PEAK_GBS, unit="GB/s")
        d["note"] = ("one engine context alone on the GPU, proofs and instances resident in HBM, 12 random factors per proof uploaded per check; "
                     "the oracle has no batched verifier: cpu_verify_per_proof is its per-proof verifier")
    return d


def other_ell(c0, e3, b3, seed, what, c1=None):
    """another proof size as throughput on one engine context (its own CRS, fixed-base table and synthetic batch): prove + verify passes,
    and the prover's and the verifier's rates apart."""
    import curdleproofs_amd as cpx
    try:
        d3 = synth_batch(c0, e3, b3, seed=seed)
        c0.load_batch(d3["vec_R"], d3["vec_S"], d3["vec_T"], d3["vec_U"], d3["M"])
        a3 = [c0.marshal(d3[k_]) for k_ in ("perms", "k", "mbl", "prand", "vrand")]
        pr_ = c0.prove_batch(a3[0], a3[1], a3[2], a3[3], raw=True)   # warm-up (buffers, plans)
        okall = all(v == cpx.CPX_OK for v in c0.verify_batch(pr_, a3[4]))
        reps, tp, tv = 0, 0.0, 0.0
        t0 = time.perf_counter()
        while reps < 2 or (time.perf_counter() - t0 < 2.0 and reps < 20):
            t1 = time.perf_counter()
            pr_ = c0.prove_batch(a3[0], a3[1], a3[2], a3[3], raw=True)
            t2 = time.perf_counter()
            okall &= all(v == cpx.CPX_OK for v in c0.verify_batch(pr_, a3[4]))
            tv += time.perf_counter() - t2
            tp += t2 - t1
            reps += 1
        dt_ = time.perf_counter() - t0
        res = dict(value=b3 * reps / dt_, unit="proofs/s", batch=b3, ms_per_pass=1e3 * dt_ / reps, passes=reps, all_verified=bool(okall),
                   prove_per_s=b3 * reps / tp, verify_per_s=b3 * reps / tv, prove_ms_per_pass=1e3 * tp / reps, verify_ms_per_pass=1e3 * tv / reps,
                   note="%s: ell=%d, %d proofs per pass, one engine context; value = prove + verify, prove_per_s / verify_per_s = CurdleproofsProof::new / "
                        "::verify alone (as criterion reports them)" % (what, e3, b3))
        if c1 is not None:
            # the same shape driven the way the headline is: a second engine context with a batch of its own (the same instances) on a second host
            # thread — one context's latency-bound kernels beside the other's MSM waves
            import threading
            c1.set_crs(e3, d3["crs"])
            c1.load_batch(d3["vec_R"], d3["vec_S"], d3["vec_T"], d3["vec_U"], d3["M"])
            b1 = [c1.marshal(d3[k_]) for k_ in ("perms", "k", "mbl", "prand", "vrand")]
            ok2 = [True, True]

            def passes(c, a, slot, n):
                try:
                    for _ in range(n):
                        pr = c.prove_batch(a[0], a[1], a[2], a[3], raw=True)
                        ok2[slot] &= all(v == cpx.CPX_OK for v in c.verify_batch(pr, a[4]))
                except Exception:   # (a failure on the helper thread must not read as a fast pass)
                    ok2[slot] = False
                    raise

            def both(n):
                th = threading.Thread(target=passes, args=(c1, b1, 1, n))
                th.start()
                passes(c0, a3, 0, n)
                th.join()

            both(1)
            n2 = max(2, min(8, int(round(1.5 / max(dt_ / reps, 1e-3)))))
            t0 = time.perf_counter()
            both(n2)
            dt2 = time.perf_counter() - t0
            res["two_contexts"] = dict(value=2 * b3 * n2 / dt2, unit="proofs/s", batch_per_context=b3, ms_per_pass_of_both=1e3 * dt2 / n2, passes=n2, all_verified=bool(all(ok2)),
                                       note="two engine contexts x %d proofs, one host thread each, passes back to back (how the headline runs its 2 x 8192)" % b3)
        return res
    except Exception as e:   # e.g. not enough free HBM for this ell's fixed-base table beside the bench's buffers
        return dict(value=None, error=repr(e))


PROD_PER_ADD = 3542.0 / 392.0   # modular products of one XYZZ mixed addition of the bucket loops (8M + 2S, two products of Y3 under one reduction: 3542 multiply-adds of 392)


def msm_pair_cost(kernel):
    """modular products one (scalar, point) pair costs in MSM kernel `kernel` (a name as Context.KERNELS / rocprofv3 spell it), 0.0 for a kernel that
    moves no MSM pairs.  ONE table for every place that prices a kernel: until round 5 the dominant-kernel lookup had its own, without
    k_msm_tblw_pair — whenever that kernel happened to be the longest MSM kernel of a small two-rank run, rank 0 died of a KeyError after the timed
    region (the 'launcher failure' of tests/test_gpu_parity.py::test_bench_two_ranks_on_one_gpu)."""
    base = kernel.split("<")[0]
    if base in ("k_msm_tblw", "k_msm_tblw_pair"):
        return 32 * 255.0 / 256 * PROD_PER_ADD            # 32 radix-256 windows, a zero digit once in 256
    if base == "k_msm_fix":
        return (14 if kernel.startswith("k_msm_fix<19") else 16) * PROD_PER_ADD
    if base == "k_msm_accw":
        return 64 * 15.0 / 16 * 11                         # radix-16 windows, Jacobian additions (11 products)
    return 0.0


def _parse_opts(kvs):
    return {kv.split("=", 1)[0]: int(kv.split("=", 1)[1]) for kv in kvs}


def traffic_child(args):
    """What `rocprofv3 --pmc` wraps when bench.py measures roofline.traffic (measure_traffic below): ONE engine context, --batch proofs, one
    prove + verify pass — the launches of the timed region's context 0 (same plans, same pairs per launch).  64 distinct synthetic instances
    tiled over the batch with prover randomness of its own for every proof: from the first challenge on every proof's scalars, hence its table
    gathers, differ like those of the timed batch."""
    import curdleproofs_amd as cpx
    ell, Bk, n = args.ell, args.batch, args.ell + 4
    c = cpx.Context(int(args.traffic_child), options=_parse_opts(args.opt))
    c.set_host_threads(2)
    base = min(64, Bk)
    assert Bk % base == 0
    d, rep = synth_batch(c, ell, base, seed=4242), Bk // base
    c.load_batch(d["vec_R"] * rep, d["vec_S"] * rep, d["vec_T"] * rep, d["vec_U"] * rep, d["M"] * rep)
    proofs = c.prove_batch(d["perms"] * rep, d["k"] * rep, d["mbl"] * rep, synth_scalars(77, Bk * (3 * n + 9)), raw=True)
    ok = all(v == cpx.CPX_OK for v in c.verify_batch(proofs, synth_scalars(78, Bk * 8)))
    print(json.dumps(dict(traffic_child_ok=bool(ok))))
    sys.exit(0 if ok else 3)


def measure_traffic(kernel, ell, Bk, device, opts, timeout_s=240):
    """HBM bytes per launch of `kernel` from rocprofv3 PMC passes taken NOW, on this box, by this command (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and
    WRITE_SIZE do not fit one pass — two child processes, counters only, no trace domain beside them; KiB units; FETCH_SIZE doubled: gfx950
    tallies the 128-byte requests of wide reads at 64 bytes).  Returns (bytes per launch or None, a dict that says how it was taken or why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, dict(skipped="this process already runs under rocprofv3")
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, dict(skipped="rocprofv3 not found")
    res, info = {}, dict(method="two rocprofv3 --pmc child passes of this run (FETCH_SIZE, WRITE_SIZE), one engine context x %d proofs, one prove + verify pass each; "
                                "bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / dispatches" % Bk, kernel=kernel)
    work = tempfile.mkdtemp(prefix="cpx_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--traffic-child", str(device),
                   "--ell", str(ell), "--batch", str(Bk)] + [a for kv in opts.items() for a in ("--opt", "%s=%d" % kv)]
            t0 = time.perf_counter()
            p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                log, _ = p.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)   # (our own process group: start_new_session)
                p.wait()
                return None, dict(info, failed="%s pass exceeded %d s" % (counter, timeout_s))
            if p.returncode != 0 or b"traffic_child_ok\": true" not in log:
                return None, dict(info, failed="%s pass: exit code %d" % (counter, p.returncode), log_tail=log[-400:].decode(errors="replace"))
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, dict(info, failed="%s pass left no counter_collection.csv" % counter)
            tot, disp = 0.0, set()
            with open(files[0]) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] == counter and r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cpx::", "") == kernel:
                        tot += float(r["Counter_Value"])
                        disp.add(r["Dispatch_Id"])
            if not disp:
                return None, dict(info, failed="%s pass saw no dispatch of %s" % (counter, kernel))
            res[counter] = dict(total_kib=tot, dispatches=len(disp), seconds=round(time.perf_counter() - t0, 1))
        info.update(res)
        return (2.0 * res["FETCH_SIZE"]["total_kib"] / res["FETCH_SIZE"]["dispatches"] + res["WRITE_SIZE"]["total_kib"] / res["WRITE_SIZE"]["dispatches"]) * 1024.0, info
    except Exception as e:   # a measurement aid must not take the bench line down
        return None, dict(info, failed=repr(e))
    finally:
        shutil.rmtree(work, ignore_errors=True)


def run_ab(args, variants, B, barrier, world):
    """--ab / --ab-lib: blocks of prove + verify passes alternate between the variants in ONE process on ONE box (order reversed every round: A B, B A, ...).
    variants: [(label, contexts, shares, options)].  Before a block the variant's options are set on its contexts (only when they differ from what the
    context holds: an option change drops the plans) and one pass runs untimed; then `passes` timed passes, all contexts of the variant side by
    side as in the headline.  The ratio of the means decides; the per-round ratios show the noise it has to beat."""
    import threading
    import curdleproofs_amd as cpx
    held = {}

    def apply(ctxs, opts):
        for c in ctxs:
            for k, v in opts.items():
                if held.get((id(c), k)) != v:
                    c.set_option(k, v)
                    held[(id(c), k)] = v

    def passes(ctxs, shares, n):
        bad = [0] * len(ctxs)

        def run(i):
            c, sh = ctxs[i], shares[i]
            for _ in range(n):
                proofs = c.prove_batch(sh["perms"], sh["k"], sh["mbl"], sh["prand"], raw=True)
                bad[i] += sum(1 for v in c.verify_batch(proofs, sh["vrand"]) if v != cpx.CPX_OK)
        th = [threading.Thread(target=run, args=(i,)) for i in range(1, len(ctxs))]
        for t in th:
            t.start()
        run(0)
        for t in th:
            t.join()
        return sum(bad)

    nbad = 0
    for label, ctxs, shares, opts in variants:   # buffers and plans of every variant exist before anything is timed
        apply(ctxs, opts)
        nbad += passes(ctxs, shares, 1)
    t0 = time.perf_counter()
    nbad += passes(variants[0][1], variants[0][2], 1)
    n = args.ab_passes or max(2, int(round(2.0 / max(time.perf_counter() - t0, 1e-3))))
    rates = [[] for _ in variants]
    for r in range(args.ab_rounds):
        order = list(range(len(variants)))
        if r % 2:
            order.reverse()
        for vi in order:
            label, ctxs, shares, opts = variants[vi]
            apply(ctxs, opts)
            nbad += passes(ctxs, shares, 1)
            barrier()
            t0 = time.perf_counter()
            nbad += passes(ctxs, shares, n)
            barrier()
            rates[vi].append(world * B * n / (time.perf_counter() - t0))
    mean = [sum(x) / len(x) for x in rates]
    per_round = [[rates[vi][r] / rates[0][r] for r in range(args.ab_rounds)] for vi in range(len(variants))]
    return dict(variants=[dict(label=variants[vi][0], options=variants[vi][3], proofs_per_s_by_round=rates[vi], mean=mean[vi], ratio_to_first=mean[vi] / mean[0],
                               ratio_by_round=per_round[vi], ratio_min=min(per_round[vi]), ratio_max=max(per_round[vi])) for vi in range(len(variants))],
                rounds=args.ab_rounds, passes_per_block=n, order="A B, B A, ... (one untimed pass before every block)", all_verified=nbad == 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32768, help="instances per GPU per step (32 x the BASELINE configs[3] batch, split over the engine contexts: 8192 per context; "
                    "smaller batches amortise the latency-bound kernels of a pass over fewer proofs: 4096 per context measured 3 %% lower)")
    ap.add_argument("--ell", type=int, default=252)
    ap.add_argument("--fused-verify", action="store_true", help="measure BASELINE config 5 instead: the batched verifier (one accumulated MSM over "
                    "all proofs of all GPUs, one all-gather of 144-byte partial sums); not the headline metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-threads", type=int, default=0, help="host worker threads per engine context (0 = cores / (ranks x contexts), clamped to [2, 24])")
    ap.add_argument("--synth-cache", default="", help="pickle file for the synthetic batch (created if missing): keeps profiled runs free of setup kernels")
    ap.add_argument("--ctxs", type=int, default=4, help="engine contexts (HIP streams, one host thread each) per GPU; the batch is split over them: the "
                    "protocol is device-resident, several contexts let one context's latency-bound kernels (reductions, finalisation, step "
                    "kernels) and the tails of its launches run beside the others' MSM waves.  4 x 8192 since round 6 (one box, processes alternating: "
                    "2 x 8192 16.44 k, 4 x 8192 16.84 k proofs/s, +2.4 %% in both pairs, profiles/r06_ctxs_2_vs_4.txt; 233 of the 288 GB of HBM; round 5 "
                    "kept 2 x 8192 for the footprint).  --ab-lib doubles the contexts: run it with --ctxs 2 --batch 16384")
    ap.add_argument("--no-extra-shapes", action="store_true", help="skip the extra keys: throughput at BASELINE's literal batch shapes (128 / 1024 proofs per "
                    "GPU), ell=1020 throughput and the all-cores CPU baseline")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="engine option for every context (cpx_ctx_set_option), e.g. "
                    "--opt late_rounds=0 for an A/B run")
    ap.add_argument("--ab", default="", metavar="KEY=V0,V1[,V2...]", help="in-process interleaved A/B of an engine option on the SAME contexts, batches and box: blocks of "
                    "passes alternate between the values (order reversed every round), every block after one untimed pass under the new value (plans are rebuilt "
                    "after an option change); prints the rate of every block, the means and their ratio instead of the bench line.  Honours --batch / --ctxs / --ell / --opt")
    ap.add_argument("--ab-lib", default="", metavar="PATH", help="the same A/B between two BUILDS of the library (variant 0 = the default libcpx.so, variant 1 = PATH, e.g. "
                    "curdleproofs_amd.build.build_variant): a second set of contexts with the same instances lives beside the first, blocks of passes alternate")
    ap.add_argument("--ab-rounds", type=int, default=6, help="rounds of the A/B (each round = one block per variant)")
    ap.add_argument("--ab-passes", type=int, default=0, help="timed passes per block (0 = about 2 s worth)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child passes that measure roofline.traffic (then null)")
    ap.add_argument("--traffic-child", default="", help=argparse.SUPPRESS)   # internal: the process rocprofv3 wraps (one pass of one context under counters)
    args = ap.parse_args()
    if args.traffic_child:
        return traffic_child(args)

    # --gpus N without a launcher around us: start the N ranks ourselves (one process per GPU, the same command line the
    # driver uses for N > 1) and hand over; under torch.distributed.run WORLD_SIZE is set and must agree with --gpus.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        # --standalone: the launcher's own c10d rendezvous on a port IT binds (localhost:0) — no port is picked here and handed over (the
        # bind-close-reuse of rounds 1-5 left a window for another process to take it); --local-addr: the container hostname may not resolve
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    # test hooks (a 1-GPU box can rehearse the multi-rank path): CPX_BENCH_DEVICE pins every rank to one device,
    # CPX_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on the same GPU)
    backend = os.environ.get("CPX_BENCH_BACKEND", "nccl")
    if "CPX_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["CPX_BENCH_DEVICE"])
    red_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        if not os.environ.get("CPX_BENCH_DRYRUN"):
            torch.cuda.set_device(local_rank)
        import datetime
        # (rank 0 runs the extra shapes and the CPU baseline while the other ranks wait at the final barrier: minutes, not seconds)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=60))
        else:
            dist.init_process_group(backend=backend, timeout=datetime.timedelta(minutes=60))
        assert dist.get_world_size() == args.gpus, "process group size %d != --gpus %d" % (dist.get_world_size(), args.gpus)
        world = dist.get_world_size()   # n_gpus in the output line is what the process group (RCCL) reports
    if os.environ.get("CPX_BENCH_DRYRUN"):
        # launcher rehearsal without GPUs (tests/test_bench_contract_cpu.py): the ranks meet, take the max over ranks of a
        # dummy step time like the real run does, rank 0 reports the world it saw.  No proof is produced: value is null.
        dt = 1.0 + rank
        if dist is not None:
            dist.barrier()
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if rank == 0:
            print(json.dumps(dict(dry_run=True, metric="shuffle proofs/sec (prove+verify) at ell=%d, BLS12-381" % args.ell, value=None, n_gpus=world,
                                  steps=args.steps, warmup=args.warmup, max_over_ranks_dt=dt,
                                  host_threads_per_context=args.host_threads or max(2, min(24, (os.cpu_count() or 64) // (world * max(1, args.ctxs)))))))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    import curdleproofs_amd as cpx
    import threading
    ell, B, K = args.ell, args.batch, max(1, args.ctxs)
    assert B % K == 0, "--batch must be a multiple of --ctxs"
    engine_opts = _parse_opts(args.opt)
    n_ = ell + 4

    def make_shares(cs, data, Bk, first):
        """every context of `cs` gets the CRS and its share of the batch (instances resident in HBM, witnesses in page-locked host memory)"""
        def part(buf, unit, k):
            return buf[k * Bk * unit:(k + 1) * Bk * unit]
        shs = []
        for k, c in enumerate(cs):
            if c is not first:
                c.set_crs(ell, data["crs"])
            sh = dict(vec_R=part(data["vec_R"], ell * cpx.AFF, k), vec_S=part(data["vec_S"], ell * cpx.AFF, k),
                      vec_T=part(data["vec_T"], ell * cpx.AFF, k), vec_U=part(data["vec_U"], ell * cpx.AFF, k), M=part(data["M"], cpx.JAC, k),
                      perms=data["perms"][k * Bk * ell:(k + 1) * Bk * ell], k=part(data["k"], cpx.FR, k), mbl=part(data["mbl"], 4 * cpx.FR, k),
                      prand=part(data["prand"], (3 * n_ + 9) * cpx.FR, k), vrand=part(data["vrand"], 8 * cpx.FR, k))
            # host-side buffers, marshalled once into page-locked memory: the witnesses of every prove, and the public instance (R, S, T, U, M)
            # that cpx_batch_load moves — once before the timed region, and once per pass in the value_incl_instance_upload leg
            for key in sh:
                sh[key] = c.marshal(sh[key])
            c.load_batch(sh["vec_R"], sh["vec_S"], sh["vec_T"], sh["vec_U"], sh["M"])   # inputs now resident in HBM
            shs.append(sh)
        return shs

    def setup(K_, B_):
        """K_ engine contexts holding B_ synthetic instances between them, and ONE prove + verify pass on all of them at once: every buffer of the
        timed region exists when this returns (the engine allocates its scratch at the first pass of a shape)"""
        cs = []
        try:
            if K_ > int(os.environ.get("CPX_BENCH_FAIL_SETUP_ABOVE", "1000000")):   # test hook: rehearses the fallback without exhausting HBM
                raise cpx.CpxError(cpx.CPX_ERR_HIP, "CPX_BENCH_FAIL_SETUP_ABOVE: set-up with %d contexts refused" % K_)
            cs = [cpx.Context(local_rank, options=engine_opts) for _ in range(K_)]          # raises without a GPU: no CPU fallback
            # host worker threads per context: the cores of the node divided over ranks x contexts (all ranks share one node)
            ht = args.host_threads or max(2, min(24, effective_cores() // (world * K_)))
            for c in cs:
                c.set_host_threads(ht)
            data_ = None
            cache = args.synth_cache and ("%s.r%d" % (args.synth_cache, rank))
            if cache and os.path.exists(cache):
                import pickle
                with open(cache, "rb") as f:
                    data_ = pickle.load(f)
                if data_.get("ell") != ell or data_.get("B") != B_ or "setup_s" not in data_:
                    data_ = None
                else:
                    cs[0].set_crs(ell, data_["crs"])
                    data_["setup_s"] = dict(data_["setup_s"], from_cache=True)
            if data_ is None:
                data_ = synth_batch(cs[0], ell, B_, seed=1234 + rank, helpers=cs[1:])
                data_["ell"], data_["B"] = ell, B_
                if cache:
                    import pickle
                    with open(cache, "wb") as f:
                        pickle.dump(data_, f)
            shs = make_shares(cs, data_, B_ // K_, cs[0])
            errs = []

            def probe(k):
                try:
                    pr_ = cs[k].prove_batch(shs[k]["perms"], shs[k]["k"], shs[k]["mbl"], shs[k]["prand"], raw=True)
                    cs[k].verify_batch(pr_, shs[k]["vrand"])
                except Exception as e:
                    errs.append(e)
            th = [threading.Thread(target=probe, args=(k,)) for k in range(K_)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if errs:
                raise errs[0]
            return cs, ht, data_, shs
        except Exception:
            for c in cs:
                try:
                    c.close()
                except Exception:
                    pass
            raise

    # Four contexts x 8192 proofs hold 233 of the 288 GB of HBM.  Should that not fit (something else on the device), every rank falls back to half
    # the contexts and half the batch — together: `value` counts world x B proofs per step — and the line says so (config.fallback).
    fallback = None
    while True:
        err = None
        try:
            ctxs, host_threads, data, shares = setup(K, B)
        except cpx.CpxError as e:
            err = e
        ok = 0 if err else 1
        if dist is not None:
            t = torch.tensor([ok], dtype=torch.int32, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        if ok:
            break
        if K <= 2 or B % 2 or (B // 2) % (K // 2):
            raise err if err else RuntimeError("another rank could not set up its contexts")
        if not err:
            for c in ctxs:
                c.close()
        fallback = dict(from_contexts=K, from_batch=B, reason=repr(err) if err else "another rank's set-up failed")
        K, B = K // 2, B // 2
    ctx = ctxs[0]
    Bk = B // K

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.ab or args.ab_lib:
        variants = []
        if args.ab:
            key, vals = args.ab.split("=", 1)
            variants = [("%s=%s" % (key, v), ctxs, shares, {key: int(v)}) for v in vals.split(",")]
        if args.ab_lib:
            ctxs_b = [cpx.Context(local_rank, options=engine_opts, lib=args.ab_lib) for _ in range(K)]
            for c in ctxs_b:
                c.set_host_threads(host_threads)
            shares_b = make_shares(ctxs_b, data, Bk, None)
            base = variants or [("", ctxs, shares, {})]
            variants = [((lbl + " " if lbl else "") + "libcpx.so", c_, s_, o_) for lbl, c_, s_, o_ in base] + \
                       [((lbl + " " if lbl else "") + os.path.basename(args.ab_lib), ctxs_b, shares_b, o_) for lbl, _, _, o_ in base]
        res = run_ab(args, variants, B, barrier, world)
        if rank == 0:
            free_b, total_b = torch.cuda.mem_get_info()
            print(json.dumps(dict(ab=res, metric="shuffle proofs/sec (prove+verify) at ell=%d, BLS12-381" % ell, unit="proofs/s", n_gpus=world,
                                  config=dict(ell=ell, batch_per_gpu=B, engine_contexts_per_gpu=K, host_threads_per_context=host_threads, options=engine_opts,
                                              hbm_used_gb=round((total_b - free_b) / 1e9, 1)))))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(0 if res["all_verified"] else 3)

    if args.fused_verify:
        return run_config5(args, ctxs, shares, data, ell, B, K, world, rank, dist, torch, red_dev)

    # everything after the timed region (isolated pass, literal shapes, CPU baseline sample) draws on context 0's share only: the other contexts' instances
    # live on in their page-locked buffers, the Python-side copies go (8 ranks of one node share its host memory: 2.4 GB less per rank)
    for key, unit in (("vec_R", ell * cpx.AFF), ("vec_S", ell * cpx.AFF), ("vec_T", ell * cpx.AFF), ("vec_U", ell * cpx.AFF), ("M", cpx.JAC), ("k", cpx.FR),
                      ("mbl", 4 * cpx.FR), ("prand", (3 * n_ + 9) * cpx.FR), ("vrand", 8 * cpx.FR)):
        data[key] = data[key][:Bk * unit]
    data["perms"] = data["perms"][:Bk * ell]

    def step_one(k, out):
        c, sh = ctxs[k], shares[k]
        proofs = c.prove_batch(sh["perms"], sh["k"], sh["mbl"], sh["prand"], raw=True)
        out[k] = (proofs, c.verify_batch(proofs, sh["vrand"]))

    def steps(nsteps):
        """nsteps passes over the batch.  The engine contexts are independent (each owns a quarter of the instances): every
        context thread runs its nsteps prove+verify passes back to back, so the contexts drift out of phase and one
        context's host work keeps overlapping the others' kernels across step boundaries.  Verdicts of every pass are kept."""
        out = [None] * K
        bad = [0] * K

        def run(k):
            for _ in range(nsteps):
                step_one(k, out)
                bad[k] += sum(1 for v in out[k][1] if v != cpx.CPX_OK)
        if K == 1:
            run(0)
        else:
            th = [threading.Thread(target=run, args=(k,)) for k in range(K)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        return [o[0] for o in out], sum(bad)

    warm = {}   # kernel statistics of the warm-up launches (only used for the all-launches average rocprofv3 reports)
    if args.warmup:
        for c in ctxs:
            c.set_profiling(True)
            c.reset_stats()
        steps(args.warmup)
        for c in ctxs:
            for name, st in c.stats().items():
                acc = warm.setdefault(name, dict(launches=0, ms=0.0))
                acc["launches"] += st["launches"]
                acc["ms"] += st["ms"]
    for c in ctxs:
        c.set_profiling(True)
        c.reset_stats()
    barrier()
    t0 = time.perf_counter()
    proofs, nbad = steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    all_ok = nbad == 0
    stats = {}
    for c in ctxs:
        for name, st in c.stats().items():
            acc = stats.setdefault(name, dict(launches=0, ms=0.0, alg_bytes=0.0, units=0.0))
            for f in acc:
                acc[f] += st[f]
        c.set_profiling(False)
    # The same passes with the public instance crossing PCIe inside the timed region (CurdleproofsProof::new / verify take vec_R, vec_S, vec_T,
    # vec_U, M as arguments, curdleproofs.rs:59-70,197-207; `value` has them resident).  Two ways: "overlapped" — every pass stages the NEXT pass's
    # instance with cpx_batch_load_begin before its prove and makes it the loaded batch with cpx_batch_load_end after its verify (97 KB per proof from
    # page-locked memory on an upload stream beside the kernels) — and "sync": cpx_batch_load at the top of every pass.
    up_steps = max(2, min(args.steps, 5))

    def steps_with_upload(nsteps, overlapped):
        bad = [0] * K

        def run(k):
            c, sh = ctxs[k], shares[k]
            inst = (sh["vec_R"], sh["vec_S"], sh["vec_T"], sh["vec_U"], sh["M"])
            for _ in range(nsteps):
                if overlapped:
                    c.load_batch_begin(*inst)
                else:
                    c.load_batch(*inst)
                proofs_k = c.prove_batch(sh["perms"], sh["k"], sh["mbl"], sh["prand"], raw=True)
                bad[k] += sum(1 for v in c.verify_batch(proofs_k, sh["vrand"]) if v != cpx.CPX_OK)
                if overlapped:
                    c.load_batch_end()
        th = [threading.Thread(target=run, args=(k,)) for k in range(1, K)]
        for t in th:
            t.start()
        run(0)
        for t in th:
            t.join()
        return sum(bad)

    def timed_upload_leg(overlapped):
        steps_with_upload(1, overlapped)   # (staging buffers, streams)
        barrier()
        t0 = time.perf_counter()
        nb_ = steps_with_upload(up_steps, overlapped)
        barrier()
        d_ = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([d_], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d_ = float(t.item())
        return d_, nb_
    dt_up, nbad_up = timed_upload_leg(True)
    dt_up_sync, nbad_sync = timed_upload_leg(False)
    nbad_up += nbad_sync
    hbm_used_gb = None
    if torch.cuda.is_available():
        free_b, total_b = torch.cuda.mem_get_info()
        hbm_used_gb = round((total_b - free_b) / 1e9, 1)   # with every context's batch resident: the working set of the timed region
    for c in ctxs[2:]:   # what follows (isolated pass, literal shapes, other proof sizes) runs on one or two contexts: the others give their HBM back
        c.close()
    ctxs = ctxs[:2]
    iso = None
    if rank == 0:   # one more pass of context 0 ALONE, outside the timed region, with its side stream's kernels in line on the main stream
        c = ctxs[0]  # (option serial_streams): every kernel then has the GPU to itself and its duration is its own — in the timed region a
        c.set_option("serial_streams", 1)   # kernel shares the GPU with the other context AND with its own context's side stream
        c.set_profiling(True)
        c.reset_stats()
        step_one(0, [None] * K)             # (plans are rebuilt after an option change: this pass only counts for the all-launches average)
        torch.cuda.synchronize()
        for name, st in c.stats().items():
            acc = warm.setdefault(name, dict(launches=0, ms=0.0))
            acc["launches"] += st["launches"]
            acc["ms"] += st["ms"]
        c.reset_stats()
        step_one(0, [None] * K)
        torch.cuda.synchronize()
        iso = c.stats()
        c.set_profiling(False)
        c.set_option("serial_streams", 0)

    if rank == 0:
        n = ell + 4
        L = n.bit_length() - 1
        # algorithmic bytes per prove+verify: MSM point 128 B, fold element 288 B, rescale element 224 B
        b_prove = 128 * (6 * ell + 15 * n + 2) + 288 * 5 * (n - 1) + 224 * n
        b_verify = 128 * (5 * ell + 8) + 10 * L * 176
        value = world * B * args.steps / dt
        # kernel names are the instantiations rocprofv3 reports (k_msm_tblw<32, false> = 32 windows per wave over shifted tables, ...)
        # (the dominant kernel is an MSM kernel: on a small batch the lone k_smul chain of the side stream can be the longest launch, but it
        # moves no MSM points and says nothing about the path's roofline)
        msm_kernels = [k for k in stats if k.startswith(("k_msm_fix", "k_msm_tblw", "k_msm_accw")) and stats[k]["ms"] > 0]
        dom = max(msm_kernels or [k for k in stats if k.startswith("k_smul")] or list(stats), key=lambda k: stats[k]["ms"])
        st = stats[dom]
        avg_ms = st["ms"] / max(st["launches"], 1)
        bytes_per_launch = st["alg_bytes"] / max(st["launches"], 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms else 0.0
        traffic, traffic_src = None, dict(skipped="--no-traffic")   # measured at the end of this run (measure_traffic), or null
        roofline = dict(bound="hbm", kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                        traffic=traffic, traffic_source=traffic_src, avg_launch_ms=avg_ms, launches=st["launches"], alg_bytes_per_launch=bytes_per_launch,
                        units_per_launch=st["units"] / max(st["launches"], 1),
                        # every launch of this process (warm-up + timed region + the isolated pass): the population
                        # `rocprofv3 --kernel-trace --stats` averages over when it wraps this command
                        avg_launch_ms_all_launches=(st["ms"] + warm.get(dom, {}).get("ms", 0.0) + (iso or {}).get(dom, {}).get("ms", 0.0))
                        / max(1, st["launches"] + warm.get(dom, {}).get("launches", 0) + (iso or {}).get(dom, {}).get("launches", 0)),
                        note="integer big-number path: VALU (v_mad_i64_i32) bound, the HBM fraction is tiny by construction; "
                             "roofline.valu gives the modular-multiplication roofline (DESIGN.md section 5)")
        # The honest bound of this integer path: modular multiplications on the VALU.  Peak = dependent
        # Montgomery-product chains on the table kernels' 28-bit-limb field at full occupancy, measured now on this GPU
        # (k_bench_f28mul: 392 multiply-adds per product); achieved = the dominant kernel's point additions x the product equivalents
        # of one mixed addition.  The extended-Jacobian mixed addition of the bucket loops is 8M + 2S with the two products of Y3
        # under one reduction: 3542 multiply-adds = 9.04 products of 392 (rounds 1-2 of this build counted 11, the Jacobian 7M + 4S,
        # for an addition that cost 9.5: their fractions read 16 % too high).
        mulmods_per_point = msm_pair_cost(dom)
        ctxs[0].set_option("bench_field", 28)
        peak_mm = ctxs[0].bench_fpmul(2048, 1000, 2)
        ach_mm = st["units"] * mulmods_per_point / (st["ms"] * 1e-3) if st["ms"] else 0.0
        roofline["valu"] = dict(achieved=ach_mm, peak=peak_mm, unit="381-bit modular products/s", frac=ach_mm / peak_mm if peak_mm else None,
                                mulmods_per_point=mulmods_per_point,
                                note="kernel time is measured with all engine contexts sharing the GPU, so it includes the other contexts' kernels; "
                                     "'isolated' repeats the measurement after the timed region with one context alone and its side stream's kernels in line "
                                     "(option serial_streams): the kernel has the GPU to itself")
        pair_cost = msm_pair_cost
        # The whole pass against the same peak: the products of the MSM additions of a prove + verify (every point-scalar pair of every MSM kernel of
        # the timed region x the additions a pair costs x 9.04 products) x the measured rate.  Everything else a pass executes — per-proof table
        # build, late-round materialisation, bucket reductions, decompression, transcripts — is overhead by this count, and so is idle issue time.
        useful = sum(v["units"] * pair_cost(k) for k, v in stats.items() if msm_pair_cost(k) and v["launches"]) / max(1.0, float(B * args.steps))
        roofline["valu"]["whole_pass"] = dict(frac=useful * (value / world) / peak_mm if peak_mm else None, useful_products_per_proof=useful,
                                              useful_products_per_s_per_gpu=useful * value / world,
                                              note="useful = modular products of the MSM additions only; per GPU against the product peak measured in this run")
        if iso:   # every MSM kernel, one context alone: point-scalar pairs, additions x their product equivalents against the measured peak
            roofline["valu"]["isolated_by_kernel"] = {
                k: dict(pairs_per_proof=v["units"] / (B // K), ms_per_proof_batch=v["ms"],
                        frac=(v["units"] * pair_cost(k) / (v["ms"] * 1e-3) / peak_mm) if v["ms"] and peak_mm else None,
                        # the same measurement in the accounting of rounds 1-2a (11 products per addition), for comparison with their numbers
                        frac_at_11_products_per_addition=(v["units"] * pair_cost(k) * (11.0 / PROD_PER_ADD if k.split("<")[0] != "k_msm_accw" else 1.0)
                                                          / (v["ms"] * 1e-3) / peak_mm) if v["ms"] and peak_mm else None)
                for k, v in iso.items() if msm_pair_cost(k) and v["ms"] and v["launches"] > 0}
        if iso and dom in iso and iso[dom]["ms"]:
            ist = iso[dom]
            iso_mm = ist["units"] * mulmods_per_point / (ist["ms"] * 1e-3)
            roofline["valu"]["isolated"] = dict(achieved=iso_mm, frac=iso_mm / peak_mm if peak_mm else None, launches=ist["launches"],
                                                avg_launch_ms=ist["ms"] / max(ist["launches"], 1))
            # the isolated pass as a whole: one context, one stream — the pass is the plain sum of its kernels
            roofline["valu"]["isolated_pass"] = dict(proofs=B // K, kernel_ms={k: v["ms"] for k, v in iso.items() if v["launches"] and not k.startswith("host_")},
                                                     wall_ms=iso.get("host_prove_wall", {}).get("ms", 0.0) + iso.get("host_verify_wall", {}).get("ms", 0.0))
        out = dict(metric="shuffle proofs/sec (prove+verify) at ell=%d, BLS12-381" % ell, value=value, unit="proofs/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="int64", data="synthetic",
                   config=dict(workload="ell=%d (n=%d, log2 n=%d): %d independent shuffle instances per GPU per step, each proven "
                                        "(CurdleproofsProof::new) then verified; BASELINE configs[1] instance shape, per-GPU share "
                                        "of configs[3]" % (ell, n, L, B),
                               ell=ell, batch_per_gpu=B, global_batch=B * world, engine_contexts_per_gpu=K, host_threads_per_context=host_threads,
                               fix_bits=ctxs[0].get_option("fix_bits"), fix_bits_effective=ctxs[0].get_option("fix_bits_effective"), gpu_max_hw_queues=os.environ.get("GPU_MAX_HW_QUEUES"), hbm_used_gb=hbm_used_gb, fallback=fallback, parallelism="proofs sharded over %d GPU(s), no collective" % world),
                   all_verified=all_ok and nbad_up == 0,
                   # the same metric with the 97 KB public instance of every proof uploaded (cpx_batch_load) inside the timed region on every pass
                   value_incl_instance_upload=world * B * up_steps / dt_up,
                   instance_upload=dict(steps=up_steps, ms_per_step=1e3 * dt_up / up_steps, ratio_to_value=(world * B * up_steps / dt_up) / value,
                                        bytes_per_proof=4 * ell * cpx.AFF + cpx.JAC,
                                        sync_load=dict(value=world * B * up_steps / dt_up_sync, ratio_to_value=(world * B * up_steps / dt_up_sync) / value,
                                                       note="cpx_batch_load at the top of every pass: the context's streams idle during its own upload"),
                                        note="value: instances resident in HBM before the clock starts (the contract's 'inputs already resident'); here every pass of "
                                             "every context uploads an instance of its share from page-locked host memory: cpx_batch_load_begin before the prove (the next "
                                             "pass's instance, beside this pass's kernels), cpx_batch_load_end after the verify; measured right after the timed region"),
                   setup_s=data.get("setup_s"), host_rss_peak_gb=round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1e6, 2),
                   algorithmic_bytes_per_proof=b_prove + b_verify,
                   algorithmic_GBps=value * (b_prove + b_verify) / 1e9,
                   kernel_ms_per_step={k: v["ms"] / args.steps for k, v in stats.items() if v["launches"]},
                   vs_readme_i7_8550u=value / README_I7_PROOFS_PER_S, roofline=roofline)
        # Everything below runs on rank 0 only, outside the timed region, at ANY N (the other ranks wait at the final barrier): the
        # literal BASELINE shapes, the batched verifier, the other proof sizes and the CPU baseline belong to every line.  Every stage is guarded: a
        # failure in an extra (round 5: a KeyError after the timed region took rank 0 down, and the line with it) is recorded under `errors` and the
        # contract line is still printed.
        AFF, FR, JAC = cpx.AFF, cpx.FR, cpx.JAC
        c0, n_ = ctxs[0], ell + 4
        errors = {}

        def stage(name, fn):
            try:
                fn()
            except Exception as e:
                import traceback
                errors[name] = "%r | %s" % (e, traceback.format_exc(limit=4).replace("\n", " / "))

        def st_config5():
            # BASELINE configs[4] (the batched verifier), driver-visible: context 0 still holds its share of the timed batch and the proofs
            # of the last pass; ONE accumulated check over all of them, verdict included (cpx_batch_verify_fused + cpx_g1_sum_jac; with
            # N GPUs the 144-byte partial sums are all-gathered first: --fused-verify)
            out["config5"] = fused_check(c0, proofs[0], B // K, ell, peak_mm, PROD_PER_ADD)

        def st_single():
            # BASELINE configs[1]: the same instance shape as ONE proof (latency, not throughput): instance 0 alone on context 0
            c0.load_batch(data["vec_R"][:ell * AFF], data["vec_S"][:ell * AFF], data["vec_T"][:ell * AFF], data["vec_U"][:ell * AFF], data["M"][:JAC])
            tp, tv = [], []
            for _ in range(4):
                t0 = time.perf_counter()
                one = c0.prove_batch(data["perms"][:ell], data["k"][:FR], data["mbl"][:4 * FR], data["prand"][:(3 * n_ + 9) * FR])
                t1 = time.perf_counter()
                v1 = c0.verify_batch(one, data["vrand"][:8 * FR])
                t2 = time.perf_counter()
                tp.append(t1 - t0)
                tv.append(t2 - t1)
            out["single_proof"] = dict(prove_ms=1e3 * min(tp), verify_ms=1e3 * min(tv), accepted=v1 == [cpx.CPX_OK],
                                       note="BASELINE configs[1] as a latency figure: one ell=%d proof alone on the GPU (best of 4)" % ell)

        def st_shapes():
            # BASELINE configs[3] at its literal shapes: 1024 proofs total = 128 per GPU on 8 GPUs; and 1024 on one GPU.  One context, proofs of the
            # bench batch; steady-state prove + verify passes (not the headline: smaller batches fill the GPU less)
            shapes = {}
            for nb in (128, 1024):
                if nb <= B:
                    shapes["batch_%d_per_gpu" % nb] = shape_passes(c0, data, ell, nb)
            if len(ctxs) > 1 and 2048 <= B:
                # the 1024 proofs of configs[3] on one GPU driven like the headline — two engine contexts: the literal batch split 2 x 512, and two batches
                # of 1024 in flight (what a service that always has a next batch sees)
                ctxs[1].set_crs(ell, data["crs"])
                shapes["batch_1024_per_gpu"]["split_over_two_contexts"] = shape_passes_two(c0, ctxs[1], data, ell, 512)
                shapes["batch_1024_per_gpu"]["two_contexts"] = shape_passes_two(c0, ctxs[1], data, ell, 1024)
            out["baseline_shapes"] = dict(note="BASELINE configs[3] (1024 independent ell=%d proofs) at its literal shapes, one engine context: 128 proofs per GPU "
                                               "(the per-GPU share on 8 GPUs) and 1024 proofs on one GPU" % ell, **shapes)

        def st_config5_literal():
            # BASELINE configs[4] at its literal shapes: 4096 proofs in one accumulated check = 512 per GPU on 8 GPUs; and all 4096 on one GPU
            lit = {}
            for nb in (512, 4096):
                if nb <= B:
                    load_first(c0, data, ell, nb)
                    a_ = [c0.marshal(x) for x in (data["perms"][:nb * ell], data["k"][:nb * FR], data["mbl"][:nb * 4 * FR], data["prand"][:nb * (3 * n_ + 9) * FR])]
                    lit["proofs_%d_per_check" % nb] = fused_check(c0, c0.prove_batch(a_[0], a_[1], a_[2], a_[3], raw=True), nb, ell, peak_mm, PROD_PER_ADD, brief=True)
            if "config5" in out:
                out["config5"]["literal_shapes"] = dict(note="BASELINE configs[4] is 4096 proofs in ONE accumulated check: 512 per GPU on 8 GPUs (then one all-gather of "
                                                             "148 bytes per rank), or all 4096 on one GPU; one engine context each", **lit)

        def st_other_ells():
            # the reference's own bench shape (benches/perf.rs:29: N = 512, ell = 508), prover and verifier rates apart as criterion prints
            # them; and BASELINE configs[2] as throughput: ell = 1020 (n = 1024, 10 rounds)
            c1 = ctxs[1] if len(ctxs) > 1 else None
            out["ell_508"] = other_ell(c0, 508, 4096, 5081, "the reference's in-tree criterion bench shape (benches/perf.rs:29, N = 512)", c1)
            out["ell_1020"] = other_ell(c0, 1020, 2048, 4321, "BASELINE configs[2] shape (ell=1020, n=1024, 10 rounds) as throughput", c1)

        def st_cpu():
            out["cpu_baseline"] = cpu_baseline(ell, data, proofs[0])   # (+ a sample of 8 proofs of context 0's last pass against the oracle's bytes)
            if "config5" in out:
                vs = out["cpu_baseline"]["verify_s"]
                out["config5"]["cpu_verify_per_proof"] = dict(value=1.0 / vs, unit="proofs/s", cores=1, kind="port",
                                                              sample="CurdleproofsProof::verify of instance 0, oracle/ single thread, %.4f s each" % vs)

        def st_cpu_all():
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(ell, data)

        def st_traffic():
            for c in ctxs:   # the counter passes run in child processes with a context of their own: give the HBM back first
                c.close()
            tr, how = measure_traffic(dom, ell, B // K, local_rank, engine_opts)
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = tr, how
            if tr:
                out["roofline"]["traffic_over_algorithmic"] = tr / bytes_per_launch if bytes_per_launch else None

        out["cpu_baseline"] = None
        if not args.no_extra_shapes:
            stage("config5", st_config5)
        stage("single_proof", st_single)
        if not args.no_extra_shapes:
            stage("baseline_shapes", st_shapes)
            stage("config5_literal_shapes", st_config5_literal)
            if ell == 252:
                stage("other_proof_sizes", st_other_ells)
        if not args.no_cpu_baseline:   # the CPU oracle beside the GPU number in the same run, on rank 0 at every N
            stage("cpu_baseline", st_cpu)
            if not args.no_extra_shapes:
                stage("cpu_baseline_all_cores", st_cpu_all)
        if not args.no_traffic and os.environ.get("CPX_BENCH_TRAFFIC", "1") != "0":
            stage("traffic", st_traffic)
        if errors:
            out["errors"] = errors
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not all_ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
