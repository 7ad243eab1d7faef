"""The newest committed bench line (profiles/rNN_bench.json, produced by `python bench.py` on an MI355X) carries every field
of the driver's contract, and bench.py's command line accepts the driver's flags."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench.json")))[-1]
    with open(newest) as f:
        d = json.loads(f.read())
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].split(",")[0].startswith(base["metric"].split(" at ")[0])      # "shuffle proofs/sec (prove+verify)"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    assert d["all_verified"] is True and c["gpu_proof_bit_exact"] is True
    if os.path.basename(newest) >= "r03_bench.json":   # round 3 on: the batched verifier (BASELINE configs[4]) is part of the default line
        c5 = d["config5"]
        assert c5["all_verified"] is True and c5["rejects_a_batch_with_one_wrong_proof"] is True and c5["proofs_per_check"] >= 4096
        assert 0 < c5["valu"]["frac"] < 1 and c5["cpu_verify_per_proof"]["kind"] == "port"
        assert "traffic_source" in r


def test_bench_cli_accepts_driver_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_bench_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks (--gpus was parsed and ignored).
    Rehearsed without GPUs: CPX_BENCH_DRYRUN skips the proofs, CPX_BENCH_BACKEND=gloo replaces RCCL; rank 0 reports the world size
    the process group saw and the max over ranks of the per-rank dummy time (1 + rank)."""
    env = dict(os.environ, CPX_BENCH_DRYRUN="1", CPX_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # exactly one line, from rank 0
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["max_over_ranks_dt"] == 2.0 and d["steps"] == 2 and d["warmup"] == 1


def test_bench_gpus_8_dry_run():
    """The driver's one-shot scaling run ends at N = 8: the same rehearsal with eight ranks (gloo, no GPU) — the launcher's own rendezvous
    (--standalone: it binds the port itself), eight process-group members, the max over ranks, one line from rank 0, two host threads per
    engine context under the 16-core quota."""
    env = dict(os.environ, CPX_BENCH_DRYRUN="1", CPX_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["max_over_ranks_dt"] == 8.0 and d["steps"] == 20 and d["warmup"] == 5
    assert d["host_threads_per_context"] >= 2


def test_bench_under_the_drivers_launcher_command_line():
    """The driver does not use bench.py's own launcher for N > 1: it runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`.  The same command line, dry, with N = 2."""
    import socket
    env = dict(os.environ, CPX_BENCH_DRYRUN="1", CPX_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    with socket.socket() as sk:   # (the driver picks its port; any free one will do for the rehearsal)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_synthetic_scalars_are_seeded_and_below_the_group_order():
    sys.path.insert(0, ROOT)
    import bench
    from curdleproofs_amd import params as pr
    a, b = bench.synth_scalars(7, 1000), bench.synth_scalars(7, 1000)
    assert a == b and a != bench.synth_scalars(8, 1000) and len(a) == 32000
    vals = [int.from_bytes(a[32 * i:32 * i + 32], "little") for i in range(1000)]
    assert all(0 < v < pr.R for v in vals) and len(set(vals)) == 1000 and max(vals).bit_length() == 254


def test_bench_refuses_a_world_that_disagrees_with_gpus():
    env = dict(os.environ, CPX_BENCH_DRYRUN="1", CPX_BENCH_BACKEND="gloo", WORLD_SIZE="1", RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def test_every_msm_kernel_the_engine_reports_has_a_price():
    """bench.py prices the dominant MSM kernel of the timed region through ONE table (msm_pair_cost).  Until round 5 the dominant-kernel lookup had a
    table of its own that lacked k_msm_tblw_pair: when that kernel happened to be the longest MSM kernel of a small two-rank run, rank 0 died of a
    KeyError after the timed region — the unexplained 'launcher failure' the two-rank GPU test retried."""
    sys.path.insert(0, ROOT)
    import bench
    import curdleproofs_amd as cpx
    msm = [k for k in cpx.Context.KERNELS if k.startswith(("k_msm_fix", "k_msm_tblw", "k_msm_accw"))]
    assert "k_msm_tblw_pair" in msm and len(msm) >= 10
    for k in msm:
        assert bench.msm_pair_cost(k) > 0, k
    assert bench.msm_pair_cost("k_msm_fix<19, 7>") == 14 * bench.PROD_PER_ADD and bench.msm_pair_cost("k_msm_fix<16, 16>") == 16 * bench.PROD_PER_ADD
    for k in cpx.Context.KERNELS:
        if not k.startswith(("k_msm_fix", "k_msm_tblw", "k_msm_accw")):
            assert bench.msm_pair_cost(k) == 0.0, k   # (k_msm_tail, k_smul, k_reduce_sets ...: no pairs)
