"""world_size-2 gloo test of the multi-GPU plumbing (partitioning, max-over-ranks timing, verdict gather).
Workers verify their shard with the CPU oracle as a stand-in for the device engine (tests may use the oracle)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import torch.distributed as dist
from curdleproofs_amd.dist import env_world, partition, max_over_ranks, gather_lists, all_gather_partials
from tests.oracle_lib import Oracle
world, rank, _ = env_world()
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
orc = Oracle()
ell, total = 28, 5
crs = orc.generate_crs_points(ell)
mine = list(partition(total, world, rank))
t0 = time.perf_counter()
verdicts = []
for seed in mine:                      # BASELINE config 4 recipe: proof i uses StdRng(i)
    inst = orc.make_instance(ell, seed, crs)
    verdicts.append((seed, orc.verify(ell, crs, inst["vec_R"], inst["vec_S"], inst["vec_T"], inst["vec_U"], inst["M"], inst["proof"], inst["verifier_rand"])))
dt = time.perf_counter() - t0
tmax = max_over_ranks(dt, dist)
allv = gather_lists(verdicts, dist)
# config 5 exchange step: rank r contributes (+/-) k * G so that the partial sums cancel; the oracle stands in for the GPU adder
k = orc.fr_from_u64(987654321)
mk = orc.fr_from_canonical_bytes((0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001 - 987654321).to_bytes(32, "little"))
part = orc.g1_msm(orc.g1_generator(), k if rank == 0 else mk)
parts, bad = all_gather_partials(part, rank, dist)          # rank r reports r invalid proofs
total = parts[:144]
for r in range(1, world):
    total = orc.g1_add_jac(total, parts[144 * r:144 * (r + 1)])
if rank == 0:
    print(json.dumps(dict(world=world, order=[s for s, _ in allv], ok=[v for _, v in allv], tmax_ge_local=tmax >= dt - 1e-9,
                          gathered=len(parts) // 144, bad=bad, cancels=orc.g1_compress_jac(total)[0] == 0xc0,
                          first_is_mine=parts[:144] == part)))
dist.barrier()
dist.destroy_process_group()
'''


def test_partition_properties():
    from curdleproofs_amd.dist import partition
    for total in (0, 1, 5, 8, 1024, 1027):
        for world in (1, 2, 3, 8):
            parts = [list(partition(total, world, r)) for r in range(world)]
            assert [x for p in parts for x in p] == list(range(total))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_two_rank_gloo_run(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29613", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["world"] == 2 and res["order"] == [0, 1, 2, 3, 4] and res["ok"] == [1] * 5 and res["tmax_ge_local"]
    # config 5: all-gather of the 144-byte partial sums (+ invalid counts) in rank order, partials cancel
    assert res["gathered"] == 2 and res["bad"] == 1 and res["cancels"] and res["first_is_mine"]
