"""Dependent modular-multiply chains on the 32-bit-limb field and the 28-bit-limb lazy field (k_bench_fpmul /
k_bench_f28mul): products per second at 1, 2 and 4 waves per SIMD."""
import os
import subprocess
import sys

if len(sys.argv) > 1:
    import curdleproofs_amd as cp
    ctx = cp.Context(0)
    for blocks in (256, 512, 1024, 2048):   # x4 waves per block over 256 CUs x 4 SIMDs
        print("field=%s blocks=%d products/s=%.4g" % (os.environ.get("CPX_BENCH_FIELD", "32"), blocks, ctx.bench_fpmul(blocks, 2000, 3)))
else:
    for f in ("32", "28"):
        subprocess.check_call([sys.executable, __file__, "run"], env=dict(os.environ, CPX_BENCH_FIELD=f))
