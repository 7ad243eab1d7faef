"""Dependent modular-multiply chains on the 32-bit-limb field and the 28-bit-limb lazy field (k_bench_fpmul /
k_bench_f28mul): products per second at 1, 2 and 4 waves per SIMD."""
import curdleproofs_amd as cp

ctx = cp.Context(0)
for field in (12, 28):      # option bench_field: 12 x 32-bit limbs (standard form) / 14 x 28-bit limbs (table form)
    ctx.set_option("bench_field", field)
    for blocks in (256, 512, 1024, 2048):   # x4 waves per block over 256 CUs x 4 SIMDs
        print("field=%s blocks=%d products/s=%.4g" % ("32" if field == 12 else "28", blocks, ctx.bench_fpmul(blocks, 2000, 3)))
