"""BLS12-381 constants and byte-encoding helpers (pure Python ints) for callers of the C-ABI:
Montgomery wire forms of scalars / the G1 generator, used to build synthetic inputs."""
import secrets

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
G1_X = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
G1_Y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1


def fr_to_wire(x):
    """integer mod r -> 32-byte Montgomery form (4 x u64 LE), arkworks' in-memory Fr"""
    return ((x % R) * (1 << 256) % R).to_bytes(32, "little")


def fr_from_wire(b):
    return int.from_bytes(b, "little") * pow(1 << 256, -1, R) % R


def fp_to_wire(x):
    return ((x % P) * (1 << 384) % P).to_bytes(48, "little")


def g1_generator_wire():
    return fp_to_wire(G1_X) + fp_to_wire(G1_Y)


def random_fr_wire(rng, n):
    """n uniform non-zero scalars in wire form.  rng=None draws from the OS CSPRNG (`secrets`): what a caller needs for the
    verifier's accumulate_check factors (msm_accumulator.rs:44).  A seeded `random.Random` is accepted for reproducible
    synthetic benchmark inputs only — never for verification factors."""
    if rng is None:
        return b"".join(fr_to_wire(1 + secrets.randbelow(R - 1)) for _ in range(n))
    return b"".join(fr_to_wire(rng.randrange(1, R)) for _ in range(n))
